#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3e; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --maxfail=12 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -25 $O/pytest.log
