#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3o; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
