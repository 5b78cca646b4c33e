#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3z; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_baseline_full.py -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest_full.txt
