#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3j; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_vs_oracle.py -m gpu -x -q -k "long_kernel_generations" 2>&1 | tail -15 | tee $O/pytest_long2.txt
