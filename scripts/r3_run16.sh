#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3h; mkdir -p $O
timeout 900 python scripts/r3_long3.py 2>&1 | tee $O/long3.txt
timeout 600 python -m pytest tests/test_gpu_vs_oracle.py -m gpu -x -q -k "long or gaussian or uniform or sep" 2>&1 | tail -5 | tee $O/pytest_long.txt
