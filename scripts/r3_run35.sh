#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3z; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_vs_oracle.py -m gpu -x -q -k "anisotropic or generations" 2>&1 | tail -12 | tee $O/pytest_aniso.txt
timeout 300 python scripts/r3_aniso.py 2>&1 | tee $O/aniso.txt
