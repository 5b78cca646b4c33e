#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3i; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "affine or lds or Dprime" 2>&1 | tail -2 | tee $O/pytest_affine.txt
timeout 300 python scripts/r3_affine_ab.py 2>&1 | tee $O/affine_ab3.txt
