#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3i; mkdir -p $O
timeout 300 python scripts/r3_affine_ab.py 2>&1 | tee $O/affine_ab.txt
