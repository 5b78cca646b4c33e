#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3z; mkdir -p $O
timeout 300 python scripts/r3_slab_kernels.py 2>&1 | tee $O/slab_kernels.txt
