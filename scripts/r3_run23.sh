#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3i; mkdir -p $O
timeout 300 python scripts/r3_long3_small_taps.py 2>&1 | tee $O/long3_small_taps.txt
