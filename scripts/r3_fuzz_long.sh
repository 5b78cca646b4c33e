#!/bin/bash
# a long differential run against scipy.ndimage on the final library of the round
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3z; mkdir -p $O
timeout 700 python scripts/fuzz_vs_scipy.py 560 20261003 2>&1 | tail -4 | tee $O/fuzz_long_final.txt
FUZZ_BIG=1 timeout 400 python scripts/fuzz_vs_scipy.py 300 918273 2>&1 | tail -3 | tee $O/fuzz_big_final.txt
