#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3fz; mkdir -p $O
timeout 760 python scripts/fuzz_vs_scipy.py 700 987654 2>&1 | tail -4 | tee $O/fuzz_987654.txt
FUZZ_BIG=1 timeout 360 python scripts/fuzz_vs_scipy.py 300 24680 2>&1 | tail -4 | tee $O/fuzz_big_24680.txt
