#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3h; mkdir -p $O
timeout 120 scripts/diag/bin/f64_rate 2>&1 | tee $O/f64_rate.txt
