#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3h; mkdir -p $O
timeout 120 scripts/diag/bin/pkfma_rate 2>&1 | tee $O/pkfma_rate.txt
