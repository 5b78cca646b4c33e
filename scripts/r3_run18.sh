#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3h; mkdir -p $O
timeout 600 python scripts/r3_long_rows_settled.py 2>&1 | tee $O/long_rows_settled.txt
