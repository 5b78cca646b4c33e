#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3z; mkdir -p $O
timeout 900 python scripts/r3_long3.py 2>&1 | tee $O/long3_settled.txt
