#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3z; mkdir -p $O
timeout 600 python scripts/r3_map_variants_settled.py 2>&1 | tee $O/map_variants_settled.txt
