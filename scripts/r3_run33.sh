#!/bin/bash
cd $GRAFT_REPO_ROOT
VAR=1 REPS=8 timeout 1500 bash scripts/pmc_interp.sh r3k_affine > gpurun_out/r3k_affine.log 2>&1; tail -40 gpurun_out/r3k_affine/summary.txt
cd $GRAFT_REPO_ROOT
timeout 1000 bash scripts/pmc_any.sh r3k_long scripts/prof_long17.py > gpurun_out/r3k_long.txt 2>&1; cat gpurun_out/r3k_long.txt | tail -50
