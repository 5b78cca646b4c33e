#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3z; mkdir -p $O
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_final.json 2> $O/bench_final.err; echo "rc=$?"; wc -l $O/bench_final.json; cut -c1-330 $O/bench_final.json; grep -o '"device_load_before_warmup": "[^"]*"' $O/bench_final.json; grep -o '"parity[^}]*}' $O/bench_final.json | head -2
