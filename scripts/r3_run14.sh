#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3n; mkdir -p $O
timeout 300 python bench.py --self-loop --steps 10 --warmup 3 --no-cpu > $O/bench_selfloop.json 2> $O/selfloop.err; echo "self-loop rc=$? stdout lines: $(wc -l < $O/bench_selfloop.json)"; cut -c1-300 $O/bench_selfloop.json
