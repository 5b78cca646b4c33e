#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3j; mkdir -p $O
timeout 300 python bench.py --self-loop --steps 20 --warmup 5 > $O/bench_selfloop.json 2> $O/selfloop.err; cut -c1-900 $O/bench_selfloop.json; tail -2 $O/selfloop.err
timeout 600 python bench.py --config E --steps 5 --warmup 2 > $O/bench_E.json 2> $O/E.err; cut -c1-900 $O/bench_E.json; tail -2 $O/E.err
