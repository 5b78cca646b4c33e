#!/bin/bash
# usage: scripts/prof_stats.sh <tag> <bench_configs --only value>   (run via gpurun)
# per-kernel average durations (rocprofv3 --kernel-trace --stats) of scripts/bench_configs.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o c -- python3 $R/scripts/bench_configs.py --reps 5 --only $2 > $O/bench.jsonl 2> $O/stats.err
cat $O/bench.jsonl
cut -d, -f1-6 $O/stats/*kernel_stats.csv | head -20
