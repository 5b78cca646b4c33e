#!/bin/bash
# usage: [env for prof_one.py] scripts/kstat.sh <tag>   -- per-kernel average durations of scripts/prof_one.py (run via gpurun)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $R/scripts/prof_one.py > /dev/null 2> $O/err.txt
cut -d, -f1-7 $O/stats/*kernel_stats.csv | head -8
