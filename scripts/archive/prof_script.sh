#!/bin/bash
# usage: scripts/prof_script.sh <tag> <python script> [args...]   (run via gpurun)
# per-kernel average durations (rocprofv3 --kernel-trace --stats) of any script
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o c -- python3 $R/$@ > $O/out.txt 2> $O/stats.err
cut -d, -f1-4 $O/stats/*kernel_stats.csv | head -40
