#!/bin/bash
# usage (through gpurun): scripts/sweep_slab_nchannels.sh <tag>
# r5: per-rank slab step (8-rank share of the headline volume, self-loop communicator) with RCCL restricted to 1 / 2 / 4 channels:
# how many workgroups (CUs) the exchange kernel takes and what a step costs in every schedule.  NCCL_* are read when the
# communicator is created: one process per setting.  Output: gpurun_out/<tag>/slab_nchannels.txt -> profiles/r5_slab_step.txt
set -eu
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}"
: "${1:?usage: scripts/sweep_slab_nchannels.sh <tag>}"
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
O="$R/gpurun_out/$1"
mkdir -p "$O"
: > "$O/slab_nchannels.txt"
for ch in default 1 2 4; do
  if [ "$ch" = default ]; then unset NCCL_MAX_NCHANNELS NCCL_MIN_NCHANNELS; else export NCCL_MAX_NCHANNELS=$ch NCCL_MIN_NCHANNELS=$ch; fi
  echo "# NCCL_MAX_NCHANNELS = NCCL_MIN_NCHANNELS = $ch" >> $O/slab_nchannels.txt
  timeout 300 python3 "$R/scripts/bench_slab_step.py" --ranks 8 --size 5 --side 512 --graphs 1 >> "$O/slab_nchannels.txt" 2>> "$O/err.txt" \
    || echo "# FAILED (rc $?): bench_slab_step.py with $ch channel(s) -- no figures for this setting" >> "$O/slab_nchannels.txt"
  # the exchange kernel's grid (workgroups) and duration from a kernel trace of a short run
  rm -rf "$O/trace_$ch"
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$O/trace_$ch" -o t -- python3 "$R/scripts/bench_slab_step.py" --ranks 8 --size 5 --side 512 --steps 40 > /dev/null 2>> "$O/err.txt" \
    || echo "# FAILED (rc $?): kernel trace with $ch channel(s)" >> "$O/slab_nchannels.txt"
  python3 - "$O/trace_$ch" >> "$O/slab_nchannels.txt" <<'PY'
import csv, glob, sys, collections
agg = collections.OrderedDict()
for f in glob.glob(sys.argv[1] + '/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'nccl' in n.lower() or 'rccl' in n.lower() or 'sep3d' in n:
            wg = int(r['Workgroup_Size_X']) * int(r.get('Workgroup_Size_Y', 1) or 1) * int(r.get('Workgroup_Size_Z', 1) or 1)
            grid = int(r['Grid_Size_X']) * int(r.get('Grid_Size_Y', 1) or 1) * int(r.get('Grid_Size_Z', 1) or 1)
            agg.setdefault((n[:60], grid // max(wg, 1), wg), []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for (n, wgs, wg), v in agg.items():
    v.sort()
    print('#   kernel %-60s workgroups %4d x %4d threads   n=%5d  median %.1f us' % (n, wgs, wg, len(v), v[len(v) // 2]))
PY
done
cat "$O/slab_nchannels.txt"
