#!/bin/bash
# usage: [env] scripts/kstat_any.sh <tag> <script.py>   -- per-launch kernel durations of a script (run via gpurun)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $R/$2 > /dev/null 2> $O/err.txt
python3 - $O <<'PY'
import csv,glob,sys,collections
d=sys.argv[1]
agg=collections.OrderedDict()
for f in glob.glob(d+'/stats/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        agg.setdefault(r['Kernel_Name'][:60],[]).append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in agg.items():
    if len(v)>=3: print(k, "n=%d mean %.1f min %.1f" % (len(v), sum(v)/len(v), min(v)), [round(x) for x in v[:16]])
PY
