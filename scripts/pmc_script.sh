#!/bin/bash
# usage: scripts/pmc_script.sh <tag> <python script>   (run via gpurun, wrap in timeout)
# SQ / memory counters per kernel for any script, three small passes
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -d $O/pmc1 -o s -- python3 $R/$1 > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_INSTS_BRANCH -d $O/pmc2 -o s -- python3 $R/$1 > /dev/null 2>&1
# FETCH_SIZE takes 3 of the 4 TCC slots: it gets a pass of its own (with TCC_HIT/MISS in the same pass rocprofv3 hung)
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc3 -o s -- python3 $R/$1 > /dev/null 2>&1
cd $O && python3 - <<'PY'
import csv,glob,collections
for f in sorted(glob.glob('*/*counter_collection.csv')):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r['Kernel_Name'][:44],r['Counter_Name'])].append(float(r['Counter_Value']))
    for (k,c),v in sorted(agg.items()):
        print(f.split('/')[0], k, c, "%.4g"%(sum(v)/len(v)), len(v))
PY
