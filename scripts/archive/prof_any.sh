#!/bin/bash
# usage: scripts/prof_any.sh <tag> <bench_configs --only value>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
A="$R/scripts/bench_configs.py --reps 3 --only $2"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -d $O/p1 -o c -- python3 $A > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA -d $O/p2 -o c -- python3 $A > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/p3 -o c -- python3 $A > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $O/p4 -o c -- python3 $A > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum -d $O/p5 -o c -- python3 $A > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_GATE_EN1_sum TCP_TA_DATA_STALL_CYCLES_sum -d $O/p6 -o c -- python3 $A > /dev/null 2>&1
cd $O && python3 - <<'PY'
import csv,glob,collections
for f in sorted(glob.glob('*/*counter_collection.csv')):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r['Kernel_Name'][:48],r['Counter_Name'], r.get('VGPR_Count','?'))].append(float(r['Counter_Value']))
    for (k,c,vg),v in sorted(agg.items()):
        print(f.split('/')[0], k, 'vgpr',vg, c, "%.4g"%(sum(v)/len(v)), len(v))
PY
