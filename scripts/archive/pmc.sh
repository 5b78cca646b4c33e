#!/bin/bash
# usage: scripts/pmc.sh <outdir-name> ; env CFG ZCH etc are passed to prof_one.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $R/scripts/prof_one.py > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -d $O/pmc1 -o s -- python3 $R/scripts/prof_one.py > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_SMEM -d $O/pmc2 -o s -- python3 $R/scripts/prof_one.py > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc3 -o s -- python3 $R/scripts/prof_one.py > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $O/pmc4 -o s -- python3 $R/scripts/prof_one.py > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_MISC SQ_INSTS_FLAT SQ_INSTS_VALU_MFMA_MOPS_F32 -d $O/pmc5 -o s -- python3 $R/scripts/prof_one.py > /dev/null 2>&1
cd $O && python3 - <<'PY'
import csv,glob,collections
for f in sorted(glob.glob('*/*counter_collection.csv')):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r['Kernel_Name'][:40],r['Counter_Name'])].append(float(r['Counter_Value']))
    for (k,c),v in sorted(agg.items()):
        print(f.split('/')[0], k, c, "%.4g"%(sum(v)/len(v)), len(v))
for f in glob.glob('stats/*kernel_stats.csv'):
    print(open(f).read())
PY
