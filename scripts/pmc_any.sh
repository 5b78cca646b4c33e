#!/bin/bash
# usage: [env] scripts/pmc_any.sh <tag> <script.py>  -- SQ / TCC counters per kernel (separate passes; FETCH_SIZE and WRITE_SIZE each alone)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -d $O/pmc1 -o s -- python3 $R/$2 > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_SMEM -d $O/pmc2 -o s -- python3 $R/$2 > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc3 -o s -- python3 $R/$2 > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/pmc4 -o s -- python3 $R/$2 > /dev/null 2>&1
cd $O && python3 - <<'PY'
import csv,glob,collections
for f in sorted(glob.glob('*/*counter_collection.csv')):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r['Kernel_Name'][:40],r['Counter_Name'])].append(float(r['Counter_Value']))
    for (k,c),v in sorted(agg.items()):
        if 'copy' in k.lower() or 'fill' in k.lower(): continue
        print(f.split('/')[0], k, c, "%.4g"%(sum(v)/len(v)), len(v))
PY
