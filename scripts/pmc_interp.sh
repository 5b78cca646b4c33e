#!/bin/bash
# usage: [VAR=..] [MAP=1] scripts/pmc_interp.sh <tag>  -- TA / TCP / SQ counters of the order-1 interpolation kernels (separate passes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
S=$R/scripts/prof_interp.py
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $S > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -d $O/pmc1 -o s -- python3 $S > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM -d $O/pmc2 -o s -- python3 $S > /dev/null 2>&1
# (the TA_* counter pass is left out: those counters hang rocprofv3 on this pool -- a 40-minute loss in round 3)
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum -d $O/pmc4 -o s -- python3 $S > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TD_STORE_WAVEFRONT_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum -d $O/pmc5 -o s -- python3 $S > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc6 -o s -- python3 $S > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $O/pmc7 -o s -- python3 $S > /dev/null 2>&1
cd $O && python3 - <<'PY' > summary.txt
import csv,glob,collections,os
print("# VAR=%s MAP=%s  (scripts/pmc_interp.sh; per-dispatch averages; SQ_* cycle counters in quad-cycles; FETCH/WRITE_SIZE KiB)" % (os.environ.get("VAR","1"), os.environ.get("MAP","")))
for f in glob.glob('stats/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'map_coords' in r['Name'] or 'affine' in r['Name']:
            print("kernel_stats", r['Name'][:90], "calls", r['Calls'], "avg_ns", r['AverageNs'], "min_ns", r['MinNs'], "max_ns", r['MaxNs'])
for f in sorted(glob.glob('pmc*/*counter_collection.csv')):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'map_coords' in r['Kernel_Name'] or 'affine' in r['Kernel_Name']:
            agg[(r['Kernel_Name'][:44],r['Counter_Name'])].append(float(r['Counter_Value']))
    for (k,c),v in sorted(agg.items()):
        print(f.split('/')[0], k, c, "%.5g"%(sum(v)/len(v)), len(v))
PY
cat summary.txt
