#!/bin/bash
# usage: scripts/pmc_configs.sh <tag> [--only H,B,C,D,Daff,E]     (run via gpurun, under timeout)
# rocprofv3 counter passes over scripts/bench_configs.py --no-parity for every BASELINE config: SQ instruction / wait
# counters, then FETCH_SIZE and WRITE_SIZE each in a pass of their own (TCC slots), --kernel-trace only (gpurun refuses
# --pmc with the tracing domains).  Prints per-kernel averages; HBM bytes = 2 x FETCH_SIZE KiB (gfx950 counts wide
# coalesced reads at half, MI355X_MICROARCH.md) + WRITE_SIZE KiB.
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
PROG=${PMC_PROG:-scripts/bench_configs.py}              # PMC_PROG=scripts/bench_cubic_affine.py PMC_ARGS=--counters: another program
ARGS=${PMC_ARGS:-"--no-parity --reps 6 $@"}
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -d $O/pmc1 -o s -- python3 $R/$PROG $ARGS > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM -d $O/pmc2 -o s -- python3 $R/$PROG $ARGS > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc3 -o s -- python3 $R/$PROG $ARGS > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/pmc4 -o s -- python3 $R/$PROG $ARGS > /dev/null 2>&1
cd $O && python3 - <<'PY' | tee counters.txt
import csv, glob, collections
rows = collections.OrderedDict()
for f in sorted(glob.glob('pmc*/*counter_collection.csv')):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r['Kernel_Name'][:48], r['Counter_Name'])].append(float(r['Counter_Value']))
    for (k, c), v in sorted(agg.items()):
        if any(s in k.lower() for s in ('copy', 'fill', 'synth')) or len(v) < 4:
            continue
        rows.setdefault(k, collections.OrderedDict())[c] = (sum(v) / len(v), len(v))
for k, cs in rows.items():
    print("==", k)
    for c, (v, n) in cs.items():
        print("   %-22s %14.5g  (n=%d)" % (c, v, n))
    if 'FETCH_SIZE' in cs and 'WRITE_SIZE' in cs:
        rd, wr = 2 * cs['FETCH_SIZE'][0] * 1024, cs['WRITE_SIZE'][0] * 1024
        print("   HBM bytes per launch: read %.4g (2 x FETCH_SIZE) + write %.4g = %.4g" % (rd, wr, rd + wr))
PY
rm -rf $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4
