#!/bin/bash
# Profiles `python3 bench.py` with rocprofv3 on the GPU box (run via gpurun):
#   pass 1: --kernel-trace --stats          (per-kernel average duration)
#   pass 2: --pmc FETCH_SIZE                (HBM-side read bytes, separate pass)
#   pass 3: --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
#   pass 4: SQ counters
# Summaries land in gpurun_out/<tag>/summary.txt; copy the ones to be judged into profiles/.
TAG=${1:-prof}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
ARGS="$R/bench.py --steps 20 --warmup 5 --no-cpu"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python3 $ARGS > $O/bench_stats.json 2> $O/stats.err
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc_fetch -o b -- python3 $ARGS > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $O/pmc_write -o b -- python3 $ARGS > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -d $O/pmc_sq -o b -- python3 $ARGS > /dev/null 2>&1
cd $O && python3 - <<'PY' > summary.txt
import csv, glob, collections
print("# rocprofv3 summary for: python3 bench.py --steps 20 --warmup 5 --no-cpu")
for f in glob.glob('stats/*kernel_stats.csv'):
    print("## kernel stats (rocprofv3 --kernel-trace --stats)")
    print(open(f).read())
print("## PMC counters (per dispatch averages; separate passes)")
for f in sorted(glob.glob('pmc_*/*counter_collection.csv')):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r['Kernel_Name'], r['Counter_Name'])].append(float(r['Counter_Value']))
    for (k, c), v in sorted(agg.items()):
        print("%s | %s | %s | avg %.6g | n=%d" % (f.split('/')[0], k[:70], c, sum(v) / len(v), len(v)))
print("""
## how to read
FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE reports exactly half of the bytes of a wide
(16 B/lane) coalesced read stream (MI355X_MICROARCH.md, HBM section): HBM read bytes = 2 * FETCH_SIZE * 1024.
WRITE_SIZE is taken as is.  Algorithmic bytes per launch: 512^3 * 8 B = 1,073,741,824.""")
PY
cat summary.txt
