#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3j; mkdir -p $O
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_line2.json 2> $O/bench2.err
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu > $O/bench_line3.json 2>> $O/bench2.err
timeout 400 python bench.py --no-cpu > $O/bench_line4.json 2>> $O/bench2.err
for f in $O/bench_line2.json $O/bench_line3.json $O/bench_line4.json; do python - $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d['steps'], d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['roofline'].get('copy_kernel_GBps_same_bytes'), d['roofline'].get('frac_of_copy_kernel'), (d.get('parity') or d.get('cpu_baseline') or {}) if False else '')
PY
done
tail -3 $O/bench2.err
