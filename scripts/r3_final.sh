#!/bin/bash
# end-of-round validation: full GPU suite, the bench lines, the per-config table with parity, kernel stats, a mid-size fuzz
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3z; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err
timeout 400 python bench.py > $O/bench_line_50steps.json 2>> $O/bench.err
cut -c1-1100 $O/bench_line.json; echo; cut -c1-400 $O/bench_line_50steps.json; echo
timeout 900 python scripts/bench_configs.py > $O/configs_bench.jsonl 2> $O/configs.err; cat $O/configs_bench.jsonl
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/cfgstats -o c -- python3 $GRAFT_REPO_ROOT/scripts/bench_configs.py --no-parity --reps 12 > /dev/null 2>&1)
cut -d, -f1-7 $O/cfgstats/*kernel_stats.csv 2>/dev/null | head -14
FUZZ_BIG=1 timeout 260 python scripts/fuzz_vs_scipy.py 200 31337 2>&1 | tail -3 | tee $O/fuzz_big.txt
timeout 200 python scripts/fuzz_vs_scipy.py 150 2026 2>&1 | tail -3 | tee $O/fuzz_2026.txt
