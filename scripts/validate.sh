#!/bin/bash
# End-of-round validation on a GPU box (run through gpurun): the whole GPU suite, the bench lines (driver flags, defaults,
# self-loop dry run of the N > 1 path, config E on one GPU), the per-config table with whole-volume parity, rocprofv3
# kernel stats of both, a mid-size differential fuzz.  usage: scripts/validate.sh <tag>   -> gpurun_out/<tag>/
TAG=${1:-validate}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --maxfail=12 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -v "^RCCL\|^HIP \|^ROCm\|^Hostname\|^Librccl" $O/pytest.log | tail -6
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"
timeout 400 python bench.py > $O/bench_line_50steps.json 2>> $O/bench.err
timeout 400 python bench.py --self-loop --steps 48 --warmup 10 > $O/bench_selfloop_dryrun.json 2>> $O/bench.err
timeout 600 python bench.py --config E --steps 5 --warmup 2 > $O/bench_config_E_n1.json 2>> $O/bench.err
cut -c1-1500 $O/bench_line.json; echo
timeout 900 python scripts/bench_configs.py > $O/configs_bench.jsonl 2> $O/configs.err; cat $O/configs_bench.jsonl
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/cfgstats -o c -- python3 $GRAFT_REPO_ROOT/scripts/bench_configs.py --no-parity --reps 12 > /dev/null 2>&1)
cp $O/cfgstats/*kernel_stats.csv $O/configs_kernel_stats.csv 2>/dev/null; cut -d, -f1-7 $O/configs_kernel_stats.csv | head -14
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/benchstats -o b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2>/dev/null)
cp $O/benchstats/*kernel_stats.csv $O/bench_kernel_stats.csv 2>/dev/null; cut -d, -f1-7 $O/bench_kernel_stats.csv | head -6
rm -rf $O/cfgstats $O/benchstats
timeout 300 python scripts/bench_slab_step.py --ranks 8 --graphs 1 2>/dev/null | grep "^{" > $O/slab_step_8.json; cut -c1-900 $O/slab_step_8.json
FUZZ_BIG=1 timeout 260 python scripts/fuzz_vs_scipy.py 200 31337 2>&1 | tail -3 | tee $O/fuzz_big.txt
timeout 200 python scripts/fuzz_vs_scipy.py 150 2026 2>&1 | tail -3 | tee $O/fuzz_2026.txt
# r6: the kernels added this round -- bit-packed binary morphology, the dense 3^3 / 5^3 stencil, min / max on ragged rows --
# their rocprofv3 kernel statistics and counters, and the headline's traffic counters (profiles/r6_traffic.json)
timeout 300 python scripts/bench_bitmorph.py > $O/bitmorph_bench.txt 2>&1; grep "shape\|cross  \|3^3\|ball2\|x3\|masked\|opening cross " $O/bitmorph_bench.txt | head -20
timeout 300 python scripts/bench_stencil.py > $O/stencil_bench.txt 2>&1; grep "3x3x3\|5x5x5" $O/stencil_bench.txt
timeout 300 python scripts/bench_fill_holes.py > $O/fill_holes.txt 2>&1; head -5 $O/fill_holes.txt | cut -c1-120
timeout 200 python scripts/survey_mni.py > $O/mni_survey.txt 2>&1
timeout 300 python scripts/bench_ragged_long.py > $O/ragged_long.txt 2>&1; head -3 $O/ragged_long.txt | cut -c1-140
timeout 300 python scripts/bench_ragged_minmax.py > $O/ragged_minmax.txt 2>&1; head -4 $O/ragged_minmax.txt
bash scripts/kstat_any.sh $TAG/binary_stats scripts/prof_bitmorph.py > $O/binary_kstat.txt 2>&1; cat $O/binary_kstat.txt
timeout 600 bash scripts/pmc_script.sh $TAG/binary_pmc scripts/prof_bitmorph.py > $O/binary_pmc.txt 2>&1; tail -30 $O/binary_pmc.txt
timeout 900 bash scripts/profile_bench.sh $TAG/headline_prof > $O/headline_prof.txt 2>&1; tail -14 $O/headline_prof.txt
cd $GRAFT_REPO_ROOT
timeout 400 python scripts/fuzz_r6.py 240 606 2>&1 | grep -v "_kernel" | tail -8 | tee $O/fuzz_r6.txt
# r5: the targeted fuzz of the round-5 routes and the tables of the kernels added this round
timeout 260 python scripts/fuzz_r5.py 200 505 2>&1 | tail -60 | tee $O/fuzz_r5.txt
timeout 300 python scripts/bench_ragged_rows.py > $O/ragged_rows.txt 2>&1
timeout 300 python scripts/bench_constant_mode.py > $O/constant_mode.txt 2>&1
timeout 600 python scripts/probe_median.py > $O/rank_filters.txt 2>&1; tail -4 $O/rank_filters.txt
