#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3j; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --maxfail=8 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err
timeout 400 python bench.py > $O/bench_line_50steps.json 2>> $O/bench.err
cut -c1-1400 $O/bench_line.json; echo; cut -c1-300 $O/bench_line_50steps.json; echo
bash scripts/profile_bench.sh r3j > $O/profile.log 2>&1; tail -30 $O/profile.log
