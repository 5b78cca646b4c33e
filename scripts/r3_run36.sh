#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3z; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --maxfail=8 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
timeout 300 python scripts/fuzz_vs_scipy.py 150 77001 2>&1 | tail -2 | cut -c1-140 | tee $O/fuzz_after_aniso.txt
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err; cut -c1-260 $O/bench_line.json
