#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3h; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --maxfail=12 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
timeout 300 python scripts/check_spline_ties.py 2>&1 | tee $O/ties.txt
timeout 260 python scripts/fuzz_vs_scipy.py 200 4242 2>&1 | tail -4 | tee $O/fuzz_4242.txt
FUZZ_ONLY=map1,affine3,zoom,shift,spline_filter timeout 220 python scripts/fuzz_vs_scipy.py 150 777111 2>&1 | tail -4 | tee $O/fuzz_777111.txt
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; cut -c1-900 $O/bench.json
