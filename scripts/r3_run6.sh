#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_vs_oracle.py tests/test_gpu_halo.py tests/test_gpu_fuzz.py -m gpu -q --maxfail=10 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
timeout 300 python scripts/check_spline_ties.py 2>&1 | tee $O/ties.txt
timeout 260 python scripts/fuzz_vs_scipy.py 200 4242 2>&1 | tail -8 | tee $O/fuzz_4242.txt
timeout 360 python scripts/fuzz_vs_scipy.py 300 13579 2>&1 | tail -8 | tee $O/fuzz_13579.txt
FUZZ_ONLY=map1,affine3,zoom,shift,spline_filter timeout 220 python scripts/fuzz_vs_scipy.py 150 777111 2>&1 | tail -8 | tee $O/fuzz_777111.txt
timeout 600 bash scripts/profile_bench.sh r3f/prof > $O/profile.log 2>&1
cat gpurun_out/r3f/prof/traffic.json; head -12 gpurun_out/r3f/prof/summary.txt
