#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3d; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_vs_oracle.py tests/test_gpu_baseline_full.py tests/test_gpu_halo.py -m gpu -x -q -k "long or gaussian or uniform or minmax or min_max or B_ or E_ or H_ or slab or halo or plane" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
timeout 600 python scripts/r3_explore.py long 2>&1 | tee $O/explore_long.txt
timeout 300 python scripts/bench_configs.py --only H,B,E --no-parity 2>&1 | tee $O/configs.jsonl
timeout 200 python scripts/bench_morph3d.py 2>&1 | tail -12 | tee $O/morph3d.txt
