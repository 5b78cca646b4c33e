#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3i; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "interp or order or affine or map_coord or finite or zoom or shift or baseline_full or fuzz or lds" 2>&1 | tail -5 | tee $O/pytest_interp.txt
timeout 600 python scripts/bench_configs.py --only D,Daff 2>&1 | tee $O/configs_D.jsonl
timeout 300 python scripts/fuzz_vs_scipy.py 200 424242 2>&1 | tail -3 | tee $O/fuzz.txt
