#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3i; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "interp or order or affine or map_coord or finite or zoom or shift or baseline_full or lds" 2>&1 | tail -3 | tee $O/pytest_interp.txt
timeout 200 python scripts/fuzz_vs_scipy.py 100 626262 2>&1 | tail -2 | cut -c1-160 | tee $O/fuzz.txt
