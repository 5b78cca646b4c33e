#!/bin/bash
# first GPU pass of round 3: full GPU test suite, the bench line, the per-config table with parity, exploration sweep
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
O=gpurun_out/r3a
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
timeout 900 python scripts/bench_configs.py > $O/configs.jsonl 2> $O/configs.err
timeout 900 python scripts/r3_explore.py > $O/explore.txt 2>&1
(cd /tmp && export TMPDIR=/tmp && rocprofv3 -L > $GRAFT_REPO_ROOT/$O/counters_list.txt 2>&1)
tail -5 $O/pytest.log; cat $O/bench.json | cut -c1-1500; cat $O/configs.jsonl; cat $O/explore.txt
