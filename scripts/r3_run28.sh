#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3z; mkdir -p $O
timeout 900 python scripts/bench_configs.py > $O/configs_bench2.jsonl 2> $O/configs2.err; cat $O/configs_bench2.jsonl | cut -c1-330
