#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3n; mkdir -p $O
timeout 300 python bench.py --self-loop --steps 10 --warmup 3 --no-cpu > $O/bench_selfloop.json 2> $O/selfloop.err; echo "self-loop rc=$?"; cut -c1-1100 $O/bench_selfloop.json; tail -3 $O/selfloop.err
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 3 --warmup 1 > $O/torchrun2.out 2> $O/torchrun2.err; echo "2-rank dry run rc=$? (ncclCommInitRank refuses two ranks on one device: expected)"; grep -h "RCCL error\|invalid usage\|Error" $O/torchrun2.err | head -3
