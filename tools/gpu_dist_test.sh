#!/bin/bash
# multi-process bench path on the 1-GPU box: 2 ranks (gloo, both on device 0), then 1 rank through torchrun with nccl
cd "$GRAFT_REPO_ROOT"; export PYTHONUNBUFFERED=1
echo "== 2 ranks, gloo, one GPU"
ADC_BENCH_BACKEND=gloo ADC_BENCH_DEVICE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 6 --warmup 2 2>&1 | tail -3 | cut -c1-700
echo "== 1 rank via torchrun, nccl"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -2 | cut -c1-400
echo "== plain default bench (with cpu baseline)"
timeout 900 python bench.py 2>&1 | tail -1
