#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
SGK_BENCH_BACKEND=gloo SGK_BENCH_ONE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 300 --warmup 50 --envs-per-gpu 262144 > gpurun_out/bench_2rank_gloo.log 2>&1
echo "rc=$?"; tail -3 gpurun_out/bench_2rank_gloo.log | cut -c1-900
