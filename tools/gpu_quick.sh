#!/bin/bash
# scratch: the last ad-hoc command sequence sent to the GPU box
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
# the multi-rank control flow of bench.py end to end on a 1-GPU box: both ranks on GPU 0, gloo collectives
SGK_BENCH_BACKEND=gloo SGK_BENCH_ONE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 500 --warmup 100 --envs-per-gpu 524288 > gpurun_out/bench_2rank_gloo.log 2>&1
tail -1 gpurun_out/bench_2rank_gloo.log | cut -c1-700
