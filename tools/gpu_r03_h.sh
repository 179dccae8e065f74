#!/bin/bash
# round 3, call h: regression of the de-waterfalled tile stores (full GPU suite + bench) and the ring-rate time series
cd "$GRAFT_REPO_ROOT" || exit 1
export SGK_NO_BUILD=1 TMPDIR=/tmp
mkdir -p gpurun_out/h
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/h/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/h/pytest_gpu.log
tail -3 gpurun_out/h/pytest_gpu.log
timeout 300 python tools/exp_ring_time_series.py 3000 ring > gpurun_out/h/ring_time_series.log 2>&1
timeout 300 python tools/exp_ring_time_series.py 3000 own > gpurun_out/h/own_time_series.log 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/h/bench_20.json 2> gpurun_out/h/bench_20.err
timeout 300 python bench.py --steps 200 --warmup 5 > gpurun_out/h/bench_200.json 2> gpurun_out/h/bench_200.err
cat gpurun_out/h/ring_time_series.log
