#!/bin/bash
# bench line + its contract tests on the GPU
cd "$GRAFT_REPO_ROOT" || exit 1
export SGK_NO_BUILD=1 TMPDIR=/tmp
mkdir -p gpurun_out/q
timeout 900 python -m pytest tests/test_bench_contract.py -x -q > gpurun_out/q/contract.log 2>&1; tail -3 gpurun_out/q/contract.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/q/bench.json 2> gpurun_out/q/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/q/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["device_us_per_step"], d["other_ring_allocations"], d["gpu_leg_device_ms"])
PY
