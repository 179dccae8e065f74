#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export SGK_NO_BUILD=1 TMPDIR=/tmp
mkdir -p gpurun_out/x
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/x/pytest_gpu.log 2>&1; tail -3 gpurun_out/x/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/x/bench.json 2> gpurun_out/x/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/x/bench.json").read().strip().splitlines()[-1])
print("%.3e"%d["value"], d["roofline"]["device_us_per_step"], d["ring_allocation"], d["probed_ring_allocation"], d["other_ring_allocations"]["device_us_per_lockstep_step"], d["gpu_leg_device_ms"])
PY
