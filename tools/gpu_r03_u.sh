#!/bin/bash
# round 3, call u: the probing ring allocator: tests, the bench line at the driver's flags (three processes), plain for comparison
cd "$GRAFT_REPO_ROOT" || exit 1
export SGK_NO_BUILD=1 TMPDIR=/tmp
O=gpurun_out/u; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_bench_contract.py -x -q -k "probed or bench or ring or tile_stores" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for i in 1 2 3; do timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$i.json 2> $O/bench_$i.err; done
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --ring-candidates 1 > $O/bench_plain.json 2> $O/bench_plain.err
python - <<'PY'
import json
for f in ("bench_1","bench_2","bench_3","bench_plain"):
    d=json.loads(open("gpurun_out/u/%s.json"%f).read().strip().splitlines()[-1])
    print(f, "%.3e"%d["value"], round(d["roofline"]["device_us_per_step"],2), round(d["roofline"]["frac"],3), [round(x,2) for x in d["ring_allocation"]["candidates_probe_us_per_slice"]], d["ring_allocation"]["chosen"], [round(x,2) for x in d["other_ring_allocations"]["device_us_per_lockstep_step"]])
PY
