#!/bin/bash
# round 3, call y: rings from the library's VMM-backed allocator: tests, allocator tool, bench at the driver's flags x3
cd "$GRAFT_REPO_ROOT" || exit 1
export SGK_NO_BUILD=1 TMPDIR=/tmp
O=gpurun_out/y; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_bench_contract.py -x -q -k "probed or bench or ring or tile_stores" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
ONLY_AB=1 timeout 300 python tools/exp_ring_allocator.py 2>&1 | grep -v amdgpu > $O/ring_allocator_vmm.log; head -8 $O/ring_allocator_vmm.log
for i in 1 2 3; do timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$i.json 2> $O/bench_$i.err; done
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --ring-backing torch > $O/bench_torch.json 2> $O/bench_torch.err
python - <<'PY'
import json
for f in ("bench_1","bench_2","bench_3","bench_torch"):
    d=json.loads(open("gpurun_out/y/%s.json"%f).read().strip().splitlines()[-1])
    print(f, "%.3e"%d["value"], round(d["roofline"]["device_us_per_step"],2), round(d["roofline"]["frac"],3), d["ring_allocation"]["backing"][:14], [round(x,2) for x in d["other_ring_allocations"]["device_us_per_lockstep_step"]], d["parity_sample_bit_exact"], d["ring_slices_checked_bit_exact"])
PY
