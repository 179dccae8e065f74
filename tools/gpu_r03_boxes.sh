#!/bin/bash
# one box's sample for the ring-backing statistics: bench at the driver's flags, primary ring from sgk_ring_alloc and from torch.empty
cd "$GRAFT_REPO_ROOT" || exit 1
export SGK_NO_BUILD=1 TMPDIR=/tmp
mkdir -p gpurun_out/boxes
for b in ring ring torch; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fused --ring-backing $b 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('backing=$b value %.3e device %.2f us per step frac %.3f | plain rings %s | in place %.2f launches %.2f' % (d['value'], d['roofline']['device_us_per_step'], d['roofline']['frac'], [round(x,2) for x in d['other_ring_allocations']['device_us_per_lockstep_step']], d['rewritten_in_place']['device_us_per_lockstep_step'], d['per_step_launches']['device_us_per_lockstep_step']))"
done | tee -a gpurun_out/boxes/samples.log
