#!/bin/bash
# round 3, call aa: tile groups by ticket (persistent resident-sized grid) in the streamed rollout: A/B, tests, bench
cd "$GRAFT_REPO_ROOT" || exit 1
export SGK_NO_BUILD=1 TMPDIR=/tmp
O=gpurun_out/aa; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_sizes.py -x -q -k "stream or ring or tile or probed" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
cat > /tmp/ab.py <<'PY'
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "safe-grid-agents_amd")
import torch
import safe_grid_agents_amd as S
def kernel_us(env, b, r, launches=5):
    st = env.torch_stream()
    (env.rollout_random_stream(100, boards=b, recs=r) if b is not None else env.step_random(100, fused="stream"))
    env.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(launches):
        (env.rollout_random_stream(100, boards=b, recs=r) if b is not None else env.step_random(100, fused="stream"))
    e1.record(st)
    env.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (100 * launches)
for name, n in (("BoatRace-v0", 1 << 20), ("BoatRace-v0", 1 << 19), ("IslandNavigation-v0", 1 << 20), ("SideEffectsSokoban-v0", 1 << 20), ("TomatoWatering-v0", 1 << 20)):
    env = S.BatchedGridworldEnv(name, n, seed=1)
    b, r, _ = env.alloc_trajectory_ring(100)
    line = "%s n=%d ring probe %.2f |" % (name, n, env.probe_trajectory_ring(b, r))
    for rep in range(2):
        for tk in ("0", "1"):
            os.environ["SGK_STREAM_TICKETS"] = tk
            line += " tickets=%s ring %.2f own %.2f |" % (tk, kernel_us(env, b, r), kernel_us(env, None, None))
    print(line, flush=True)
    del b, r
    env.close()
PY
timeout 600 python /tmp/ab.py 2>&1 | grep -v amdgpu > $O/tickets_ab.log; cat $O/tickets_ab.log
SGK_LIB_PATH=$PWD/safe-grid-agents_amd/lib/libsgk_timeline.so timeout 300 python tools/exp_stream_timeline.py 2>&1 | grep -v amdgpu > $O/stream_timeline_tickets.log; grep -A9 "K = 100" $O/stream_timeline_tickets.log | head -11
for i in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fused 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3e'%d['value'], round(d['roofline']['device_us_per_step'],2), round(d['roofline']['frac'],3), [round(x,2) for x in d['other_ring_allocations']['device_us_per_lockstep_step']], round(d['rewritten_in_place']['device_us_per_lockstep_step'],2), d['parity_sample_bit_exact'], d['ring_slices_checked_bit_exact'])"; done
