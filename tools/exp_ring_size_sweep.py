"""Ring-size sweep of the streamed rollout: 1 M BoatRace envs, 100 steps per launch into trajectory rings
of 1 ... 100 slices (30 MB ... 3 GB). Device time per lockstep step; the knee near 256 MiB is where the ring stops fitting the
Infinity Cache. PMC bytes for the same sizes come from tools/gpu_r03_a.sh (one rocprofv3 --pmc WRITE_SIZE pass per size)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "safe-grid-agents_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import safe_grid_agents_amd as S  # noqa: E402


def timed(env, fn, reps):
    st = env.torch_stream()
    fn()
    env.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        fn()
    e1.record(st)
    env.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


name = sys.argv[1] if len(sys.argv) > 1 else "BoatRace-v0"
n, K = 1 << 20, 100
env = S.BatchedGridworldEnv(name, n, seed=1)
nt = (n + 63) // 64
own = timed(env, lambda: env.step_random(K, fused="stream"), 10) / K
print("%s n=%d own buffers (26 MB rewritten in place): %.2f us per step" % (name, n, own), flush=True)
for rep in range(2):
    for slices in (1, 2, 4, 8, 16, 32, 64, 100):
        for layout in ("slice", "tile"):
            if layout == "tile":
                b = torch.empty((nt, slices, 64, env.n_cells), dtype=torch.int8, device="cuda")
                r = torch.empty((nt, slices, 64, 4), dtype=torch.int8, device="cuda")
            else:
                b = torch.empty((slices, n, env.n_cells), dtype=torch.int8, device="cuda")
                r = torch.empty((slices, n, 4), dtype=torch.int8, device="cuda")
            us = timed(env, lambda: env.rollout_random_stream(K, boards=b, recs=r, layout=layout), 10) / K
            mb = (b.numel() + r.numel()) / 1e6
            print("ring %3d slices (%7.1f MB, %5.1f MiB) %-5s: %.2f us per step = %.2f TB/s of kept outputs" % (
                slices, mb, mb * 1e6 / 2**20, layout, us, n * (env.n_cells + 4) / us / 1e6), flush=True)
            del b, r
env.close()
