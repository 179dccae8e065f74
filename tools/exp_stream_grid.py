"""Streamed rollout at 1 M BoatRace envs: device time per lockstep step by destination (own buffers / 32-slice ring / 100-slice
ring), for the library and SGK_STREAM_GRID in force (tools/gpu_r03_b.sh loops over builds with different __launch_bounds__ and
over grids); and with the 100 steps of a pass issued as several shorter launches (--chunks)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "safe-grid-agents_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import safe_grid_agents_amd as S  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--env", default="BoatRace-v0")
ap.add_argument("--n", type=int, default=1 << 20)
ap.add_argument("--rings", default="32,100")
ap.add_argument("--chunks", default="100")
ap.add_argument("--layout", default="slice")
ap.add_argument("--reps", type=int, default=10)
args = ap.parse_args()


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


n, K = args.n, 100
env = S.BatchedGridworldEnv(args.env, n, seed=1)
nt = (n + 63) // 64
tag = "lib=%s grid=%s" % (os.path.basename(os.environ.get("SGK_LIB_PATH", "libsgk.so")), os.environ.get("SGK_STREAM_GRID", "default"))
line = "%s %s n=%d: own %.2f" % (tag, args.env, n, timed(env, lambda: env.step_random(K, fused="stream"), args.reps) / K)
for slices in (int(x) for x in args.rings.split(",")):
    if args.layout == "tile":
        b = torch.empty((nt, slices, 64, env.n_cells), dtype=torch.int8, device="cuda")
        r = torch.empty((nt, slices, 64, 4), dtype=torch.int8, device="cuda")
    else:
        b = torch.empty((slices, n, env.n_cells), dtype=torch.int8, device="cuda")
        r = torch.empty((slices, n, 4), dtype=torch.int8, device="cuda")
    for ck in (int(x) for x in args.chunks.split(",")):
        def go():
            for c in range(0, K, ck):
                env.rollout_random_stream(ck, boards=b, recs=r, first_slice=c % slices, layout=args.layout)
        us = timed(env, go, args.reps) / K
        line += " | ring %d%s %.2f us (%.2f TB/s)" % (slices, "" if ck == K else " in %d-step launches" % ck, us, n * (env.n_cells + 4) / us / 1e6)
    del b, r
print(line, flush=True)
env.close()
