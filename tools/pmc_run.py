"""Workload for the rocprofv3 --pmc passes (run on the GPU box): the bench's kernels at the bench size, plus reset_kernel
launches whose HBM reads are a known byte count in the SAME access pattern (8 B/lane state words), used to calibrate
FETCH_SIZE as MI355X_MICROARCH.md prescribes.

    python3 tools/pmc_run.py <env> <layout> <n_envs> [launch|stream|ring|ringtile|fused] [ring slices]

launch: eager per-step launches (step_kernel);  stream: 100-step streaming rollout launches into the env's own buffers;
ring: the same into a trajectory ring of [ring slices] (default 100) slices; ringtile: the tile-major ring; fused: the
outputs-once rollout kernel, 1000 steps per launch."""
import os
import sys

os.environ["SGK_NO_GRAPH"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "safe-grid-agents_amd")):
    sys.path.insert(0, p)
import safe_grid_agents_amd as S

name = sys.argv[1] if len(sys.argv) > 1 else "BoatRace-v0"
layout = sys.argv[2] if len(sys.argv) > 2 else "compact"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 20
mode = sys.argv[4] if len(sys.argv) > 4 else "launch"
slices = int(sys.argv[5]) if len(sys.argv) > 5 else 100
env = S.BatchedGridworldEnv(name, n, seed=0x5AFE, layout=layout)


def work(reps):
    if mode == "launch":
        for _ in range(reps):
            env.step_random(1, auto_reset=True)
    elif mode == "stream":
        for _ in range(max(1, reps // 6)):
            env.step_random(100, auto_reset=True, fused="stream")
    elif mode == "fused":
        for _ in range(max(1, reps // 6)):
            env.step_random(1000, auto_reset=True, fused=True)
    else:
        import torch

        if mode == "ringtile":
            nt = (n + 63) // 64
            boards = torch.empty((nt, slices, 64, env.n_cells), dtype=torch.int8, device="cuda")
            recs = torch.empty((nt, slices, 64, 4), dtype=torch.int8, device="cuda")
        else:
            boards = torch.empty((slices, n, env.n_cells), dtype=torch.int8, device="cuda")
            recs = torch.empty((slices, n, 4), dtype=torch.int8, device="cuda")
        for _ in range(max(1, reps // 6)):
            env.rollout_random_stream(100, boards=boards, recs=recs, layout="tile" if mode == "ringtile" else "slice")
        env.synchronize()


work(30)
for _ in range(10):
    env.reset_done()  # reads n state words (8 B each), writes n boards; nothing is over -> no state writes
work(30)
env.synchronize()
print("done", name, layout, n, mode)
