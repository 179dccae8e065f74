"""Trajectory-ring streaming taken apart: both rings / boards ring only / records ring only / the env's own buffers, per lockstep
step at 1 M envs (the ring rate is the DRAM write pattern's, not the split between the two streams)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "safe-grid-agents_amd")):
    sys.path.insert(0, p)
import torch
import safe_grid_agents_amd as S
def timed(env, fn, reps):
    st = env.torch_stream(); fn(); env.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps): fn()
    e1.record(st); env.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for name in ("BoatRace-v0", "IslandNavigation-v0"):
    n = 1 << 20; K = 100
    env = S.BatchedGridworldEnv(name, n, seed=1)
    boards = torch.empty((K, n, env.n_cells), dtype=torch.int8, device="cuda")
    recs = torch.empty((K, n, 4), dtype=torch.int8, device="cuda")
    for rep in range(2):
        both = timed(env, lambda: env.rollout_random_stream(K, boards=boards, recs=recs), 10) / K
        bo = timed(env, lambda: env.rollout_random_stream(K, boards=boards), 10) / K
        ro = timed(env, lambda: env.rollout_random_stream(K, recs=recs), 10) / K
        own = timed(env, lambda: env.step_random(K, fused="stream"), 10) / K
        nt = (n + 63) // 64
        tb = torch.empty((nt, K, 64, env.n_cells), dtype=torch.int8, device="cuda")
        tr = torch.empty((nt, K, 64, 4), dtype=torch.int8, device="cuda")
        tile = timed(env, lambda: env.rollout_random_stream(K, boards=tb, recs=tr, layout="tile"), 10) / K
        del tb, tr
        nb = env.n_cells * n / 1e6
        print("%s ring both %.2f us (%.2f TB/s) | boards only %.2f us (%.2f TB/s) | recs only %.2f us | own %.2f us | TILE-MAJOR ring both %.2f us (%.2f TB/s)" % (name, both, (nb + 4.19) / both, bo, nb / bo, ro, own, tile, (nb + 4.19) / tile), flush=True)
    env.close()
