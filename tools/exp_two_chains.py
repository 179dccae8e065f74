"""Does splitting the batch into independent launch chains on separate streams hide a streamed launch's fixed cost (launch
boundary, start-up, the draining tail: ~88 us of a ~550 us launch at 1 M envs)? P env objects of 1 M / P BoatRace
envs each, every one with its own stream and its own 100-slice ring, launches enqueued round-robin with no cross-stream events;
against one env object of 1 M. Device time until ALL chains have finished, per lockstep step of the whole 1 M batch."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "safe-grid-agents_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import safe_grid_agents_amd as S  # noqa: E402

N, K, LAUNCHES = 1 << 20, 100, 12
backing = sys.argv[1] if len(sys.argv) > 1 else "ring"
for P in (1, 2, 4, 1, 2, 4, 8):
    envs, rings, streams = [], [], []
    for p in range(P):
        e = S.BatchedGridworldEnv("BoatRace-v0", N // P, seed=1, env_index_base=p * (N // P))
        st = torch.cuda.Stream()
        e.bind_torch_stream(st)
        b, r, _ = e.alloc_trajectory_ring(100, backing=backing)
        envs.append(e); rings.append((b, r)); streams.append(st)
    for e, (b, r), st in zip(envs, rings, streams):
        with torch.cuda.stream(st):
            e.rollout_random_stream(K, boards=b, recs=r)
    torch.cuda.synchronize()
    start = [torch.cuda.Event(enable_timing=True) for _ in range(P)]
    stop = [torch.cuda.Event(enable_timing=True) for _ in range(P)]
    t0 = time.perf_counter()
    for p in range(P):
        start[p].record(streams[p])
    for _ in range(LAUNCHES):
        for e, (b, r), st in zip(envs, rings, streams):
            with torch.cuda.stream(st):
                e.rollout_random_stream(K, boards=b, recs=r)
    for p in range(P):
        stop[p].record(streams[p])
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    per_chain = [start[p].elapsed_time(stop[p]) * 1e3 / (K * LAUNCHES) for p in range(P)]
    print("%d chain(s) of %7d envs: wall %.3f us per lockstep step of the whole batch; per chain %s" % (
        P, N // P, wall * 1e6 / (K * LAUNCHES), " ".join("%.2f" % u for u in per_chain)), flush=True)
    for e in envs:
        e.close()
    del envs, rings, streams
    torch.cuda.empty_cache()
