#!/usr/bin/env python3
"""Per-lockstep-step device time of the three random-rollout forms at several batch sizes: one launch per step (hipGraph),
the streaming rollout (one launch per K steps, every step's boards + records materialised) and the fused rollout (outputs
once per launch). HIP events on the library's stream; prints one line per (env, n)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "safe-grid-agents_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import safe_grid_agents_amd as S  # noqa: E402


def timed(env, fn, reps):
    st = env.torch_stream()
    # (the env's stream is torch's current stream while timing: otherwise the wrapper's cross-stream event hops leave ~25 us
    # of idle GPU between back-to-back launches -- a third of a 100-step launch at 65 536 envs)
    with torch.cuda.stream(st):
        fn()
        env.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps):
            fn()
        e1.record(st)
        env.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps  # us per call


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", default="BoatRace-v0")
    ap.add_argument("--sizes", default="65536,131072,262144,1048576")
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--ring", type=int, default=0, help="also stream into a trajectory ring of this many slices")
    ap.add_argument("--modes", default="launch,stream,fused", help="which forms to time (ring is added by --ring)")
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    modes = args.modes.split(",")
    for name in args.envs.split(","):
        for n in (int(x) for x in args.sizes.split(",")):
            env = S.BatchedGridworldEnv(name, n, seed=1)
            K = args.k
            line = "%s n=%d:" % (name, n)
            if "launch" in modes:
                g = timed(env, lambda: env.step_random(K, auto_reset=True), args.reps) / K
                line += " per-step launches %.2f us/step |" % g
            if "stream" in modes:
                s = timed(env, lambda: env.step_random(K, auto_reset=True, fused="stream"), args.reps) / K
                line += " streamed %.2f us/step (%.3g env-steps/s) |" % (s, n / s * 1e6)
            if "fused" in modes:
                f = timed(env, lambda: env.step_random(K, auto_reset=True, fused=True), args.reps) / K
                line += " fused %.3f us/step" % f
            if args.ring:
                boards = torch.empty((args.ring, n, env.n_cells), dtype=torch.int8, device="cuda")
                recs = torch.empty((args.ring, n, 4), dtype=torch.int8, device="cuda")
                r = timed(env, lambda: env.rollout_random_stream(K, boards=boards, recs=recs), args.reps) / K
                line += " | streamed into a %d-slice ring %.2f us/step" % (args.ring, r)
            print(line, flush=True)
            env.close()


if __name__ == "__main__":
    main()
