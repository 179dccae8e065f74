"""What BatchedGridworldEnv.alloc_trajectory_ring finds: 1 M BoatRace envs, 100-slice rings, 8 candidates back to back and with
16 / 24 GiB spacers between them; the streamed kernel timed on the chosen ring and on a plain one."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "safe-grid-agents_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import safe_grid_agents_amd as S  # noqa: E402


def kernel_us(env, b, r, launches=3):
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


n = 1 << 20
env = S.BatchedGridworldEnv("BoatRace-v0", n, seed=1)
if "k" in sys.argv:  # launch length: fixed costs per launch (workgroup start-up, final flush) against per-step costs
    st = env.torch_stream()
    vb, vr, _ = env.alloc_trajectory_ring(100)
    print("sgk_ring_alloc ring: probe %.2f" % env.probe_trajectory_ring(vb, vr), flush=True)
    for rep in range(3):
        if rep == 2:
            env.bind_torch_stream()  # the library's kernels on torch's current stream: no cross-stream events around each call
            st = torch.cuda.current_stream()
            print("  -- bound to torch's current stream", flush=True)
        for K in (100, 200, 400, 1000):
            env.rollout_random_stream(K, boards=vb, recs=vr)
            env.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            launches = max(2, 600 // K)
            for _ in range(launches):
                env.rollout_random_stream(K, boards=vb, recs=vr)
            e1.record(st)
            env.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / launches
            print("  K = %4d steps per launch: %.1f us per launch = %.3f us per step" % (K, us, us / K), flush=True)
    sys.exit(0)
if "nt" in sys.argv:  # on rings from the library's allocator: board stores non-temporal write-through (default above 1.5 GB) or write-through
    for trial in range(3):
        vb, vr, _ = env.alloc_trajectory_ring(100)
        line = "sgk_ring_alloc ring %d: probe %.2f |" % (trial, env.probe_trajectory_ring(vb, vr))
        for nt in ("1", "0", "1", "0"):
            os.environ["SGK_RING_NT"] = nt
            line += " nt=%s kernel %.2f" % (nt, kernel_us(env, vb, vr, 5))
        print(line, flush=True)
        del vb, vr
    os.environ.pop("SGK_RING_NT")
    sys.exit(0)
for trial in range(4):  # the library's ring allocator (HIP VMM, 256 MiB chunks): probe and kernel
    vb, vr, _ = env.alloc_trajectory_ring(100)
    print("sgk_ring_alloc ring %d: probe %.2f, kernel %.2f us per step" % (trial, env.probe_trajectory_ring(vb, vr), kernel_us(env, vb, vr, 5)), flush=True)
    del vb, vr
if os.environ.get("ONLY_AB"):
    sys.argv.append("ab")
pb = torch.empty((100, n, env.n_cells), dtype=torch.int8, device="cuda")
pr = torch.empty((100, n, 4), dtype=torch.int8, device="cuda")
print("plain ring: probe %.2f, kernel %.2f us per step" % (env.probe_trajectory_ring(pb, pr), kernel_us(env, pb, pr)), flush=True)
for spread in (() if "ab" in sys.argv else (0, 16, 24)):
    for cand in (8, 12):
        b, r, info = env.alloc_trajectory_ring(100, candidates=cand, spread_gib=spread, backing="torch")
        print("candidates %2d, spacers %2d GiB: probes %s -> chosen %d: kernel %.2f us per step" % (
            cand, spread, " ".join("%.2f" % u for u in info["candidates_us"]), info["chosen"], kernel_us(env, b, r)), flush=True)
        del b, r
        torch.cuda.empty_cache()
env.close()

# ---- on ONE fast ring: does the residency of the streamed kernel matter? (SGK_STREAM_RESIDENT is read per launch)
env = S.BatchedGridworldEnv("BoatRace-v0", n, seed=1)
b, r, info = env.alloc_trajectory_ring(100, candidates=8, spread_gib=16, backing="torch")
print("ring for the residency A/B: probe %.2f us per step (candidates %s)" % (
    info["candidates_us"][info["chosen"]], " ".join("%.2f" % u for u in info["candidates_us"])), flush=True)
for rep in range(2):
    for res in ("-1", "5", "4", "3", "2", "0"):
        os.environ["SGK_STREAM_RESIDENT"] = res
        print("  SGK_STREAM_RESIDENT=%2s: kernel %.2f us per step into the ring, %.2f in place" % (
            res, kernel_us(env, b, r, 5), kernel_us(env, None, None, 5)), flush=True)
print("  the probe again: %.2f" % env.probe_trajectory_ring(b, r))
env.close()
