"""When does each workgroup of ONE streamed-rollout launch start, reach its step loop, and end? (a library built with
-DSGK_DBG_TIMELINE=1, loaded through SGK_LIB_PATH, stamps wall_clock64() -- 100 MHz -- at the three points.) 1 M BoatRace envs,
100 steps into a 100-slice ring: the launch's fixed cost (88 us of a ~600 us launch: T(K) = f + b K over K = 100 ... 1000) is
start-up, rounds of workgroups, or the tail in which the last ones finish alone?"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "safe-grid-agents_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import safe_grid_agents_amd as S  # noqa: E402
from safe_grid_agents_amd import _lib  # noqa: E402

n = 1 << 20
env = S.BatchedGridworldEnv("BoatRace-v0", n, seed=1)
b, r, _ = env.alloc_trajectory_ring(100)
print("ring probe %.2f us per step" % env.probe_trajectory_ring(b, r))
lib = _lib.load()
n_wg = n // 256
buf = (ctypes.c_ulonglong * (3 * n_wg))()
for K in (100, 400):
    for rep in range(2):
        env.rollout_random_stream(K, boards=b, recs=r)
        env.synchronize()
        st = env.torch_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        env.rollout_random_stream(K, boards=b, recs=r)
        e1.record(st)
        env.synchronize()
        assert lib.sgk_debug_timeline(buf, 3 * n_wg) == 0
        t = np.frombuffer(buf, dtype=np.uint64).reshape(n_wg, 3).astype(np.float64) / 100.0  # us
        t0 = t[:, 0].min()
        start, loop, end = t[:, 0] - t0, t[:, 1] - t0, t[:, 2] - t0
        total = end.max()
        print("K = %d: event time %.1f us; first workgroup start -> last end %.1f us" % (K, e0.elapsed_time(e1) * 1e3, total))
        print("  start-up (start -> step loop) per workgroup: median %.1f us, p95 %.1f, max %.1f" % (
            np.median(loop - start), np.percentile(loop - start, 95), (loop - start).max()))
        print("  workgroup lifetime: median %.1f us, min %.1f, max %.1f" % (np.median(end - start), (end - start).min(), (end - start).max()))
        # how many workgroups are in their step loop at time x
        xs = np.linspace(0, total, 41)
        active = [(int(((loop <= x) & (end > x)).sum())) for x in xs]
        print("  in the step loop at t = 0 ... %.0f us in 40 intervals: %s" % (total, " ".join(str(a) for a in active)))
        order = np.sort(start)
        print("  workgroup starts: 1st %.1f, 1536th %.1f, 1537th %.1f, 3072nd %.1f, last %.1f us" % (
            order[0], order[min(1535, n_wg - 1)], order[min(1536, n_wg - 1)], order[min(3071, n_wg - 1)], order[-1]))
        wg = np.arange(n_wg)
        life = end - start
        print("  lifetime by XCD (workgroup b runs on XCD b % 8), median us: " + " ".join("%.0f" % np.median(life[wg % 8 == x]) for x in range(8)))
        print("  last end by XCD: " + " ".join("%.0f" % end[wg % 8 == x].max() for x in range(8)))
        first = start < 5.0  # the workgroups of the first round
        print("  first-round workgroups (%d): lifetime p5 %.0f, median %.0f, p95 %.0f us; by XCD median: %s" % (
            int(first.sum()), np.percentile(life[first], 5), np.median(life[first]), np.percentile(life[first], 95),
            " ".join("%.0f" % np.median(life[first & (wg % 8 == x)]) for x in range(8))))
        ends = np.sort(end)
        print("  workgroup ends: first %.1f, median %.1f, 95 %% %.1f, 99 %% %.1f, last %.1f us" % (
            ends[0], np.median(ends), np.percentile(ends, 95), np.percentile(ends, 99), ends[-1]))
env.close()
