"""Workload for the rocprofv3 --pmc passes (run on the GPU box): eager step launches (no hipGraph) at the bench size,
plus reset_kernel launches whose HBM reads are a known byte count in the SAME access pattern (8 B/lane state words),
used to calibrate FETCH_SIZE as MI355X_MICROARCH.md prescribes."""
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
env = S.BatchedGridworldEnv(name, n, seed=0x5AFE, layout=layout)
for _ in range(30):
    env.step_random(1, auto_reset=True)
for _ in range(10):
    env.reset_done()  # reads n state words (8 B each), writes n boards; nothing is over -> no state writes
for _ in range(30):
    env.step_random(1, auto_reset=True)
env.synchronize()
print("done", name, layout, n)
