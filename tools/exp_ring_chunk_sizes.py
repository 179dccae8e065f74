"""The library's ring memory (sgk_ring_alloc) at other physical chunk sizes (SGK_RING_CHUNK_MIB, read per allocation): store-only
probe on fresh 100-slice rings of 1 M BoatRace envs, next to a torch.empty ring. One line per box."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "safe-grid-agents_amd")):
    sys.path.insert(0, p)
import safe_grid_agents_amd as S  # noqa: E402

env = S.BatchedGridworldEnv("BoatRace-v0", 1 << 20, seed=1)
out = []
for rep in range(2):
    for chunk in ("torch", "8", "32", "128", "256", "512"):
        if chunk == "torch":
            b, r, _ = env.alloc_trajectory_ring(100, backing="torch")
        else:
            os.environ["SGK_RING_CHUNK_MIB"] = chunk
            b, r, _ = env.alloc_trajectory_ring(100)
        out.append("%s %.2f" % (chunk, env.probe_trajectory_ring(b, r)))
        del b, r
print("probe us per step by chunk MiB: " + " | ".join(out), flush=True)
