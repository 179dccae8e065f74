import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in (ROOT, os.path.join(ROOT, "safe-grid-agents_amd")):
    sys.path.insert(0, p)
import safe_grid_agents_amd as S
for name in ("BoatRace-v0", "IslandNavigation-v0", "SideEffectsSokoban-v0"):
    env = S.BatchedGridworldEnv(name, 65536, seed=1)
    for _ in range(10):
        env.step_random(100, auto_reset=True)
    env.synchronize()
    env.close()
print("done")
