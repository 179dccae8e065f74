"""Workload for rocprofv3 --pmc on the fused tabular-Q kernel (config 3 shape) and the step kernel."""
import os
import sys
import types

os.environ["SGK_NO_GRAPH"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "safe-grid-agents_amd")):
    sys.path.insert(0, p)
import safe_grid_agents_amd as S

n = 262144
args = types.SimpleNamespace(lr=0.5, discount=0.99, epsilon=0.01, epsilon_anneal=100000)
env = S.BatchedGridworldEnv("IslandNavigation-v0", n, seed=0x5AFE)
agent = S.BatchedTabularQAgent(env, args)
for _ in range(4):
    agent.rollout(500)
env.synchronize()
env2 = S.BatchedGridworldEnv("BoatRace-v0", 1 << 20, seed=0x5AFE, layout="compact")
for _ in range(20):
    env2.step_random(1, auto_reset=True)
env2.synchronize()
print("done")
