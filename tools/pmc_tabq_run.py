"""Workload for rocprofv3 (--kernel-trace / --pmc) on the fused tabular-Q rollout at BASELINE config 3's shape:
IslandNavigation, 262 144 private agents, LDS-resident tables (tabq_rollout_kernel).

    python3 tools/pmc_tabq_run.py [n_agents] [steps_per_launch] [launches] [kernel: auto|lds|hbm] [env]"""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "safe-grid-agents_amd")):
    sys.path.insert(0, p)
import safe_grid_agents_amd as S

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 500
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 4
kernel = sys.argv[4] if len(sys.argv) > 4 else "lds"
name = sys.argv[5] if len(sys.argv) > 5 else "IslandNavigation-v0"
args = types.SimpleNamespace(lr=0.5, discount=0.99, epsilon=0.01, epsilon_anneal=100000)
env = S.BatchedGridworldEnv(name, n, seed=0x5AFE)
agent = S.BatchedTabularQAgent(env, args)
for _ in range(launches):
    agent.rollout(steps, kernel=kernel)
env.synchronize()
print("done", name, n, steps, launches, kernel)
