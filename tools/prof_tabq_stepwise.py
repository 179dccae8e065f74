"""Workload for rocprofv3 --kernel-trace: the drop-in tabular-Q call sequence (act_explore / step / learn / reset_done) at the
config 3 shape (IslandNavigation, 262 144 private agents, float64 tables in HBM), replayed from the library's hipGraph
(sgk_tabq_learn_steps); `calls` as a second argument makes the four calls from Python instead."""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "safe-grid-agents_amd")):
    sys.path.insert(0, p)
import safe_grid_agents_amd as S

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
how = sys.argv[2] if len(sys.argv) > 2 else "graph"
args = types.SimpleNamespace(lr=0.5, discount=0.99, epsilon=0.01, epsilon_anneal=100000)
env = S.BatchedGridworldEnv("IslandNavigation-v0", n, seed=0x5AFE)
agent = S.BatchedTabularQAgent(env, args)
if how == "graph":
    for _ in range(5):
        agent.learn_steps(100)
else:
    for _ in range(300):
        a = agent.act_explore()
        env.step(a, auto_reset=False, write_boards=False)
        agent.learn(action=a)
        env.reset_done()
env.synchronize()
print("done")
