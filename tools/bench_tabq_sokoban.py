"""Sokoban + tabular-Q (tables too large for LDS): the fused HBM-resident rollout vs the four-launch drop-in sequence."""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "safe-grid-agents_amd")):
    sys.path.insert(0, p)
import safe_grid_agents_amd as S

for n in (65536, 262144):
    args = types.SimpleNamespace(lr=0.5, discount=0.99, epsilon=0.01, epsilon_anneal=100000)
    env = S.BatchedGridworldEnv("SideEffectsSokoban-v0", n, seed=0x5AFE)
    agent = S.BatchedTabularQAgent(env, args)
    agent.rollout(200, cheat=True)
    env.synchronize()
    t0 = time.perf_counter()
    agent.rollout(1000, cheat=True)
    env.synchronize()
    dt = (time.perf_counter() - t0) / 1000

    def stepwise():
        a = agent.act_explore()
        env.step(a, auto_reset=False, write_boards=False)
        agent.learn(action=a, cheat=True)
        env.reset_done()

    for _ in range(20):
        stepwise()
    env.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        stepwise()
    env.synchronize()
    ds = (time.perf_counter() - t0) / 200
    print(f"Sokoban tabular-q n={n}: fused HBM-resident rollout {dt * 1e6:.1f} us/step = {n / dt:.3e} agent-steps/s; "
          f"four launches per step {ds * 1e6:.1f} us/step = {n / ds:.3e}", flush=True)
    agent.close(); env.close()
