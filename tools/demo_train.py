"""Batched tabular-Q training demo (runs on the GPU box): N private agents per env, reference hyper-parameters except a
shorter epsilon anneal; prints the aggregate meters the reference would plot (returns / safeties / margins)."""
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "safe-grid-agents_amd")):
    sys.path.insert(0, p)
import safe_grid_agents_amd as S

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
names = [a for a in sys.argv[2:] if a.endswith("-v0")] or ["BoatRace-v0", "IslandNavigation-v0"]
cheat = "--cheat" in sys.argv  # learn from the hidden reward and the executed action (reference learn.py:72-79)
for name, episodes in [(nm, 300) for nm in names]:
    args = types.SimpleNamespace(lr=0.5, discount=0.99, epsilon=0.01, epsilon_anneal=20000)
    env = S.BatchedGridworldEnv(name, n, seed=1)
    agent = S.BatchedTabularQAgent(env, args)
    t0 = time.perf_counter()
    for ep in range(episodes):
        env.metrics_reset()
        agent.rollout(100, cheat=cheat)
        if ep % 50 == 49 or ep == 0:
            bm = S.BatchMetrics(env.metrics())
            ev = S.batched_default_eval(agent, env, 200)
            env.reset()
            print(json.dumps({"env": name, "cheat": cheat, "agents": n, "episode_x100steps": ep + 1, "epsilon": round(agent.epsilon, 4),
                              "train_return": round(bm.meter("returns")["avg"], 2), "train_safety": round(bm.meter("safeties")["avg"], 2),
                              "eval_return": round(ev.meter("returns")["avg"], 2), "eval_safety": round(ev.meter("safeties")["avg"], 2),
                              "eval_margin": round(ev.meter("margins")["avg"], 2)}), flush=True)
    env.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"env": name, "wall_s": round(dt, 2), "agent_steps": n * episodes * 100,
                      "agent_steps_per_s_incl_evals": n * episodes * 100 / dt}), flush=True)
    agent.close(); env.close()
