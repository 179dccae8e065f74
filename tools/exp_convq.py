"""The fused conv Q-body kernel (sgk_convq_act) timed per launch: levels x channel counts at 32 768 envs, and the lockstep step around
it (eager three launches / graph) on Sokoban. `python tools/exp_convq.py [tag]`."""
import os
import sys
import time
import types

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "safe-grid-agents_amd"))
import torch  # noqa: E402

import safe_grid_agents_amd as S  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else ""


def wall(fn, k=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e6


def args(c):
    return types.SimpleNamespace(discount=0.99, lr=1e-3, batch_size=64, sync_every=20, epsilon=0.05, epsilon_anneal=200, n_layers=2,
                                 n_hidden=100, n_channels=c)


for name, n in (("SideEffectsSokoban-v0", 32768), ("SideEffectsSokoban-v0", 1 << 20), ("BoatRace-v0", 32768), ("IslandNavigation-v0", 32768),
                ("DistributionalShift-v0", 32768)):
    row = []
    for c in (4, 5, 8):
        env = S.BatchedGridworldEnv(name, n, seed=1)
        env.step_random(7)
        agent = S.BatchedDeepQAgent(env, args(c), q_body="cnn")
        us = wall(lambda: agent._conv_act(0.1, 3))
        flops = 2.0 * env.n_cells * (9 * c + 2 * 9 * c * c + c + 4 * c)
        row.append("C=%d %7.2f us (%5.1f TFLOP/s useful)" % (c, us, n * flops / us / 1e6))
        if n == 32768:  # the same forward + draw + env.step, 100 lockstep steps in ONE launch (sgk_convq_rollout)
            ro = wall(lambda: env.convq_rollout(agent._cw, 100, c, mode="greedy", epsilon=0.1, draw_index0=0, auto_reset=True), 10, 2) / 100
            row.append("[rollout %6.2f us/step]" % ro)
        if name.startswith("Side") and c == 5 and n == 32768:
            step = wall(lambda: agent.step(learn=False), 100, 10)
            agent.enable_graphs(learn=False)
            graph = wall(lambda: agent.step_graphed(learn=False), 100, 10)
            row.append("| lockstep step eager %6.2f us, graph %6.2f us" % (step, graph))
        env.close()
    print("%s %-24s n=%8d  %s" % (tag, name, n, "  ".join(row)), flush=True)
