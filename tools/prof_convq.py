"""Workload for rocprofv3: the fused conv Q-body kernel (sgk_convq_act) at two env counts (Sokoban, 5 channels, 20 launches each)."""
import os
import sys
import types

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "safe-grid-agents_amd"))
import torch  # noqa: E402

import safe_grid_agents_amd as S  # noqa: E402

a = types.SimpleNamespace(discount=0.99, lr=1e-3, batch_size=64, sync_every=20, epsilon=0.05, epsilon_anneal=200, n_layers=2, n_hidden=100,
                          n_channels=int(os.environ.get("CONVQ_CHANNELS", "5")))
for n in (32768, 1048576):
    env = S.BatchedGridworldEnv(os.environ.get("CONVQ_ENV", "SideEffectsSokoban-v0"), n, seed=1)
    env.step_random(7)
    agent = S.BatchedDeepQAgent(env, a, q_body="cnn")
    assert agent.fused_conv
    for _ in range(20):
        agent._conv_act(0.1, 3)
    torch.cuda.synchronize()
    env.close()
