"""Workload for rocprofv3 (--kernel-trace / --pmc): BASELINE config 4's acting kernel -- sgk_policy_rollout, SideEffectsSokoban, 32 768
envs, the reference's MLP (36-100-100-4), 1000 steps per launch, a few launches."""
import os
import sys
import types

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "safe-grid-agents_amd"))
import torch

import safe_grid_agents_amd as S

a = types.SimpleNamespace(discount=0.99, lr=1e-3, batch_size=64, sync_every=10000, epsilon=0.01, epsilon_anneal=100000, n_layers=2,
                          n_hidden=100)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
env = S.BatchedGridworldEnv("SideEffectsSokoban-v0", n, seed=0x5AFE, layout="compact")
env.bind_torch_stream()
dq = S.BatchedDeepQAgent(env, a, sgd_steps=1, replay_slices=8)
dq.warmup(8)
for _ in range(4):
    dq.act_rollout(1000, epsilon=0.01)
torch.cuda.synchronize()
print("done", n)
