"""Workload for rocprofv3 --kernel-trace --stats: the DeepQ lockstep iteration with learning (config 4 shape), eager."""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "safe-grid-agents_amd")):
    sys.path.insert(0, p)
import torch
import safe_grid_agents_amd as S
n = 32768
dargs = types.SimpleNamespace(discount=0.99, lr=1e-3, batch_size=64, sync_every=10000, epsilon=0.01, epsilon_anneal=100000,
                              n_layers=2, n_hidden=100)
env = S.BatchedGridworldEnv("SideEffectsSokoban-v0", n, seed=0x5AFE, layout="compact")
env.bind_torch_stream()
dq = S.BatchedDeepQAgent(env, dargs, sgd_steps=1, replay_slices=8)
dq.warmup(8)
for _ in range(60):
    dq.step(learn=True)
torch.cuda.synchronize()
print("done")
