"""Workload for rocprofv3 --kernel-trace: the fused policy kernel at three env counts (Sokoban, 20 launches each)."""
import sys, os, types
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "safe-grid-agents_amd"))
import torch
import safe_grid_agents_amd as S

a = types.SimpleNamespace(discount=0.99, lr=1e-3, batch_size=64, sync_every=20, epsilon=0.05, epsilon_anneal=200, n_layers=2, n_hidden=100)
for n in (4096, 32768, 1048576):
    env = S.BatchedGridworldEnv("SideEffectsSokoban-v0", n, seed=1)
    env.bind_torch_stream()
    env.step_random(7)
    agent = S.BatchedDeepQAgent(env, a)
    agent._refresh_fused_weights()
    out = torch.empty(n, dtype=torch.uint8, device="cuda")
    for _ in range(20):
        env.policy_act(agent._fw, 0.1, 3, out=out)
    torch.cuda.synchronize()
    env.close()
