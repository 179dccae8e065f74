"""Workload for rocprofv3 --kernel-trace --stats: PPO iterations (gather one rollout, learn 16 epochs x 64 rows, sync) with
the default ppo-mlp topology: sgk_policy_rollout + post-processing + ONE sgk_ppo_epochs launch per iteration."""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "safe-grid-agents_amd")):
    sys.path.insert(0, p)
import torch
import safe_grid_agents_amd as S
name = sys.argv[1] if len(sys.argv) > 1 else "BoatRace-v0"
env = S.BatchedGridworldEnv(name, 32768, seed=5)
env.bind_torch_stream()
a = types.SimpleNamespace(discount=0.99, lr=1e-3, batch_size=64, rollouts=1, epochs=16, clipping=0.2, entropy_bonus=0.01,
                          critic_coeff=1.0, n_layers=2, n_hidden=100, n_channels=5, device=0, log_gradients=False, cheat=False)
agent = S.BatchedPPOAgent(env, a)
for _ in range(30):
    ro = agent.gather_rollout()
    agent.learn(ro)
    agent.sync()
torch.cuda.synchronize()
print("done")
