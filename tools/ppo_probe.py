"""Scratch probe: does the batched PPO improve the BoatRace return? (not part of the product or the tests)"""
import sys, os, types
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "safe-grid-agents_amd"))
import torch
import safe_grid_agents_amd as S

def run(lr, batch, epochs, iters, eb, n=2048, seed=0, cc=1.0):
    torch.manual_seed(seed)
    env = S.BatchedGridworldEnv("BoatRace-v0", n, seed=5)
    env.bind_torch_stream()
    a = types.SimpleNamespace(discount=0.99, lr=lr, batch_size=batch, rollouts=1, epochs=epochs, clipping=0.2, entropy_bonus=eb,
                              critic_coeff=cc, n_layers=2, n_hidden=100, n_channels=5, device=0, log_gradients=False, cheat=False)
    agent = S.BatchedPPOAgent(env, a)
    out = []
    for i in range(iters):
        bm = S.batched_ppo_learn(agent, env, None)
        out.append(round(bm.meter("returns")["avg"], 1))
    print("lr", lr, "batch", batch, "epochs", epochs, "eb", eb, "cc", cc, "seed", seed, "->", out[::max(1, iters // 12)], out[-1], flush=True)
    env.close()

for seed in (0, 1):
    run(1e-3, 4096, 16, 30, 0.01, cc=0.01, seed=seed)
    run(3e-3, 4096, 16, 30, 0.01, cc=0.001, seed=seed)
    run(1e-3, 4096, 16, 30, 0.0, cc=0.0001, seed=seed)
