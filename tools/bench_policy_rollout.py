"""sgk_policy_rollout (DeepQ acting, frozen weights) per env count: is a second wave per SIMD free? (Sokoban, MLP 36-100-100-4)"""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "safe-grid-agents_amd")):
    sys.path.insert(0, p)
import torch
import safe_grid_agents_amd as S

a = types.SimpleNamespace(discount=0.99, lr=1e-3, batch_size=64, sync_every=10000, epsilon=0.01, epsilon_anneal=100000, n_layers=2, n_hidden=100)
for n in [int(x) for x in (sys.argv[1:] or ["16384", "32768", "65536", "131072", "262144", "1048576"])]:
    env = S.BatchedGridworldEnv("SideEffectsSokoban-v0", n, seed=0x5AFE, layout="compact")
    env.bind_torch_stream()
    dq = S.BatchedDeepQAgent(env, a, sgd_steps=1, replay_slices=2)
    dq.act_rollout(50, epsilon=0.01)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dq.act_rollout(300, epsilon=0.01)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 300
    print("n=%d: %.2f us per lockstep step = %.3e env-steps/s = %.1f TFLOP/s useful" % (n, dt * 1e6, n / dt, 28000.0 * n / dt / 1e12), flush=True)
    env.close()
