"""Time the fused policy kernels (sgk_policy_act / sgk_policy_sample) at a few env counts with torch events (includes the
host-side launch cost; tools/prof_policy.py under rocprofv3 --kernel-trace gives the pure kernel durations).
Run on the GPU box: python tools/bench_policy.py"""
import sys, os, types
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "safe-grid-agents_amd"))
import torch
import safe_grid_agents_amd as S

a = types.SimpleNamespace(discount=0.99, lr=1e-3, batch_size=64, sync_every=20, epsilon=0.05, epsilon_anneal=200, n_layers=2, n_hidden=100)
for name in ("SideEffectsSokoban-v0", "BoatRace-v0", "IslandNavigation-v0"):
    for n in (4096, 32768, 262144, 1048576):
        env = S.BatchedGridworldEnv(name, n, seed=1)
        env.bind_torch_stream()
        env.step_random(7)
        agent = S.BatchedDeepQAgent(env, a)
        agent._refresh_fused_weights()
        out = torch.empty(n, dtype=torch.uint8, device="cuda")
        for mode in ("act", "sample"):
            f = (lambda: env.policy_act(agent._fw, 0.1, 3, out=out)) if mode == "act" else (lambda: env.policy_sample(agent._fw, 3, out=out))
            for _ in range(5):
                f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                f()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 50
            flops = 2.0 * n * (env.n_cells * 100 + 100 * 100 + 100 * 4)
            print(f"{name} n={n} {mode}: {us:.1f} us  {flops / us * 1e-6:.1f} TFLOP/s f32", flush=True)
        env.close()
