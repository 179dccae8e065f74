"""sgk_dqn_sgd_step (config 4's learner: Sokoban 36-100-100-4, batch 64, replay of 8 x 32 768 transitions), device time per call from HIP
events over 300 back-to-back calls, and one lockstep step of dqn_learn with learning (graph replay, host clock). Default: the
one-workgroup kernel + Adam as a second, chip-wide launch; SGK_DQN_ONE_LAUNCH=1: Adam inside the one kernel (rounds 1-5's form).
(Against a -DSGK_DQN_MULTI_WG build -- SGK_LIB_PATH, tools/gpu_dqn_timeline.sh ... multi -- SGK_DQN_WORKGROUPS=4 picks the four-workgroup experiment.)"""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "safe-grid-agents_amd")):
    sys.path.insert(0, p)
import torch
import safe_grid_agents_amd as S

n = 32768
for name, hidden, batch in (("SideEffectsSokoban-v0", 100, 64), ("SideEffectsSokoban-v0", 64, 64), ("BoatRace-v0", 100, 64), ("IslandNavigation-v0", 100, 32)):
    dargs = types.SimpleNamespace(discount=0.99, lr=1e-3, batch_size=batch, sync_every=10000, epsilon=0.01, epsilon_anneal=100000, n_layers=2,
                                  n_hidden=hidden)
    env = S.BatchedGridworldEnv(name, n, seed=0x5AFE, layout="compact")
    dq = S.BatchedDeepQAgent(env, dargs, sgd_steps=1, replay_slices=8)
    dq.warmup(8)
    for _ in range(20):
        dq.learn_batch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300):
        dq.learn_batch()
    e1.record()
    torch.cuda.synchronize()
    sgd_us = e0.elapsed_time(e1) * 1e3 / 300
    dq.enable_graphs(learn=True)
    for _ in range(30):
        dq.step_graphed(learn=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        dq.step_graphed(learn=True)
    torch.cuda.synchronize()
    step_us = (time.perf_counter() - t0) / 300 * 1e6
    print("%s %-22s hidden %3d batch %2d: sgk_dqn_sgd_step %6.2f us per call | lockstep step with learning (graph) %6.2f us = %.3g env-steps/s"
          % ("adam-inside-the-kernel" if os.environ.get("SGK_DQN_ONE_LAUNCH") == "1" else "two-launches(default)", name, hidden, batch, sgd_us, step_us, n / step_us * 1e6), flush=True)
    env.close()
