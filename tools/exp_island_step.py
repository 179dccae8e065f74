"""Where does the per-step launch of a level with irregular episode ends spend its time? (library calls, HIP events)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "safe-grid-agents_amd")):
    sys.path.insert(0, p)
import torch
import safe_grid_agents_amd as S


def timed(env, fn, reps=20):
    st = env.torch_stream()
    with torch.cuda.stream(st):
        fn(); env.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps):
            fn()
        e1.record(st)
        env.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps / 100


for name in ("BoatRace-v0", "IslandNavigation-v0", "SideEffectsSokoban-v0"):
    for n in (1024, 65536):
        env = S.BatchedGridworldEnv(name, n, seed=1)
        a = timed(env, lambda: env.step_random(100, auto_reset=True))
        b = timed(env, lambda: env.step_random(100, auto_reset=True, write_boards=False))
        ep0 = int(env.metrics()[4])
        env.step_random(100, auto_reset=True)
        ep1 = int(env.metrics()[4])
        c = timed(env, lambda: env.step_random(100, auto_reset=False))  # after the first call every episode is over: envs idle
        print("%-24s n=%6d  auto-reset %.2f us | no boards %.2f | all envs idle (episodes over) %.2f | episodes per lockstep step: %.1f"
              % (name, n, a, b, c, (ep1 - ep0) / 100.0), flush=True)
        env.close()
