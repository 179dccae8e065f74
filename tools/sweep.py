"""Size / layout / mode sweep of the step path (runs on the GPU box). Prints one line per configuration."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "safe-grid-agents_amd")):
    sys.path.insert(0, p)
import torch

import safe_grid_agents_amd as S

B_ALG = {"BoatRace-v0": 78, "SideEffectsSokoban-v0": 100, "IslandNavigation-v0": 124, "DistributionalShift-v0": 154, "WhiskyGold-v0": 124, "AbsentSupervisor-v0": 124}  # 2 H W + 28


def timed(env, fn, reps):
    s = env.torch_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    env.synchronize()
    e0.record(s)
    for _ in range(reps):
        fn()
    e1.record(s)
    env.synchronize()
    return e0.elapsed_time(e1) / reps


NAMES = [a for a in sys.argv[1:] if a.endswith("-v0")] or ["BoatRace-v0", "IslandNavigation-v0", "SideEffectsSokoban-v0", "DistributionalShift-v0", "WhiskyGold-v0", "AbsentSupervisor-v0"]
SIZES = [int(a) for a in sys.argv[1:] if a.isdigit()] or [1 << 10, 1 << 16, 1 << 18, 1 << 20, 1 << 22]
for name in NAMES:
    for layout in ("compact", "pitched"):
        for n in SIZES:
            env = S.BatchedGridworldEnv(name, n, seed=1, layout=layout)
            acts = torch.randint(0, 4, (n,), dtype=torch.uint8, device="cuda")
            ms_graph = timed(env, lambda: env.step_random(100, auto_reset=True), 5) / 100
            ms_nob = timed(env, lambda: env.step_random(100, auto_reset=True, write_boards=False), 5) / 100
            ms_given = timed(env, lambda: env.step(acts, auto_reset=True), 200)
            ms_rep = timed(env, lambda: env.step_repeat(acts, 100, auto_reset=True), 5) / 100
            ms_fused = timed(env, lambda: env.step_random(200, auto_reset=True, fused=True), 3) / 200
            gbs = B_ALG[name] * n / (ms_graph / 1e3) / 1e9
            print("%-22s %-8s n=%8d  step(graph) %8.2f us  %7.2e steps/s  %6.0f GB/s-alg | no-boards %8.2f us | "
                  "given-actions(eager,torch) %8.2f us | repeat(eager C loop, no RNG) %8.2f us | fused %8.3f us/step %7.2e steps/s"
                  % (name, layout, n, ms_graph * 1e3, n / (ms_graph / 1e3), gbs, ms_nob * 1e3, ms_given * 1e3, ms_rep * 1e3,
                     ms_fused * 1e3, n / (ms_fused / 1e3)), flush=True)
            env.close()
