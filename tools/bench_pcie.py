#!/usr/bin/env python3
"""The batched step with a HOST-side caller: per lockstep step the actions come from pinned host memory (H2D), the step kernel
runs, and the boards + step records go back to pinned host memory (D2H) -- what a policy that lives on the CPU would pay. The
C-ABI's data path takes device pointers (sgk_step); this is the PCIe-inclusive rate DESIGN.md quotes beside the device-resident
ones. One line per batch size: device-resident step, + actions in, + records out, + boards out."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "safe-grid-agents_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import safe_grid_agents_amd as S  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "BoatRace-v0"
    sizes = [int(x) for x in os.environ.get("SGK_BENCH_SIZES", "1024,65536,1048576").split(",")]
    for n in sizes:
        env = S.BatchedGridworldEnv(name, n, seed=1, layout="compact")  # boards [n][cells]: one contiguous copy
        st = env.torch_stream()
        host_actions = torch.randint(0, 4, (n,), dtype=torch.uint8).pin_memory()
        dev_actions = torch.empty(n, dtype=torch.uint8, device="cuda")
        host_boards = torch.empty((n, env.n_cells), dtype=torch.int8).pin_memory()
        host_recs = torch.empty((n, 4), dtype=torch.int8).pin_memory()
        reps = 200 if n <= 65536 else 50
        rec_dev = env._device_views()["rec"]  # int8 [n][4]: reward, hidden reward, done, executed action

        def loop(copy_in, recs_out, boards_out):
            with torch.cuda.stream(st):
                for warm in (True, False):
                    env.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(3 if warm else reps):
                        if copy_in:
                            dev_actions.copy_(host_actions, non_blocking=True)
                        boards, reward, done, info = env.step(dev_actions, auto_reset=True)
                        if recs_out:
                            host_recs.copy_(rec_dev, non_blocking=True)
                        if boards_out:
                            host_boards.copy_(boards.reshape(n, env.n_cells), non_blocking=True)
                        if copy_in or recs_out or boards_out:
                            env.synchronize()  # the host policy needs the observation before it can choose the next action
                    env.synchronize()
                    dt = time.perf_counter() - t0
            return dt / reps * 1e6

        a = loop(False, False, False)
        b = loop(True, False, False)
        c = loop(True, True, False)
        d = loop(True, True, True)
        print("%s n=%d: device-resident %.1f us/step (%.3g env-steps/s) | + actions from the host %.1f | + records to the host %.1f | "
              "+ boards to the host %.1f us/step = %.3g env-steps/s PCIe-inclusive (%.1f GB/s over the link)"
              % (name, n, a, n / a * 1e6, b, c, d, n / d * 1e6, n * (env.n_cells + 5) / d / 1e3))
        env.close()


if __name__ == "__main__":
    main()
