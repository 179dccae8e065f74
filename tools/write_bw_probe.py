"""Achievable WRITE-ONLY HBM bandwidth on this chip (torch fills over buffers far larger than the caches) next to the copy
rate: what a kernel that only streams stores, like the streaming rollout into a trajectory ring, can be held against."""
import torch


def t(f, reps=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


for mb in (30, 256, 1024, 3000):
    n = mb * 1000 * 1000
    x = torch.empty(n, dtype=torch.uint8, device="cuda")
    y = torch.empty_like(x)
    us_fill = t(lambda: x.fill_(3))
    us_copy = t(lambda: y.copy_(x))
    print("%5d MB: fill %.1f us = %.2f TB/s written;  copy %.1f us = %.2f TB/s (read + written)" % (
        mb, us_fill, n / us_fill / 1e6, us_copy, 2 * n / us_copy / 1e6), flush=True)
