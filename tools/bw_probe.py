"""HBM bandwidth probes with torch ops (write-only fill, copy) at the byte counts of one 1M-env BoatRace step."""
import torch
def t(f, reps=200):
    for _ in range(20): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps
for mb in (16, 37, 45, 64, 128, 512):
    n = mb * 1000 * 1000
    x = torch.empty(n, dtype=torch.uint8, device="cuda"); y = torch.empty_like(x)
    us_fill = t(lambda: x.zero_())
    us_copy = t(lambda: y.copy_(x))
    print(f"{mb} MB: fill {us_fill:.1f} us = {n/us_fill/1e6:.2f} TB/s write;  copy {us_copy:.1f} us = {2*n/us_copy/1e6:.2f} TB/s (r+w)", flush=True)
