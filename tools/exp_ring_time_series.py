"""Is the trajectory-ring rate a function of HOW LONG the chip has been writing? 1 M BoatRace envs, 100 steps per launch into a
100-slice ring, thousands of launches back to back with a HIP event every `GROUP` launches: the series of device us per lockstep
step, next to the clocks and the power the driver reports while it runs (sysfs: pp_dpm_sclk / pp_dpm_mclk / pp_dpm_fclk, hwmon
power1_average). A rate that starts high and settles lower after tens of milliseconds is power/clock management, not the write
pattern; bench.py's 20-launch bracket and a 200-launch one would then measure different things.
  python tools/exp_ring_time_series.py [n_launches] [own|ring]"""
import glob
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "safe-grid-agents_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import safe_grid_agents_amd as S  # noqa: E402

N_LAUNCH = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
FORM = sys.argv[2] if len(sys.argv) > 2 else "ring"
GROUP = 10


def read(path):
    try:
        with open(path) as f:
            return f.read()
    except OSError:
        return ""


def active_level(text):
    for line in text.splitlines():
        if line.rstrip().endswith("*"):
            return line.split(":", 1)[1].replace("*", "").strip()
    return "?"


def device_dirs():
    out = []
    for d in sorted(glob.glob("/sys/class/drm/card*/device")):
        if os.path.exists(os.path.join(d, "pp_dpm_sclk")):
            out.append(d)
    return out


class Sampler(threading.Thread):
    def __init__(self, dev):
        super().__init__(daemon=True)
        self.dev, self.rows, self.stop = dev, [], False
        hw = glob.glob(os.path.join(dev, "hwmon", "hwmon*"))
        self.hw = hw[0] if hw else None

    def run(self):
        t0 = time.perf_counter()
        while not self.stop:
            row = [time.perf_counter() - t0]
            for f in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk"):
                row.append(active_level(read(os.path.join(self.dev, f))))
            p = read(os.path.join(self.hw, "power1_average")) if self.hw else ""
            if not p and self.hw:
                p = read(os.path.join(self.hw, "power1_input"))
            row.append("%.0f W" % (int(p) / 1e6) if p.strip().isdigit() else "?")
            self.rows.append(row)
            time.sleep(0.02)


n, K = 1 << 20, 100
env = S.BatchedGridworldEnv("BoatRace-v0", n, seed=1)
st = env.torch_stream()
if FORM == "ring":
    b = torch.empty((100, n, env.n_cells), dtype=torch.int8, device="cuda")
    r = torch.empty((100, n, 4), dtype=torch.int8, device="cuda")
    fn = lambda: env.rollout_random_stream(K, boards=b, recs=r)  # noqa: E731
else:
    fn = lambda: env.step_random(K, fused="stream")  # noqa: E731
fn()
env.synchronize()
devs = device_dirs()
print("sysfs devices with clocks:", devs, flush=True)
for phase in ("cold (after 3 s idle)", "again (after 3 s idle)", "again (no idle)"):
    if "no idle" not in phase:
        time.sleep(3.0)
    sam = Sampler(devs[0]) if devs else None
    if sam:
        sam.start()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(N_LAUNCH // GROUP + 1)]
    t0 = time.perf_counter()
    evs[0].record(st)
    for i in range(N_LAUNCH // GROUP):
        for _ in range(GROUP):
            fn()
        evs[i + 1].record(st)
    env.synchronize()
    wall = time.perf_counter() - t0
    if sam:
        sam.stop = True
        sam.join()
    us = [evs[i].elapsed_time(evs[i + 1]) * 1e3 / (GROUP * K) for i in range(len(evs) - 1)]
    print("== %s, %s: %d launches in %.3f s; us per lockstep step per group of %d launches" % (FORM, phase, N_LAUNCH, wall, GROUP))
    marks = [0, 1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256]
    print("  group: " + " ".join("%6d" % m for m in marks if m < len(us)))
    print("  us   : " + " ".join("%6.2f" % us[m] for m in marks if m < len(us)))
    q = len(us) // 4
    for k in range(4):
        seg = us[k * q:(k + 1) * q]
        print("  quarter %d: mean %.2f  min %.2f  max %.2f" % (k + 1, sum(seg) / len(seg), min(seg), max(seg)))
    if sam and sam.rows:
        step = max(1, len(sam.rows) // 12)
        for row in sam.rows[::step]:
            print("  t=%.2fs sclk %s mclk %s fclk %s power %s" % tuple(row))
env.close()
