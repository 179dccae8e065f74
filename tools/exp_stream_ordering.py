"""What one ordering call between the handle's stream and torch's costs on the host (EXPERIMENTS R5.13): event record / stream wait on
torch's default (legacy) stream, on a torch side stream and on the library's own stream, and the library's sgk_stream_wait /
sgk_stream_signal pairs against both kinds of torch stream."""
import os, sys, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "safe-grid-agents_amd")):
    sys.path.insert(0, p)
import torch
import safe_grid_agents_amd as S
from safe_grid_agents_amd import _lib

env = S.BatchedGridworldEnv("BoatRace-v0", 1024, seed=1)
null = torch.cuda.default_stream()
side = torch.cuda.Stream()
own = env.torch_stream()
ev = torch.cuda.Event()
x = torch.zeros(8, device="cuda")


def t(fn, reps=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    dt = (time.perf_counter() - t0) / reps * 1e6
    torch.cuda.synchronize()
    return dt


print("event record on the legacy stream      %6.2f us" % t(lambda: ev.record(null)))
print("event record on a torch side stream    %6.2f us" % t(lambda: ev.record(side)))
print("event record on the library's stream   %6.2f us" % t(lambda: ev.record(own)))
ev.record(own)
print("legacy stream waits for an event       %6.2f us" % t(lambda: null.wait_event(ev)))
print("side stream waits for an event         %6.2f us" % t(lambda: side.wait_event(ev)))
print("library's stream waits for an event    %6.2f us" % t(lambda: own.wait_event(ev)))
h = env._h.ptr
for name, st in (("legacy", null), ("side", side)):
    p = ctypes.c_void_p(st.cuda_stream)
    print("sgk_stream_wait   against the %-6s stream %6.2f us" % (name, t(lambda: env.lib.sgk_stream_wait(h, p))))
    print("sgk_stream_signal against the %-6s stream %6.2f us" % (name, t(lambda: env.lib.sgk_stream_signal(h, p))))
# the same pairs with a kernel between them (the real pattern: wait, launch, signal)
a = torch.zeros(1024, dtype=torch.uint8, device="cuda")
for name, st in (("legacy", null), ("side", side)):
    with torch.cuda.stream(st):
        print("env.step(actions) with torch on the %-6s stream %6.2f us per call" % (name, t(lambda: env.step(a, auto_reset=True), 1000)))
env.bind_torch_stream(side)
with torch.cuda.stream(side):
    print("env.step(actions) bound to the side stream          %6.2f us per call" % t(lambda: env.step(a, auto_reset=True), 1000))
env.close()
