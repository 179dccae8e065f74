"""Where a single-env `env.step()` of the drop-in goes (BASELINE config 1: one env, a host-side agent in the loop): the C entry
point alone (sgk_step_host through ctypes), the GridworldEnv wrapper around it, and the reference-shaped train() loop on it
(long enough that creating the env does not show). The CPU side of the comparison is bench.py's cpu_baseline leg
(`reference_shaped_python_loop_1core`): only tests/, smoke() and that leg touch oracle/."""
import contextlib
import io
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "safe-grid-agents_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402

import safe_grid_agents_amd as S  # noqa: E402
from safe_grid_agents_amd import _lib  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "BoatRace-v0"
N = 20000
env = S.make(name)
env.reset()
b = env._b
act = np.zeros(1, dtype=np.uint8)
rec = np.zeros((1, 4), dtype=np.int8)
board = np.zeros((1, b.n_cells), dtype=np.int8)
ret = np.zeros(1, dtype=np.int32)
lib = b.lib
fn, args = lib.sgk_step_host, (b.handle, act.ctypes.data, 1, rec.ctypes.data, board.ctypes.data, ret.ctypes.data)  # (pointers looked up once)
for _ in range(200):
    fn(*args)
t0 = time.perf_counter()
for i in range(N):
    act[0] = i & 3
    fn(*args)
t_c = (time.perf_counter() - t0) / N * 1e6
env.reset()
t0 = time.perf_counter()
for i in range(N):
    s, r, d, info = env.step(i & 3)
    if d:
        env.reset()
t_w = (time.perf_counter() - t0) / N * 1e6
t0 = time.perf_counter()
for i in range(2000):  # an episode of one step: what env.reset() costs between episodes (the resident server does it)
    env.reset()
    env.step(i & 3)
t_r = (time.perf_counter() - t0) / 2000 * 1e6 - t_w
print("%s: sgk_step_host (ctypes) %.1f us | GridworldEnv.step %.1f us | GridworldEnv.reset %.1f us" % (name, t_c, t_w, t_r), flush=True)
a = S.prepare_parser().parse_args(["-S", "7", "-E", "400", "-EE", "1000", "-V", "100", "-EV", "0", "boat", "tabular-q", "-l", ".5"])
with contextlib.redirect_stdout(io.StringIO()):
    t0 = time.perf_counter()
    _, hist, _ = S.train(a, env_factory=lambda nm: env if env.reset() is not None else env, writer_factory=lambda d: S.NullWriter(d))
    dt = time.perf_counter() - t0
print("train() boat tabular-q, 400 episodes on the HIP single env: %.0f steps/s (%.1f us per step)" % ((hist["t"] + 100) / dt, dt / (hist["t"] + 100) * 1e6), flush=True)

# deep-q on the single env: the host DeepQAgent's torch kernels (forward per act, forward/backward/Adam per learn, all on the GPU)
# are in flight around every env.step while the step server's wave stays resident on the same device -- the case the advisor asked
# about (a device-wide synchronisation would wait for the server to idle out; stream-ordered torch work does not)
a = S.prepare_parser().parse_args(["-S", "3", "-E", "12", "-EE", "1000", "-V", "100", "-EV", "0", "boat", "deep-q", "-l", "1e-3", "-r", "300",
                                   "-b", "32", "-s", "200"])
env2 = S.make(name)
for attempt in range(2):  # the first pass pays torch's cold start (kernel loading, allocator): the second one is reported
    with contextlib.redirect_stdout(io.StringIO()):
        t0 = time.perf_counter()
        _, hist, _ = S.train(a, env_factory=lambda nm: env2 if env2.reset() is not None else env2, writer_factory=lambda d: S.NullWriter(d))
        dt = time.perf_counter() - t0
steps = hist["t"] + 100 + 300  # + the final evaluation + dqn_warmup's replay_capacity random steps
print("train() boat deep-q (torch MLP on %s), 12 episodes + warm-up on the HIP single env: %.0f steps/s (%.0f us per step incl. the agent's "
      "torch kernels; SGK_STEP_SERVER=%s)" % (a.device, steps / dt, dt / steps * 1e6, os.environ.get("SGK_STEP_SERVER", "1")), flush=True)
