"""Barrier-to-barrier timeline of sgk::dqn_sgd_kernel (config 4's learner: Sokoban 36-100-100-4, batch 64) from a -DSGK_LEARN_TIMELINE
build of the library (tools/gpu_dqn_timeline.sh): lane 0's wall_clock64() (100 MHz) behind every barrier, averaged over steps."""
import ctypes, os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "safe-grid-agents_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
import safe_grid_agents_amd as S

LABELS = ["entry: step counter -> minibatch draw -> gather of scalars and both boards (one chain); target W1^T, small tensors; Adam's bias corrections",
          "target layer 1 (target W2^T requested first)", "target W2^T into LDS", "target layer 2 (W1^T requested first)", "target head | W1^T into LDS",
          "Q layer 1 (W2^T requested first)", "W2^T into LDS", "Q layer 2 (W2 requested first)", "Q head | W2 into LDS", "loss (block sum) + dL/dq",
          "dL/dh2, W3 / b3 gradient", "dL/dh1, W2 / b2 gradient tiles", "W1 / b1 gradient tiles + norm (block sum); the first quad's Adam state requested",
          "Adam: one quad ahead (m / v / vmax / w in, five tensors out)"]
MULTI = ["entry: minibatch draw + gather, four weight matrices and the small tensors into LDS (one batch of loads)", "layer 1, target and Q side by side",
         "layer 2, target and Q side by side", "heads", "targets out + grid barrier B1 (the minibatch's mean target)", "mean target, loss, dL/dq",
         "dL/dh2 | W3, b3 gradient", "dL/dh1 | W2, b2 gradient tiles", "W1, b1 gradient tiles", "grid barrier B2 (partial gradients)",
         "sum of the partials, squared norm, Adam state in", "grid barrier B3 (norm)", "Adam on a quarter of the parameters, transposed copies"]
multi = os.environ.get("SGK_DQN_WORKGROUPS") == "4"  # the -DSGK_DQN_MULTI_WG experiment (tools/gpu_dqn_timeline.sh multi)
one_launch = os.environ.get("SGK_DQN_ONE_LAUNCH") == "1"
if multi:
    LABELS = MULTI
elif not one_launch:  # the product's default: the kernel ends behind the norm, Adam is a second launch (dqn_adam_kernel)
    LABELS = LABELS[:12] + ["W1 / b1 gradient tiles; the gradient (55 KB, flat) and the clip coefficient out; norm (block sum)"]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
dargs = types.SimpleNamespace(discount=0.99, lr=1e-3, batch_size=64, sync_every=10000, epsilon=0.01, epsilon_anneal=100000, n_layers=2, n_hidden=100)
env = S.BatchedGridworldEnv("SideEffectsSokoban-v0", n, seed=0x5AFE, layout="compact")
dq = S.BatchedDeepQAgent(env, dargs, sgd_steps=1, replay_slices=8)
dq.warmup(8)
lib = env.lib
lib.sgk_debug_learn_stamps.argtypes = [ctypes.c_void_p]
stamps = np.zeros(32, dtype=np.uint64)
rows = []
clocks = []
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev = []
for k in range(60):
    e0.record()
    dq.learn_batch()
    e1.record()
    assert lib.sgk_debug_learn_stamps(stamps.ctypes.data) == 0
    if k >= 10:
        rows.append(stamps[:len(LABELS) + 1].astype(np.int64).copy())
        clocks.append(float(int(stamps[31]) - int(stamps[30])) / float(int(stamps[len(LABELS)]) - int(stamps[0])) * 100.0)
        ev.append(e0.elapsed_time(e1) * 1e3)
d = np.diff(np.stack(rows), axis=1) / 100.0  # us
print(("dqn_sgd_multi_kernel (4 workgroups; workgroup 0's stamps)" if multi else "dqn_sgd_kernel (1 workgroup%s)" % (", Adam inside" if one_launch else "; Adam follows as dqn_adam_kernel, not in this timeline")) + " <%d, 100>, batch 64, %d envs x 8 slices in the replay: us between barriers, lane 0 (mean / min / max over %d steps)" % (env.n_cells, n, len(rows)))
for i, lab in enumerate(LABELS):
    print("  %2d  %6.2f  %6.2f  %6.2f  %s" % (i, d[:, i].mean(), d[:, i].min(), d[:, i].max(), lab))
print("  in-kernel shader clock (s_memtime over s_memrealtime): %.0f MHz (min %.0f, max %.0f)" % (np.mean(clocks), np.min(clocks), np.max(clocks)))
print("  sum %6.2f us inside the kernel; HIP events around the call %6.2f us (min %.2f)" % (d.sum(axis=1).mean(), np.mean(ev), np.min(ev)))
