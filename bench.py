#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched lockstep gridworld step path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A bench "step" is one pass of the hot path over the whole env batch = one EPISODE for every env: `--lockstep-per-step`
(default 100 = BoatRace's fixed horizon; SURVEY.md 8(d) states the measurement in whole 100-step episodes) lockstep
steps, in each of which every env of every rank takes one RandomAgent action (counter RNG, in-kernel) -- transition,
observed reward, hidden safety reward, episode bookkeeping, auto-reset. `value` is env-steps/s (envs x lockstep steps /
host wall seconds); `ms_per_step` is per bench step, `us_per_lockstep_step` beside it.

What is timed -- `--path`:
  ring    (default, `value`)  sgk_rollout_random_stream, 100 lockstep steps per launch, env state words in registers between
          steps, EVERY step's successor board (int8 tile, write-through stores) and step record KEPT in a caller-owned
          trajectory ring of 100 slices (boards [100][n][cells] + records [100][n]: 3.04 GB at 1 M BoatRace envs, twelve
          times the 256 MiB Infinity Cache): the batched dqn_warmup (reference warmup.py:14-21 keeps every random-action
          transition) and the only streamed form whose per-step outputs a consumer can read afterwards.
  own     the same kernel into the env's own board / record buffers: step k overwrites step k-1's outputs inside the launch
          (what 100 per-step launches leave). The 30 MB working set is rewritten in place inside the Infinity Cache, the 4-byte
          records coalesce in L2: its rate is a fabric / Infinity-Cache write rate, not an HBM rate. Round 2's `value`.
  launch  one step-kernel launch per lockstep step, replayed from a hipGraph (state words through memory every step). Round
          1's `value` was this with one lockstep step per bench step.
The default run reports the other two as secondary objects (`rewritten_in_place`, `per_step_launches`) and the outputs-once
kernel (`fused_rollout`: no per-step output exists), each with its own counter-based traffic figures.
The ring's rate differs from allocation to allocation of the 3 GB, so the primary measurement is made
`--rings` times (default 5: 15 GB of the 288), each on a FRESH ring of the same backing (the library's ring allocator `sgk_ring_alloc`: one virtual
range mapped from 256 MiB physical chunks; `--ring-backing torch` for plain blocks), earlier rings held while the next is timed:
`value` is the MEDIAN ring's whole-job rate, `value_min` / `value_max` and `primary_rings` give the spread, and every ring
is also timed with the library's store-only probe (`sgk_ring_probe`: the kernel's stores and nothing else) so that
`roofline.kernel_over_probe` says how far the kernel is from what THIS memory can take. `other_ring_allocations` times three
PLAIN rings (torch.empty) in the same process. Nothing is picked anywhere.

Workload = BASELINE.json's metric config: BoatRace, 1 048 576 concurrent envs in the whole job at every GPU count (the batch
shards by env id, one contiguous block per rank, no data-path collective; the only exchange is one int64 metrics all-reduce
at the end of the timed region). At N > 1 a secondary object reports the weak-scaling form (1 048 576 envs on every GPU).

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` and `cpu_baseline` objects added.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "safe-grid-agents_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

# SURVEY.md 8(d)'s per-step-launch contract, 2 H W + 28 bytes per env-step (board read + written, action, counters, returns,
# rewards, done): what ONE LAUNCH PER STEP would have to move. Kept for the `survey_8d` figures of the line.
B_ALG = {"BoatRace-v0": 78, "SideEffectsSokoban-v0": 100, "IslandNavigation-v0": 124, "DistributionalShift-v0": 154,
         "WhiskyGold-v0": 124, "AbsentSupervisor-v0": 124, "SafeInterruptibility-v0": 140, "ConveyorBelt-v0": 126, "TomatoWatering-v0": 154, "FriendFoe-v0": 88}
CELLS = {k: (v - 28) // 2 for k, v in B_ALG.items()}
REC_BYTES = 4  # sgk_step_rec: reward i8, hidden reward i8, done u8, executed action u8 (8(d)'s reward / hidden / done, packed)
STATE_BYTES = 8  # the packed env state word
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
GRAPH_CHUNK = 100      # lockstep steps per launch (ring / own) or per hipGraph replay (launch)
RING_SLICES = 100      # trajectory ring of the primary path: one 100-step episode of every env


def algorithmic_bytes_per_env_step(env_name, path):
    """Bytes per env-step a kernel of this FORM cannot avoid moving (DESIGN.md 3.2 / 5 state them):
    ring / own: the outputs of 8(d)'s step contract that a consumer receives -- successor board H W + packed step record 4 --;
    the state word, the counters and the returns stay in registers for the whole launch (their one round trip per launch is
    16 B / 100 steps); launch: the same + the state word's read and write every step."""
    out = CELLS[env_name] + REC_BYTES
    return out + (2 * STATE_BYTES if path == "launch" else 0)


def cpu_baseline(env_name, seed, target_seconds=10.0):
    """The oracle's scalar C engine (kind "port": the reference's env is absent) on the host cores, on a
    bounded sample of the same workload (random-action lockstep rollout with reset-on-done): first ONE thread, then one
    thread per host core (contiguous env ranges, no sharing). `value`/`cores` report the all-core run."""
    from oracle import oracle as O

    def timed(n, steps, threads):
        envs = O.EnvBatch(env_name, n)
        t0 = time.perf_counter()
        if threads == 1:
            envs.rollout(steps, seed=seed, auto_reset=True)
        else:
            threads = O.rollout_mt(envs, steps, threads, seed=seed, auto_reset=True)
        return n * steps / (time.perf_counter() - t0), threads

    n1 = 16384
    probe, _ = timed(n1, 50, 1)
    steps1 = int(max(100, min(20000, target_seconds * probe / n1)))
    one, _ = timed(n1, steps1, 1)
    cores = os.cpu_count() or 1
    n_all = max(n1, 256 * cores)
    steps_all = int(max(100, min(20000, target_seconds * one * cores * 0.5 / n_all)))
    allc, used = timed(n_all, steps_all, cores)
    # the reference-shaped single-process Python loop (train(): TabularQAgent + env.step per step)
    # on the oracle env -- what `python main.py boat tabular-q --lr .5` does, minus the upstream env's own cost
    import safe_grid_agents_amd as S
    from oracle.gym_shim import OracleGridworldEnv

    # (400 episodes: long enough that creating the env and the first calls do not show -- 40 episodes read 3 x lower)
    a = S.prepare_parser().parse_args(["-S", "7", "-E", "400", "-EE", "1000", "-V", "100", "-EV", "0", "boat", "tabular-q",
                                       "-l", ".5"])
    import contextlib
    import io

    import torch  # noqa: F401  (keep its import time out of the measurement)

    with contextlib.redirect_stdout(io.StringIO()):  # default_eval prints a banner; bench.py owns stdout (ONE JSON line)
        t0 = time.perf_counter()
        _, hist, _ = S.train(a, env_factory=OracleGridworldEnv, writer_factory=lambda d: S.NullWriter(d))
        py_loop = (hist["t"] + 100) / (time.perf_counter() - t0)
    return {
        "value": allc, "unit": "env-steps/s", "cores": used, "kind": "port",
        "reference_shaped_python_loop_1core": py_loop,
        "sample": "%s random rollout, oracle C engine (gcc -O2): %d envs x %d steps on %d threads; 1 thread: %d envs x %d "
                  "steps" % (env_name, n_all, steps_all, used, n1, steps1),
        "one_core_value": one, "host_cpus": cores,
    }


def parity_sample(env, env_name, seed, base, total_steps, block=2048):
    """Bit-exact check in the same run: the first and the last `block` envs of this rank's shard --
    boards, episode returns, hidden returns, frame counters and the last episode's return / performance -- against the
    oracle stepped through the same `total_steps` lockstep steps."""
    from oracle import oracle as O

    n = env.n_envs
    boards = env.boards_host().reshape(n, -1)
    st = env.episode_state_host()
    le = env.last_episode_host()
    checked = 0
    for lo in sorted({0, max(0, n - block)}):
        hi = min(n, lo + block)
        orc = O.EnvBatch(env_name, hi - lo, seed=seed, env_begin=base + lo)  # keyed like the shard from the first reset on
        orc.rollout(total_steps, seed=seed, env_begin=base + lo, t_begin=0, auto_reset=True)
        ok = ((boards[lo:hi] == orc.boards()).all()
              and (st["episode_return"][lo:hi] == orc.field("episode_return")).all()
              and (st["hidden_return"][lo:hi] == orc.field("hidden_return")).all()
              and (st["frame"][lo:hi] == orc.field("frame")).all()
              and (le["n_episodes"][lo:hi] == orc.field("n_episodes")).all()
              and (le["last_return"][lo:hi] == orc.field("last_episode_return")).all()
              and (le["last_performance"][lo:hi] == [orc.last_performance(i) or 0 for i in range(hi - lo)]).all())
        if not ok:
            return False, checked
        checked += hi - lo
    return True, checked


def ring_parity_sample(env_name, seed, base, ring, first_step, n_slices, block=512, ring_start=0):
    """The trajectory ring's CONTENTS against the oracle: slice (first_step + k - ring_start) % slices must hold the board and the
    step record of lockstep step first_step + k for the first `block` envs of the shard (the last `n_slices` steps written;
    `ring_start` = the lockstep step whose outputs went into slice 0 first)."""
    import numpy as np

    from oracle import oracle as O

    boards, recs = ring
    slices, n = int(boards.shape[0]), int(boards.shape[1])
    m = min(block, n)
    orc = O.EnvBatch(env_name, m, seed=seed, env_begin=base)
    if first_step:
        orc.rollout(first_step, seed=seed, env_begin=base, t_begin=0, auto_reset=True)
    for k in range(n_slices):
        rec = orc.rollout(1, seed=seed, env_begin=base, t_begin=first_step + k, auto_reset=True)
        sl = (first_step + k - ring_start) % slices
        if not ((boards[sl, :m].cpu().numpy() == orc.boards()).all() and (recs[sl, :m].cpu().numpy() == np.asarray(rec)).all()):
            return False
    return True


def chunk_schedule(k):
    """Chunk sizes `run(k)` issues, in order: GRAPH_CHUNK-step launches (ring / own) / hipGraph replays (launch) and one tail."""
    out = [GRAPH_CHUNK] * (k // GRAPH_CHUNK)
    if k % GRAPH_CHUNK:
        out.append(k % GRAPH_CHUNK)
    return out


def timed_steps(env, steps, warmup, barrier, global_metrics, path="ring", ring=None, slice_next=None):
    """W untimed warm-up steps, then EXACTLY `steps` lockstep steps + the metrics flush between barrier + synchronize pairs.
    Every hipGraph the timed region replays is captured and instantiated BEFORE the region (sgk_step_random_prepare for each
    chunk size of the schedule), and the first HIP event is recorded immediately before the first launch, so neither the
    host clock nor the device clock sees a capture. Returns (elapsed_s, kernel_ms, BatchMetrics): the host clock stops when
    this rank holds the all-reduced metrics and its stream is idle -- the all-reduce is the synchronisation; the barrier that
    follows is outside the clock."""
    import torch

    stream = env.torch_stream()
    slice_next = slice_next if slice_next is not None else [0]

    def run(k):
        for c in chunk_schedule(k):
            if path == "ring":  # every step's boards + records kept in the caller's trajectory ring
                env.rollout_random_stream(c, boards=ring[0], recs=ring[1], first_slice=slice_next[0])
                slice_next[0] = (slice_next[0] + c) % ring[0].shape[0]
            elif path == "own":
                env.step_random(c, auto_reset=True, fused="stream")
            else:
                env.step_random(c, auto_reset=True)

    if path == "launch":
        for c in sorted(set(chunk_schedule(warmup) + chunk_schedule(steps))):
            env.prepare_step_random(c, auto_reset=True)
    # The env's stream is torch's CURRENT stream for the whole region. Otherwise every call of the Python wrapper orders the
    # library's stream against torch's current stream with a pair of events (so that torch ops before / after see the right
    # data), and two such hops between back-to-back launches cost 25 us of idle GPU each time (rocprofv3 kernel trace of this
    # file, profiles/r03/bench_launch_gaps.log) -- with nothing of torch's in between to order against.
    with torch.cuda.stream(stream):
        run(warmup)
        env.metrics_reset()  # the line's episode figures (episodes_finished, mean_return, mean_safety) are the timed region's
        env.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # ---- timed region -----------------------------------------------------------------------------------------------
        barrier()
        torch.cuda.synchronize()
        env.synchronize()
        t0 = time.perf_counter()
        ev0.record(stream)
        run(steps)
        t_enq = time.perf_counter()
        ev1.record(stream)
        gm = global_metrics(env)  # syncs the stream; one int64 all-reduce (RCCL over xGMI) when world > 1
        t_met = time.perf_counter()
        env.synchronize()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        barrier()
    if os.environ.get("SGK_BENCH_TRACE") == "1":  # where the host clock goes (stderr; stdout stays ONE JSON line)
        sys.stderr.write("bench trace: enqueue %.1f us, +metrics %.1f us, +syncs %.1f us, device %.1f us\n" % (
            (t_enq - t0) * 1e6, (t_met - t_enq) * 1e6, (t0 + elapsed - t_met) * 1e6, ev0.elapsed_time(ev1) * 1e3))
    # ---------------------------------------------------------------------------------------------------------------------
    return elapsed, ev0.elapsed_time(ev1), gm  # HIP events on the stream the step kernels run on


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks with torch.distributed.run as a CHILD process, before
    anything in this process has touched the GPU, and exit with its code."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd)


def traffic_bytes(env_name, layout, n_local, path, steps_per_launch):
    """HBM-side bytes one launch of the path's kernel moves, from the committed rocprofv3 PMC passes (profiles/traffic.json:
    FETCH_SIZE x the calibration the guide prescribes + WRITE_SIZE), scaled to this run's steps per launch; None if that size
    was not measured."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tpath):
        return None
    with open(tpath) as f:
        tj = json.load(f)
    key = "%s/%s/%d" % (env_name, layout, n_local)
    if path == "launch":
        return tj.get(key)
    per100 = tj.get("%s/%s%d" % (key, {"ring": "ring", "own": "stream"}[path], GRAPH_CHUNK))
    return None if per100 is None else per100 * steps_per_launch / GRAPH_CHUNK


PATH_KERNEL = {"ring": "sgk::rollout_random_kernel<%s, stream> -> trajectory ring", "own": "sgk::rollout_random_kernel<%s, stream> -> own buffers",
               "launch": "sgk::step_kernel<%s>"}
PATH_BOUND = {
    # where the bytes land decides what the rate can be held against
    "ring": "hbm",  # 3 GB of fresh addresses per pass: every byte reaches DRAM (profiles/r03/ring_size_sweep.log)
    "own": "fabric / infinity-cache write",   # 30 MB rewritten in place inside the 256 MiB Infinity Cache
    "launch": "fabric / infinity-cache write",  # the same buffers, rewritten by every launch
}


def path_figures(env_name, layout, n_local, path, lockstep_steps, kernel_ms):
    """The roofline-style figures of one measured path: algorithmic rate, counter-based traffic rate, what bounds it."""
    launches = len(chunk_schedule(lockstep_steps)) if path != "launch" else lockstep_steps
    steps_per_launch = lockstep_steps / launches
    launch_s = kernel_ms / 1e3 / launches  # average duration of one launch of the dominant kernel incl. its launch gap
    b_alg = algorithmic_bytes_per_env_step(env_name, path)
    achieved = b_alg * n_local * steps_per_launch / launch_s / 1e9
    traffic = traffic_bytes(env_name, layout, n_local, path, steps_per_launch)
    b8d = B_ALG[env_name]
    return {
        "bound": PATH_BOUND[path], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        # a fraction of the HBM peak only where the bytes demonstrably reach DRAM (the ring): buffers rewritten in place live in
        # the Infinity Cache and "achieved / 8 TB/s" is above 1 there -- not a utilisation, so not given (traffic_frac is: it
        # says how busy the fabric's write path was)
        "frac": achieved / HBM_PEAK_GBS if PATH_BOUND[path] == "hbm" else None,
        # what the fabric really moved (rocprofv3 FETCH_SIZE x calibration + WRITE_SIZE per launch, profiles/traffic.json) over the
        # same launch time
        "traffic": traffic,
        "traffic_gbs": None if traffic is None else traffic / launch_s / 1e9,
        "traffic_frac": None if traffic is None else traffic / launch_s / 1e9 / HBM_PEAK_GBS,
        "kernel": PATH_KERNEL[path] % env_name,
        "algorithmic_bytes_per_env_step": b_alg, "steps_per_launch": steps_per_launch,
        "algorithmic_bytes_per_launch": b_alg * n_local * steps_per_launch,
        "avg_launch_us": launch_s * 1e6, "device_us_per_step": launch_s * 1e6 / steps_per_launch,
        "frac_of_measured_copy_peak_6290": achieved / 6290.0,
        # SURVEY 8(d)'s figure for a one-launch-per-step design (2 H W + 28: it charges a board read and a state / counter round
        # trip per step that a K-step kernel does not make): quoted for continuity with rounds 1-2, NOT a utilisation
        "survey_8d_bytes_per_env_step": b8d, "survey_8d_gbs": b8d * n_local * steps_per_launch / launch_s / 1e9,
    }


def issue_roofline(env_name, n_local, steps, seconds, peak_now=None):
    """`fused_rollout`'s bound is instruction issue, not memory: instructions per env-step from the committed SQ counter pass
    (profiles/issue.json, tools/make_issue_json.py: a property of the code) x this run's rate, against the chip's issue peaks
    MEASURED IN THIS PROCESS (sgk_issue_peak: the peaks move with the clock the box runs at); the committed peaks of another box
    are kept beside them as `peak_committed`. None when no counter pass for this env is committed."""
    ipath = os.path.join(ROOT, "profiles", "issue.json")
    if not os.path.exists(ipath):
        return None
    with open(ipath) as f:
        ij = json.load(f)
    k = ij.get("%s/outputs_once" % env_name)
    if not k:
        return None
    wave_steps = (n_local / 64.0) * steps / seconds  # wave-level steps per second
    valu, salu = k["valu_per_wave_step"] * wave_steps, k["salu_per_wave_step"] * wave_steps
    cv, cs = ij["peak"]["valu_wave_instr_per_s"], ij["peak"]["salu_wave_instr_per_s"]
    pv, ps = peak_now if peak_now else (cv, cs)
    bound = "salu-issue" if salu / ps >= valu / pv else "valu-issue"
    return {"bound": bound, "achieved": (salu if bound == "salu-issue" else valu) / 1e9, "peak": (ps if bound == "salu-issue" else pv) / 1e9,
            "unit": "G wave-instructions/s", "frac": max(salu / ps, valu / pv), "valu_frac": valu / pv, "salu_frac": salu / ps,
            "peak_source": "sgk_issue_peak, this process" if peak_now else "profiles/issue.json (committed)",
            "peak_in_run": None if not peak_now else {"valu": pv / 1e9, "salu": ps / 1e9},
            "peak_committed": {"valu": cv / 1e9, "salu": cs / 1e9, "frac": max(salu / cs, valu / cv)},
            "valu_per_wave_step": k["valu_per_wave_step"], "salu_per_wave_step": k["salu_per_wave_step"],
            "lds_per_wave_step": k.get("lds_per_wave_step"), "source": k.get("source")}


def median_index(vals):
    """Index of the median element (the lower one for an even count)."""
    order = sorted(range(len(vals)), key=lambda i: vals[i])
    return order[(len(vals) - 1) // 2]


MFMA_F32_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: dense fp32 matrix peak


def _threads_timed(make_worker, n_threads):
    """Run make_worker(i)() on n_threads Python threads (the oracle's C calls release the GIL); returns wall seconds."""
    import threading

    workers = [make_worker(i) for i in range(n_threads)]
    ts = [threading.Thread(target=w) for w in workers]
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    return time.perf_counter() - t0


def cpu_baseline_tabq(env_name, seed, target_seconds=10.0):
    """Config 3's CPU side: the oracle's tabular-Q rollout (scalar C: act_explore -> env.step -> learn per agent-step, dictionary
    Q-tables like the reference's defaultdict), one agent block per host thread."""
    from oracle import oracle as O

    args = dict(lr=0.5, discount=0.99, eps0=0.01, anneal=100000)

    def block(n, steps):
        envs = O.EnvBatch(env_name, n, seed=seed)
        agents = [O.TabQ(envs.H * envs.W, args["lr"], args["discount"], args["eps0"], args["anneal"]) for _ in range(n)]
        return lambda: O.tabq_rollout(envs, agents, steps, seed=seed)

    n1 = 1024
    w = block(n1, 20)
    t0 = time.perf_counter(); w(); probe = n1 * 20 / (time.perf_counter() - t0)
    steps1 = int(max(50, min(5000, target_seconds * 0.4 * probe / n1)))
    w = block(n1, steps1)
    t0 = time.perf_counter(); w(); one = n1 * steps1 / (time.perf_counter() - t0)
    cores = os.cpu_count() or 1
    threads = min(cores, 64)
    steps_all = int(max(50, min(5000, target_seconds * 0.6 * one / n1)))
    dt = _threads_timed(lambda i: block(n1, steps_all), threads)
    return {"value": threads * n1 * steps_all / dt, "unit": "agent-steps/s", "cores": threads, "kind": "port", "one_core_value": one,
            "host_cpus": cores,
            "sample": "%s + tabular-q, oracle C (gcc -O2): %d threads x %d agents x %d steps; 1 thread: %d agents x %d steps"
                      % (env_name, threads, n1, steps_all, n1, steps1)}


def cpu_baseline_mlp(env_name, seed, n_hidden, target_seconds=10.0):
    """Config 4's CPU side: the same MLP forward in torch fp32 on the host cores (torch's own thread pool) + the oracle's env.step
    on the greedy actions, for a block of envs. (No exploration draws: a baseline of the arithmetic, not of the stream.)"""
    import numpy as np
    import torch

    from oracle import oracle as O

    H, W = O.shape(env_name)
    n = 4096
    envs = O.EnvBatch(env_name, n, seed=seed)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(H * W, n_hidden), torch.nn.ReLU(), torch.nn.Linear(n_hidden, n_hidden), torch.nn.ReLU(),
                              torch.nn.Linear(n_hidden, 4))

    def run(steps):
        t0 = time.perf_counter()
        with torch.no_grad():
            for k in range(steps):
                obs = torch.from_numpy(envs.boards().astype(np.float32))
                a = net(obs).argmax(1).numpy().astype(np.uint8)
                envs.rollout(1, seed=seed, t_begin=k, auto_reset=True, actions=a.reshape(1, n))
        return time.perf_counter() - t0

    probe = n * 5 / run(5)
    steps = int(max(10, min(2000, target_seconds * probe / n)))
    dt = run(steps)
    return {"value": n * steps / dt, "unit": "env-steps/s", "cores": torch.get_num_threads(), "kind": "port", "host_cpus": os.cpu_count(),
            "sample": "%s: torch fp32 CPU forward of the %d-%d-%d-4 MLP (%d torch threads) + oracle env.step (1 thread), %d envs x %d "
                      "steps, greedy actions" % (env_name, H * W, n_hidden, n_hidden, torch.get_num_threads(), n, steps)}


def run_config(args):
    """BASELINE.json configs 2 / 3 / 4 on one GPU: one JSON object each, with the roofline of the config's dominant kernel and a CPU
    baseline on the box's host cores. Every fraction follows from numbers inside the object."""
    import types

    import torch

    import safe_grid_agents_amd as S

    dev = 0
    torch.cuda.set_device(dev)

    def ev_time(env, fn, reps):
        st = env.torch_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        env.synchronize(); torch.cuda.synchronize()
        with torch.cuda.stream(st):
            e0.record(st)
            for _ in range(reps):
                fn()
            e1.record(st)
            env.synchronize(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 1e3 / reps

    out = {"config": args.config, "n_gpus": 1, "data": "synthetic", "higher_is_better": True}
    if args.config == 2:
        name, n = "BoatRace-v0", 65536
        env = S.BatchedGridworldEnv(name, n, device=dev, seed=args.seed, layout="compact", stream="own")
        env.step_random(200, auto_reset=True)
        per = {}
        per["launch"] = ev_time(env, lambda: env.step_random(100, auto_reset=True), 20) / 100
        per["own"] = ev_time(env, lambda: env.step_random(100, auto_reset=True, fused="stream"), 20) / 100
        per["fused"] = ev_time(env, lambda: env.step_random(1000, auto_reset=True, fused=True), 5) / 1000
        b = algorithmic_bytes_per_env_step(name, "launch")
        gbs = b * n / per["launch"] / 1e9
        out.update({
            "workload": "BoatRace random-action rollout, 65 536 envs lockstep, step kernel (one launch per lockstep step, hipGraph x100)",
            "metric": "env-steps/s", "unit": "env-steps/s", "value": n / per["launch"], "us_per_lockstep_step": per["launch"] * 1e6,
            "dtype": "int8",
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_env_step": b, "kernel": "sgk::step_kernel<BoatRace-v0>",
                         "note": "2.9 MB per launch: the working set lives in L2 / Infinity Cache and a launch is ~1 dependent-launch "
                                 "boundary long -- the config is launch-latency-bound, the HBM fraction says how far from bandwidth"},
            "other_forms": {"streamed_100_steps_per_launch_us": per["own"] * 1e6, "outputs_once_1000_steps_per_launch_us": per["fused"] * 1e6,
                            "streamed_value": n / per["own"], "outputs_once_value": n / per["fused"]}})
        env.close()
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(name, args.seed)
    elif args.config == 3:
        name, n = "IslandNavigation-v0", 262144
        targs = types.SimpleNamespace(lr=0.5, discount=0.99, epsilon=0.01, epsilon_anneal=100000)
        env = S.BatchedGridworldEnv(name, n, device=dev, seed=args.seed, stream="own")
        agent = S.BatchedTabularQAgent(env, targs)
        agent.rollout(200)
        steps_per_launch = 1000
        dt = ev_time(env, lambda: agent.rollout(steps_per_launch), 3) / steps_per_launch
        # the per-launch fixed cost (table load + store), from two launch lengths: t(K) = fixed + K * marginal
        per_launch_us = {}
        for K in (250, 500, 1000, 2000):
            per_launch_us[str(K)] = ev_time(env, lambda: agent.rollout(K), 3) * 1e6 / K
        marginal_us = (per_launch_us["2000"] * 2000 - per_launch_us["500"] * 500) / 1500.0
        fixed_us = per_launch_us["500"] * 500 - marginal_us * 500
        table_mb = n * agent.n_states * 4 * 8 / 1e6
        peak1 = S._lib.issue_peak(dev, 1)  # the kernel's residency: its Q image leaves room for ONE wave per SIMD
        peak8 = S._lib.issue_peak(dev, 8)
        kpath = os.path.join(ROOT, "profiles", "issue.json")
        k = json.load(open(kpath)).get("%s/tabq_rollout" % name) if os.path.exists(kpath) else None
        roof = None
        if k:
            wave_steps = (n / 64.0) / dt
            valu = k["valu_per_wave_step"] * wave_steps
            # frac: against what the CHIP can issue (8 waves per SIMD), measured in this process. The kernel's own residency -- ONE
            # wave per SIMD, all its 40 KB Q image per 64 agents leaves room for -- is the design's choice, not the chip's limit:
            # the figure against that ceiling is kept beside it as frac_at_kernel_occupancy.
            roof = {"bound": "valu-issue", "achieved": valu / 1e9, "peak": peak8[0] / 1e9, "unit": "G wave-instructions/s",
                    "frac": valu / peak8[0], "peak_is": "sgk_issue_peak at 8 waves per SIMD, this process (the chip's VALU issue rate)",
                    "frac_at_kernel_occupancy": valu / peak1[0], "peak_at_kernel_occupancy": peak1[0] / 1e9,
                    "kernel_occupancy": "1 wave per SIMD: the LDS-resident Q image (40 960 B per 64 agents) allows 4 waves per CU",
                    "valu_per_wave_step": k["valu_per_wave_step"], "salu_per_wave_step": k["salu_per_wave_step"],
                    "lds_per_wave_step": k.get("lds_per_wave_step"), "source": k.get("source"), "kernel": "sgk::tabq_rollout_kernel<2>",
                    "steps_per_launch": steps_per_launch, "us_per_step_in_a_launch_of": per_launch_us,
                    "fixed_us_per_launch": fixed_us, "marginal_us_per_step": marginal_us,
                    "fixed_cost_is": "loading every agent's table into LDS at entry and storing it at exit: %.0f MB in + %.0f MB out per launch"
                                     % (table_mb, table_mb),
                    "survey_8d_bytes_per_agent_step": 196, "survey_8d_gbs": 196 * n / dt / 1e9,
                    "note": "tables resident in LDS: no HBM traffic per step (8(d)'s 196 B per agent-step would be %.1f TB/s)" % (196 * n / dt / 1e12)}
        # what a caller who needs every step's observation pays instead (device time per lockstep step, 100-step hipGraphs): ONE launch
        # per step (sgk_tabq_step's kernel; round 6) with and without the boards, and the four launches of the drop-in call sequence
        per_step_api = {
            "one_launch_per_step_us": ev_time(env, lambda: agent.learn_steps(100), 10) / 100 * 1e6,
            "one_launch_per_step_with_boards_us": ev_time(env, lambda: agent.learn_steps(100, write_boards=True), 10) / 100 * 1e6,
            "four_launches_per_step_us": ev_time(env, lambda: agent.learn_steps(100, separate_launches=True), 10) / 100 * 1e6,
            "kernel": "sgk::tabq_step_kernel (act_explore + env.step + learn + reset of finished envs)",
            "bound": "HBM: ~230 B per agent-step (kept row + tag 80 B, state + record 20 B, the successor row's 128-B line, the 8-B update)"}
        out.update({"workload": "IslandNavigation + tabular-q, 262 144 private agents, fused LDS-resident rollout (1000 steps per launch)",
                    "per_step_api": per_step_api,
                    "metric": "agent-steps/s", "unit": "agent-steps/s", "value": n / dt, "us_per_lockstep_step": dt * 1e6, "dtype": "f64",
                    "roofline": roof})
        agent.close(); env.close()
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_tabq(name, args.seed)
    else:
        name, n, nh = "SideEffectsSokoban-v0", 32768, 100
        dargs = types.SimpleNamespace(discount=0.99, lr=1e-3, batch_size=64, sync_every=10000, epsilon=0.01, epsilon_anneal=100000,
                                      n_layers=2, n_hidden=nh)
        env = S.BatchedGridworldEnv(name, n, device=dev, seed=args.seed, layout="compact")
        env.bind_torch_stream()
        dq = S.BatchedDeepQAgent(env, dargs, sgd_steps=1, replay_slices=8)
        dq.warmup(8)
        dq.act_rollout(100, epsilon=0.01)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            dq.act_rollout(1000, epsilon=0.01)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3000
        def wall(fn, reps, warm=20):
            for _ in range(warm):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps

        # with learning: one lockstep iteration of dqn_learn for all envs = {replay.store(states), forward + act_explore,
        # env.step, replay.store(successors ...), ONE SGD step (sgk_dqn_sgd_step), reset_done}. Eager (six library calls from
        # Python) and replayed from one hipGraph; then each piece alone, back to back on the device (HIP events), so that
        # what the pieces do not add up to is the host / launch time between them.
        dt_learn = wall(lambda: dq.step(learn=True), 300)
        dq.enable_graphs(learn=True)
        dt_learn_graph = wall(lambda: dq.step_graphed(learn=True), 300)
        acts = dq._actions
        pieces = {
            "forward_and_act_explore": ev_time(env, lambda: env.policy_act(dq._fw, 0.01, 0, out=acts), 200),
            # env.step (+ the replay add's second half) and reset_done (+ the add's first half for the next step): the two fused
            # launches of round 6, timed as ONE piece in the order the lockstep step makes them -- timed apart and back to back most
            # timed steps would run on envs whose episode is over and every reset after the first would have nothing to reset
            "env_step_store_and_reset_done_store": ev_time(env, lambda: (dq.replay.step_store(env, acts), dq.replay.reset_store(env)), 200),
            "sgd_step": ev_time(env, lambda: dq.learn_batch(), 200),
        }
        device_sum = sum(pieces.values())
        dt_best = min(dt_learn, dt_learn_graph)
        with_learning = {
            "us_per_lockstep_step": dt_best * 1e6, "value": n / dt_best,
            "how": "three library calls (four launches: policy_act, step_store, the SGD kernel, Adam + reset_done_store) per lockstep step from Python" if dt_learn <= dt_learn_graph else "one hipGraph replay per lockstep step",
            "eager_us_per_lockstep_step": dt_learn * 1e6, "eager_value": n / dt_learn,
            "graph_us_per_lockstep_step": dt_learn_graph * 1e6, "graph_value": n / dt_learn_graph,
            "breakdown_us": {k: v * 1e6 for k, v in pieces.items()},
            "breakdown_device_sum_us": device_sum * 1e6,
            "acting_and_replay_store_us": (pieces["forward_and_act_explore"] + pieces["env_step_store_and_reset_done_store"]) * 1e6,
            "sgd_us": pieces["sgd_step"] * 1e6,
            "whole_step_minus_pieces_us": {"graph": (dt_learn_graph - device_sum) * 1e6, "eager": (dt_learn - device_sum) * 1e6},
            "note": "one SGD step (batch 64, Adam amsgrad: sgk::dqn_sgd_kernel, one 1 024-lane workgroup, 30 us + sgk::dqn_adam_kernel over the chip, "
                    "5 us; profiles/r06/dqn_learn_kernel_stats.csv, dqn_timeline.log) per lockstep step of all 32 768 envs; the reference's ratio is one SGD step per "
                    "SINGLE env-step (value.py:113-117). Each piece is timed alone, 200 calls back to back from Python: a piece reads "
                    "max(its kernel, one Python call ~ 5 us), so the pieces can add up to MORE than the whole step, whose calls overlap "
                    "the previous kernels (whole_step_minus_pieces_us < 0)."}
        nc = env.n_cells
        flops = 2.0 * (nc * nh + nh * nh + nh * 4)  # useful multiply-adds of one forward, x 2
        tf = flops * n / dt / 1e12
        out.update({"workload": "SideEffectsSokoban + deep-q (the reference's MLP %d-%d-%d-4, fp32), 32 768 envs: acting with frozen weights, "
                                "1000 x {forward + eps-greedy + env.step + auto-reset} per launch (sgk_policy_rollout)" % (nc, nh, nh),
                    "metric": "env-steps/s", "unit": "env-steps/s", "value": n / dt, "acting_only": True,
                    "value_with_learning": n / dt_best, "us_per_lockstep_step": dt * 1e6, "dtype": "f32",
                    "q_body": "mlp (the reference's DeepQAgent, value.py:148-158; BASELINE.json's wording 'conv policy' has no counterpart "
                              "in the reference's deep-q: the conv body below is a labelled non-parity option)",
                    "roofline": {"bound": "mfma", "achieved": tf, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TFLOPS,
                                 "useful_flops_per_env_step": flops, "kernel": "sgk::policy_rollout_kernel",
                                 "note": "useful multiply-adds only (the kernel pads the hidden width to MFMA tiles)"},
                    "with_learning": with_learning})
        dq = None
        # the conv Q-body (policy_cnn.py:17-81's trunk with a Q head): NOT the reference's deep-q, no parity claim. Acting: the forward +
        # act_explore in one hand-written kernel (sgk_convq_act, im2col GEMMs on fp32 MFMA); the torch / MIOpen composition timed beside it
        try:
            cq = S.BatchedDeepQAgent(env, dargs, sgd_steps=1, replay_slices=8, q_body="cnn")
            cq.warmup(8)
            dt_cnn_act = wall(lambda: cq.step(learn=False), 300, warm=30)
            dt_cnn_learn = wall(lambda: cq.step(learn=True), 100, warm=10)
            dt_cnn_kernel = wall(lambda: cq.act_explore(), 100, warm=10)
            dt_cnn_rollout = wall(lambda: cq.act_rollout(100, epsilon=0.05), 5, warm=1) / 100  # frozen weights, fixed epsilon, one launch
            C, cells = cq.n_channels, env.n_cells
            conv_flops = 2.0 * cells * (9 * C + 2 * 9 * C * C + C + 4 * C)  # multiply-adds x 2 per board: L1, L2 + head, 1x1, linear
            cnn = {"q_body": "cnn", "parity": "none (not the reference's DeepQAgent)", "n_channels": C, "fused_kernel": bool(cq.fused_conv),
                   "acting": {"us_per_lockstep_step": dt_cnn_act * 1e6, "value": n / dt_cnn_act,
                              "how": "eager, two launches: sgk_convq_act (conv forward + act_explore), sgk_step with auto-reset",
                              "forward_and_act_explore_us": dt_cnn_kernel * 1e6, "useful_flops_per_board": conv_flops,
                              "forward_tflops": n * conv_flops / dt_cnn_kernel / 1e12,
                              "rollout_us_per_lockstep_step": dt_cnn_rollout * 1e6, "rollout_value": n / dt_cnn_rollout,
                              "rollout_how": "100 lockstep steps of {conv forward, eps-greedy draw, env.step + auto-reset} per launch "
                                             "(sgk_convq_rollout: state in registers, boards in LDS)"},
                   "acting_plus_sgd": {"us_per_lockstep_step": dt_cnn_learn * 1e6, "value": n / dt_cnn_learn,
                                       "how": "the same with the replay add fused into step / reset + one torch autograd SGD step (batch 64, Adam "
                                              "amsgrad fused)"}}
            try:
                cq.enable_graphs(learn=False)
                dt_cnn_act_g = wall(lambda: cq.step_graphed(learn=False), 100, warm=10)
                cnn["acting"]["graph_us_per_lockstep_step"] = dt_cnn_act_g * 1e6
                cnn["acting"]["graph_value"] = n / dt_cnn_act_g
            except Exception as exc:  # noqa: BLE001 -- a labelled extra: its failure must not cost the config's line
                cnn["acting"]["graph_error"] = repr(exc)[:200]
            cq = None
            try:
                tq = S.BatchedDeepQAgent(env, dargs, sgd_steps=1, replay_slices=8, q_body="cnn", fused_conv=False)
                dt_t = wall(lambda: tq.step(learn=False), 100, warm=10)
                cnn["acting_torch_composition"] = {
                    "us_per_lockstep_step": dt_t * 1e6, "value": n / dt_t,
                    "how": "eager: obs cast, torch / MIOpen conv forward on all boards, sgk_epsilon_greedy, sgk_step, reset_done"}
                tq.enable_graphs(learn=False)
                dt_tg = wall(lambda: tq.step_graphed(learn=False), 100, warm=10)
                cnn["acting_torch_composition"]["graph_us_per_lockstep_step"] = dt_tg * 1e6
                tq = None
            except Exception as exc:  # noqa: BLE001
                cnn["acting_torch_composition"] = {"error": repr(exc)[:200]}
            out["conv_q_body_non_parity"] = cnn
            cq = None
        except Exception as exc:  # noqa: BLE001
            out["conv_q_body_non_parity"] = {"error": repr(exc)[:300]}
        env.close()
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_mlp(name, args.seed, nh)
    print(json.dumps(out))
    sys.stdout.flush()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20, help="timed bench steps (each --lockstep-per-step lockstep steps)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--lockstep-per-step", type=int, default=GRAPH_CHUNK,
                    help="lockstep steps in one bench step (default 100: one BoatRace episode for every env)")
    ap.add_argument("--env", default="BoatRace-v0")
    ap.add_argument("--total-envs", type=int, default=1 << 20,
                    help="env batch of the whole job, sharded over the GPUs (BASELINE.json: 1 048 576 at every GPU count)")
    ap.add_argument("--envs-per-gpu", type=int, default=0,
                    help="weak-scaling form instead: this many envs on EVERY GPU (total = N times it)")
    ap.add_argument("--layout", default=os.environ.get("SGK_BENCH_LAYOUT", "compact"), choices=["pitched", "compact"])
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x5AFE)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fused", action="store_true")
    ap.add_argument("--path", default="ring", choices=["ring", "own", "launch"],
                    help="ring: streamed rollout, every step's board + record KEPT in a 100-slice trajectory ring (default); own: "
                         "the same kernel into the env's own buffers (rewritten in place); launch: one step-kernel launch per step")
    ap.add_argument("--no-secondary", action="store_true", help="skip the measurements of the other two paths")
    ap.add_argument("--ring-backing", choices=("ring", "torch"), default="ring",
                    help="memory of the PRIMARY trajectory ring: the library's ring allocator (sgk_ring_alloc: HIP virtual memory "
                         "management, 256 MiB physical chunks) or a plain torch.empty block")
    ap.add_argument("--rings", type=int, default=5,
                    help="how many fresh primary rings the timed region is repeated on (value = the median ring; path ring only)")
    ap.add_argument("--no-weak-line", action="store_true", help="skip the secondary 1M-envs-per-GPU measurement at N > 1")
    ap.add_argument("--sustain-seconds", type=float, default=3.0,
                    help="after the timed region: the primary path's launches back to back for about this long (device time), as "
                         "`sustained` -- long enough for a coarse busy / clock sampler to see the kernel; 0 skips it")
    ap.add_argument("--config", type=int, default=0, choices=(0, 2, 3, 4),
                    help="instead of the headline: BASELINE.json's config 2 (BoatRace 65 536 envs, step kernel), 3 (IslandNavigation + "
                         "tabular-q, 262 144 agents) or 4 (Sokoban + deep-q MLP, 32 768 envs) on ONE GPU, as one JSON object with its own "
                         "roofline and cpu_baseline (tools/gpu_configs.sh collects them into profiles/rNN/configs.json)")
    args = ap.parse_args()
    if args.config:
        from safe_grid_agents_amd import dist as sdist

        sys.exit(sdist.fail_fast(lambda: run_config(args)) or 0)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))  # nothing here has initialised the GPU yet; the child's code is ours

    from safe_grid_agents_amd import dist as sdist

    # One rank's body under fail_fast: an exception on ANY rank is printed and that process leaves at once with a non-zero code --
    # torch.distributed.run then ends the job, and a rank nobody watches fails in its next collective after the process group's
    # 120 s timeout instead of parking there until the launcher's limit. No JSON line is printed unless every rank got to the end.
    code = sdist.fail_fast(lambda: run_rank(args))
    sys.exit(code or 0)


def run_rank(args):
    import torch

    import safe_grid_agents_amd as S
    from safe_grid_agents_amd import dist as sdist

    rank, local_rank, world = sdist.env_from_torchrun()
    if world != args.gpus:
        raise RuntimeError("bench.py: --gpus %d but WORLD_SIZE=%d; launch with torch.distributed.run --nproc-per-node == --gpus"
                           % (args.gpus, world))
    # test-only knobs for exercising the multi-rank control flow on a 1-GPU box: every rank on GPU 0, gloo collectives
    backend = os.environ.get("SGK_BENCH_BACKEND", "nccl")
    if os.environ.get("SGK_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    if world > 1:
        sdist.init_process_group(backend)
    import torch.distributed as tdist

    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank
    weak = args.envs_per_gpu > 0
    if weak:
        n_total = args.envs_per_gpu * world
        base, end = rank * args.envs_per_gpu, (rank + 1) * args.envs_per_gpu
    else:  # the metric's configuration: ONE batch of --total-envs, contiguous env-id blocks per rank (dist.shard_range)
        n_total = args.total_envs
        base, end = sdist.shard_range(n_total, rank, world)
    n_local = end - base

    def barrier():
        if world > 1:
            tdist.barrier()

    def max_over_ranks(*vals):
        if world == 1:
            return vals
        t = torch.tensor(vals, dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
        return tuple(float(x) for x in t)

    def gather_over_ranks(val):
        if world == 1:
            return [val]
        t = torch.tensor([val], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        out = [torch.zeros_like(t) for _ in range(world)]
        tdist.all_gather(out, t)
        return [float(x[0]) for x in out]

    def alloc_ring(e, backing):
        """A trajectory ring for env `e` from `backing` ("ring": sgk_ring_alloc; "torch": torch.empty) -> ((boards, recs), info);
        a driver without HIP virtual memory management says so and gets a plain block."""
        try:
            rb, rr, info = e.alloc_trajectory_ring(RING_SLICES, backing=backing)
        except S._lib.SgkError as err:
            sys.stderr.write("bench: sgk_ring_alloc failed (%s); this ring is a torch.empty block\n" % err)
            rb, rr, info = e.alloc_trajectory_ring(RING_SLICES, backing="torch")
        return (rb, rr), info

    # (stream="own": the handle's private stream, made torch's current stream for the timed region -- the configuration every round
    # measured; the wrapper's default would be torch's default stream, the device's NULL stream)
    env = S.BatchedGridworldEnv(args.env, n_local, device=local_rank, seed=args.seed, env_index_base=base,
                                layout=args.layout, stream="own")
    stream = env.torch_stream()
    comm = sdist.library_comm(env)  # the RCCL communicator of the metrics all-reduce is made HERE (N > 1), not inside a timed region
    sdist.require_library_comm(comm, world, backend)  # N > 1 over RCCL never passes on torch.distributed's all-reduce unnoticed
    rccl_ranks = sdist.library_comm_ranks(env) if comm is not None else None
    if comm is not None:
        sdist.global_metrics(env)  # the communicator's FIRST collective (RCCL sets its channels up in it) happens here, outside every clock
    L = max(1, args.lockstep_per_step)
    k_lock, w_lock = args.steps * L, args.warmup * L  # the timed region / the warm-up in lockstep steps
    gpu_leg_ms = 0.0
    # The rate a persistent kernel writes a multi-GB ring at depends on how the ring's physical memory is made up:
    # the same kernel measures 4.5-6.1 us per step from ring to ring. So the timed region (W warm-up + K timed steps, same bracket)
    # is repeated on --rings fresh rings of the same backing, the earlier ones held (a freed block would be handed out again);
    # `value` is the median ring. Each ring is then timed with the store-only probe: the kernel's stores and nothing else.
    ring, ring_alloc, primary = None, None, []
    held = []
    total_steps = 0
    slice_next = [0]
    n_primary = max(1, args.rings) if args.path == "ring" else 1
    for r_i in range(n_primary):
        if args.path == "ring":
            ring, ring_alloc = alloc_ring(env, args.ring_backing)
            held.append(ring)
        slice_next, ring_start = [0], total_steps
        el, kms, gm_r = timed_steps(env, k_lock, w_lock, barrier, sdist.global_metrics, path=args.path, ring=ring, slice_next=slice_next)
        per_rank = [x * 1e3 for x in gather_over_ranks(kms)]
        el, kms = max_over_ranks(el, kms)
        gpu_leg_ms += kms
        total_steps += w_lock + k_lock
        ok_r, probe_us = None, None
        if ring is not None:  # the ring really holds the last steps' outputs (checked outside every clock) ...
            ok_r = ring_parity_sample(args.env, args.seed, base, ring, total_steps - min(3, total_steps), min(3, total_steps),
                                      ring_start=ring_start)
            try:  # ... and what its memory takes from a kernel that only stores (zeros the ring: after the check)
                mine_us = env.probe_trajectory_ring(ring[0], ring[1])
            except S._lib.SgkError:
                mine_us = float("inf")  # (every rank still enters the collective below)
            probe_us = max_over_ranks(mine_us)[0]
            probe_us = None if probe_us == float("inf") else probe_us
        primary.append({"elapsed": el, "kernel_ms": kms, "gm": gm_r, "per_rank_device_us": per_rank, "ring_ok": ok_r,
                        "probe_us": probe_us, "backing": None if ring_alloc is None else ring_alloc["backing"]})
    mi = median_index([p["elapsed"] for p in primary])
    elapsed, kernel_ms, gm = primary[mi]["elapsed"], primary[mi]["kernel_ms"], primary[mi]["gm"]
    per_rank_device_us = primary[mi]["per_rank_device_us"]
    ring_ok = None if args.path != "ring" else all(p["ring_ok"] for p in primary)

    secondary = {}
    if not args.no_secondary:
        # the other two paths, same bracket, a bounded number of steps (device time per step is what these report; the host-clock
        # figure of the primary path is `value`)
        k2 = min(k_lock, 400)
        w2 = min(w_lock, 100)
        names = {"ring": "kept_in_trajectory_ring", "own": "rewritten_in_place", "launch": "per_step_launches"}
        notes = {
            "ring": "sgk_rollout_random_stream into boards [%d][n][cells] + records [%d][n]: every step's outputs KEPT (the batched "
                    "dqn_warmup, reference warmup.py:14-21); nothing is overwritten within a launch" % (RING_SLICES, RING_SLICES),
            "own": "sgk_rollout_random_stream into the env's own buffers: %d steps per launch, step k overwrites step k-1's board and "
                   "record (boards write-through, records coalesce in L2); the working set sits in the Infinity Cache" % GRAPH_CHUNK,
            "launch": "sgk_step_random: one step-kernel launch per lockstep step (hipGraph x%d), state words through memory every step"
                      % GRAPH_CHUNK,
        }
        for other in ("ring", "own", "launch"):
            if other == args.path:
                continue
            r2 = alloc_ring(env, args.ring_backing)[0] if other == "ring" else None
            o_el, o_ms, _ = timed_steps(env, k2, w2, barrier, sdist.global_metrics, path=other, ring=r2)
            o_el, o_ms = max_over_ranks(o_el, o_ms)
            del r2
            total_steps += k2 + w2
            gpu_leg_ms += o_ms
            fig = path_figures(args.env, args.layout, n_local, other, k2, o_ms)
            secondary[names[other]] = {
                "value": n_total * k2 / o_el, "unit": "env-steps/s", "lockstep_steps": k2,
                "us_per_lockstep_step": o_el * 1e6 / k2, "device_us_per_lockstep_step": o_ms * 1e3 / k2,
                "device_value": n_total * k2 / (o_ms / 1e3),
                "bound": fig["bound"], "algorithmic_bytes_per_env_step": fig["algorithmic_bytes_per_env_step"],
                # a fraction of the HBM peak only where HBM is what bounds the path; the in-place forms are held against nothing
                # here: their traffic figures say what the fabric moved
                "achieved_gbs": fig["achieved"], "frac": fig["frac"] if fig["bound"] == "hbm" else None,
                "traffic": fig["traffic"], "traffic_gbs": fig["traffic_gbs"], "traffic_frac": fig["traffic_frac"],
                "note": notes[other]}
    ring_spread = None
    if args.path == "ring" and not args.no_secondary:
        # what PLAIN rings (torch.empty: one hipMalloc block each) get in this very process, next to the primary rings' figures;
        # the earlier rings stay allocated while the next is timed
        spread_us = []
        for _ in range(3):
            r3, _info = alloc_ring(env, "torch")
            held.append(r3)
            _, s_ms, _ = timed_steps(env, 3 * GRAPH_CHUNK, GRAPH_CHUNK, barrier, sdist.global_metrics, path="ring", ring=r3)
            _, s_ms = max_over_ranks(0.0, s_ms)
            total_steps += 4 * GRAPH_CHUNK
            gpu_leg_ms += s_ms
            spread_us.append(s_ms * 1e3 / (3 * GRAPH_CHUNK))
        del r3
        ring_spread = {"device_us_per_lockstep_step": spread_us, "rings": len(spread_us), "lockstep_steps_each": 3 * GRAPH_CHUNK,
                       "backing": "torch.empty",
                       "note": "PLAIN %d-slice rings (torch.empty) allocated and timed one after another in this process with the "
                               "same kernel (device clock, max over ranks); the primary rings' own figures are in primary_rings"
                               % RING_SLICES}
    fused = None
    if not args.no_fused:
        # same workload through the outputs-once rollout kernel (state in registers, boards materialised once per launch)
        env.synchronize()
        f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        env.step_random(100, auto_reset=True, fused=True)
        f0.record(stream)
        fused_steps = 1000
        env.step_random(fused_steps, auto_reset=True, fused=True)
        f1.record(stream)
        env.synchronize()
        fms = f0.elapsed_time(f1)
        try:  # the issue ceilings of THIS box at THIS moment, right after the kernel they are held against
            peak_now = S._lib.issue_peak(local_rank)
        except S._lib.SgkError:
            peak_now = None
        total_steps += 100 + fused_steps
        gpu_leg_ms += fms
        fused = {"value": n_total * fused_steps / (fms / 1e3), "unit": "env-steps/s",
                 "ms_per_launch": fms, "steps_per_launch": fused_steps,
                 "roofline": issue_roofline(args.env, n_local, fused_steps, fms / 1e3, peak_now),
                 "note": "sgk_rollout_random: %d lockstep steps in ONE launch, outputs once at the end (no per-step observation "
                         "exists); rank 0's device time" % fused_steps}
    ok, n_checked = parity_sample(env, args.env, args.seed, base, total_steps)

    sustained = None
    if args.sustain_seconds > 0:
        # The timed region is K x 100 lockstep steps = milliseconds of device work: a one-second busy sampler sees an idle GPU. The
        # same launches, back to back for seconds, on the last primary ring (rewritten lap after lap; after the parity sample: the
        # oracle does not follow these steps) -- whether the rate holds when the chip is warm and the clocks have settled.
        k_s = int(args.sustain_seconds / (kernel_ms * 1e-3 / k_lock) / GRAPH_CHUNK + 0.5) * GRAPH_CHUNK  # (kernel_ms: max over ranks)
        k_s = max(GRAPH_CHUNK, min(k_s, 4000 * GRAPH_CHUNK))
        s_el, s_ms, _ = timed_steps(env, k_s, 0, barrier, sdist.global_metrics, path=args.path, ring=ring)
        s_el, s_ms = max_over_ranks(s_el, s_ms)
        gpu_leg_ms += s_ms
        sustained = {"value": n_total * k_s / s_el, "unit": "env-steps/s", "lockstep_steps": k_s, "seconds": s_el,
                     "us_per_lockstep_step": s_el * 1e6 / k_s, "device_us_per_lockstep_step": s_ms * 1e3 / k_s,
                     "frac": None if args.path != "ring" else algorithmic_bytes_per_env_step(args.env, args.path) * n_local
                     / (s_ms * 1e-3 / k_s) / 1e9 / HBM_PEAK_GBS,
                     "note": "the primary path's launches back to back on the last primary ring, host clock around all of them "
                             "(not part of `value`; not followed by the oracle)"}

    weak_line = None
    if world > 1 and not weak and not args.no_weak_line:
        # secondary: the weak-scaling form (1 048 576 envs on EVERY GPU), same K / W, same bracket
        env.close()
        del ring, held
        per = 1 << 20
        wenv = S.BatchedGridworldEnv(args.env, per, device=local_rank, seed=args.seed, env_index_base=rank * per,
                                     layout=args.layout, stream="own")
        wring, winfo = alloc_ring(wenv, args.ring_backing) if args.path == "ring" else (None, None)
        w_el, w_ms, _ = timed_steps(wenv, k_lock, w_lock, barrier, sdist.global_metrics, path=args.path, ring=wring)
        w_el, w_ms = max_over_ranks(w_el, w_ms)
        gpu_leg_ms += w_ms
        weak_line = {"value": per * world * k_lock / w_el, "unit": "env-steps/s", "envs_per_gpu": per,
                     "total_envs": per * world, "ms_per_step": w_el * 1e3 / args.steps,
                     "us_per_lockstep_step": w_el * 1e6 / k_lock, "device_us_per_lockstep_step": w_ms * 1e3 / k_lock,
                     "device_value": per * world * k_lock / (w_ms / 1e3), "scaling": "weak",
                     "ring_backing": None if winfo is None else winfo["backing"], "rings_timed": 1}
        del wring
        wenv.close()

    # every rank's parity sample counts: MIN over the ranks (also the last collective: every rank got here)
    mine = bool(ok) and ring_ok is not False
    ok_all = mine if world == 1 else max_over_ranks(0.0 if mine else 1.0)[0] == 0.0  # (max of "failed" over the ranks)
    if world > 1:
        tdist.barrier()
        tdist.destroy_process_group()
    if rank != 0:
        return 0 if ok_all else 3
    value = n_total * k_lock / elapsed  # the median ring's
    roofline = path_figures(args.env, args.layout, n_local, args.path, k_lock, kernel_ms)
    if args.path == "ring":
        # the same ring under a kernel that only stores (sgk_ring_probe: the streamed kernel's store instructions over every slice,
        # no env work): what THIS allocation takes, measured in this process right after the timed region
        pr = primary[mi]["probe_us"]
        roofline["store_only_probe_us_per_step"] = pr
        roofline["kernel_over_probe"] = None if not pr else roofline["device_us_per_step"] / pr
        roofline["store_only_probe_frac"] = None if not pr else roofline["algorithmic_bytes_per_env_step"] * n_local / (pr * 1e-6) / 1e9 / HBM_PEAK_GBS
    what = {
        "ring": "streaming rollout kernel (%d steps per launch, env state in registers between steps), auto-reset, every step's "
                "successor board (write-through tile stores) and step record KEPT in a %d-slice trajectory ring in HBM (%.2f GB per "
                "GPU: the batched dqn_warmup)" % (GRAPH_CHUNK, RING_SLICES, RING_SLICES * n_local * (CELLS[args.env] + 4) / 1e9),
        "own": "streaming rollout kernel (%d steps per launch, env state in registers between steps), auto-reset, every step's board "
               "written through into the env's own buffer (rewritten in place: step k overwrites step k-1; the %.0f MB working set "
               "sits in the Infinity Cache), step records coalesced in L2" % (GRAPH_CHUNK, n_local * (CELLS[args.env] + 4) / 1e6),
        "launch": "step kernel (one launch per step, hipGraph x%d), auto-reset, state words / step records / boards of the env's own "
                  "buffers rewritten by every launch" % GRAPH_CHUNK,
    }[args.path]
    out = {
        "metric": "env-steps/sec at 1M concurrent BoatRace envs" if args.env == "BoatRace-v0" else "env-steps/sec",
        "value": value,
        "unit": "env-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed * 1e3 / args.steps,
        "lockstep_steps_per_step": L, "us_per_lockstep_step": elapsed * 1e6 / k_lock,
        "higher_is_better": True,
        "scaling": "weak" if (weak or world == 1) else "strong",
        "vs_baseline": None,
        "dtype": "int8",
        "data": "synthetic",
        "config": {
            "workload": "%s random-action rollout, %d concurrent envs in lockstep (%d per GPU), %s" % (args.env, n_total, n_local, what),
            "path": args.path,
            "step": "one pass over the batch = %d lockstep steps for every env (%s)" % (
                L, "one full BoatRace episode each" if (L == 100 and args.env == "BoatRace-v0") else "--lockstep-per-step"),
            "envs_per_gpu": n_local, "total_envs": n_total, "board_layout": args.layout,
            "parallelism": "env-sharded x%d, int64 metrics all-reduce" % world,
        },
        "roofline": roofline,
        # every primary ring of this run (allocation order): host-clock rate, device time per lockstep step, the store-only probe
        # on the same ring; `value` is the median of these, nothing is picked
        "value_min": min(n_total * k_lock / p["elapsed"] for p in primary),
        "value_max": max(n_total * k_lock / p["elapsed"] for p in primary),
        "value_is": "median of %d fresh primary ring%s" % (len(primary), "" if len(primary) == 1 else "s") if args.path == "ring" else "one measurement",
        "primary_rings": [{"value": n_total * k_lock / p["elapsed"], "us_per_lockstep_step": p["elapsed"] * 1e6 / k_lock,
                           "device_us_per_lockstep_step": p["kernel_ms"] * 1e3 / k_lock,
                           "frac": None if args.path != "ring" else algorithmic_bytes_per_env_step(args.env, args.path) * n_local
                           / (p["kernel_ms"] * 1e-3 / k_lock) / 1e9 / HBM_PEAK_GBS,
                           "store_only_probe_us_per_step": p["probe_us"],
                           "kernel_over_probe": None if not p["probe_us"] else p["kernel_ms"] * 1e3 / k_lock / p["probe_us"],
                           "backing": p["backing"], "median": i == mi} for i, p in enumerate(primary)],
        # the same throughput from the max-over-ranks HIP-event time of the timed launches (no host latency in it), each rank's
        # device time, and how many ranks the library's RCCL communicator spans (None: one rank, or gloo in the CPU / one-GPU tests)
        "device_value": n_total * k_lock / (kernel_ms / 1e3),
        "per_rank_device_us": per_rank_device_us,
        "rccl_ranks": rccl_ranks,
        "metrics_collective": ("sgk_metrics_allreduced (RCCL)" if comm is not None else
                               ("torch.distributed (%s)" % backend if world > 1 else "none (one rank)")),
        # device milliseconds of ALL timed GPU work of this run (primary + secondaries + fused + weak line): why a coarse busy
        # sampler may see an idle GPU -- the rest of the wall clock is imports, allocation, parity checks and the CPU baseline
        "gpu_leg_device_ms": gpu_leg_ms,
        "episodes_finished": gm.episodes,
        "mean_return": gm.meter("returns")["avg"], "mean_safety": gm.meter("safeties")["avg"],
        "parity_sample_bit_exact": ok_all, "parity_sample_envs": n_checked, "parity_sample_ranks": world,
        "ring_slices_checked_bit_exact": ring_ok,
    }
    out.update(secondary)
    if fused:
        out["fused_rollout"] = fused
    if ring_alloc is not None:
        out["ring_allocation"] = {
            "backing": ("sgk_ring_alloc: one virtual range mapped from 256 MiB physical chunks (HIP virtual memory management); nothing "
                        "timed, nothing picked" if ring_alloc["backing"] == "ring" else "torch.empty: one hipMalloc block per ring"),
            "bytes": ring_alloc["bytes"]}
    if ring_spread:
        out["other_ring_allocations"] = ring_spread
    if sustained:
        out["sustained"] = sustained
    if weak_line:
        out["weak_1m_per_gpu"] = weak_line
    if not args.no_cpu_baseline and world == 1:  # a reported baseline of the N = 1 line only
        out["cpu_baseline"] = cpu_baseline(args.env, args.seed)
    print(json.dumps(out))
    sys.stdout.flush()
    return 0 if out["parity_sample_bit_exact"] else 3


if __name__ == "__main__":
    main()
