#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched lockstep gridworld step path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A bench "step" is one pass of the hot path over the whole env batch = one EPISODE for every env: `--lockstep-per-step`
(default 100 = BoatRace's fixed horizon, SURVEY.md 8(d) states the measurement in whole 100-step episodes) lockstep
steps, in each of which every env of every rank takes one RandomAgent action (counter RNG, in-kernel) -- transition,
observed reward, hidden safety reward, episode bookkeeping, auto-reset -- and its outputs are MATERIALISED in HBM every
lockstep step: the successor board (int8 cells, streaming tile stores) and the step record. `value` is env-steps/s
(envs x lockstep steps / wall seconds); `ms_per_step` is per bench step, `us_per_lockstep_step` is beside it. (Round 1
counted ONE lockstep step per bench step: `--lockstep-per-step 1` is that definition; at the driver's `--steps 20` its
timed region is 80 us of device work behind ~70 us of launch + synchronise latency.) Default path ("stream"): the streaming rollout kernel, 100 lockstep steps per launch with
the env state words in registers between them (sgk_rollout_random_stream into the env's own buffers: each step
overwrites the previous one's outputs, exactly what 100 per-step launches leave). `--path launch`: one step-kernel
launch per step replayed from a hipGraph (the state word makes a round trip through HBM per step); reported as a
secondary object by the default run, as is the streamed rollout into a TRAJECTORY RING (every step's boards and records
kept: the batched dqn_warmup). Workload = BASELINE.json's metric config: BoatRace, 1 048 576 concurrent
envs in the whole job at every GPU count (the batch shards by env id, one contiguous block per rank, no data-path
collective; the only exchange is one int64 metrics all-reduce at the end of the timed region). At N > 1 a secondary
object reports the weak-scaling form (1 048 576 envs on every GPU) measured in the same run.

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

B_ALG = {"BoatRace-v0": 78, "SideEffectsSokoban-v0": 100, "IslandNavigation-v0": 124, "DistributionalShift-v0": 154,
         "WhiskyGold-v0": 124, "AbsentSupervisor-v0": 124, "SafeInterruptibility-v0": 124, "ConveyorBelt-v0": 126, "TomatoWatering-v0": 154, "FriendFoe-v0": 88}  # 2*H*W + 28
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
GRAPH_CHUNK = 100      # lockstep steps per hipGraph replay


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

    a = S.prepare_parser().parse_args(["-S", "7", "-E", "40", "-EE", "1000", "-V", "100", "-EV", "0", "boat", "tabular-q",
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


def chunk_schedule(k):
    """Chunk sizes `run(k)` issues, in order: GRAPH_CHUNK-step launches (stream) / hipGraph replays (launch) and one tail."""
    out = [GRAPH_CHUNK] * (k // GRAPH_CHUNK)
    if k % GRAPH_CHUNK:
        out.append(k % GRAPH_CHUNK)
    return out


def timed_steps(env, steps, warmup, barrier, global_metrics, path="stream", ring=None, ring_layout="slice"):
    """W untimed warm-up steps, then EXACTLY `steps` lockstep steps + the metrics flush between barrier + synchronize pairs.
    Every hipGraph the timed region replays is captured and instantiated BEFORE the region (sgk_step_random_prepare for each
    chunk size of the schedule), and the first HIP event is recorded immediately before the first replay, so neither the
    host clock nor the device clock sees a capture. Returns (elapsed_s, kernel_ms, BatchMetrics)."""
    import torch

    stream = env.torch_stream()

    slice_next = [0]

    def run(k):
        for c in chunk_schedule(k):
            if ring is not None:  # every step's boards + records kept in the caller's trajectory ring
                env.rollout_random_stream(c, boards=ring[0], recs=ring[1], first_slice=slice_next[0], layout=ring_layout)
                slice_next[0] = (slice_next[0] + c) % ring[0].shape[1 if ring_layout == "tile" else 0]
            elif path == "stream":
                env.step_random(c, auto_reset=True, fused="stream")
            else:
                env.step_random(c, auto_reset=True)

    if path == "launch" and ring is None:
        for c in sorted(set(chunk_schedule(warmup) + chunk_schedule(steps))):
            env.prepare_step_random(c, auto_reset=True)
    run(warmup)
    env.metrics_reset()  # the line's episode figures (episodes_finished, mean_return, mean_safety) are the timed region's
    env.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # ---- timed region ---------------------------------------------------------------------------------------------------
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
    barrier()
    elapsed = time.perf_counter() - t0
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
    ap.add_argument("--path", default="stream", choices=["stream", "launch"],
                    help="stream: 100 lockstep steps per launch, every step's outputs materialised (default); launch: one "
                         "step-kernel launch per step (hipGraph)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the per-step-launch and trajectory-ring measurements")
    ap.add_argument("--no-weak-line", action="store_true", help="skip the secondary 1M-envs-per-GPU measurement at N > 1")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))  # nothing here has initialised the GPU yet

    import torch

    import safe_grid_agents_amd as S
    from safe_grid_agents_amd import dist as sdist

    rank, local_rank, world = sdist.env_from_torchrun()
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d; launch with torch.distributed.run --nproc-per-node == --gpus"
                 % (args.gpus, world))
    # test-only knobs for exercising the multi-rank control flow on a 1-GPU box: every rank on GPU 0, gloo collectives
    backend = os.environ.get("SGK_BENCH_BACKEND", "nccl")
    if os.environ.get("SGK_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    if world > 1:
        sdist.init_process_group(backend)
    import torch.distributed as tdist

    torch.cuda.set_device(local_rank)
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

    env = S.BatchedGridworldEnv(args.env, n_local, device=local_rank, seed=args.seed, env_index_base=base,
                                layout=args.layout)
    stream = env.torch_stream()
    sdist.library_comm(env)  # the RCCL communicator of the metrics all-reduce is made HERE (N > 1), not inside a timed region
    L = max(1, args.lockstep_per_step)
    k_lock, w_lock = args.steps * L, args.warmup * L  # the timed region / the warm-up in lockstep steps
    elapsed, kernel_ms, gm = timed_steps(env, k_lock, w_lock, barrier, sdist.global_metrics, path=args.path)
    elapsed, kernel_ms = max_over_ranks(elapsed, kernel_ms)

    total_steps = w_lock + k_lock
    secondary = {}
    if not args.no_secondary:
        # the other path and the trajectory-ring form, same bracket, a bounded number of steps (device time per step is what
        # these report; the host-clock figure of the primary path is `value`)
        k2 = min(k_lock, 400)
        w2 = min(w_lock, 100)
        other = "launch" if args.path == "stream" else "stream"
        o_el, o_ms, _ = timed_steps(env, k2, w2, barrier, sdist.global_metrics, path=other)
        o_el, o_ms = max_over_ranks(o_el, o_ms)
        total_steps += k2 + w2
        secondary["per_step_launches" if other == "launch" else "streamed"] = {
            "value": n_total * k2 / o_el, "unit": "env-steps/s", "lockstep_steps": k2,
            "us_per_lockstep_step": o_el * 1e6 / k2, "device_us_per_lockstep_step": o_ms * 1e3 / k2,
            "note": ("sgk_step_random: one step-kernel launch per lockstep step (hipGraph x%d), state words through HBM every step"
                     % GRAPH_CHUNK) if other == "launch" else "sgk_rollout_random_stream: %d steps per launch" % GRAPH_CHUNK}
        slices = GRAPH_CHUNK
        ring = (torch.empty((slices, n_local, env.n_cells), dtype=torch.int8, device="cuda:%d" % local_rank),
                torch.empty((slices, n_local, 4), dtype=torch.int8, device="cuda:%d" % local_rank))
        r_el, r_ms, _ = timed_steps(env, k2, w2, barrier, sdist.global_metrics, ring=ring)
        r_el, r_ms = max_over_ranks(r_el, r_ms)
        total_steps += k2 + w2
        secondary["streamed_into_trajectory_ring"] = {
            "value": n_total * k2 / r_el, "unit": "env-steps/s", "lockstep_steps": k2,
            "us_per_lockstep_step": r_el * 1e6 / k2, "device_us_per_lockstep_step": r_ms * 1e3 / k2, "ring_slices": slices,
            "ring_bytes": int(ring[0].numel() + ring[1].numel()),
            "note": "sgk_rollout_random_stream into boards [%d][n][cells] + records [%d][n]: every step's outputs KEPT "
                    "(the batched dqn_warmup, reference warmup.py:14-21); nothing is overwritten within a launch" % (slices, slices)}
        del ring
        # the same with the rings laid out TILE-major ([n_tiles][ring][64][...]: one contiguous run per wave and launch)
        n_tiles = (n_local + 63) // 64
        tring = (torch.empty((n_tiles, slices, 64, env.n_cells), dtype=torch.int8, device="cuda:%d" % local_rank),
                 torch.empty((n_tiles, slices, 64, 4), dtype=torch.int8, device="cuda:%d" % local_rank))
        t_el, t_ms, _ = timed_steps(env, k2, w2, barrier, sdist.global_metrics, ring=tring, ring_layout="tile")
        t_el, t_ms = max_over_ranks(t_el, t_ms)
        total_steps += k2 + w2
        secondary["streamed_into_tile_major_trajectory_ring"] = {
            "value": n_total * k2 / t_el, "unit": "env-steps/s", "lockstep_steps": k2,
            "us_per_lockstep_step": t_el * 1e6 / k2, "device_us_per_lockstep_step": t_ms * 1e3 / k2, "ring_slices": slices,
            "note": "SGK_F_RING_TILE_MAJOR: boards [n_tiles][%d][64][cells] + records [n_tiles][%d][64]" % (slices, slices)}
        del tring
    fused = None
    if not args.no_fused:
        # same workload through the fused rollout kernel (state in registers, boards materialised once per launch)
        env.synchronize()
        f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        env.step_random(100, auto_reset=True, fused=True)
        f0.record(stream)
        fused_steps = 1000
        env.step_random(fused_steps, auto_reset=True, fused=True)
        f1.record(stream)
        env.synchronize()
        fms = f0.elapsed_time(f1)
        total_steps += 100 + fused_steps
        fused = {"value": n_total * fused_steps / (fms / 1e3), "unit": "env-steps/s",
                 "ms_per_launch": fms, "steps_per_launch": fused_steps,
                 "note": "sgk_rollout_random: %d lockstep steps in ONE launch; rank 0's device time" % fused_steps}
    ok, n_checked = parity_sample(env, args.env, args.seed, base, total_steps)

    weak_line = None
    if world > 1 and not weak and not args.no_weak_line:
        # secondary: the weak-scaling form (1 048 576 envs on EVERY GPU), same K / W, same bracket
        env.close()
        per = 1 << 20
        wenv = S.BatchedGridworldEnv(args.env, per, device=local_rank, seed=args.seed, env_index_base=rank * per,
                                     layout=args.layout)
        w_el, w_ms, _ = timed_steps(wenv, k_lock, w_lock, barrier, sdist.global_metrics, path=args.path)
        w_el, w_ms = max_over_ranks(w_el, w_ms)
        weak_line = {"value": per * world * k_lock / w_el, "unit": "env-steps/s", "envs_per_gpu": per,
                     "total_envs": per * world, "ms_per_step": w_el * 1e3 / args.steps,
                     "us_per_lockstep_step": w_el * 1e6 / k_lock, "device_us_per_lockstep_step": w_ms * 1e3 / k_lock,
                     "scaling": "weak"}
        wenv.close()

    if world > 1:
        tdist.barrier()
        tdist.destroy_process_group()
    if rank != 0:
        return
    value = n_total * k_lock / elapsed
    launches = len(chunk_schedule(k_lock)) if args.path == "stream" else k_lock
    steps_per_launch = k_lock / launches
    launch_s = kernel_ms / 1e3 / launches  # average duration of one launch of the dominant kernel incl. its launch gap
    b_alg = B_ALG[args.env]
    achieved = b_alg * n_local * steps_per_launch / launch_s / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        with open(tpath) as f:
            tj = json.load(f)
        if args.path == "stream":  # measured per 100-step launch; scaled to this run's steps per launch
            per100 = tj.get("%s/%s/%d/stream%d" % (args.env, args.layout, n_local, GRAPH_CHUNK))
            traffic = None if per100 is None else per100 * steps_per_launch / GRAPH_CHUNK
        else:
            traffic = tj.get("%s/%s/%d" % (args.env, args.layout, n_local))
    roofline = {
        "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
        "traffic": traffic,
        # what the fabric really moved (rocprofv3 FETCH_SIZE x calibration + WRITE_SIZE per launch, profiles/traffic.json)
        # over the same launch time: the UTILISATION figure. `frac` is algorithmic (2 H W + 28 bytes per env-step,
        # which charges a board read this design never makes) and can exceed 1.
        "traffic_gbs": None if traffic is None else traffic / launch_s / 1e9,
        "traffic_frac": None if traffic is None else traffic / launch_s / 1e9 / HBM_PEAK_GBS,
        "kernel": ("sgk::rollout_random_kernel<%s, stream>" if args.path == "stream" else "sgk::step_kernel<%s>") % args.env,
        "algorithmic_bytes_per_env_step": b_alg, "steps_per_launch": steps_per_launch,
        "algorithmic_bytes_per_launch": b_alg * n_local * steps_per_launch,
        "avg_launch_us": launch_s * 1e6, "device_us_per_step": launch_s * 1e6 / steps_per_launch,
        "frac_of_measured_copy_peak_6290": achieved / 6290.0,
    }
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
            "workload": ("%s random-action rollout, %d concurrent envs in lockstep (%d per GPU), " % (args.env, n_total, n_local))
                        + ("streaming rollout kernel (%d steps per launch, env state in registers between steps)" % GRAPH_CHUNK
                           if args.path == "stream" else "step kernel (one launch per step, hipGraph x%d)" % GRAPH_CHUNK)
                        + ", auto-reset, every step's board and step record materialised in HBM",
            "path": args.path,
            "step": "one pass over the batch = %d lockstep steps for every env (%s)" % (
                L, "one full BoatRace episode each" if (L == 100 and args.env == "BoatRace-v0") else "--lockstep-per-step"),
            "envs_per_gpu": n_local, "total_envs": n_total, "board_layout": args.layout,
            "parallelism": "env-sharded x%d, int64 metrics all-reduce" % world,
        },
        "roofline": roofline,
        "episodes_finished": gm.episodes,
        "mean_return": gm.meter("returns")["avg"], "mean_safety": gm.meter("safeties")["avg"],
        "parity_sample_bit_exact": ok, "parity_sample_envs": n_checked,
    }
    out.update(secondary)
    if fused:
        out["fused_rollout"] = fused
    if weak_line:
        out["weak_1m_per_gpu"] = weak_line
    if not args.no_cpu_baseline and world == 1:  # a reported baseline of the N = 1 line only
        out["cpu_baseline"] = cpu_baseline(args.env, args.seed)
    print(json.dumps(out))
    if not ok:
        sys.exit(3)


if __name__ == "__main__":
    main()
