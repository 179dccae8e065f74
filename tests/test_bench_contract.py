"""bench.py's contract (the task statement's JSON line + the `roofline` / `cpu_baseline` objects): the pure parts on the CPU, the
line itself on a GPU at a size that takes seconds -- one rank, and two ranks on the one GPU of a test box (gloo collectives: the
multi-rank control flow, sharding included; RCCL itself needs two GPUs)."""
import importlib.util
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_module():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_chunk_schedule_covers_exactly_the_requested_steps():
    b = _bench_module()
    for k in (0, 1, 5, 20, 99, 100, 101, 250, 2000):
        sched = b.chunk_schedule(k)
        assert sum(sched) == k and all(0 < c <= b.GRAPH_CHUNK for c in sched)
        assert sched[:-1] == [b.GRAPH_CHUNK] * (len(sched) - 1) if sched else k == 0
    # algorithmic bytes per env-step: SURVEY.md 8(d)'s 2 H W + 28 for every level
    from oracle import oracle as O

    for name in O.ENV_IDS:
        H, W = O.shape(O.ENV_IDS[name])
        assert b.B_ALG[name] == 2 * H * W + 28, name
        # what a kernel of each form cannot avoid moving: the kept outputs (board + packed record); + the state word's round trip
        # when every step is a launch
        assert b.algorithmic_bytes_per_env_step(name, "ring") == b.algorithmic_bytes_per_env_step(name, "own") == H * W + 4
        assert b.algorithmic_bytes_per_env_step(name, "launch") == H * W + 4 + 16


def test_a_multi_gpu_run_over_rccl_never_passes_on_the_torch_fallback_unnoticed(monkeypatch):
    """dist.require_library_comm: more than one rank on an RCCL-backed group without the library's communicator is an error (bench.py
    calls it right after library_comm), unless SGK_METRICS_COLLECTIVE=torch asks for torch.distributed's all-reduce."""
    from safe_grid_agents_amd import dist as sdist

    monkeypatch.delenv("SGK_METRICS_COLLECTIVE", raising=False)
    monkeypatch.delenv("SGK_BENCH_REQUIRE_RCCL", raising=False)
    with pytest.raises(RuntimeError, match="sgk_comm_create"):
        sdist.require_library_comm(None, 8, "nccl")
    sdist.require_library_comm(object(), 8, "nccl")  # the communicator exists
    sdist.require_library_comm(None, 1, "nccl")      # one rank: nothing to reduce across
    sdist.require_library_comm(None, 2, "gloo")      # the CPU tests' backend: torch.distributed is the collective there
    monkeypatch.setenv("SGK_BENCH_REQUIRE_RCCL", "1")
    with pytest.raises(RuntimeError):
        sdist.require_library_comm(None, 2, "gloo")
    monkeypatch.setenv("SGK_METRICS_COLLECTIVE", "torch")
    sdist.require_library_comm(None, 8, "nccl")
    sdist.require_library_comm(None, 2, "gloo")
    assert "sdist.require_library_comm(comm, world, backend)" in open(os.path.join(ROOT, "bench.py")).read()


def _line(args, env=None, timeout=600):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout,
                       env=dict(os.environ, **(env or {})), cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]  # ONE JSON line on stdout
    return json.loads(lines[0])


def _check_contract(d, n_gpus, steps, warmup):
    assert d["metric"].startswith("env-steps/sec") and d["unit"] == "env-steps/s" and d["higher_is_better"] is True
    assert (d["n_gpus"], d["steps"], d["warmup"]) == (n_gpus, steps, warmup)
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "int8"
    assert d["scaling"] in ("weak", "strong") and "workload" in d["config"] and "model" not in d["config"]
    L = d["lockstep_steps_per_step"]
    total_envs = d["config"]["total_envs"]
    # value = units all ranks processed / the timed region
    assert abs(d["value"] - total_envs * steps * L / (d["ms_per_step"] * steps / 1e3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    path = d["config"]["path"]
    assert r["bound"] == ("hbm" if path == "ring" else "fabric / infinity-cache write") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    if path == "ring":
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    else:  # buffers rewritten in place never leave the Infinity Cache: no fraction of the HBM peak is claimed for them
        assert r["frac"] is None
    assert ("KEPT in a 100-slice trajectory ring" in d["config"]["workload"]) == (path == "ring")
    # the device-clock form of `value`, per rank, and the GPU leg's total
    assert len(d["per_rank_device_us"]) == n_gpus and all(x > 0 for x in d["per_rank_device_us"])
    assert abs(d["device_value"] - total_envs * steps * L / (max(d["per_rank_device_us"]) * 1e-6)) <= 1e-6 * d["device_value"]
    assert d["device_value"] >= d["value"] and d["gpu_leg_device_ms"] >= max(d["per_rank_device_us"]) / 1e3
    per_launch = r["algorithmic_bytes_per_env_step"] * d["config"]["envs_per_gpu"] * r["steps_per_launch"]
    assert r["algorithmic_bytes_per_launch"] == per_launch
    assert abs(r["achieved"] - per_launch / (r["avg_launch_us"] * 1e-6) / 1e9) <= 1e-6 * r["achieved"]
    assert "traffic" in r and (r["traffic"] is None) == (r["traffic_frac"] is None)
    assert d["parity_sample_bit_exact"] is True and d["parity_sample_envs"] > 0
    # `value` is the median of the primary rings of THIS run; every fraction of the line follows from numbers inside the line
    pr = d["primary_rings"]
    assert d["value_min"] <= d["value"] <= d["value_max"] and sum(1 for p in pr if p["median"]) == 1
    assert len(pr) == (5 if path == "ring" else 1) and [p for p in pr if p["median"]][0]["value"] == d["value"]
    assert sorted(p["value"] for p in pr)[(len(pr) - 1) // 2] == d["value"]
    if path == "ring":
        assert all(p["store_only_probe_us_per_step"] > 0 and p["kernel_over_probe"] > 0 and 0 < p["frac"] for p in pr)
        assert r["store_only_probe_us_per_step"] > 0
        assert abs(r["kernel_over_probe"] - r["device_us_per_step"] / r["store_only_probe_us_per_step"]) < 1e-9
        mp = [p for p in pr if p["median"]][0]
        assert abs(mp["frac"] - r["frac"]) < 1e-9 and mp["store_only_probe_us_per_step"] == r["store_only_probe_us_per_step"]


@pytest.mark.gpu
def test_bench_line_one_rank():
    d = _line(["--steps", "3", "--warmup", "1", "--total-envs", "8192", "--lockstep-per-step", "100"])
    _check_contract(d, 1, 3, 1)
    assert d["scaling"] == "weak" and d["config"]["envs_per_gpu"] == 8192 and d["roofline"]["steps_per_launch"] == 100
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "env-steps/s" and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    assert d["config"]["path"] == "ring" and d["ring_slices_checked_bit_exact"] is True and d["rccl_ranks"] is None
    for k in ("per_step_launches", "rewritten_in_place"):  # every secondary object carries its own counter-based figures
        assert d[k]["value"] > 0 and d[k]["device_value"] > 0 and d[k]["bound"] == "fabric / infinity-cache write"
        assert {"traffic", "traffic_gbs", "traffic_frac", "frac", "algorithmic_bytes_per_env_step"} <= set(d[k]) and d[k]["frac"] is None
    assert d["fused_rollout"]["value"] > 0 and "roofline" in d["fused_rollout"]
    fr = d["fused_rollout"]["roofline"]  # issue-bound: the peak is measured in the same process (sgk_issue_peak), the committed one kept beside it
    assert fr["peak_source"].startswith("sgk_issue_peak") and fr["peak_in_run"]["valu"] > 0 and fr["peak_in_run"]["salu"] > 0
    assert abs(fr["frac"] - max(fr["valu_frac"], fr["salu_frac"])) < 1e-12 and fr["peak_committed"]["frac"] > 0
    assert fr["peak"] in (fr["peak_in_run"]["valu"], fr["peak_in_run"]["salu"])
    assert d["ring_allocation"]["backing"].startswith("sgk_ring_alloc") and d["ring_allocation"]["bytes"] == 100 * 8192 * 29
    spread = d["other_ring_allocations"]  # what other fresh rings get in the same process: the allocation lottery, shown
    assert spread["rings"] == 3 and len(spread["device_us_per_lockstep_step"]) == 3 and min(spread["device_us_per_lockstep_step"]) > 0
    su = d["sustained"]  # the same launches back to back for seconds (capped at 4 000 launches: this batch is small), not `value`
    assert su["lockstep_steps"] % 100 == 0 and 100 <= su["lockstep_steps"] <= 400000 and su["frac"] > 0
    assert abs(su["value"] - 8192 * su["lockstep_steps"] / su["seconds"]) <= 1e-6 * su["value"]
    assert su["device_us_per_lockstep_step"] <= su["us_per_lockstep_step"] and d["gpu_leg_device_ms"] >= su["device_us_per_lockstep_step"] * su["lockstep_steps"] / 1e3
    # another path as the primary one, and round 1's step definition
    d2 = _line(["--steps", "40", "--warmup", "10", "--total-envs", "8192", "--path", "launch", "--lockstep-per-step", "1",
                "--no-cpu-baseline", "--no-fused", "--sustain-seconds", "0.05"])
    _check_contract(d2, 1, 40, 10)
    assert d2["roofline"]["steps_per_launch"] == 1 and "cpu_baseline" not in d2 and d2["sustained"]["frac"] is None
    assert d2["kept_in_trajectory_ring"]["bound"] == "hbm" and d2["rewritten_in_place"]["value"] > 0
    d3 = _line(["--steps", "3", "--warmup", "1", "--total-envs", "8192", "--path", "own", "--no-cpu-baseline", "--no-fused",
                "--no-secondary", "--sustain-seconds", "0"])
    _check_contract(d3, 1, 3, 1)
    assert "sustained" not in d3
    d4 = _line(["--steps", "3", "--warmup", "1", "--total-envs", "8192", "--ring-backing", "torch", "--no-cpu-baseline", "--no-fused",
                "--no-secondary"])
    _check_contract(d4, 1, 3, 1)
    assert d4["ring_allocation"]["backing"].startswith("torch.empty")


@pytest.mark.gpu
def test_bench_line_two_ranks_on_one_gpu_shards_the_batch():
    """`python bench.py --gpus 2` starts its ranks itself (torch.distributed.run as a child, before any GPU call); the batch is
    cut into two contiguous env-id blocks; the line reports the whole job."""
    d = _line(["--gpus", "2", "--steps", "3", "--warmup", "1", "--total-envs", "8192", "--no-fused", "--sustain-seconds", "0.2"],
              env={"SGK_BENCH_BACKEND": "gloo", "SGK_BENCH_ONE_DEVICE": "1"})
    _check_contract(d, 2, 3, 1)
    assert d["scaling"] == "strong" and d["config"]["envs_per_gpu"] == 4096 and d["config"]["total_envs"] == 8192
    assert d["weak_1m_per_gpu"]["total_envs"] == 2 << 20 and "cpu_baseline" not in d
    assert d["episodes_finished"] == 8192 * 3  # every env of both shards finished one episode per bench step
    assert d["metrics_collective"] == "torch.distributed (gloo)" and d["rccl_ranks"] is None and d["sustained"]["value"] > 0


@pytest.mark.gpu
def test_bench_exits_non_zero_when_the_librarys_communicator_is_missing():
    """The exit path of require_library_comm end to end, with the failure stubbed: two ranks on the one GPU (gloo: no RCCL
    communicator can exist) and SGK_BENCH_REQUIRE_RCCL=1 -> no JSON line, a non-zero exit code, the reason on stderr; the same
    launch with SGK_METRICS_COLLECTIVE=torch goes through."""
    args = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--total-envs", "8192", "--no-fused", "--no-secondary", "--no-weak-line",
            "--no-cpu-baseline", "--sustain-seconds", "0", "--rings", "1"]
    env = dict(os.environ, SGK_BENCH_BACKEND="gloo", SGK_BENCH_ONE_DEVICE="1", SGK_BENCH_REQUIRE_RCCL="1")
    env.pop("SGK_METRICS_COLLECTIVE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode != 0 and not [ln for ln in p.stdout.splitlines() if ln.startswith("{")], p.stdout[-500:]
    assert "sgk_comm_create" in p.stderr and "SGK_METRICS_COLLECTIVE=torch" in p.stderr, p.stderr[-1500:]
    d = _line(args, env={"SGK_BENCH_BACKEND": "gloo", "SGK_BENCH_ONE_DEVICE": "1", "SGK_BENCH_REQUIRE_RCCL": "1", "SGK_METRICS_COLLECTIVE": "torch"})
    assert d["n_gpus"] == 2 and d["rccl_ranks"] is None


@pytest.mark.gpu
def test_bench_line_eight_ranks_dry_run_on_one_gpu():
    """The driver's 8-GPU launch, rehearsed on the one GPU of a test box (gloo collectives): eight ranks, eight contiguous env-id
    shards, one line with eight device times (what an 8-GPU node adds is RCCL and seven more devices, not control flow)."""
    d = _line(["--gpus", "8", "--steps", "2", "--warmup", "1", "--total-envs", "16384", "--no-fused", "--no-weak-line", "--sustain-seconds", "0.2"],
              env={"SGK_BENCH_BACKEND": "gloo", "SGK_BENCH_ONE_DEVICE": "1"}, timeout=900)
    _check_contract(d, 8, 2, 1)
    assert d["config"]["envs_per_gpu"] == 2048 and d["scaling"] == "strong" and d["episodes_finished"] == 16384 * 2


@pytest.mark.gpu
@pytest.mark.parametrize("k", [2, 3, 4])
def test_bench_config_objects_carry_their_own_roofline_and_cpu_baseline(k):
    """`bench.py --config k` (BASELINE.json configs 2-4 on one GPU): one JSON object with value, the dominant kernel's roofline --
    every fraction following from numbers inside the object -- and a CPU baseline on the host cores."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", str(k)], capture_output=True, text=True, timeout=900,
                       cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["config"] == k and d["value"] > 0 and d["n_gpus"] == 1 and d["data"] == "synthetic"
    n = {2: 65536, 3: 262144, 4: 32768}[k]
    assert abs(d["value"] - n / (d["us_per_lockstep_step"] * 1e-6)) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == {2: "hbm", 3: "valu-issue", 4: "mfma"}[k] and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1.5
    if k == 2:
        assert abs(r["achieved"] - r["algorithmic_bytes_per_env_step"] * d["value"] / 1e9) <= 1e-6 * r["achieved"]
    if k == 3:  # instructions per wave-step (committed SQ pass) x this run's wave-steps per second, against the CHIP's in-run issue peak
        assert abs(r["achieved"] - r["valu_per_wave_step"] * (d["value"] / 64) / 1e9) <= 1e-6 * r["achieved"]
        # frac is the chip-level figure (8 waves per SIMD); the one against the kernel's own residency is the larger, secondary one
        assert "8 waves per SIMD" in r["peak_is"] and r["frac"] < r["frac_at_kernel_occupancy"] < 1.2
        assert abs(r["frac_at_kernel_occupancy"] - r["achieved"] / r["peak_at_kernel_occupancy"]) < 1e-9
        # the per-launch fixed cost (table load + store) explains why a shorter launch costs more per step
        assert r["fixed_us_per_launch"] > 0 and r["marginal_us_per_step"] > 0
        per = r["us_per_step_in_a_launch_of"]
        assert per["250"] > per["2000"]
        for K in (250, 1000):  # t(K) = fixed + K * marginal, fitted on K = 500 and 2000, predicts the other two within 10 %
            assert abs(per[str(K)] - (r["fixed_us_per_launch"] / K + r["marginal_us_per_step"])) <= 0.1 * per[str(K)]
    if k == 4:
        assert abs(r["achieved"] - r["useful_flops_per_env_step"] * d["value"] / 1e12) <= 1e-6 * r["achieved"] and r["peak"] == 157.3
        assert d["acting_only"] is True and 0 < d["value_with_learning"] < d["value"]
        w = d["with_learning"]
        assert w["value"] == d["value_with_learning"] == max(w["eager_value"], w["graph_value"]) > 0
        b = w["breakdown_us"]
        assert set(b) == {"forward_and_act_explore", "env_step_store_and_reset_done_store", "sgd_step"}
        assert abs(w["breakdown_device_sum_us"] - sum(b.values())) < 1e-6 and abs(w["acting_and_replay_store_us"] + w["sgd_us"] - sum(b.values())) < 1e-6
        c = d["conv_q_body_non_parity"]
        assert c["parity"].startswith("none") and c["acting"]["value"] > 0 and c["acting_plus_sgd"]["value"] > 0
        assert c["fused_kernel"] is True and c["acting"]["rollout_value"] > c["acting"]["value"]  # one launch beats two per step
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["sample"]


def test_check_scale_reads_bench_lines_and_says_pass_or_fail():
    """tools/check_scale.py -- the verdict for a first `bench.py --gpus N` run against BASELINE.md section 10 -- on the committed
    lines: the one-GPU gloo dry runs of 2 and 8 ranks FAIL as multi-GPU results (no RCCL communicator, a torch.distributed
    collective, a value far below the prediction) and PASS the checks that still mean something with --allow-collective gloo; a
    synthetic 8-GPU line built to the prediction passes, and each single defect fails it."""
    import copy
    import importlib.util
    import io
    from contextlib import redirect_stdout

    spec = importlib.util.spec_from_file_location("check_scale", os.path.join(ROOT, "tools", "check_scale.py"))
    cs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cs)
    logs = [os.path.join(ROOT, "profiles", "r05", "bench_%drank_one_gpu_gloo.log" % k) for k in (2, 8)]

    def run(argv):
        buf = io.StringIO()
        with redirect_stdout(buf):
            rc = cs.main(argv)
        return rc, buf.getvalue()

    rc, out = run(logs)
    assert rc == 1 and out.count("FAIL n_gpus=") == 2 and "SCALE FAIL (2 lines)" in out
    assert "collective FAIL (torch.distributed (gloo))" in out and "ranks FAIL" in out and "episodes ok" in out and "parity ok" in out
    rc, out = run(["--allow-collective", "gloo"] + logs)
    assert rc == 0 and "SCALE PASS (2 lines)" in out and "value n/a" in out
    # a line as BASELINE section 10 predicts it for eight GPUs
    base = cs.load_lines(logs[1])[0]
    good = copy.deepcopy(base)
    good.update(value=1.6e12, rccl_ranks=8, metrics_collective=cs.LIBRARY_COLLECTIVE, per_rank_device_us=[62.0 + 0.2 * i for i in range(8)])
    ok, res = cs.check(good)
    assert ok and all(r is True for _, r, _ in res), res
    for defect in ({"rccl_ranks": 4}, {"metrics_collective": "torch.distributed (nccl)"}, {"episodes_finished": good["episodes_finished"] - 1},
                   {"parity_sample_bit_exact": False}, {"per_rank_device_us": [60.0] * 7 + [66.0]}, {"value": 1.2e12}, {"value": 2.0e12}):
        bad = dict(good, **defect)
        ok, res = cs.check(bad)
        assert not ok and sum(r is False for _, r, _ in res) == 1, (defect, res)
    # a driver record keeps only the contract's keys under "parsed": judged on what it holds
    rc, out = run([os.path.join(ROOT, "BENCH_r05.json")])
    assert rc == 0 and "value ok" in out and "episodes n/a (not in this record)" in out
