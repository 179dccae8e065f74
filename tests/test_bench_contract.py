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
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    per_launch = r["algorithmic_bytes_per_env_step"] * d["config"]["envs_per_gpu"] * r["steps_per_launch"]
    assert r["algorithmic_bytes_per_launch"] == per_launch
    assert abs(r["achieved"] - per_launch / (r["avg_launch_us"] * 1e-6) / 1e9) <= 1e-6 * r["achieved"]
    assert "traffic" in r and (r["traffic"] is None) == (r["traffic_frac"] is None)
    assert d["parity_sample_bit_exact"] is True and d["parity_sample_envs"] > 0


@pytest.mark.gpu
def test_bench_line_one_rank():
    d = _line(["--steps", "3", "--warmup", "1", "--total-envs", "8192", "--lockstep-per-step", "100"])
    _check_contract(d, 1, 3, 1)
    assert d["scaling"] == "weak" and d["config"]["envs_per_gpu"] == 8192 and d["roofline"]["steps_per_launch"] == 100
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "env-steps/s" and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    for k in ("per_step_launches", "streamed_into_trajectory_ring", "streamed_into_tile_major_trajectory_ring", "fused_rollout"):
        assert d[k]["value"] > 0
    # the other path as the primary one, and round 1's step definition
    d2 = _line(["--steps", "40", "--warmup", "10", "--total-envs", "8192", "--path", "launch", "--lockstep-per-step", "1",
                "--no-cpu-baseline", "--no-fused"])
    _check_contract(d2, 1, 40, 10)
    assert d2["roofline"]["steps_per_launch"] == 1 and "streamed" in d2 and "cpu_baseline" not in d2


@pytest.mark.gpu
def test_bench_line_two_ranks_on_one_gpu_shards_the_batch():
    """`python bench.py --gpus 2` starts its ranks itself (torch.distributed.run as a child, before any GPU call); the batch is
    cut into two contiguous env-id blocks; the line reports the whole job."""
    d = _line(["--gpus", "2", "--steps", "3", "--warmup", "1", "--total-envs", "8192", "--no-fused"],
              env={"SGK_BENCH_BACKEND": "gloo", "SGK_BENCH_ONE_DEVICE": "1"})
    _check_contract(d, 2, 3, 1)
    assert d["scaling"] == "strong" and d["config"]["envs_per_gpu"] == 4096 and d["config"]["total_envs"] == 8192
    assert d["weak_1m_per_gpu"]["total_envs"] == 2 << 20 and "cpu_baseline" not in d
    assert d["episodes_finished"] == 8192 * 3  # every env of both shards finished one episode per bench step
