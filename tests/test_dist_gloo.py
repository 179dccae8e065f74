"""The N>1 path on CPU: two processes (gloo), each owning a contiguous env-id shard, one metrics all-reduce.
Shard metrics here come from the oracle (no GPU in CI); the collective, the shard arithmetic and the
sharding-invariance of the counter RNG are exactly what the GPU job uses."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle as O
from safe_grid_agents_amd import dist as sdist
from safe_grid_agents_amd.metering import BatchMetrics

ENV, TOTAL, STEPS, SEED = "IslandNavigation-v0", 301, 150, 77


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _shard_metrics(begin, end):
    envs = O.EnvBatch(ENV, end - begin)
    m = O.metrics_new()
    envs.rollout(STEPS, seed=SEED, env_begin=begin, auto_reset=True, metrics=m)
    return envs, m


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, lr, w = sdist.init_process_group("gloo")
    assert (r, w) == (rank, world)
    begin, end = sdist.shard_range(TOTAL, rank, world)
    envs, m = _shard_metrics(begin, end)
    red = sdist.allreduce_metrics(torch.as_tensor(m))
    np.save(os.path.join(out_dir, "m%d.npy" % rank), red.numpy())
    np.save(os.path.join(out_dir, "b%d.npy" % rank), envs.boards())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions_exactly():
    for total, world in [(301, 2), (1 << 20, 8), (7, 8), (1000, 3)]:
        spans = [sdist.shard_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [e - b for b, e in spans]
        assert max(sizes) - min(sizes) <= 1


def test_two_rank_allreduce_equals_unsharded(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    whole, m = _shard_metrics(0, TOTAL)
    got = [np.load(tmp_path / ("m%d.npy" % r)) for r in range(world)]
    assert got[0].tolist() == got[1].tolist() == m.tolist()
    boards = np.concatenate([np.load(tmp_path / ("b%d.npy" % r)) for r in range(world)])
    assert (boards == whole.boards()).all()  # env-id keyed RNG: shards reproduce the unsharded streams
    bm = BatchMetrics(got[0])
    assert bm.episodes == m[O.M_EPISODES] > 0 and bm.meter("returns")["avg"] == m[O.M_SUM_RETURN] / m[O.M_EPISODES]


def test_allreduce_is_identity_without_process_group():
    v = torch.arange(16, dtype=torch.int64)
    assert sdist.allreduce_metrics(v).tolist() == v.tolist()


_FAIL_SCRIPT = """
import os, sys, time
sys.path[:0] = [%r, %r]
import torch, torch.distributed as dist
from safe_grid_agents_amd import dist as sdist

def body():
    rank, _, world = sdist.init_process_group("gloo", timeout_s=20)
    dist.barrier()  # the rendezvous worked: every rank is here
    if rank == 1:
        raise RuntimeError("rank 1 breaks after the rendezvous")
    t = torch.ones(1)
    dist.all_reduce(t)  # rank 0 waits for a rank that is gone: the timeout (or the launcher) ends it
    return 0

sys.exit(sdist.fail_fast(body) or 0)
"""


def test_a_rank_that_dies_after_the_rendezvous_ends_the_job_quickly(tmp_path):
    """A rank that raises after the rendezvous leaves at once with a non-zero code (dist.fail_fast); the launcher ends the job --
    and even an unwatched survivor fails in its next collective after the process group's timeout: the job is over in well under
    150 s with a non-zero code, instead of parking in a collective until the launcher's limit."""
    import subprocess
    import sys
    import time

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "fail_rank.py"
    script.write_text(_FAIL_SCRIPT % (root, os.path.join(root, "safe-grid-agents_amd")))
    t0 = time.time()
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(_free_port()), str(script)], capture_output=True, text=True, timeout=300)
    took = time.time() - t0
    assert p.returncode != 0 and took < 150, (p.returncode, took, p.stderr[-2000:])
    assert "rank 1 failed" in p.stderr and "rank 1 breaks after the rendezvous" in p.stderr
    # without a launcher that watches: rank 0 alone must come back by itself (collective timeout), non-zero
    env0 = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="2")
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env0, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [q.communicate(timeout=200) for q in procs]
    took = time.time() - t0
    assert procs[1].returncode == 1 and procs[0].returncode != 0 and took < 150, ([q.returncode for q in procs], took, outs[0][1][-1500:])


_LATE_SCRIPT = """
import os, sys, time
sys.path[:0] = [%r, %r]
import torch, torch.distributed as dist
from safe_grid_agents_amd import dist as sdist

rank, _, world = sdist.init_process_group("gloo", timeout_s=4)  # (rendezvous: the default 600 s)
dist.barrier()
t0 = time.time()
if rank == 1:
    time.sleep(20)  # alive, but not coming
    os._exit(0)
try:
    dist.all_reduce(torch.ones(1))
    print("returned", flush=True)
except Exception as err:
    print("failed after %%.1f s" %% (time.time() - t0), flush=True)
os._exit(0)
"""


def test_collectives_time_out_after_timeout_s_not_after_the_rendezvous_timeout(tmp_path):
    """init_process_group gives the rendezvous 600 s (ranks of a fresh box arrive minutes apart) and every collective after it
    `timeout_s`: a rank waiting for a peer that is alive but never comes fails after timeout_s."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "late_rank.py"
    script.write_text(_LATE_SCRIPT % (root, os.path.join(root, "safe-grid-agents_amd")))
    env0 = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env0, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [q.communicate(timeout=120) for q in procs]
    line = [ln for ln in outs[0][0].splitlines() if ln.startswith(("failed", "returned"))]
    assert line and line[0].startswith("failed after"), outs[0]
    assert 3.0 <= float(line[0].split()[2]) <= 15.0, line
