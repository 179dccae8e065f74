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
