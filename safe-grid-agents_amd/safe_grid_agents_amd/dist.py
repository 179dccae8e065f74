"""Sharding of the env batch across the GPUs of one node and the one collective the path has.

Every env (and its private agent) is independent -- the reference is literally one env + one agent per process
(reference train.py:51-54) -- so the batch splits into contiguous env-id blocks, one block per rank / GPU, and no
board, state or Q-table byte ever crosses xGMI. The only exchange is the aggregate episode metrics: the fields of the
four AverageMeters track_metrics maintains (reference meters.py:20-35,58-63,76-84) as a 16-word int64 vector, summed
([0..7]) and max-ed ([8..11]) across ranks. With backend "nccl" that is one RCCL all-reduce of 64 B + one of 32 B over
xGMI: pure latency, issued once per metrics flush, never per step. Integer sums/maxima make the result independent of
the number of ranks. The counter RNG is keyed by GLOBAL env index (env_index_base), so a sharded run reproduces the
unsharded action streams exactly.
"""
import os

import numpy as np

from .metering import BatchMetrics, METRICS_LEN


def shard_range(total_envs, rank, world_size):
    """Contiguous block [begin, end) of env ids owned by `rank`; remainders go to the lowest ranks."""
    base, rem = divmod(int(total_envs), int(world_size))
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def env_from_torchrun():
    """(rank, local_rank, world_size) from the torch.distributed.run environment; (0, 0, 1) when not launched by it."""
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)))


def init_process_group(backend=None, timeout_s=None, rendezvous_s=None):
    """One process per GPU. backend 'nccl' IS RCCL on ROCm; 'gloo' for the CPU tests. Collectives time out after `timeout_s`
    seconds (default 120, SGK_DIST_TIMEOUT_S overrides): the path's only exchange is a 96-byte all-reduce, so a rank that waits
    longer than that is waiting for a rank that died -- it fails instead of parking the job until a launcher's limit. The
    RENDEZVOUS gets `rendezvous_s` (default 600, SGK_DIST_RENDEZVOUS_S; never less than timeout_s): ranks of a fresh box reach it
    minutes apart while the image pages in, and nobody has died yet."""
    import datetime

    import torch
    import torch.distributed as dist

    rank, local_rank, world = env_from_torchrun()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl" and os.environ.get("SGK_BENCH_ONE_DEVICE") != "1":
            torch.cuda.set_device(local_rank)
        if timeout_s is None:
            timeout_s = float(os.environ.get("SGK_DIST_TIMEOUT_S", "120"))
        if rendezvous_s is None:
            rendezvous_s = float(os.environ.get("SGK_DIST_RENDEZVOUS_S", "600"))
        rendezvous_s = max(float(rendezvous_s), float(timeout_s))
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=rendezvous_s))
        if rendezvous_s > timeout_s:
            # every rank is here: from now on a wait is a wait for a dead rank. (A private torch call -- there is no public one that
            # changes a live group's timeout; a torch without it keeps the rendezvous' figure and says so.)
            try:
                from torch.distributed.distributed_c10d import _set_pg_timeout

                _set_pg_timeout(datetime.timedelta(seconds=timeout_s), dist.group.WORLD)
            except Exception as err:  # noqa: BLE001
                import sys

                sys.stderr.write("safe_grid_agents_amd.dist: collectives keep the rendezvous timeout of %.0f s (%s)\n" % (rendezvous_s, err))
    return rank, local_rank, world


def fail_fast(body):
    """Run one rank's `body()`; on ANY exception print it to stderr and leave the process at once with a non-zero code -- no
    interpreter shutdown, no process-group destructor that would wait for the other ranks. A launcher that watches its children
    (torch.distributed.run does) then ends the job; ranks it does not watch run into the collective timeout of
    init_process_group. Returns body()'s value otherwise."""
    import sys
    import traceback

    try:
        return body()
    except SystemExit:
        raise
    except BaseException:  # noqa: BLE001  (a rank must not outlive its failure, whatever it was)
        rank = os.environ.get("RANK", "0")
        sys.stderr.write("rank %s failed:\n%s" % (rank, traceback.format_exc()))
        sys.stderr.flush()
        sys.stdout.flush()
        os._exit(1)


def allreduce_metrics(vec):
    """All-reduce a metrics vector (torch int64 tensor of METRICS_LEN, on the GPU for nccl / on the CPU for gloo).

    Returns a new tensor; a no-op copy when torch.distributed is not initialised (single GPU)."""
    import torch
    import torch.distributed as dist

    out = vec.clone()
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return out
    sums = out[:8].contiguous()
    maxs = out[8:12].contiguous()
    dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    dist.all_reduce(maxs, op=dist.ReduceOp.MAX)
    out[:8] = sums
    out[8:12] = maxs
    return out


_COMMS = {}  # device -> sgk_comm* (one RCCL communicator per process and GPU, made on first use)


def library_comm(env):
    """The C-ABI's RCCL communicator (sgk_comm_create) for this process: rank 0 draws the unique id, torch.distributed ships
    its 128 bytes to the other ranks, every rank creates its end on its own GPU. None when the process group is not RCCL-backed
    (gloo tests, single rank) or the library cannot set one up -- the caller then reduces through torch.distributed, which is
    the same RCCL underneath. SGK_METRICS_COLLECTIVE=torch forces that path."""
    import ctypes
    import sys

    import torch
    import torch.distributed as dist

    from . import _lib

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1 or dist.get_backend() != "nccl":
        return None
    if os.environ.get("SGK_METRICS_COLLECTIVE", "sgk") == "torch":
        return None
    if env.device in _COMMS:
        return _COMMS[env.device]
    lib = _lib.load()
    rank, world = dist.get_rank(), dist.get_world_size()
    # Every rank first checks that librccl loads and resolves (sgk_comm_available: side-effect free), rank 0 also draws the id
    # (ncclGetUniqueId opens a bootstrap listener and a thread per call: only the rank whose id is used makes one). All ranks
    # agree on the outcome before anyone enters ncclCommInitRank -- a rank that cannot load RCCL would return at once and leave the
    # others waiting in the rendezvous.
    ident = (ctypes.c_uint8 * _lib.COMM_ID_BYTES)()
    mine = lib.sgk_comm_unique_id(ident) if rank == 0 else lib.sgk_comm_available(None)
    ok = torch.tensor([int(mine == _lib.SGK_OK)], dtype=torch.int32, device="cuda:%d" % env.device)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    t = torch.tensor(list(ident), dtype=torch.uint8, device="cuda:%d" % env.device)
    dist.broadcast(t, src=0)
    comm = ctypes.c_void_p()
    made = 0
    if int(ok.item()) == 1:
        ident = (ctypes.c_uint8 * _lib.COMM_ID_BYTES)(*t.cpu().tolist())
        made = int(lib.sgk_comm_create(ident, rank, world, env.device, ctypes.byref(comm)) == _lib.SGK_OK)
    agree = torch.tensor([made], dtype=torch.int32, device="cuda:%d" % env.device)
    dist.all_reduce(agree, op=dist.ReduceOp.MIN)  # all ranks take the same path
    if int(agree.item()) != 1:
        if made:
            lib.sgk_comm_destroy(comm)
        if rank == 0:
            sys.stderr.write("safe_grid_agents_amd.dist: sgk_comm_create failed (%s); metrics go through torch.distributed\n"
                             % (lib.sgk_last_error() or b"?").decode())
        _COMMS[env.device] = None
        return None
    _COMMS[env.device] = comm
    return comm


def require_library_comm(comm, world, backend):
    """A multi-GPU run over RCCL must use the LIBRARY's communicator (sgk_comm_create / sgk_metrics_allreduced): a silent fall back to
    torch.distributed's all-reduce would let a first 8-GPU run pass without ever executing the C-ABI's collective. Raises when there
    is more than one rank on an RCCL-backed process group and `comm` (library_comm's result) is None -- unless the caller asked for
    torch's path with SGK_METRICS_COLLECTIVE=torch. SGK_BENCH_REQUIRE_RCCL=1 applies the rule whatever the backend (the tests'
    way to reach this exit on a one-GPU box, where two ranks can only talk over gloo)."""
    forced = os.environ.get("SGK_BENCH_REQUIRE_RCCL") == "1"
    if comm is not None or world <= 1 or (backend != "nccl" and not forced):
        return
    if os.environ.get("SGK_METRICS_COLLECTIVE", "sgk") == "torch":
        return
    raise RuntimeError("the library's RCCL communicator could not be made (sgk_comm_create; see the message above) and the metrics "
                       "all-reduce would go through torch.distributed instead: refusing to report a multi-GPU line that never ran "
                       "sgk_metrics_allreduced. Set SGK_METRICS_COLLECTIVE=torch to accept torch.distributed's all-reduce.")


def library_comm_ranks(env):
    """How many ranks the library's RCCL communicator of this process spans (RCCL's own ncclCommCount through sgk_comm_info),
    or None when the metrics go through torch.distributed / there is one rank."""
    import ctypes

    from . import _lib

    comm = library_comm(env)
    if comm is None:
        return None
    world = ctypes.c_int32(0)
    _lib.check(env.lib.sgk_comm_info(comm, None, ctypes.byref(world), None))
    return int(world.value)


def global_metrics(env):
    """BatchMetrics over all ranks for a BatchedGridworldEnv shard: the library's own RCCL all-reduce
    (sgk_metrics_allreduced) when the process group is RCCL-backed, torch.distributed otherwise (gloo in the CPU tests)."""
    import ctypes

    import torch
    import torch.distributed as dist

    from . import _lib

    comm = library_comm(env)
    if comm is not None:
        out = np.zeros(METRICS_LEN, dtype=np.int64)
        _lib.check(env.lib.sgk_metrics_allreduced(env.handle, comm, out.ctypes.data_as(ctypes.c_void_p)))
        return BatchMetrics(out, getattr(env, "reward_scale", 1.0))
    local = np.asarray(env.metrics(), dtype=np.int64)  # device sums + the host-side step counter
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        device = "cuda:%d" % env.device if dist.get_backend() == "nccl" else "cpu"
        vec = torch.as_tensor(local, device=device)
        local = allreduce_metrics(vec).cpu().numpy()
    assert local.shape[0] == METRICS_LEN
    return BatchMetrics(local, getattr(env, "reward_scale", 1.0))
