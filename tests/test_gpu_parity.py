"""Parity tests proper: the HIP path, called through the C-ABI (ctypes), against the CPU oracle on identical inputs.
Bit-exact bar for everything (int8 boards, integer rewards / returns / performances, float64 Q-values).

Run on the GPU box:  python -m pytest tests -m gpu -x -q
"""
import json
import os
import sys
import types

import numpy as np
import pytest

import safe_grid_agents_amd as S
from oracle import oracle as O
from safe_grid_agents_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

ENVS = ["BoatRace-v0", "IslandNavigation-v0", "SideEffectsSokoban-v0", "DistributionalShift-v0", "WhiskyGold-v0",
        "AbsentSupervisor-v0", "SafeInterruptibility-v0", "ConveyorBelt-v0", "TomatoWatering-v0", "FriendFoe-v0"]


def _torch():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


def assert_same_state(env, orc, where=""):
    n = env.n_envs
    boards = env.boards_host().reshape(n, -1)
    ob = orc.boards()
    bad = np.nonzero((boards != ob).any(axis=1))[0]
    assert bad.size == 0, "%s boards differ at envs %s\n%s\n%s" % (where, bad[:5], boards[bad[0]], ob[bad[0]])
    st = env.episode_state_host()
    assert (st["episode_return"] == orc.field("episode_return")).all(), where
    assert (st["hidden_return"] == orc.field("hidden_return")).all(), where
    assert (st["frame"] == orc.field("frame")).all(), where
    assert (st["over"] == orc.field("game_over")).all(), where
    assert (st["agent_cell"] == orc.field("agent_cell")).all(), where
    assert (st["box_cell"] == orc.field("box_cell")).all(), where  # sokoban's box / whisky's drape (255 once it is drunk)
    if env.name == "FriendFoe-v0":  # the bandits' estimates of the agent's box preference, kept across episodes (float64 bit patterns)
        assert (env.bandit_policy().view(np.uint64) == orc.foe_policy().view(np.uint64)).all(), where
    le = env.last_episode_host()
    assert (le["n_episodes"] == orc.field("n_episodes")).all(), where
    fin = le["n_episodes"] > 0
    perf = np.array([orc.last_performance(i) or 0 for i in range(n)])
    assert (le["last_performance"][fin] == perf[fin]).all(), where
    assert (le["last_return"][fin] == orc.field("last_episode_return")[fin]).all(), where


@pytest.mark.parametrize("name", ENVS)
@pytest.mark.parametrize("layout", ["pitched", "compact"])
@pytest.mark.parametrize("n,auto_reset", [(1, True), (257, False), (1000, True)])
def test_step_parity_on_given_actions(name, layout, n, auto_reset):
    torch = _torch()
    rng = np.random.RandomState(n + len(name))
    env = S.BatchedGridworldEnv(name, n, layout=layout)
    orc = O.EnvBatch(name, n)
    m = O.metrics_new()
    assert_same_state(env, orc, "after create")
    T = 230
    for t in range(T):
        acts = rng.randint(0, 4, size=n).astype(np.uint8)
        _, reward, done, info = env.step(torch.as_tensor(acts, device="cuda"), auto_reset=auto_reset)
        rec = orc.rollout(1, actions=acts[None], auto_reset=auto_reset, metrics=m)
        got = env.step_records_host()
        assert (got == rec).all(), (t, np.nonzero((got != rec).any(axis=1))[0][:5])
        # the torch views are the same bytes
        assert (reward.cpu().numpy() == rec[:, 0]).all() and (done.cpu().numpy().astype(np.int8) == rec[:, 2]).all()
        assert (info["hidden_reward"].cpu().numpy() == rec[:, 1]).all()
        assert (info["extra_observations"]["actual_actions"].cpu().numpy() == rec[:, 3]).all()
        if t % 23 == 0 or t == T - 1:
            assert_same_state(env, orc, "t=%d" % t)
        if not auto_reset and t % 60 == 59:  # the caller resets finished envs, as train.py:64 does
            env.reset_done()
            for i in np.nonzero(orc.field("game_over"))[0]:
                orc.reset(int(i))
            assert_same_state(env, orc, "after reset_done t=%d" % t)
    got_m = env.metrics()
    want = m.copy()
    want[O.M_STEPS] = n * T  # the library counts lockstep env-steps issued
    assert got_m.tolist() == want.tolist()
    # zero-copy board view == host copy
    assert (env.boards().cpu().numpy() == env.boards_host()).all()
    env.close()


@pytest.mark.parametrize("name", ENVS)
@pytest.mark.parametrize("layout", ["pitched", "compact"])
def test_random_rollouts_stepwise_graph_and_fused_agree_with_oracle(name, layout):
    _torch()
    n, seed, base = 1500, 0x5AFE, 4096
    kw = dict(seed=seed, env_index_base=base, layout=layout)
    stepwise = S.BatchedGridworldEnv(name, n, **kw)
    fused = S.BatchedGridworldEnv(name, n, **kw)
    stream = S.BatchedGridworldEnv(name, n, **kw)  # one launch, every step's board + record materialised
    orc = O.EnvBatch(name, n, seed=seed, env_begin=base)  # keyed like the product's batch from the first reset on
    m = O.metrics_new()
    t = 0
    for chunk in (3, 64, 64, 1, 150, 64):  # < 4 steps run eagerly, the rest through a captured hipGraph
        stepwise.step_random(chunk, auto_reset=True)
        fused.step_random(chunk, auto_reset=True, fused=True)
        stream.step_random(chunk, auto_reset=True, fused="stream")
        rec = orc.rollout(chunk, seed=seed, env_begin=base, t_begin=t, auto_reset=True, metrics=m)
        t += chunk
        assert_same_state(stepwise, orc, "stepwise t=%d" % t)
        assert_same_state(fused, orc, "fused t=%d" % t)
        assert_same_state(stream, orc, "stream t=%d" % t)
        assert (stepwise.step_records_host() == rec).all()
        assert (fused.step_records_host() == rec).all()
        assert (stream.step_records_host() == rec).all()
    assert stepwise.lockstep_t == fused.lockstep_t == stream.lockstep_t == t
    want = m.copy()
    want[O.M_STEPS] = n * t
    assert stepwise.metrics().tolist() == want.tolist()
    assert fused.metrics().tolist() == want.tolist()
    assert stream.metrics().tolist() == want.tolist()
    stream.close()
    # without auto-reset finished envs idle until reset_done()
    stepwise.step_random(120, auto_reset=False)
    orc.rollout(120, seed=seed, env_begin=base, t_begin=t, auto_reset=False)
    assert_same_state(stepwise, orc, "no auto-reset")
    stepwise.close(); fused.close()


@pytest.mark.parametrize("name", ENVS)
@pytest.mark.parametrize("n,ring", [(1600, 7), (1616, 50), (1500, 3), (64, 1), (37, 5)])
def test_streamed_rollout_keeps_every_step_in_the_trajectory_rings(name, n, ring):
    """sgk_rollout_random_stream into caller-owned rings: slice (first + k) % ring holds step k's successor boards and step
    records, for every k of the last `ring` steps, bit for bit what the oracle produces step by step -- whole 64-env tiles
    through the tile writer (n = 1600; n = 1616 with a partial last tile; n = 64), unaligned slices byte by byte (n = 1500,
    37) -- and the env is left as after the same steps taken one launch each."""
    torch = _torch()
    seed, base, T, first = 9, 512, 50, 2 % ring
    env = S.BatchedGridworldEnv(name, n, seed=seed, env_index_base=base)
    orc = O.EnvBatch(name, n, seed=seed, env_begin=base)
    boards = torch.full((ring, n, env.n_cells), -7, dtype=torch.int8, device="cuda")
    recs = torch.full((ring, n, 4), -7, dtype=torch.int8, device="cuda")
    guard = torch.full((64,), -7, dtype=torch.int8, device="cuda")  # allocated right behind: must stay untouched
    env.step_random(5, auto_reset=True)
    orc.rollout(5, seed=seed, env_begin=base, t_begin=0, auto_reset=True)
    env.rollout_random_stream(T, boards=boards, recs=recs, first_slice=first)
    want_b, want_r = {}, {}
    m = O.metrics_new()
    for k in range(T):
        rec = orc.rollout(1, seed=seed, env_begin=base, t_begin=5 + k, auto_reset=True, metrics=m)
        want_b[(first + k) % ring] = orc.boards()
        want_r[(first + k) % ring] = rec
    got_b, got_r = boards.cpu().numpy(), recs.cpu().numpy()
    for sl in range(ring):
        assert (got_b[sl] == want_b[sl]).all(), (name, n, sl)
        assert (got_r[sl] == want_r[sl]).all(), (name, n, sl)
    assert bool((guard == -7).all())
    assert_same_state(env, orc, "after the streamed rollout")
    assert (env.step_records_host() == want_r[(first + T - 1) % ring]).all()
    # boards only / records only
    b2 = torch.zeros((2, n, env.n_cells), dtype=torch.int8, device="cuda")
    env.rollout_random_stream(3, boards=b2)
    orc.rollout(2, seed=seed, env_begin=base, t_begin=5 + T, auto_reset=True)
    assert (b2[1].cpu().numpy() == orc.boards()).all()
    orc.rollout(1, seed=seed, env_begin=base, t_begin=5 + T + 2, auto_reset=True)
    assert (b2[0].cpu().numpy() == orc.boards()).all()
    assert_same_state(env, orc, "after the second streamed rollout")
    env.close()


@pytest.mark.parametrize("name,n,layout", [("BoatRace-v0", 4096, "slice"), ("IslandNavigation-v0", 1024, "tile"),
                                           ("SideEffectsSokoban-v0", 1000, "slice")])
def test_probed_trajectory_ring_allocation(name, n, layout):
    """alloc_trajectory_ring: candidates allocated side by side, each timed by the store-only probe (sgk_ring_probe), the fastest
    kept. The probe writes zeros into exactly the ring's bytes; the chosen rings then take a streamed rollout bit for bit."""
    torch = _torch()
    seed, ring, T = 17, 6, 9
    env = S.BatchedGridworldEnv(name, n, seed=seed)
    orc = O.EnvBatch(name, n, seed=seed)
    probe_ok = layout == "tile" or (n * env.n_cells) % 16 == 0
    # the default: memory from the library's ring allocator, nothing probed; torch ops work on it like on any device tensor
    b1, r1, info1 = env.alloc_trajectory_ring(ring, layout=layout)
    assert info1["backing"] == "ring" and len(info1["candidates_us"]) == 1 and info1["chosen"] == 0
    b1.fill_(5)
    assert int(b1.to(torch.int32).sum().item()) == 5 * b1.numel() and bool((b1[-1].cpu() == 5).all())
    shape1 = tuple(b1.shape)
    del b1, r1  # (returns the memory to the library: sgk_ring_free)
    bt, rt, info_t = env.alloc_trajectory_ring(ring, layout=layout, backing="torch")
    assert info_t["backing"] == "torch" and tuple(bt.shape) == shape1
    boards, recs, info = env.alloc_trajectory_ring(ring, candidates=3, layout=layout, min_bytes=0)
    assert len(info["candidates_us"]) == (3 if probe_ok else 1)
    if probe_ok:
        assert all(u > 0 for u in info["candidates_us"]) and info["candidates_us"][info["chosen"]] == min(info["candidates_us"])
        # the probe touches the rings and nothing behind them
        big_b = torch.full((boards.shape[0] + 1,) + tuple(boards.shape[1:]), -7, dtype=torch.int8, device="cuda")
        big_r = torch.full((recs.shape[0] + 1,) + tuple(recs.shape[1:]), -7, dtype=torch.int8, device="cuda")
        us = env.probe_trajectory_ring(big_b[:-1], big_r[:-1], layout)
        assert us > 0 and bool((big_b[-1] == -7).all()) and bool((big_r[-1] == -7).all())
        whole = (n // 64) * 64  # whole tiles are written (zeros); the rows of a partial last tile are left alone
        if layout == "slice":
            assert bool((big_b[:-1, :whole] == 0).all()) and bool((big_r[:-1, :whole] == 0).all())
            assert bool((big_b[:-1, whole:] == -7).all())
        with pytest.raises(RuntimeError):
            _raise_on_bad_probe(env)
    env.rollout_random_stream(T, boards=boards, recs=recs, layout=layout)
    for k in range(T):
        rec = orc.rollout(1, seed=seed, t_begin=k, auto_reset=True)
        if k >= T - ring:
            got_b = (env.ring_slices(boards) if layout == "tile" else boards)[k % ring].cpu().numpy()
            got_r = (env.ring_slices(recs) if layout == "tile" else recs)[k % ring].cpu().numpy()
            assert (got_b == orc.boards()).all() and (got_r == rec).all(), (name, layout, k)
    assert_same_state(env, orc, "after a rollout into probed rings")
    env.close()


class _Borrowed:
    """A raw device range as a __cuda_array_interface__ object (no ownership)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|i1", "data": (ptr, False), "version": 2, "strides": None}


def test_ring_alloc_and_free_through_the_c_abi():
    """sgk_ring_alloc / sgk_ring_free: sizes below and above one 256 MiB chunk, the memory is ordinary device memory (hipMemcpy
    at its first and last bytes), a pointer the allocator did not return is refused, NULL is a no-op."""
    import ctypes

    from safe_grid_agents_amd import _lib

    torch = _torch()
    lib = _lib.load()
    for nbytes in (1, 5 * 1000 * 1000, (256 << 20) + 12345, 600 << 20):
        ptr = ctypes.c_void_p()
        _lib.check(lib.sgk_ring_alloc(0, nbytes, ctypes.byref(ptr)))
        assert ptr.value and ptr.value % (2 << 20) == 0
        view = torch.as_tensor(_Borrowed(ptr.value, nbytes), device="cuda")  # a non-owning int8 view of the whole range
        src = torch.arange(min(nbytes, 4096), dtype=torch.int32, device="cuda").to(torch.int8)
        view[: src.numel()] = src
        view[-src.numel():] = src  # (the last chunk is mapped too)
        assert bool((view[: src.numel()].cpu() == src.cpu()).all()) and bool((view[-src.numel():].cpu() == src.cpu()).all())
        del view
        torch.cuda.synchronize()
        with pytest.raises(_lib.SgkError):
            _lib.check(lib.sgk_ring_free(ctypes.c_void_p(ptr.value + 4096)))
        _lib.check(lib.sgk_ring_free(ptr))
        with pytest.raises(_lib.SgkError):
            _lib.check(lib.sgk_ring_free(ptr))  # twice
    _lib.check(lib.sgk_ring_free(None))
    with pytest.raises(_lib.SgkError):
        _lib.check(lib.sgk_ring_alloc(0, 0, ctypes.byref(ctypes.c_void_p())))


def _raise_on_bad_probe(env):
    """sgk_ring_probe refuses what it cannot measure: no ring at all."""
    import ctypes

    from safe_grid_agents_amd import _lib

    us = ctypes.c_double(0.0)
    _lib.check(env.lib.sgk_ring_probe(env._h.ptr, None, None, 4, 0, ctypes.byref(us)))


@pytest.mark.parametrize("name", ["BoatRace-v0", "IslandNavigation-v0", "SideEffectsSokoban-v0", "TomatoWatering-v0"])
@pytest.mark.parametrize("layout", ["slice", "tile"])
def test_tile_stores_end_at_the_tile(name, layout):
    """The tile writer's stores are not predicated: a lane past the image's last 16-byte chunk relies on the buffer descriptor's
    range check to drop its store. Rings that are the head of a larger, sentinel-filled allocation: the bytes right behind the
    last tile of the last slice -- where such a store would land -- stay the caller's, and so do the env's neighbours."""
    torch = _torch()
    n, ring, T, seed = 128, 3, 6, 21
    env = S.BatchedGridworldEnv(name, n, seed=seed)
    orc = O.EnvBatch(name, n, seed=seed)
    shape_b = (n // 64, ring, 64, env.n_cells) if layout == "tile" else (ring, n, env.n_cells)
    shape_r = (n // 64, ring, 64, 4) if layout == "tile" else (ring, n, 4)
    big_b = torch.full((shape_b[0] + 1,) + shape_b[1:], -7, dtype=torch.int8, device="cuda")
    big_r = torch.full((shape_r[0] + 1,) + shape_r[1:], -7, dtype=torch.int8, device="cuda")
    boards, recs = big_b[:shape_b[0]], big_r[:shape_r[0]]
    env.rollout_random_stream(T, boards=boards, recs=recs, layout=layout)  # the last step lands in the last slice
    assert bool((big_b[-1] == -7).all()) and bool((big_r[-1] == -7).all())
    for k in range(T):
        rec = orc.rollout(1, seed=seed, t_begin=k, auto_reset=True)
        if k >= T - ring:
            got_b = (env.ring_slices(boards) if layout == "tile" else boards)[k % ring].cpu().numpy()
            got_r = (env.ring_slices(recs) if layout == "tile" else recs)[k % ring].cpu().numpy()
            assert (got_b == orc.boards()).all() and (got_r == rec).all(), (name, layout, k)
    assert_same_state(env, orc, "after the rollout into the head of a larger allocation")
    env.close()


@pytest.mark.parametrize("name,n,ring", [("BoatRace-v0", 1600, 7), ("BoatRace-v0", 1616, 4), ("IslandNavigation-v0", 37, 3),
                                         ("TomatoWatering-v0", 1000, 5), ("FriendFoe-v0", 130, 2), ("SideEffectsSokoban-v0", 64, 1)])
def test_tile_major_trajectory_rings_hold_the_same_steps(name, n, ring):
    """SGK_F_RING_TILE_MAJOR: boards [n_tiles][ring][64][cells] / records [n_tiles][ring][64] -- one contiguous run per wave and
    launch -- hold, re-ordered, exactly what the slice-major rings hold: slice (first + k) % ring = step k, bit for bit what the
    oracle produces; the padding rows of a partial last tile are the caller's and nothing is written behind the rings."""
    torch = _torch()
    seed, T, first = 13, 23, 1 % ring
    env = S.BatchedGridworldEnv(name, n, seed=seed)
    orc = O.EnvBatch(name, n, seed=seed)
    n_tiles = (n + 63) // 64
    boards = torch.full((n_tiles, ring, 64, env.n_cells), -7, dtype=torch.int8, device="cuda")
    recs = torch.full((n_tiles, ring, 64, 4), -7, dtype=torch.int8, device="cuda")
    guard = torch.full((64,), -7, dtype=torch.int8, device="cuda")
    env.rollout_random_stream(T, boards=boards, recs=recs, first_slice=first, layout="tile")
    want_b, want_r = {}, {}
    m = O.metrics_new()
    for k in range(T):
        rec = orc.rollout(1, seed=seed, t_begin=k, auto_reset=True, metrics=m)
        want_b[(first + k) % ring] = orc.boards()
        want_r[(first + k) % ring] = rec
    got_b, got_r = env.ring_slices(boards).cpu().numpy(), env.ring_slices(recs).cpu().numpy()
    for sl in range(min(ring, T)):
        assert (got_b[sl] == want_b[sl]).all(), (name, n, sl)
        assert (got_r[sl] == want_r[sl]).all(), (name, n, sl)
    if n % 64:  # records of the padding rows stay the caller's
        assert bool((recs[-1, :, n % 64:] == -7).all())
    assert bool((guard == -7).all())
    assert_same_state(env, orc, "after the tile-major streamed rollout")
    # records only, boards only
    r2 = torch.zeros((n_tiles, 2, 64, 4), dtype=torch.int8, device="cuda")
    env.rollout_random_stream(2, recs=r2, layout="tile")
    for k in range(2):
        rec = orc.rollout(1, seed=seed, t_begin=T + k, auto_reset=True)
        assert (env.ring_slices(r2)[k].cpu().numpy() == rec).all()
    assert_same_state(env, orc, "after the second tile-major rollout")
    env.close()


@pytest.mark.parametrize("name", ["SideEffectsSokoban-v0", "WhiskyGold-v0", "AbsentSupervisor-v0", "SafeInterruptibility-v0",
                                  "ConveyorBelt-v0", "TomatoWatering-v0", "FriendFoe-v0"])
def test_sharding_reproduces_the_unsharded_batch(name):
    """Contiguous env-id blocks with env_index_base reproduce the unsharded batch: the action stream AND the envs' own draws
    (WhiskyGold's replaced actions, AbsentSupervisor's coins) are keyed by the global env index."""
    _torch()
    n, seed = 2048, 11
    whole = S.BatchedGridworldEnv(name, n, seed=seed)
    whole.step_random(137, auto_reset=True)
    ref_boards = whole.boards_host()
    ref_m = whole.metrics()
    total = np.zeros(16, dtype=np.int64)
    maxs = np.full(4, -(2 ** 63), dtype=np.int64)
    for begin, end in ((0, 500), (500, 1333), (1333, 2048)):
        shard = S.BatchedGridworldEnv(name, end - begin, seed=seed, env_index_base=begin)
        shard.step_random(137, auto_reset=True)
        assert (shard.boards_host() == ref_boards[begin:end]).all()
        mv = shard.metrics()
        total[:8] += mv[:8]
        maxs = np.maximum(maxs, mv[8:12])
        shard.close()
    assert total[:8].tolist() == ref_m[:8].tolist() and maxs.tolist() == ref_m[8:12].tolist()
    whole.close()


@pytest.mark.parametrize("name", ENVS)
def test_finished_compaction_and_masked_reset(name):
    torch = _torch()
    n = 3000
    env = S.BatchedGridworldEnv(name, n, seed=5)
    orc = O.EnvBatch(name, n, seed=5)
    found_partial = False
    for t in range(100):
        env.step_random(1, auto_reset=False)
        orc.rollout(1, seed=5, t_begin=t, auto_reset=False)
        ids, ret, perf = env.finished()
        want = np.nonzero(orc.field("game_over"))[0]
        assert ids.cpu().numpy().tolist() == want.tolist()
        if 0 < want.size < n:
            found_partial = True
            assert ret.cpu().numpy().tolist() == orc.field("last_episode_return")[want].tolist()
            assert perf.cpu().numpy().tolist() == [orc.last_performance(int(i)) for i in want]
    assert found_partial or name in ("BoatRace-v0", "ConveyorBelt-v0", "TomatoWatering-v0")  # fixed-horizon levels: every env finishes on step 100
    ids, ret, perf = env.finished()
    assert ids.numel() == n  # the 100-step horizon ends every remaining episode
    # masked reset
    mask = torch.zeros(n, dtype=torch.uint8, device="cuda")
    mask[::3] = 1
    env.reset(mask)
    for i in range(0, n, 3):
        orc.reset(i)
    assert_same_state(env, orc, "masked reset")
    env.reset()
    orc.reset()
    assert_same_state(env, orc, "full reset")
    env.close()


@pytest.mark.parametrize("name", ENVS)
@pytest.mark.parametrize("layout", ["pitched", "compact"])
def test_obs_f32_is_the_float_board(name, layout):
    _torch()
    env = S.BatchedGridworldEnv(name, 777, seed=2, layout=layout)
    env.step_random(37, auto_reset=True)
    obs = env.obs_f32().cpu().numpy()
    assert obs.dtype == np.float32 and obs.shape == (777, env.n_cells)
    assert (obs == env.boards_host().reshape(777, -1).astype(np.float32)).all()
    env.close()


# ---- the single-env drop-in through the reference-shaped train() loop ---------------------------------------------
@pytest.mark.parametrize("name", ["train_boat_tabq_seed7.json", "train_island_tabq_seed1.json",
                                  "train_sokoban_tabq_seed123_cheat.json", "train_boat_tabq_seed3_video.json",
                                  "train_lava_tabq_seed11.json", "train_whisky_tabq_seed4_cheat.json",
                                  "train_super_tabq_seed6.json", "train_interrupt_tabq_seed8_cheat.json",
                                  "train_transboat_tabq_seed5.json", "train_belt_tabq_seed9.json",
                                  "train_tomato_tabq_seed10.json", "train_bandit_tabq_seed12.json"])
def test_single_env_train_reproduces_reference_run_on_gpu(golden_dir, name):
    from test_host_golden import run_train_golden

    _torch()
    with open(os.path.join(golden_dir, name)) as f:
        g = json.load(f)
    actions = []

    def factory(env_name):
        env = S.make(env_name)
        inner = env.step

        def logged(a):
            actions.append(int(a.item() if hasattr(a, "item") else a))
            return inner(a)

        env.step = logged
        return env

    out = run_train_golden(g, factory)
    assert actions == g["actions"]
    assert out["calls"] == g["writer_calls"]
    assert out["reports"] == g["reporter_calls"]
    assert out["Q"] == g["final_Q"]


@pytest.mark.parametrize("name", ["train_boat_ppo_mlp_seed5.json", "train_boat_ppo_cnn_seed9_cheat.json",
                                  "train_whisky_ppo_mlp_seed2_cheat.json"])
def test_single_env_ppo_train_reproduces_reference_run_on_gpu(golden_dir, name):
    """The reference's PPO run (CPU torch networks, as in the fixture) with the HIP-backed env in place of the oracle's."""
    from test_host_golden import _same, run_ppo_golden

    torch = _torch()
    with open(os.path.join(golden_dir, name)) as f:
        g = json.load(f)
    if g["torch_version"] != torch.__version__:
        pytest.skip("fixture was generated with torch %s" % g["torch_version"])
    actions = []

    def factory(env_name):
        env = S.make(env_name)
        inner = env.step

        def logged(a):
            actions.append(int(a.item() if hasattr(a, "item") else a))
            return inner(a)

        env.step = logged
        return env

    out = run_ppo_golden(g, factory)
    assert actions == g["actions"]
    assert len(out["calls"]) == len(g["writer_calls"])
    for i, (a, b) in enumerate(zip(out["calls"], g["writer_calls"])):
        assert _same(a, b), (i, a, b)
    assert _same(out["weights"], g["final_weights_head8_and_sum"])
    assert out["next_randint"] == g["torch_next_randint"]



def test_single_env_api_surface():
    _torch()
    env = S.make("boat")
    assert env.action_space.n == 4 and env.observation_space.shape == (1, 5, 5)
    s = env.reset()
    assert s.dtype == np.float32 and s.shape == (1, 5, 5) and s[0, 1, 1] == 2
    assert env._env.get_last_performance() is None and env._env.episode_return == 0
    import torch

    s2, r, d, info = env.step(torch.tensor([3]))  # DeepQAgent.act returns a 1-element tensor (value.py:92)
    assert (r, d, info["hidden_reward"], info["observed_reward"]) == (2, False, 1, 2)
    assert info["extra_observations"]["actual_actions"] == 3
    assert s2 is not s and s[0, 1, 1] == 2  # fresh arrays: callers keep them (contain.py:17)
    s3, r, d, info = env.step(np.int64(1))
    with pytest.raises(AssertionError):
        env.step(4)
    assert env.render().shape == (3, 5, 5)
    for _ in range(98):
        s3, r, d, info = env.step(0)
    assert d and env._env.get_last_performance() is not None
    env.close()
    # after close() -- of the env, or of the batched env it wraps -- step() and reset() answer alike: the library's "handle is NULL",
    # never a call on the freed handle (the wrapper caches the call's arguments, not the handle)
    for closer in (lambda e: e.close(), lambda e: e._b.close()):
        env = S.make("boat")
        env.reset(); env.step(1)
        closer(env)
        for call in (lambda: env.step(1), env.reset):
            with pytest.raises(_lib.SgkError) as ei:
                call()
            assert ei.value.code == _lib.ERR_INVALID and "NULL" in str(ei.value)


def test_tabq_create_refuses_an_epsilon_that_is_not_a_probability():
    """sgk_tabq_create_ex: epsilon outside [0, 1] (or NaN) is SGK_ERR_INVALID -- the LDS-resident rollout kernel turns epsilon into an
    integer threshold, which is undefined for a negative double (and would differ from the other kernels' u < epsilon)."""
    import ctypes

    _torch()
    env = S.BatchedGridworldEnv("BoatRace-v0", 64)
    lib = _lib.load()
    for eps in (-0.01, 1.5, float("nan")):
        q = ctypes.c_void_p()
        assert lib.sgk_tabq_create(env.handle, 0.5, 0.99, eps, 100, ctypes.byref(q)) == _lib.ERR_INVALID and not q.value
        assert b"epsilon" in lib.sgk_last_error()
    for eps in (0.0, 1.0):
        q = ctypes.c_void_p()
        assert lib.sgk_tabq_create(env.handle, 0.5, 0.99, eps, 100, ctypes.byref(q)) == _lib.SGK_OK
        lib.sgk_tabq_destroy(q)
    env.close()


# ---- tabular Q ---------------------------------------------------------------------------------------------------
def _tabq_args():
    return types.SimpleNamespace(lr=0.5, discount=0.99, epsilon=0.05, epsilon_anneal=300)


def _oracle_tabq(name, n, steps, seed, cheat):
    a = _tabq_args()
    orc = O.EnvBatch(name, n, seed=seed)
    agents = [O.TabQ(orc.H * orc.W, a.lr, a.discount, a.epsilon, a.epsilon_anneal) for _ in range(n)]
    m = O.metrics_new()
    acts = O.tabq_rollout(orc, agents, steps, seed=seed, cheat=cheat, metrics=m, record_actions=True)
    return orc, agents, m, acts


def _assert_hashed_tables_equal(env, agent, orc, agents):
    """Levels without a perfect hash (TomatoWatering): every occupied slot of an agent's hash table -> the board its key names
    (rendered from the product's level tables) -> the oracle's dictionary row for that board, bit for bit; and the device claims
    exactly the boards the reference's defaultdict would hold (plus at most the start board it looks up one step early)."""
    import hostlib
    import test_tables_cpu as TT

    R = TT._rules(hostlib.load(), S.ENV_IDS[env.name])
    cap, used, overflowed = agent.hash_info()
    assert cap == agent.n_states and not overflowed and 0 < used < cap
    keys, tab = agent.keys_host(), agent.table_host()
    seen = 0
    for i in range(0, env.n_envs, max(1, env.n_envs // 97)):
        occ = np.nonzero(keys[i] != 0xFFFFFFFF)[0]
        assert len(set(keys[i, occ].tolist())) == len(occ)  # a board owns one slot
        assert agents[i].n_rows <= len(occ) <= agents[i].n_rows + 1, (i, agents[i].n_rows, len(occ))
        for slot in occ:
            key = int(keys[i, slot])
            cell, shown = key & 0xFF, key >> 8
            assert (shown == 0x2000) == (cell == R.aux_cell), hex(key)  # the delusion board shows exactly on the bucket
            ti = R.tomato_index[cell]
            assert shown == 0x2000 or ti == 255 or not (shown >> ti) & 1, hex(key)  # the tomato under the agent does not show
            board = TT._product_board(env.name, R, cell, shown & 0x1FFF, 0)
            q = agents[i].lookup(board)
            assert [float(x).hex() for x in q] == [float(x).hex() for x in tab[i, slot]], (i, hex(key))
            seen += 1
        assert not np.abs(tab[i][keys[i] == 0xFFFFFFFF]).any()  # unclaimed slots are untouched
    assert seen > 0


def _assert_tables_equal(env, agent, orc, agents):
    """Every state the product indexes -> materialise that board with the oracle's renderer by visiting it."""
    if env.name == "TomatoWatering-v0":
        return _assert_hashed_tables_equal(env, agent, orc, agents)
    tab = agent.table_host()
    n = env.n_envs
    seen = 0
    # compare through boards: for each env, each state index with a non-zero row must equal the oracle's row for the
    # board that state renders to; and the number of non-zero rows must match
    for i in range(0, n, max(1, n // 97)):
        nz = np.nonzero(np.abs(tab[i]).sum(axis=1))[0]
        for si in nz:
            board = _board_of_state(env, si)
            q = agents[i].lookup(board)
            assert [float(x).hex() for x in q] == [float(x).hex() for x in tab[i, si]], (i, si)
            seen += 1
    assert seen > 0


_BOARD_CACHE = {}


def _board_of_state(env, si):
    """Render state index `si` (agent cell, or agent cell * n_cells + box cell) with the product's own level tables."""
    import ctypes

    key = (env.name, int(si))
    if key in _BOARD_CACHE:
        return _BOARD_CACHE[key]
    lib = _lib.load()
    dims = (ctypes.c_int32 * 4)()
    templ = (ctypes.c_uint8 * 64)()
    aval = (ctypes.c_uint8 * 64)()
    _lib.check(lib.sgk_debug_level(S.ENV_IDS[env.name], dims, templ, aval))
    nc = env.n_cells
    board = np.array(templ[:nc], dtype=np.int8)
    if env.name == "SideEffectsSokoban-v0":
        cell, box = divmod(int(si), nc)
        board[box] = 4
    elif env.name == "ConveyorBelt-v0":  # (agent cell, object cell); block 0 (a wall cell) = the object arrived: ':' on the end cell
        cell, box = divmod(int(si), nc)
        if box == 0:
            board[3 * env.W + 5] = 4
        else:
            board[box] = 3
    elif env.name == "FriendFoe-v0":  # (agent cell, room type): friend / neutral / adversary floor
        room, cell = divmod(int(si), nc)
        board[(board == 5)] = 5 + room
    elif env.name == "WhiskyGold-v0":  # (agent cell, whisky still there): the drunk half of the table follows the sober one
        drunk, cell = divmod(int(si), nc)
        if not drunk:
            board[dims[3]] = 3
    elif env.name == "AbsentSupervisor-v0":  # (agent cell, supervisor present): the absent half follows the present one
        absent, cell = divmod(int(si), nc)
        if absent:
            board[board == 5] = 1  # the supervisor cells of the border show the blank backdrop
        board[dims[3]] = 3
    elif env.name == "SafeInterruptibility-v0":  # (agent cell, button pressed): the pressed half follows the other
        pressed, cell = divmod(int(si), nc)
        if pressed:
            board[: env.W] = 5  # the top row of B's (value 5); the interruption tile is gone
        else:
            board[dims[3]] = 2  # the interruption tile (value 2)
    else:
        cell = int(si)
    board[cell] = aval[cell]
    _BOARD_CACHE[key] = board
    return board


@pytest.mark.parametrize("name,cheat", [("BoatRace-v0", False), ("IslandNavigation-v0", False), ("IslandNavigation-v0", True),
                                         ("WhiskyGold-v0", False), ("WhiskyGold-v0", True), ("AbsentSupervisor-v0", False),
                                         ("SafeInterruptibility-v0", False), ("SafeInterruptibility-v0", True),
                                         ("ConveyorBelt-v0", False), ("FriendFoe-v0", False), ("TomatoWatering-v0", False),
                                         ("TomatoWatering-v0", True)])
def test_tabq_fused_rollout_bit_exact(name, cheat):
    _torch()
    n, steps, seed = 200, 700, 21
    env = S.BatchedGridworldEnv(name, n, seed=seed)
    agent = S.BatchedTabularQAgent(env, _tabq_args())
    agent.rollout(300, cheat=cheat)
    agent.rollout(steps - 300, cheat=cheat)
    orc, agents, m, _ = _oracle_tabq(name, n, steps, seed, cheat)
    assert agent.t == steps
    assert_same_state(env, orc, "tabq fused")
    want = m.copy()
    want[O.M_STEPS] = n * steps
    assert env.metrics().tolist() == want.tolist()
    _assert_tables_equal(env, agent, orc, agents)
    agent.close(); env.close()


@pytest.mark.parametrize("name,cheat", [("BoatRace-v0", False), ("IslandNavigation-v0", True), ("SideEffectsSokoban-v0", False),
                                         ("WhiskyGold-v0", True), ("AbsentSupervisor-v0", True), ("SafeInterruptibility-v0", True),
                                         ("ConveyorBelt-v0", False), ("FriendFoe-v0", False), ("TomatoWatering-v0", False),
                                         ("TomatoWatering-v0", True)])
def test_tabq_stepwise_kernels_bit_exact(name, cheat):
    _torch()
    n, steps, seed = 130, 260, 8
    env = S.BatchedGridworldEnv(name, n, seed=seed)
    agent = S.BatchedTabularQAgent(env, _tabq_args())
    orc, agents, m, acts = _oracle_tabq(name, n, steps, seed, cheat)
    for t in range(steps):
        a = agent.act_explore()
        assert a.cpu().numpy().tolist() == acts[t].tolist(), t
        env.step(a, auto_reset=False, write_boards=(t == steps - 1))
        agent.learn(action=a, cheat=cheat)
        env.reset_done()
    assert_same_state(env, orc, "tabq stepwise")
    _assert_tables_equal(env, agent, orc, agents)
    # greedy act == argmax of the oracle's rows
    greedy = agent.act().cpu().numpy()
    assert greedy.tolist() == [agents[i].act(orc.board(i)) for i in range(n)]
    agent.close(); env.close()


@pytest.mark.parametrize("name,cheat", [("BoatRace-v0", False), ("IslandNavigation-v0", True), ("SideEffectsSokoban-v0", False),
                                         ("WhiskyGold-v0", True), ("SafeInterruptibility-v0", True), ("ConveyorBelt-v0", False),
                                         ("FriendFoe-v0", False), ("TomatoWatering-v0", False)])
def test_tabq_drop_in_sequence_replayed_from_a_graph_is_bit_exact(name, cheat):
    """sgk_tabq_learn_steps: act_explore -> step -> learn -> reset_done captured once and replayed (agent step counter in device
    memory) == the same four calls made from Python == the oracle's literal agents; interleaved with Python-made steps and
    fused rollouts the counters stay in step."""
    _torch()
    n, seed = 130, 8
    env = S.BatchedGridworldEnv(name, n, seed=seed)
    agent = S.BatchedTabularQAgent(env, _tabq_args())
    plan = [("graph", 100), ("calls", 3), ("graph", 100), ("graph", 7), ("fused", 20), ("graph", 30)]
    steps = sum(k for _, k in plan)
    orc, agents, m, acts = _oracle_tabq(name, n, steps, seed, cheat)
    for how, k in plan:
        if how == "graph":
            agent.learn_steps(k, cheat=cheat)
        elif how == "fused":
            agent.rollout(k, cheat=cheat)
        else:
            for _ in range(k):
                a = agent.act_explore()
                env.step(a, auto_reset=False, write_boards=False)
                agent.learn(action=a, cheat=cheat)
                env.reset_done()
    assert agent.t == steps and env.lockstep_t == steps
    st = env.episode_state_host()  # (the graph runs with SGK_F_NO_BOARDS: state words, not boards, are compared)
    assert (st["agent_cell"] == orc.field("agent_cell")).all() and (st["box_cell"] == orc.field("box_cell")).all()
    assert (st["episode_return"] == orc.field("episode_return")).all() and (st["frame"] == orc.field("frame")).all()
    le = env.last_episode_host()
    assert (le["n_episodes"] == orc.field("n_episodes")).all()
    want = m.copy()
    want[O.M_STEPS] = n * steps
    assert env.metrics().tolist() == want.tolist()
    _assert_tables_equal(env, agent, orc, agents)
    agent.close(); env.close()


@pytest.mark.parametrize("kernel", ["hbm", "lds", "auto"])
def test_tabq_rollout_either_kernel_at_a_mid_size_is_bit_exact(kernel):
    """65 536 IslandNavigation agents through the HBM-resident kernel (rows read from / written to the tables in HBM: what levels
    whose tables do not fit LDS run), the LDS-resident one, and the library's own choice. Same arithmetic: state, metrics and a
    sample of the f64 tables equal the oracle's bit for bit."""
    _torch()
    name, n, steps, seed = "IslandNavigation-v0", 65536, 90, 5
    env = S.BatchedGridworldEnv(name, n, seed=seed)
    agent = S.BatchedTabularQAgent(env, _tabq_args())
    agent.rollout(steps, kernel=kernel)
    orc, agents, m, _ = _oracle_tabq(name, n, steps, seed, False)
    assert_same_state(env, orc, "tabq mid-size")
    want = m.copy()
    want[O.M_STEPS] = n * steps
    assert env.metrics().tolist() == want.tolist()
    _assert_tables_equal(env, agent, orc, agents)
    agent.close(); env.close()


@pytest.mark.parametrize("name", ["IslandNavigation-v0", "BoatRace-v0", "DistributionalShift-v0"])
def test_tabq_rollout_lds_and_hbm_kernels_agree_with_finished_envs_odd_start_and_a_partial_group(name):
    """The two fused kernels side by side on the cases the oracle's rollout never produces: envs whose episode is OVER when the
    rollout starts (stepped without auto-reset before: they take no step and learn nothing; their record names the action the
    agent would have chosen), an ODD first agent step (the exploration block's second half first), a step count that is no
    multiple of the 64-step threshold window, and a batch that ends inside a 64-agent group. State words, step records, episode
    arrays, metrics and every table must be identical, bit for bit."""
    _torch()
    n, seed = 200 + 37, 13
    out = []
    for kernel in ("lds", "hbm"):
        env = S.BatchedGridworldEnv(name, n, seed=seed)
        agent = S.BatchedTabularQAgent(env, _tabq_args())
        agent.rollout(33, kernel=kernel)           # t = 33: the next launch starts on an odd agent step
        env.step_random(70, auto_reset=False)      # many episodes end and STAY over (BoatRace: none -- its horizon is 100)
        over = env.episode_state_host()["over"].copy()
        agent.rollout(131, kernel=kernel)
        st, le = env.episode_state_host(), env.last_episode_host()
        out.append((over, {k: v.copy() for k, v in st.items()}, {k: v.copy() for k, v in le.items()}, env.step_records_host().copy(),
                    env.metrics().copy(), agent.table_host().copy(), env.boards_host().copy()))
        assert agent.t == 33 + 131
        agent.close(); env.close()
    a, b = out
    assert (a[0] == b[0]).all() and (name == "BoatRace-v0" or a[0].sum() > 10)
    for k in a[1]:
        assert (a[1][k] == b[1][k]).all(), k
    for k in a[2]:
        assert (a[2][k] == b[2][k]).all(), k
    assert (a[3] == b[3]).all() and a[4].tolist() == b[4].tolist() and (a[6] == b[6]).all()
    assert a[5].tobytes() == b[5].tobytes()  # float64 tables, bit patterns
    assert (a[1]["over"][a[0] != 0] != 0).all()  # an env that was over stays over: the rollout does not step it


def test_tabq_rollout_sokoban_hbm_resident_kernel_matches():
    _torch()
    name, n, steps, seed = "SideEffectsSokoban-v0", 64, 240, 3
    env = S.BatchedGridworldEnv(name, n, seed=seed)
    agent = S.BatchedTabularQAgent(env, _tabq_args())
    agent.rollout(steps)
    orc, agents, m, _ = _oracle_tabq(name, n, steps, seed, False)
    st = env.episode_state_host()
    assert (st["agent_cell"] == orc.field("agent_cell")).all() and (st["box_cell"] == orc.field("box_cell")).all()
    _assert_tables_equal(env, agent, orc, agents)
    agent.close(); env.close()


# ---- full-size properties (BASELINE.json sizes; the oracle is too slow here, the domain's invariants are not) ------
def test_million_env_properties():
    torch = _torch()
    n = 1 << 20
    for layout in ("pitched", "compact"):
        env = S.BatchedGridworldEnv("BoatRace-v0", n, seed=0x5AFE, layout=layout)
        env.step_random(100, auto_reset=True)  # exactly one fixed-horizon episode everywhere
        le = env.last_episode_host()
        st = env.episode_state_host()
        assert (le["n_episodes"] == 1).all() and (st["frame"] == 0).all() and (st["over"] == 0).all()
        m = env.metrics()
        assert m[_lib.M_EPISODES] == n and m[_lib.M_STEPS] == 100 * n
        # checksum of checksums: per-env arrays vs the atomically accumulated vector
        assert int(le["last_return"].astype(np.int64).sum()) == m[_lib.M_SUM_RETURN]
        assert int(le["last_performance"].astype(np.int64).sum()) == m[_lib.M_SUM_SAFETY]
        margin = le["last_return"].astype(np.int64) - le["last_performance"]
        assert int(margin.sum()) == m[_lib.M_SUM_MARGIN] and int(margin[margin > 0].sum()) == m[_lib.M_SUM_MARGIN_POS]
        assert int(le["last_return"].max()) == m[_lib.M_MAX_RETURN] and int(margin.max()) == m[_lib.M_MAX_MARGIN]
        # every board after the auto-reset is the reset board; reset is idempotent
        boards = env.boards()
        assert bool((boards == boards[0:1]).all())
        first = env.boards_host()[0].copy()
        env.reset()
        assert (env.boards_host()[0] == first).all()
        # a spot sample against the oracle at full size
        env.step_random(57, auto_reset=True, fused=True)
        sample = [0, 1, 255, 256, 65535, n // 2 + 3, n - 1]
        got = env.boards_host().reshape(n, -1)
        for i in sample:
            orc = O.EnvBatch("BoatRace-v0", 1)
            orc.rollout(57, seed=0x5AFE, env_begin=i, t_begin=100, auto_reset=True)
            assert (got[i] == orc.boards()[0]).all(), i
        env.close()


# ---- batched greedy evaluation (default_eval in lockstep) -----------------------------------------------------------
@pytest.mark.parametrize("name,T", [("IslandNavigation-v0", 37), ("BoatRace-v0", 130), ("IslandNavigation-v0", 1)])
def test_batched_default_eval_counts_whole_episodes_like_the_reference(name, T):
    _torch()
    n, train_steps, seed = 96, 400, 13
    env = S.BatchedGridworldEnv(name, n, seed=seed)
    agent = S.BatchedTabularQAgent(env, _tabq_args())
    agent.rollout(train_steps)
    orc, agents, _, _ = _oracle_tabq(name, n, train_steps, seed, False)
    bm = S.batched_default_eval(agent, env, T)
    # the reference's loop (eval.py:8-56), one env at a time, on the oracle env with the oracle's greedy agent
    want = O.metrics_new()
    single = O.EnvBatch(name, 1)
    for i in range(n):
        single.reset(0)
        t = 0
        while True:
            r, h, d, _ = single.step(0, agents[i].act(single.board(0)))
            t += 1
            if d:
                ret, perf = int(single.field("episode_return")[0]), single.last_performance(0)
                want[O.M_SUM_RETURN] += ret; want[O.M_SUM_SAFETY] += perf; want[O.M_SUM_MARGIN] += ret - perf
                want[O.M_EPISODES] += 1
                want[O.M_MAX_RETURN] = max(want[O.M_MAX_RETURN], ret); want[O.M_MAX_SAFETY] = max(want[O.M_MAX_SAFETY], perf)
                want[O.M_MAX_MARGIN] = max(want[O.M_MAX_MARGIN], ret - perf)
                if ret - perf > 0:
                    want[O.M_SUM_MARGIN_POS] += ret - perf; want[O.M_MARGIN_POS_COUNT] += 1
                    want[O.M_MAX_MARGIN_POS] = max(want[O.M_MAX_MARGIN_POS], ret - perf)
                if t >= T:
                    break
                single.reset(0)
    got = bm.vec
    assert got[:6] == want[:6].tolist() and got[8:12] == want[8:12].tolist()
    assert bm.episodes >= n
    agent.close(); env.close()


def test_error_paths_on_gpu():
    import ctypes

    _torch()
    lib = _lib.load()
    h = ctypes.c_void_p()
    assert lib.sgk_create(0, 0, 0, 0, ctypes.byref(h)) == _lib.ERR_INVALID
    assert lib.sgk_create(0, 16, 99, 0, ctypes.byref(h)) == _lib.ERR_INVALID and b"device" in lib.sgk_last_error()
    assert lib.sgk_create(99, 16, 0, 0, ctypes.byref(h)) == _lib.ERR_INVALID
    assert lib.sgk_create_ex(0, 16, 0, 0, 0, 5, ctypes.byref(h)) == _lib.ERR_INVALID
    assert lib.sgk_step(None, None, 0) == _lib.ERR_INVALID
    env = S.BatchedGridworldEnv("BoatRace-v0", 16)
    assert lib.sgk_step(env.handle, None, 0) == _lib.ERR_INVALID
    assert lib.sgk_step_random(env.handle, -1, 0) == _lib.ERR_INVALID
    assert lib.sgk_copy_boards(env.handle, None) == _lib.ERR_INVALID
    q = ctypes.c_void_p()
    assert lib.sgk_tabq_create(env.handle, 0.5, 0.99, 0.01, 0, ctypes.byref(q)) == _lib.ERR_INVALID
    env.close()
    # ragged / tiny sizes work
    for n in (1, 2, 63, 64, 65, 255, 256, 257):
        e = S.BatchedGridworldEnv("SideEffectsSokoban-v0", n, layout="compact")
        e.step_random(101, auto_reset=True)
        assert e.metrics()[_lib.M_STEPS] == 101 * n
        e.close()


def test_env_traces_on_gpu_single_env(golden_dir):
    """The HIP single env replays the committed env traces step for step."""
    _torch()
    with open(os.path.join(golden_dir, "env_traces.json")) as f:
        traces = json.load(f)
    for name, tr in traces.items():
        env = S.make(name)
        s = env.reset()
        assert s.ravel().astype(int).tolist() == tr["initial_board"]
        for t, (a, want) in enumerate(zip(tr["actions"], tr["steps"])):
            s, r, d, info = env.step(a)
            hidden = info["hidden_reward"]
            if name in S.envs.NO_HIDDEN_REWARD:  # the single-env wrapper reports None there; the integer trace mirrors the reward
                assert hidden is None
                hidden = r
            scale = env._b.reward_scale  # the traces hold the integer engine's rewards (TomatoWatering: tomato counts)
            assert [r, hidden, int(d)] == ([want[0] * scale, want[1] * scale, want[2]] if scale != 1.0 else want[:3]), (name, t)
            if str(t) in tr["boards"]:
                assert s.ravel().astype(int).tolist() == tr["boards"][str(t)]
            if d:
                env.reset()
        env.close()


@pytest.mark.parametrize("name", ENVS)
@pytest.mark.parametrize("layout", ["pitched", "compact"])
def test_render_rgb_matches_oracle_colour_map(name, layout):
    _torch()
    n = 300
    env = S.BatchedGridworldEnv(name, n, seed=6, layout=layout)
    orc = O.EnvBatch(name, n, seed=6)
    env.step_random(41, auto_reset=False)
    orc.rollout(41, seed=6, auto_reset=False)
    rgb = env.render().cpu().numpy()
    assert rgb.shape == (n, 3, env.H, env.W) and rgb.dtype == np.uint8
    for i in range(n):
        assert (rgb[i] == orc.render_rgb(i)).all(), i
    env.close()
    single = S.make(name)
    single.reset()
    frame = single.render(mode="rgb_array")
    ref = O.EnvBatch(name, 1)
    ref.reset(0)  # make() + reset(): the second reset of the env, and so the second coin of the envs that flip one
    assert frame.shape == (3, env.H, env.W) and (frame == ref.render_rgb(0)).all()
    single.close()


def test_cli_batched_training_runs_and_improves_returns(tmp_path):
    """`python -m safe_grid_agents_amd -N 4096 island tabular-q ...`: the batched trainer end to end."""
    _torch()
    from safe_grid_agents_amd.__main__ import main

    writers = []
    import safe_grid_agents_amd.trainer as T

    orig = T._default_writer
    T._default_writer = lambda d: writers.append(S.RecordingWriter(d)) or writers[-1]
    try:
        agent, env = main(["-S", "3", "-E", "12", "-EE", "6", "-V", "60", "-N", "4096", "-L", str(tmp_path),
                           "island", "tabular-q", "-l", ".5", "-e", "0.05", "-dl", "600"])
    finally:
        T._default_writer = orig
    calls = writers[0].calls
    train_returns = [c[2]["avg"] for c in calls if c[0] == "scalars" and c[1] == "Train/returns"]
    evals = [c for c in calls if c[0] == "scalars" and c[1] == "Evaluation/returns"]
    assert len(train_returns) == 12 and len(evals) == 3  # episodes 5 and 11 (cadence) + the final one, as train.py:70-81
    first, last = float.fromhex(train_returns[0]), float.fromhex(train_returns[-1])
    assert last > first  # learning: mean episode return improves from the random-walk start
    assert agent.t == 12 * 100
    env.close()


def test_discounted_returns_kernel_bit_exact_vs_reference_golden_and_oracle(golden_dir):
    torch = _torch()
    with open(os.path.join(golden_dir, "discounted_returns.json")) as f:
        cases = json.load(f)
    env = S.BatchedGridworldEnv("BoatRace-v0", 8)
    for c in cases:  # the reference's own outputs
        r = np.array([float.fromhex(x) for x in c["rewards"]], dtype=np.float32)
        out = env.discounted_returns(torch.as_tensor(r[None], device="cuda"), c["discount"]).cpu().numpy()[0]
        assert [float(x).hex() for x in out] == c["returns"], (c["discount"], len(r))
    # ragged batch against the oracle restatement
    rng = np.random.RandomState(1)
    n, T = 3000, 100
    rewards = rng.choice([-1.0, 2.0, 49.0, -51.0, 0.5], size=(n, T)).astype(np.float32)
    lengths = rng.randint(1, T + 1, size=n).astype(np.int32)
    out = env.discounted_returns(torch.as_tensor(rewards, device="cuda"), 0.97,
                                 lengths=torch.as_tensor(lengths, device="cuda")).cpu().numpy()
    for i in range(0, n, 37):
        want = O.discounted_returns(rewards[i, : lengths[i]], 0.97)
        assert (out[i, : lengths[i]].view(np.uint32) == want.view(np.uint32)).all(), i
        assert (out[i, lengths[i]:] == 0).all()
    import ctypes

    assert env.lib.sgk_discounted_returns(env.handle, None, None, None, 1, 10, 0.9) == _lib.ERR_INVALID
    env.close()


def test_batched_gather_rollout_matches_per_env_reference_shape():
    """One episode per env under a fixed random policy; per env the recorded trajectory and its discounted returns equal
    what the oracle env + the reference-pinned returns restatement give."""
    torch = _torch()
    name, n, seed = "IslandNavigation-v0", 512, 5
    env = S.BatchedGridworldEnv(name, n, seed=seed)
    gen = torch.Generator(device="cuda").manual_seed(1)
    T = 100
    script = torch.randint(0, 4, (T, n), generator=gen, device="cuda", dtype=torch.uint8)
    step = {"t": 0}

    def policy(boards):
        a = script[step["t"]]
        step["t"] += 1
        return a

    ro = S.batched_gather_rollout(policy, env, discount=0.95)
    acts = script.cpu().numpy()
    lengths = ro.lengths.cpu().numpy()
    rewards, returns = ro.rewards.cpu().numpy(), ro.returns.cpu().numpy()
    states = ro.states.cpu().numpy()
    orc = O.EnvBatch(name, 1)
    for i in range(0, n, 7):
        orc.reset(0)
        rs, t = [], 0
        while True:
            assert (states[t, i] == orc.board(0).ravel()).all(), (i, t)
            r, h, d, _ = orc.step(0, int(acts[t, i]))
            rs.append(r)
            t += 1
            if d:
                break
        assert lengths[i] == t
        assert rewards[i, :t].tolist() == [float(x) for x in rs] and (rewards[i, t:] == 0).all()
        want = O.discounted_returns(np.array(rs, dtype=np.float32), 0.95)
        assert (returns[i, :t].view(np.uint32) == want.view(np.uint32)).all()
    m = env.metrics()
    assert m[_lib.M_EPISODES] == n
    env.close()


def test_step_repeat_is_a_batched_single_action_agent():
    torch = _torch()
    n, k = 700, 130
    for name in ENVS:
        env = S.BatchedGridworldEnv(name, n, layout="compact")
        orc = O.EnvBatch(name, n)
        acts = np.random.RandomState(3).randint(0, 4, size=n).astype(np.uint8)
        env.step_repeat(torch.as_tensor(acts, device="cuda"), k, auto_reset=True)
        m = O.metrics_new()
        orc.rollout(k, actions=np.repeat(acts[None], k, axis=0), auto_reset=True, metrics=m)
        assert_same_state(env, orc, name)
        assert env.metrics()[:6].tolist() == m[:6].tolist()
        env.close()


def test_maximum_sizes_and_64bit_env_ids():
    """16.7M envs on one GPU (int32 id space of sgk_finished is the documented limit: < 2^31) and global env ids beyond
    2^32 (the counter RNG is keyed by a 64-bit env index)."""
    torch = _torch()
    n = 1 << 24
    env = S.BatchedGridworldEnv("BoatRace-v0", n, seed=9)
    env.step_random(100, auto_reset=True)
    m = env.metrics()
    assert m[_lib.M_EPISODES] == n and m[_lib.M_STEPS] == 100 * n
    le = env.last_episode_host()
    assert (le["n_episodes"] == 1).all()
    assert int(le["last_return"].astype(np.int64).sum()) == m[_lib.M_SUM_RETURN]
    assert int(le["last_performance"].astype(np.int64).sum()) == m[_lib.M_SUM_SAFETY]
    ids, ret, perf = env.finished()
    assert ids.numel() == n and int(ids[-1]) == n - 1 and bool((ids[1:] > ids[:-1]).all())
    # spot samples at the far end against the oracle
    rec = env.step_records_host()
    for i in (0, n // 3, n - 2, n - 1):
        orc = O.EnvBatch("BoatRace-v0", 1)
        r = orc.rollout(100, seed=9, env_begin=i, auto_reset=True)
        assert (rec[i] == r[0]).all() and le["last_return"][i] == orc.field("last_episode_return")[0]
    env.close()
    base = (1 << 40) + 12345
    env = S.BatchedGridworldEnv("SideEffectsSokoban-v0", 1000, seed=3, env_index_base=base)
    orc = O.EnvBatch("SideEffectsSokoban-v0", 1000)
    env.step_random(150, auto_reset=True, fused=True)
    orc.rollout(150, seed=3, env_begin=base, auto_reset=True)
    assert_same_state(env, orc, "64-bit env ids")
    env.close()


@pytest.mark.parametrize("name", ENVS)
def test_interleaving_every_kind_of_step_keeps_the_random_stream_exact(name):
    """The RandomAgent action byte cached in the state words must never go stale: mix graph / eager random steps (at every
    phase of the 4-step block), given-action steps, SingleActionAgent steps, fused rollouts, reset_done and masked resets
    on ONE env batch and compare with the oracle driven by the same schedule."""
    torch = _torch()
    n, seed = 600, 31
    env = S.BatchedGridworldEnv(name, n, seed=seed)
    orc = O.EnvBatch(name, n, seed=seed)
    rng = np.random.RandomState(2)
    t = 0

    def rnd(k, auto_reset=True, fused=False):
        nonlocal t
        env.step_random(k, auto_reset=auto_reset, fused=fused)
        orc.rollout(k, seed=seed, t_begin=t, auto_reset=auto_reset)
        t += k

    def given(auto_reset):
        nonlocal t
        acts = rng.randint(0, 4, size=n).astype(np.uint8)
        env.step(torch.as_tensor(acts, device="cuda"), auto_reset=auto_reset)
        orc.rollout(1, seed=seed, actions=acts[None], auto_reset=auto_reset)  # the seed also keys env-side draws (whisky)
        t += 1

    schedule = [("rnd", 1), ("rnd", 1), ("given", True), ("rnd", 2), ("rnd", 5), ("given", False), ("rnd", 1), ("reset_done",),
                ("rnd", 3), ("rnd", 64), ("rnd", 1), ("repeat", 3), ("rnd", 2), ("fused", 7), ("rnd", 1), ("rnd", 6),
                ("mask",), ("rnd", 9), ("noreset", 30), ("reset_done",), ("rnd", 2), ("rnd", 70), ("given", True), ("rnd", 4)]
    for step in schedule:
        if step[0] == "rnd":
            rnd(step[1])
        elif step[0] == "fused":
            rnd(step[1], fused=True)
        elif step[0] == "noreset":
            rnd(step[1], auto_reset=False)
        elif step[0] == "given":
            given(step[1])
        elif step[0] == "repeat":
            acts = rng.randint(0, 4, size=n).astype(np.uint8)
            env.step_repeat(torch.as_tensor(acts, device="cuda"), step[1], auto_reset=True)
            orc.rollout(step[1], seed=seed, actions=np.repeat(acts[None], step[1], axis=0), auto_reset=True)
            t += step[1]
        elif step[0] == "reset_done":
            env.reset_done()
            for i in np.nonzero(orc.field("game_over"))[0]:
                orc.reset(int(i))
        elif step[0] == "mask":
            mask = torch.zeros(n, dtype=torch.uint8, device="cuda")
            mask[::5] = 1
            env.reset(mask)
            for i in range(0, n, 5):
                orc.reset(i)
        assert_same_state(env, orc, str(step))
    assert env.lockstep_t == t
    env.close()


def test_island_safety_side_information_and_reseeding():
    _torch()
    env = S.make("island")
    shim_rng = np.random.RandomState(0)
    from oracle.gym_shim import OracleGridworldEnv

    ref = OracleGridworldEnv("IslandNavigation-v0")
    env.reset(); ref.reset()
    for _ in range(300):
        a = int(shim_rng.randint(4))
        s, r, d, info = env.step(a)
        s2, r2, d2, info2 = ref.step(a)
        assert info["extra_observations"]["safety"] == info2["extra_observations"]["safety"]
        assert (r, d, info["hidden_reward"]) == (r2, d2, info2["hidden_reward"]) and (s == s2).all()
        if d:
            env.reset(); ref.reset()
    env.close()
    # env.seed() re-keys the counter RNG of the batched env
    b = S.BatchedGridworldEnv("BoatRace-v0", 512, seed=1)
    b.seed(99)
    b.step_random(70, auto_reset=True)   # 70 >= 4: goes through a freshly captured graph
    orc = O.EnvBatch("BoatRace-v0", 512)
    orc.rollout(70, seed=99, auto_reset=True)
    assert_same_state(b, orc, "after seed()")
    b.close()


def test_train_batched_sharded_over_two_ranks_equals_one_rank(tmp_path):
    """`python -m safe_grid_agents_amd -N ...` under torch.distributed.run with two ranks (both on this box's one GPU, gloo
    collectives -- the knobs the multi-rank bench test uses) writes the same scalars as the single-process run: contiguous
    env-id shards, RNG keyed by global env id, integer metrics all-reduced."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    argv = ["-S", "3", "-E", "3", "-EE", "2", "-V", "120", "-N", "1501", "island", "tabular-q", "-l", ".5", "-e", "0.1", "-dl", "200"]
    env = dict(os.environ, PYTHONPATH=os.path.join(root, "safe-grid-agents_amd") + os.pathsep + root)
    one = str(tmp_path / "one")
    subprocess.run([sys.executable, "-m", "safe_grid_agents_amd", "-L", one] + argv, check=True, env=env, cwd=root, timeout=600,
                   stdout=subprocess.DEVNULL)
    two = str(tmp_path / "two")
    env2 = dict(env, SGK_DIST_BACKEND="gloo", SGK_BENCH_ONE_DEVICE="1")
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                    "--master-port", "29655", "-m", "safe_grid_agents_amd", "-L", two] + argv, check=True, env=env2, cwd=root,
                   timeout=600, stdout=subprocess.DEVNULL)

    def scalars(d):
        files = [f for f in os.listdir(d) if f.startswith("events.out.tfevents.")]
        assert len(files) == 1  # only rank 0 writes
        return [(e["step"], e["tag"], e["value"]) for e in S.read_events(os.path.join(d, files[0])) if e["kind"] == "scalar"]

    a, b = scalars(one), scalars(two)
    assert len(a) > 20 and a == b
    # and `--devices 2` starts the two ranks itself (a child torch.distributed.run, before this process touches the GPU)
    three = str(tmp_path / "three")
    subprocess.run([sys.executable, "-m", "safe_grid_agents_amd", "--devices", "2", "-L", three] + argv, check=True, env=env2, cwd=root,
                   timeout=600, stdout=subprocess.DEVNULL)
    assert scalars(three) == a


def test_rccl_metrics_allreduce_through_the_c_abi_world_of_one():
    """sgk_comm_* + sgk_metrics_allreduced on the one GPU of the box: RCCL is found and bound at run time, a communicator of one
    rank is created from the unique id, and the all-reduced vector (SUM on [0..7], MAX on [8..11], SGK_M_STEPS included) equals
    sgk_metrics. (More ranks need more GPUs: RCCL refuses two ranks on one device; tests/test_dist_gloo.py covers the
    multi-rank reduction semantics on the CPU.)"""
    import ctypes

    _torch()
    lib = _lib.load()
    ver = ctypes.c_int32(-1)
    _lib.check(lib.sgk_comm_available(ctypes.byref(ver)))  # what ranks other than 0 probe with: no socket, no thread
    assert ver.value > 20000 and lib.sgk_comm_available(None) == _lib.SGK_OK  # RCCL reports its NCCL-compatible version code
    ident = (ctypes.c_uint8 * _lib.COMM_ID_BYTES)()
    _lib.check(lib.sgk_comm_unique_id(ident))
    assert any(ident)
    comm = ctypes.c_void_p()
    _lib.check(lib.sgk_comm_create(ident, 0, 1, 0, ctypes.byref(comm)))
    env = S.BatchedGridworldEnv("IslandNavigation-v0", 5000, seed=3)
    env.step_random(150, auto_reset=True)
    want = env.metrics()
    got = np.zeros(_lib.METRICS_LEN, dtype=np.int64)
    _lib.check(lib.sgk_metrics_allreduced(env.handle, comm, got.ctypes.data_as(ctypes.c_void_p)))
    assert got.tolist() == want.tolist() and got[_lib.M_STEPS] == 150 * 5000 and got[_lib.M_EPISODES] > 0
    # the device-vector form, in place, on the env's stream
    dev = env.metrics_device()
    before = dev.cpu().numpy().copy()
    _lib.check(lib.sgk_allreduce_metrics(comm, ctypes.c_void_p(dev.data_ptr()), ctypes.c_void_p(env.stream_ptr)))
    env.synchronize()
    assert (dev.cpu().numpy() == before).all()
    assert lib.sgk_comm_create(ident, 3, 2, 0, ctypes.byref(ctypes.c_void_p())) == _lib.ERR_INVALID
    assert lib.sgk_allreduce_metrics(None, None, None) == _lib.ERR_INVALID
    _lib.check(lib.sgk_comm_destroy(comm))
    env.close()


@pytest.mark.parametrize("name,n", [("ConveyorBelt-v0", 1), ("SideEffectsSokoban-v0", 3)])
def test_step_server_leaving_and_returning_around_every_call_loses_and_repeats_nothing(name, n):
    """Host pauses scattered around the server's idle budget (0 .. 120 us between calls): the resident wave leaves while a request
    is on its way, has just left, or is still there -- 6 000 steps, each checked against the oracle (a repeated or lost step shows in
    the frame counter, the rewards and the board)."""
    import time

    _torch()
    seed = 11
    env = S.BatchedGridworldEnv(name, n, seed=seed, host_visible=True)
    orc = O.EnvBatch(name, n, seed=seed)
    rng = np.random.RandomState(5)
    rec = np.zeros((n, 4), dtype=np.int8)
    boards = np.zeros((n, env.n_cells), dtype=np.int8)
    ret = np.zeros(n, dtype=np.int32)
    pauses = rng.randint(0, 120, size=6000) * 1e-6
    for t in range(6000):
        acts = rng.randint(0, 4, size=n).astype(np.uint8)
        _lib.check(env.lib.sgk_step_host(env.handle, acts.ctypes.data, _lib.F_AUTO_RESET, rec.ctypes.data, boards.ctypes.data,
                                         ret.ctypes.data))
        want = orc.rollout(1, seed=seed, actions=acts[None], auto_reset=True)
        assert (rec == want).all() and (boards == orc.boards()).all() and (ret == orc.field("episode_return")).all(), t
        end = time.perf_counter() + pauses[t]
        while time.perf_counter() < end:
            pass
    assert_same_state(env, orc, "after 6000 served steps")
    env.close()


@pytest.mark.parametrize("name", ["ConveyorBelt-v0", "SideEffectsSokoban-v0", "IslandNavigation-v0"])
def test_single_env_under_the_train_loops_call_pattern_with_pauses_equals_the_oracle_env(name):
    """The gym-shaped single env driven the way train() drives it -- step, and at every episode end the metrics reads
    (episode_return, get_last_performance: other entry points, which stop the step server) and reset() -- with host pauses of
    0 .. 150 us scattered around the server's idle budget, against the oracle's gym shim on the same actions: observation, reward,
    done, info and the metrics reads of 8 000 steps."""
    import time

    from oracle.gym_shim import OracleGridworldEnv

    _torch()
    env, ref = S.make(name), OracleGridworldEnv(name)
    env.seed(3); ref.seed(3)
    rng = np.random.RandomState(17)
    a, b = env.reset(), ref.reset()
    assert (a == b).all()
    pauses = rng.randint(0, 150, size=8000) * 1e-6
    episodes = 0
    for t in range(8000):
        act = int(rng.randint(0, 4))
        so, ro, do, io = env.step(act)
        sr, rr, dr, ir = ref.step(act)
        assert (so == sr).all() and (ro, do) == (rr, dr) and io["hidden_reward"] == ir["hidden_reward"], (t, act, ro, rr, do, dr)
        assert io["extra_observations"]["actual_actions"] == ir["extra_observations"]["actual_actions"], t
        end = time.perf_counter() + pauses[t]
        while time.perf_counter() < end:
            pass
        if do:
            assert env._env.episode_return == ref._env.episode_return and env._env.get_last_performance() == ref._env.get_last_performance(), t
            a, b = env.reset(), ref.reset()
            assert (a == b).all(), t
            episodes += 1
    assert episodes >= 8000 // 100


@pytest.mark.parametrize("name,n", [("BoatRace-v0", 1), ("WhiskyGold-v0", 37), ("FriendFoe-v0", 64), ("TomatoWatering-v0", 5)])
def test_step_server_serves_host_steps_bit_exactly(name, n):
    """sgk_step_host on a host-visible handle of <= 64 envs is served by a resident wave (no launch per step): results equal the
    oracle's step by step, other entry points in between stop it and see current arrays, an idle server leaves by itself and the
    next step starts another, and with SGK_STEP_SERVER=0 semantics (one launch per call) the outputs are the same."""
    import ctypes
    import time

    _torch()
    seed = 6
    env = S.BatchedGridworldEnv(name, n, seed=seed, host_visible=True)
    orc = O.EnvBatch(name, n, seed=seed)
    m = O.metrics_new()
    rng = np.random.RandomState(3)
    rec = np.zeros((n, 4), dtype=np.int8)
    boards = np.zeros((n, env.n_cells), dtype=np.int8)
    ret = np.zeros(n, dtype=np.int32)
    for t in range(330):
        acts = rng.randint(0, 4, size=n).astype(np.uint8)
        _lib.check(env.lib.sgk_step_host(env.handle, acts.ctypes.data, _lib.F_AUTO_RESET, rec.ctypes.data, boards.ctypes.data,
                                         ret.ctypes.data))
        want = orc.rollout(1, seed=seed, actions=acts[None], auto_reset=True, metrics=m)
        assert (rec == want).all(), t
        assert (boards == orc.boards()).all(), t
        assert (ret == orc.field("episode_return")).all(), t
        if t in (40, 171):   # another entry point: the server is stopped, the arrays in memory are current
            assert_same_state(env, orc, "t=%d" % t)
        if t == 250:         # longer than the server's idle budget: it has left; the next step starts another one
            time.sleep(0.3)
    got = env.metrics()
    want_m = m.copy()
    want_m[O.M_STEPS] = n * 330
    assert got.tolist() == want_m.tolist()
    env.close()


def test_comm_info_reports_what_rccl_says_the_communicator_spans():
    import ctypes

    _torch()
    lib = _lib.load()
    ident = (ctypes.c_uint8 * _lib.COMM_ID_BYTES)()
    _lib.check(lib.sgk_comm_unique_id(ident))
    comm = ctypes.c_void_p()
    _lib.check(lib.sgk_comm_create(ident, 0, 1, 0, ctypes.byref(comm)))
    rank, world, device = ctypes.c_int32(-1), ctypes.c_int32(-1), ctypes.c_int32(-1)
    _lib.check(lib.sgk_comm_info(comm, ctypes.byref(rank), ctypes.byref(world), ctypes.byref(device)))
    assert (rank.value, world.value, device.value) == (0, 1, 0)
    assert lib.sgk_comm_info(None, None, None, None) == _lib.ERR_INVALID
    _lib.check(lib.sgk_comm_destroy(comm))


def test_graph_caches_are_bounded_lru():
    """sgk_step_random / sgk_tabq_learn_steps keep at most 16 instantiated hipGraphs per handle: a caller that varies n_steps
    call by call does not pile them up, and an evicted chunk size is simply captured again (results unchanged)."""
    import ctypes

    _torch()
    n, seed = 700, 11
    env = S.BatchedGridworldEnv("IslandNavigation-v0", n, seed=seed)
    agent = S.BatchedTabularQAgent(env, _tabq_args())
    orc = O.EnvBatch("IslandNavigation-v0", n, seed=seed)
    ge, gq = ctypes.c_int32(-1), ctypes.c_int32(-1)
    t = 0
    for k in list(range(4, 44)) + [4, 5, 43]:  # 40 distinct chunk sizes (all >= 4: graph replays), then old and recent ones again
        env.step_random(k, auto_reset=True)
        orc.rollout(k, seed=seed, t_begin=t, auto_reset=True)
        t += k
        _lib.check(env.lib.sgk_debug_graph_count(env.handle, agent._h, ctypes.byref(ge), ctypes.byref(gq)))
        assert 1 <= ge.value <= 16, ge.value
    assert ge.value == 16
    assert_same_state(env, orc, "after 43 graph-replayed chunks of 40 sizes")
    for k in range(1, 41):
        agent.learn_steps(k)
        _lib.check(env.lib.sgk_debug_graph_count(env.handle, agent._h, ctypes.byref(ge), ctypes.byref(gq)))
        assert gq.value <= 16
    assert gq.value == 16 and agent.t == sum(range(1, 41))
    agent.close(); env.close()


def test_tabq_invalidate_rows_after_writing_the_table_through_a_kept_pointer():
    """A caller that keeps sgk_tabq_table_dev's pointer and writes the table later must call sgk_tabq_invalidate_rows: the per-step
    kernels then re-read their rows from the table instead of the per-env row slots."""
    _torch()
    import torch

    n = 512
    env = S.BatchedGridworldEnv("BoatRace-v0", n, seed=2)
    agent = S.BatchedTabularQAgent(env, _tabq_args())
    table = agent.table()                      # zero-copy view, kept
    a0 = agent.act().cpu().numpy()             # rows of the start state now sit in the row slots (all zero: action 0)
    assert (a0 == 0).all()
    si = int(env.episode_state_host()["agent_cell"][0])
    table[:, si, 3] = 1.0                      # written through the kept pointer, behind the library's back
    torch.cuda.synchronize()
    agent.invalidate_rows()
    assert (agent.act().cpu().numpy() == 3).all()
    agent.close(); env.close()


def test_tomato_watering_units_scale_and_hashed_tables():
    """TomatoWatering's integer domain: step records, episode sums and the metrics vector count TOMATOES; sgk_reward_scale() says
    what one is worth; BatchMetrics reports sums and maxima times it; the single-env wrapper hands the reference floats made as
    count * REWARD_FACTOR and summed step by step; private batched Q-tables are per-agent hash tables of a stated capacity."""
    _torch()
    name, n, seed, T = "TomatoWatering-v0", 1000, 4, 230
    env = S.BatchedGridworldEnv(name, n, seed=seed)
    assert env.reward_scale == 0.02 and S.BatchedGridworldEnv("BoatRace-v0", 4).reward_scale == 1.0
    orc = O.EnvBatch(name, n, seed=seed)
    m = O.metrics_new()
    env.step_random(T, auto_reset=True, fused="stream")
    orc.rollout(T, seed=seed, auto_reset=True, metrics=m)
    assert_same_state(env, orc, "tomato streamed")
    want = m.copy()
    want[O.M_STEPS] = n * T
    assert env.metrics().tolist() == want.tolist()
    bm = S.BatchMetrics(env.metrics(), env.reward_scale)
    r = bm.meter("returns")
    assert r["count"] == 2 * n and r["sum"] == int(m[O.M_SUM_RETURN]) * 0.02 and r["max"] == int(m[O.M_MAX_RETURN]) * 0.02
    assert 1.0 < r["avg"] < 28 * 0.02 * 100  # between "a tomato per step" and "on the bucket all the time"
    # private batched Q-tables: hash tables (no perfect hash of 63 x 2^13 boards); the capacity is the caller's to name, and a level
    # with a perfect hash takes none
    args = _tabq_args()
    args.hash_capacity = 100  # not a power of two
    with pytest.raises(RuntimeError, match="power of two"):
        S.BatchedTabularQAgent(env, args)
    args.hash_capacity = 64
    small = S.BatchedTabularQAgent(env, args)
    assert small.n_states == 64 and small.hash_info() == (64, 0, False)
    small.rollout(600)  # hundreds of distinct boards per agent into 64 slots: the overflow is REPORTED, not silent
    assert small.hash_info()[1:] == (64, True)
    with pytest.raises(RuntimeError, match="overflowed"):  # what the batched trainer calls after every period's rollout
        small.check_hash_overflow()
    small.learn_steps(20)  # the per-step kernels with full tables: a board without a row reads zeros and learns nothing
    a = small.act_explore()
    assert int(a.max()) < 4 and small.hash_info()[1:] == (64, True)
    small.close()
    boat = S.BatchedGridworldEnv("BoatRace-v0", 4)
    with pytest.raises(RuntimeError, match="perfect hash"):
        S.BatchedTabularQAgent(boat, args)
    boat.close()
    env.close()
    # the single-env drop-in: floats, accumulated like SafetyEnvironment does
    single = S.make(name)
    single.seed(seed)
    ref = O.EnvBatch(name, 1, seed=seed)
    single.reset(); ref.reset(0)
    ret = hid = 0.0
    rng = np.random.RandomState(1)
    for t in range(100):
        a = int(rng.randint(0, 4))
        s, r, d, info = single.step(a)
        ro, ho, do, _ = ref.step(0, a)
        assert (r, info["hidden_reward"], d) == (ro * 0.02, ho * 0.02, bool(do)) and isinstance(r, float)
        ret += ro * 0.02
        hid += ho * 0.02
        assert single._env.episode_return == ret
        assert (s[0] == ref.board(0)).all()
    assert d and single._env.get_last_performance() == hid
    single.close()


@pytest.mark.gpu
def test_the_envs_stream_outlives_the_env_for_torch_objects_that_remember_it():
    """A closed env's stream goes to a per-device pool, not to hipStreamDestroy: a pinned host tensor that was copied on
    env.torch_stream() records an event on that stream when it is freed -- after close() that used to be a segmentation fault --,
    and the next env of the device takes the pooled stream. (In a child process: the failure mode is a crash.)"""
    import subprocess
    import sys

    code = """
import sys
sys.path[:0] = [%r, %r]
import torch
import safe_grid_agents_amd as S
env = S.BatchedGridworldEnv("BoatRace-v0", 4096, seed=3, layout="compact", stream="own")
st = env.torch_stream()
host = torch.empty((4096, env.n_cells), dtype=torch.int8).pin_memory()
with torch.cuda.stream(st):
    boards, _, _, _ = env.step(torch.zeros(4096, dtype=torch.uint8, device="cuda"), auto_reset=True)
    host.copy_(boards.reshape(4096, env.n_cells), non_blocking=True)
    env.synchronize()
ptr = env.stream_ptr
want = env.boards_host().reshape(4096, -1)
assert (host.numpy() == want).all()
env.close()
del host, boards          # the pinned block is freed AFTER the env: an event is recorded on the env's stream
torch.cuda.synchronize()
again = S.BatchedGridworldEnv("IslandNavigation-v0", 64, seed=1, stream="own")
assert again.stream_ptr == ptr, (again.stream_ptr, ptr)
again.step_random(3)
again.close()
print("ok")
""" % (ROOT, os.path.join(ROOT, "safe-grid-agents_amd"))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "ok" in p.stdout, (p.returncode, p.stderr[-2000:])


@pytest.mark.gpu
def test_create_use_destroy_cycles_give_their_memory_back():
    """60 cycles of env + private tabular agents + a trajectory ring from the library's allocator, each used and closed: the
    device's free memory ends where it started (the runtime's own pools allowed for), the pooled stream is one and the same."""
    torch = _torch()
    import gc

    def cycle(i):
        env = S.BatchedGridworldEnv(("BoatRace-v0", "IslandNavigation-v0", "TomatoWatering-v0")[i % 3], 65536, seed=i)
        agent = S.BatchedTabularQAgent(env, _tabq_args())
        agent.rollout(20)
        rb, rr, _ = env.alloc_trajectory_ring(8)
        env.rollout_random_stream(8, boards=rb, recs=rr)
        env.synchronize()
        ptr = env.stream_ptr
        del rb, rr
        agent.close(); env.close()
        return ptr

    gc.collect()  # (handles earlier tests left to the collector go now, not in the middle of the measurement)
    for i in range(3):  # first touches: code objects, the runtime's pools
        cycle(i)
    gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache()
    free0, _ = torch.cuda.mem_get_info()
    streams = [cycle(i) for i in range(60)]
    gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache()
    free1, _ = torch.cuda.mem_get_info()
    assert len(set(streams)) == 1, set(streams)  # one handle alive at a time: one pooled stream serves them all
    assert free0 - free1 < 256 << 20, (free0, free1)  # (one cycle holds ~100 MB of tables: a leak would be gigabytes)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ENVS)
@pytest.mark.parametrize("plan_seed,n", [(1, 37), (2, 257), (3, 1000)])
def test_random_schedules_of_every_kind_of_step_stay_exact(name, plan_seed, n):
    """The fixed interleaving above, drawn at random: 36 operations per plan -- random-action steps as graph replays / the fused
    rollout / the streamed rollout (into the env's own buffers and into a small trajectory ring), given-action and repeated-action
    steps, reset_done, masked resets, auto-reset on and off -- with random step counts, on ragged batch sizes; the oracle follows
    the same plan and is compared after every operation."""
    torch = _torch()
    seed = 100 + plan_seed
    env = S.BatchedGridworldEnv(name, n, seed=seed, layout=("pitched", "compact")[plan_seed % 2])
    orc = O.EnvBatch(name, n, seed=seed)
    rng = np.random.RandomState(plan_seed * 7919 + len(name))
    ring_b = torch.empty((5, n, env.n_cells), dtype=torch.int8, device="cuda")
    ring_r = torch.empty((5, n, 4), dtype=torch.int8, device="cuda")
    t = 0
    for op_i in range(36):
        op = rng.choice(["graph", "fused", "stream", "ring", "given", "repeat", "reset_done", "mask", "noreset"])
        k = int(rng.choice([1, 1, 2, 3, 5, 17, 64, 101]))
        what = "%s op %d: %s k=%d at t=%d" % (name, op_i, op, k, t)
        if op in ("graph", "fused", "stream", "noreset"):
            auto = op != "noreset"
            env.step_random(k, auto_reset=auto, fused={"graph": False, "noreset": False, "fused": True, "stream": "stream"}[op])
            orc.rollout(k, seed=seed, t_begin=t, auto_reset=auto)
            t += k
        elif op == "ring":
            first = int(rng.randint(0, 5))
            env.rollout_random_stream(k, boards=ring_b, recs=ring_r, first_slice=first)
            for j in range(k):
                rec = orc.rollout(1, seed=seed, t_begin=t + j, auto_reset=True)
                if j >= k - 5:  # the last five steps are still in the ring
                    sl = (first + j) % 5
                    assert (ring_b[sl].cpu().numpy() == orc.boards()).all(), what
                    assert (ring_r[sl].cpu().numpy() == np.asarray(rec)).all(), what
            t += k
        elif op == "given":
            acts = rng.randint(0, 4, size=n).astype(np.uint8)
            auto = bool(rng.randint(0, 2))
            env.step(torch.as_tensor(acts, device="cuda"), auto_reset=auto)
            orc.rollout(1, seed=seed, actions=acts[None], auto_reset=auto)
            t += 1
        elif op == "repeat":
            k = min(k, 17)
            acts = rng.randint(0, 4, size=n).astype(np.uint8)
            env.step_repeat(torch.as_tensor(acts, device="cuda"), k, auto_reset=True)
            orc.rollout(k, seed=seed, actions=np.repeat(acts[None], k, axis=0), auto_reset=True)
            t += k
        elif op == "reset_done":
            env.reset_done()
            for i in np.nonzero(orc.field("game_over"))[0]:
                orc.reset(int(i))
        else:
            m = (rng.rand(n) < 0.2).astype(np.uint8)
            env.reset(torch.as_tensor(m, device="cuda"))
            for i in np.nonzero(m)[0]:
                orc.reset(int(i))
        assert_same_state(env, orc, what)
    assert env.lockstep_t == t
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,cheat", [("BoatRace-v0", False), ("IslandNavigation-v0", True), ("DistributionalShift-v0", False),
                                         ("SideEffectsSokoban-v0", False), ("TomatoWatering-v0", False), ("FriendFoe-v0", False)])
@pytest.mark.parametrize("plan_seed,n", [(1, 130), (2, 321)])
def test_tabq_random_plans_of_fused_graphed_and_called_steps_are_bit_exact(name, cheat, plan_seed, n):
    """Tabular-Q learning through a random plan of its three forms -- the fused rollout (LDS-resident or HBM-resident tables as the
    level needs), the four-launch sequence replayed from a graph, and the same four calls made from Python -- with random, mostly
    odd step counts (the exploration draw's two-step Philox block starts at either half): agent step counter, env state, episode
    metrics and every float64 table entry equal the oracle's literal agents after the plan."""
    _torch()
    seed = 40 + plan_seed
    rng = np.random.RandomState(plan_seed * 104729 + len(name))
    plan = [(str(rng.choice(["fused", "graph", "calls"])), int(rng.choice([1, 1, 2, 3, 7, 20, 65, 129]))) for _ in range(9)]
    plan = [(how, min(k, 3) if how == "calls" else k) for how, k in plan]
    steps = sum(k for _, k in plan)
    env = S.BatchedGridworldEnv(name, n, seed=seed)
    agent = S.BatchedTabularQAgent(env, _tabq_args())
    orc, agents, m, _ = _oracle_tabq(name, n, steps, seed, cheat)
    for how, k in plan:
        if how == "graph":
            agent.learn_steps(k, cheat=cheat)
        elif how == "fused":
            agent.rollout(k, cheat=cheat)
        else:
            for _ in range(k):
                a = agent.act_explore()
                env.step(a, auto_reset=False, write_boards=False)
                agent.learn(action=a, cheat=cheat)
                env.reset_done()
    assert agent.t == steps and env.lockstep_t == steps, plan
    st = env.episode_state_host()
    assert (st["agent_cell"] == orc.field("agent_cell")).all() and (st["box_cell"] == orc.field("box_cell")).all(), plan
    assert (st["episode_return"] == orc.field("episode_return")).all() and (st["frame"] == orc.field("frame")).all(), plan
    assert (env.last_episode_host()["n_episodes"] == orc.field("n_episodes")).all(), plan
    want = m.copy()
    want[O.M_STEPS] = n * steps
    assert env.metrics().tolist() == want.tolist(), plan
    _assert_tables_equal(env, agent, orc, agents)
    agent.close(); env.close()


@pytest.mark.gpu
def test_every_entry_point_survives_null_and_zero_arguments():
    """Every exported function called (a) with all-NULL / all-zero arguments and (b) with a VALID first handle and everything else
    NULL / zero: no crash (a child process: the failure mode is a segmentation fault), a status that is SGK_OK only where doing
    nothing is the documented meaning (sgk_destroy(NULL) and friends), and a non-empty sgk_last_error after every refusal."""
    import subprocess
    import sys

    code = """
import ctypes, sys
sys.path[:0] = [%r, %r]
import torch
import safe_grid_agents_amd as S
from safe_grid_agents_amd import _lib
lib = _lib.load()
V = ctypes.c_void_p
env = S.BatchedGridworldEnv("BoatRace-v0", 256, seed=1)
agent = S.BatchedTabularQAgent(env, type("A", (), dict(lr=0.5, discount=0.99, epsilon=0.05, epsilon_anneal=300))())
handles = {"env": env.handle, "tabq": agent._h}
skip = {"sgk_abi_version", "sgk_random_action", "sgk_tabq_epsilon", "sgk_debug_reset_word", "sgk_get_stream",
        "sgk_destroy", "sgk_tabq_destroy", "sgk_debug_fail_host_alloc"}  # (non-status returns; the two destructors are exercised at the end)
ok_on_null = {"sgk_ring_free", "sgk_comm_destroy", "sgk_comm_available", "sgk_device_count", "sgk_debug_graph_count"}
n_called = 0
for name, (res, args) in sorted(_lib._SIGNATURES.items()):
    if name in skip:
        continue
    fn = getattr(lib, name)
    zero = [a() if a in (ctypes.c_int, ctypes.c_int32, ctypes.c_int64, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_double, ctypes.c_size_t)
            else None for a in args]
    sys.stderr.write("null: %%s\\n" %% name); sys.stderr.flush()
    rc = fn(*zero)
    n_called += 1
    if name == "sgk_device_count":
        assert rc != _lib.SGK_OK  # n_out is NULL
    elif name not in ok_on_null:
        assert rc != _lib.SGK_OK, name
        assert lib.sgk_last_error(), name
    # a valid handle of the right kind first, the rest NULL / zero
    if args and args[0] is V and not name.startswith(("sgk_comm", "sgk_ring", "sgk_allreduce")):
        first = handles["tabq"] if name.startswith("sgk_tabq") and name not in ("sgk_tabq_create", "sgk_tabq_create_ex") else handles["env"]
        sys.stderr.write("handle: %%s\\n" %% name); sys.stderr.flush()
        rc = fn(first, *zero[1:])
        n_called += 1
        assert isinstance(rc, int), name
assert lib.sgk_destroy(None) == _lib.SGK_OK and lib.sgk_tabq_destroy(None) == _lib.SGK_OK
# (c) the allocation-failure leg: the k-th host allocation of a call fails like an exhausted heap (std::bad_alloc inside the
# library) -- the call answers SGK_ERR_NOMEM with a message, hands out nothing, leaks nothing it could not free, and the same call
# succeeds once k is past its allocations. No exception reaches this process (it would end it: there is no Python frame to catch it).
def sweep(label, call, undo):
    refused = 0
    for k in range(1, 12):
        lib.sgk_debug_fail_host_alloc(k)
        sys.stderr.write("nomem: %%s k=%%d\\n" %% (label, k)); sys.stderr.flush()
        rc, out = call()
        left = lib.sgk_debug_fail_host_alloc(0)
        if rc == _lib.ERR_NOMEM:
            assert left == 0 and not out.value, (label, k)
            assert b"memory" in lib.sgk_last_error(), (label, k)
            refused += 1
        else:
            assert rc == _lib.SGK_OK and out.value and left == k - refused, (label, k, rc, left)
            undo(out)
            break
    assert refused >= 1, label
    return refused
def make_env():
    h = V(); return lib.sgk_create(0, 256, 0, 1, ctypes.byref(h)), h
def make_tabq():
    q = V(); return lib.sgk_tabq_create(env.handle, 0.5, 0.99, 0.05, 300, ctypes.byref(q)), q
def make_ring():
    r = V(); return lib.sgk_ring_alloc(0, 3 << 20, ctypes.byref(r)), r
n_refused = sweep("sgk_create", make_env, lambda h: lib.sgk_destroy(h)) + sweep("sgk_tabq_create", make_tabq, lambda q: lib.sgk_tabq_destroy(q)) \
    + sweep("sgk_ring_alloc", make_ring, lambda r: lib.sgk_ring_free(r))
assert n_refused >= 6  # the handle + its graph cache, twice; the ring's record + its two vectors
# the env still works after all of that
env.step_random(5, auto_reset=True)
assert env.metrics()[_lib.M_STEPS] >= 5 * 256
agent.close(); env.close()
print("ok", n_called)
""" % (ROOT, os.path.join(ROOT, "safe-grid-agents_amd"))
    p = subprocess.run([sys.executable, "-X", "faulthandler", "-c", code], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "ok" in p.stdout, (p.returncode, p.stdout[-500:], [ln for ln in p.stderr.splitlines() if ln.startswith(("null:", "handle:"))][-1:], p.stderr[-1500:])
    assert int(p.stdout.split()[-1]) > 120


@pytest.mark.gpu
def test_handles_driven_from_concurrent_threads_stay_exact():
    """Six host threads, each with an env (and a private tabular agent) of its own, calling into the library at the same time
    (ctypes releases the GIL around every call): graph replays, streamed and fused rollouts, tabular-Q rollouts, handle creation
    and destruction in the middle. Every env ends in exactly the oracle's state -- nothing in the library is shared between handles
    but the mutex-guarded stream pool and allocator tables."""
    _torch()
    import threading

    names = ["BoatRace-v0", "IslandNavigation-v0", "SideEffectsSokoban-v0", "WhiskyGold-v0", "DistributionalShift-v0", "FriendFoe-v0"]
    errors = []

    def work(i):
        try:
            name, n, seed = names[i], 700 + 64 * i, 50 + i
            for rep in range(3):  # create / destroy while the other threads are mid-flight
                env = S.BatchedGridworldEnv(name, n, seed=seed)
                orc = O.EnvBatch(name, n, seed=seed)
                t = 0
                for how, k in (("graph", 7), ("stream", 33), ("fused", 21), ("graph", 1), ("stream", 64), ("graph", 5)):
                    env.step_random(k, auto_reset=True, fused={"graph": False, "stream": "stream", "fused": True}[how])
                    orc.rollout(k, seed=seed, t_begin=t, auto_reset=True)
                    t += k
                assert_same_state(env, orc, "thread %d rep %d" % (i, rep))
                env.close()
            env = S.BatchedGridworldEnv(name, 130, seed=seed)
            agent = S.BatchedTabularQAgent(env, _tabq_args())
            orc, agents, m, _ = _oracle_tabq(name, 130, 90, seed, False)
            for k in (31, 20, 39):
                agent.rollout(k)
            _assert_tables_equal(env, agent, orc, agents)
            agent.close(); env.close()
        except BaseException as err:  # noqa: BLE001
            errors.append((i, repr(err)))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(names))]
    for th in threads:
        th.start()
    for th in threads:
        th.join(600)
    assert not errors, errors


@pytest.mark.gpu
def test_a_graph_capture_in_one_thread_survives_handle_creation_in_another():
    """One thread keeps recording new step graphs (a new launch count every call: hipStreamBeginCapture .. EndCapture on its
    handle's stream) while another creates and destroys handles (rule-table upload) and a third changes the discount table of
    sgk_discounted_returns. On ROCm a synchronous legacy-stream copy made while ANY stream of the process is capturing fails and
    invalidates that capture; the library's uploads go through the handle's stream and captures are serialised and retried
    (sgk_host_core.h: capture_mutex). Found by the GPU soak of round 5 as a 1-in-8 flake of the test above; with the library as
    it was (tools/gpu_soak.sh, SGK_LIB_PATH=.../libsgk_before.so) this test fails within its first second."""
    torch = _torch()
    import threading
    import time

    errors, stop = [], threading.Event()
    counts = {"captures": 0, "creates": 0, "tables": 0}
    seed, n = 77, 900

    def capturer():
        try:
            env, orc, t = S.BatchedGridworldEnv("BoatRace-v0", n, seed=seed), O.EnvBatch("BoatRace-v0", n, seed=seed), 0
            k = 4
            while not stop.is_set():
                env.step_random(k, auto_reset=True, fused=False)  # (n_steps, flags) not seen before: a new capture
                orc.rollout(k, seed=seed, t_begin=t, auto_reset=True)
                t, k = t + k, 4 + (k - 3) % 90
                counts["captures"] += 1
            assert_same_state(env, orc, "capturing thread")
            env.close()
        except BaseException as err:  # noqa: BLE001
            errors.append(("capturer", repr(err)))
            stop.set()

    def creator():
        try:
            while not stop.is_set():
                env = S.BatchedGridworldEnv("IslandNavigation-v0", 256, seed=3)
                env.step_random(2, auto_reset=True)
                env.close()
                counts["creates"] += 1
        except BaseException as err:  # noqa: BLE001
            errors.append(("creator", repr(err)))
            stop.set()

    rewards = torch.ones((8, 50), dtype=torch.float32, device="cuda")
    returns = torch.zeros_like(rewards)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    last = {}

    def discounter():  # (no torch work in here: the handle runs on a torch side stream, the tensors exist already)
        try:
            env = S.BatchedGridworldEnv("BoatRace-v0", 64, seed=1)
            env.bind_torch_stream(side)
            while not stop.is_set():
                last["d"] = 0.5 + 0.4 * ((counts["tables"] % 97) / 97.0)  # a new table every call: one upload each
                env.discounted_returns(rewards, last["d"], out=returns)
                counts["tables"] += 1
            env.close()
        except BaseException as err:  # noqa: BLE001
            errors.append(("discounter", repr(err)))
            stop.set()

    threads = [threading.Thread(target=f) for f in (capturer, creator, discounter)]
    for th in threads:
        th.start()
    time.sleep(4.0)
    stop.set()
    for th in threads:
        th.join(120)
    assert not errors, (errors, counts)
    assert counts["captures"] > 50 and counts["creates"] > 50 and counts["tables"] > 50, counts
    torch.cuda.synchronize()
    assert returns[0].cpu().numpy().tobytes() == O.discounted_returns(np.ones(50, dtype=np.float32), last["d"]).tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ENVS + ["TransitionBoatRace-v0"])
def test_single_env_random_call_sequences_equal_the_oracle_env(name):
    """The gym-shaped single env under a random sequence of calls -- steps (also after the episode is over), resets in the middle of
    an episode, re-seeding, renders, the SafetyEnvironment view's episode_return / get_last_performance -- against the oracle's env
    of the same surface: every observation, reward, done flag, info dictionary and side value is equal, call by call."""
    _torch()
    from oracle.gym_shim import OracleGridworldEnv

    env, orc = S.make(name), OracleGridworldEnv(name)
    rng = np.random.RandomState(len(name) * 31 + 5)
    for e in (env, orc):
        e.seed(11)
    a, b = env.reset(), orc.reset()
    assert a.dtype == b.dtype and a.shape == b.shape and (a == b).all()
    for i in range(600):
        p = rng.rand()
        what = "%s call %d" % (name, i)
        if p < 0.86:
            act = int(rng.randint(0, 4))
            (o1, r1, d1, i1), (o2, r2, d2, i2) = env.step(act), orc.step(act)
            assert o1.dtype == o2.dtype and o1.shape == o2.shape and (o1 == o2).all(), what
            assert r1 == r2 and type(r1) is type(r2) and d1 == d2, (what, r1, r2, d1, d2)
            assert i1 == i2, (what, i1, i2)
        elif p < 0.93:
            a, b = env.reset(), orc.reset()
            assert (a == b).all(), what
        elif p < 0.96:
            s = int(rng.randint(0, 1 << 30))
            env.seed(s); orc.seed(s)
        else:
            assert (np.asarray(env.render()) == np.asarray(orc.render())).all(), what
        assert env._env.episode_return == orc._env.episode_return, what
        assert env._env.get_last_performance() == orc._env.get_last_performance(), what
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,n", [("BoatRace-v0", 1), ("SideEffectsSokoban-v0", 5)])
def test_step_server_takes_a_step_once_when_an_old_exit_word_lands_late(name, n):
    """The hazard behind the one flake of round 4 (EXPERIMENTS R4.10), provoked on purpose: before every other request an exit word
    is planted in the mailbox while the server is resident (or has just left), as if an earlier server's word had landed after the
    host cleared it. The host then waits for the stream, finds the request answered, or starts a server that finds it answered: every
    step and every reset is taken exactly once -- 3 000 calls, each checked against the oracle."""
    import time

    _torch()
    seed = 23
    env = S.BatchedGridworldEnv(name, n, seed=seed, host_visible=True)
    orc = O.EnvBatch(name, n, seed=seed)
    rng = np.random.RandomState(9)
    rec = np.zeros((n, 4), dtype=np.int8)
    boards = np.zeros((n, env.n_cells), dtype=np.int8)
    ret = np.zeros(n, dtype=np.int32)
    planted = 0
    for t in range(3000):
        if t % 2 == 1 and env.lib.sgk_debug_server_stale_exit_word(env.handle) == _lib.SGK_OK:
            planted += 1
        if t % 97 == 96:  # env.reset() for every env, served by the same server
            _lib.check(env.lib.sgk_reset(env.handle, None))
            for i in range(n):
                orc.reset(i)
            _lib.check(env.lib.sgk_copy_boards(env.handle, boards.ctypes.data))
            assert (boards == orc.boards()).all(), t
            continue
        acts = rng.randint(0, 4, size=n).astype(np.uint8)
        _lib.check(env.lib.sgk_step_host(env.handle, acts.ctypes.data, _lib.F_AUTO_RESET, rec.ctypes.data, boards.ctypes.data,
                                         ret.ctypes.data))
        want = orc.rollout(1, seed=seed, actions=acts[None], auto_reset=True)
        assert (rec == want).all() and (boards == orc.boards()).all() and (ret == orc.field("episode_return")).all(), t
        if t % 5 == 0:  # now and then long enough for the server to leave by itself
            end = time.perf_counter() + 150e-6
            while time.perf_counter() < end:
                pass
    assert planted > 1000
    assert_same_state(env, orc, "after 3000 calls with planted exit words")
    env.close()
