"""Parity at BASELINE.json's stated sizes, through the C-ABI, against the oracle (bit-exact bar).

config 2  BoatRace random-action rollout, 65 536 envs: the WHOLE batch vs the oracle for 230 steps, hipGraph-replayed step
          kernel and fused rollout kernel.
config 5  (the metric's own batch) BoatRace random-action rollout, 1 048 576 envs: the WHOLE batch -- every board, state word field,
          last-episode array and the metrics vector -- vs the oracle (threaded) after 230 steps, through each of the three
          paths (one launch per step, streamed into the env's own buffers, streamed into a trajectory ring: the last 30
          slices slice by slice), and as two 524 288-env shards; the same (streamed) for SideEffectsSokoban, WhiskyGold, ConveyorBelt and TomatoWatering.
config 3  IslandNavigation + tabular-Q, 262 144 private agents, the LDS-resident kernel FORCED (it runs ~5 rounds of 64-agent
          groups per workgroup there: the grid-stride path) -- and again at 65 536 agents forced onto the same kernel: env
          state and the f64 tables of 4 096 sampled agents (first / middle / last groups, every round of the grid-stride
          loop) vs the oracle's literal dict-keyed agents (reference value.py:33-58 semantics), plus full-size checksums.
config 4  SideEffectsSokoban + deep-q, 32 768 envs, 150 iterations of {policy_act, step, replay_store, sgd, reset_done}
          replayed from one graph: env state bit-exact vs the oracle on the executed actions every iteration; the fused SGD
          step vs torch autograd + Adam at the tolerance of test_gpu_deepq (rtol 2e-4) on the same minibatch rows.
"""
import os
import types

import numpy as np
import pytest

import safe_grid_agents_amd as S
from oracle import oracle as O
from safe_grid_agents_amd import _lib

pytestmark = pytest.mark.gpu


def _torch():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


def _tabq_args():
    return types.SimpleNamespace(lr=0.5, discount=0.99, epsilon=0.05, epsilon_anneal=300)


def _block_state_equal(st, le, boards, lo, orc, where):
    hi = lo + orc.n
    assert (boards[lo:hi] == orc.boards()).all(), where
    for key, field in (("episode_return", "episode_return"), ("hidden_return", "hidden_return"), ("frame", "frame"),
                       ("over", "game_over"), ("agent_cell", "agent_cell"), ("box_cell", "box_cell")):
        assert (st[key][lo:hi] == orc.field(field)).all(), (where, key)
    assert (le["n_episodes"][lo:hi] == orc.field("n_episodes")).all(), where
    fin = le["n_episodes"][lo:hi] > 0
    perf = np.array([orc.last_performance(i) or 0 for i in range(orc.n)])
    assert (le["last_performance"][lo:hi][fin] == perf[fin]).all(), where
    assert (le["last_return"][lo:hi][fin] == orc.field("last_episode_return")[fin]).all(), where


# ---- config 2 ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("fused", [False, True])
def test_config2_boatrace_65536_envs_whole_batch_vs_oracle(fused):
    _torch()
    name, n, seed, T = "BoatRace-v0", 65536, 0x5AFE, 230
    env = S.BatchedGridworldEnv(name, n, seed=seed)
    orc = O.EnvBatch(name, n, seed=seed)
    m = O.metrics_new()
    t = 0
    for chunk in (100, 100, 30):  # 100-step graphs (bench.py's chunk) and a tail chunk
        env.step_random(chunk, auto_reset=True, fused=fused)
        rec = O.rollout_mt(orc, chunk, 8, seed=seed, t_begin=t, auto_reset=True, metrics=m)
        t += chunk
    assert t == T
    boards = env.boards_host().reshape(n, -1)
    _block_state_equal(env.episode_state_host(), env.last_episode_host(), boards, 0, orc, "config 2 fused=%s" % fused)
    want = m.copy()
    want[O.M_STEPS] = n * T
    assert env.metrics().tolist() == want.tolist()
    # the last step's records: the oracle's single-thread form returns them
    env2 = S.BatchedGridworldEnv(name, 4096, seed=seed, env_index_base=n - 4096)
    orc2 = O.EnvBatch(name, 4096, seed=seed, env_begin=n - 4096)
    env2.step_random(T, auto_reset=True, fused=fused)
    rec = orc2.rollout(T, seed=seed, env_begin=n - 4096, auto_reset=True)
    assert (env2.step_records_host() == rec).all()
    assert (env2.boards_host().reshape(4096, -1) == boards[n - 4096:]).all()  # and the shard equals the batch's tail
    env.close(); env2.close()


# ---- config 5 / the metric's batch -------------------------------------------------------------------------------------
def _whole_batch_equal(env, orc, where):
    boards, f = orc.export()
    n = env.n_envs
    assert (env.boards_host().reshape(n, -1) == boards).all(), where
    st, le = env.episode_state_host(), env.last_episode_host()
    for key, name in (("episode_return", "episode_return"), ("hidden_return", "hidden_return"), ("frame", "frame"),
                      ("over", "game_over"), ("agent_cell", "agent_cell"), ("box_cell", "box_cell")):
        assert (st[key] == f[name]).all(), (where, key)
    assert (le["n_episodes"] == f["n_episodes"]).all(), where
    fin = f["n_episodes"] > 0
    assert (le["last_return"][fin] == f["last_episode_return"][fin]).all(), where
    assert (le["last_performance"][fin] == f["last_performance"][fin]).all(), where


@pytest.mark.parametrize("name,path", [("BoatRace-v0", "launch"), ("BoatRace-v0", "stream"), ("BoatRace-v0", "ring"),
                                       ("SideEffectsSokoban-v0", "stream"), ("WhiskyGold-v0", "stream"),
                                       ("ConveyorBelt-v0", "stream"), ("TomatoWatering-v0", "stream")])
def test_one_million_envs_whole_batch_vs_oracle(name, path):
    torch = _torch()
    n, seed, T = 1 << 20, 0x5AFE, 230
    threads = os.cpu_count() or 8
    env = S.BatchedGridworldEnv(name, n, seed=seed)
    orc = O.EnvBatch(name, n, seed=seed)
    m = O.metrics_new()
    ring = 30
    boards = recs = None
    if path == "ring":
        boards = torch.empty((ring, n, env.n_cells), dtype=torch.int8, device="cuda")
        recs = torch.empty((ring, n, 4), dtype=torch.int8, device="cuda")
    t = 0
    for chunk in (100, 100):
        if path == "launch":
            env.step_random(chunk, auto_reset=True)
        elif path == "stream":
            env.step_random(chunk, auto_reset=True, fused="stream")
        else:
            env.rollout_random_stream(chunk, boards=boards, recs=recs, first_slice=t % ring)
        O.rollout_mt(orc, chunk, threads, seed=seed, t_begin=t, auto_reset=True, metrics=m)
        t += chunk
    # the last 30 steps one oracle step at a time: the ring's slices are checked slice by slice
    if path == "ring":
        env.rollout_random_stream(T - t, boards=boards, recs=recs, first_slice=t % ring)
        got_b = boards.cpu().numpy()
        for k in range(T - t):
            O.rollout_mt(orc, 1, threads, seed=seed, t_begin=t + k, auto_reset=True, metrics=m)
            if k % 7 == 0 or k == T - t - 1:  # (a million-env export per slice costs ~0.5 s)
                want, _ = orc.export()
                assert (got_b[(t + k) % ring] == want).all(), ("slice", k)
    else:
        if path == "launch":
            env.step_random(T - t, auto_reset=True)
        else:
            env.step_random(T - t, auto_reset=True, fused="stream")
        O.rollout_mt(orc, T - t, threads, seed=seed, t_begin=t, auto_reset=True, metrics=m)
    _whole_batch_equal(env, orc, "%s %s" % (name, path))
    want = m.copy()
    want[O.M_STEPS] = n * T
    assert env.metrics().tolist() == want.tolist()
    env.close()


def test_one_million_boatrace_envs_as_two_shards_equal_the_whole_batch():
    """BASELINE config 5's sharding at its size: two contiguous 524 288-env blocks (env_index_base) reproduce the unsharded batch's
    boards, and their metrics vectors add up (SUM on [0..7], MAX on [8..11]) to the whole batch's, which equals the oracle's."""
    _torch()
    n, seed, T = 1 << 20, 0x5AFE, 130
    orc = O.EnvBatch("BoatRace-v0", n, seed=seed)
    m = O.metrics_new()
    O.rollout_mt(orc, T, os.cpu_count() or 8, seed=seed, auto_reset=True, metrics=m)
    want_boards, _ = orc.export()
    total = np.zeros(16, dtype=np.int64)
    maxs = np.full(4, -(2 ** 63), dtype=np.int64)
    for begin in (0, n // 2):
        shard = S.BatchedGridworldEnv("BoatRace-v0", n // 2, seed=seed, env_index_base=begin)
        shard.step_random(100, auto_reset=True, fused="stream")
        shard.step_random(T - 100, auto_reset=True)
        assert (shard.boards_host().reshape(n // 2, -1) == want_boards[begin:begin + n // 2]).all()
        mv = shard.metrics()
        total[:8] += mv[:8]
        maxs = np.maximum(maxs, mv[8:12])
        shard.close()
    want = m.copy()
    want[O.M_STEPS] = n * T
    assert total[:8].tolist() == want[:8].tolist() and maxs.tolist() == want[8:12].tolist()


# ---- config 3 ------------------------------------------------------------------------------------------------------
def _sample_blocks(n, block, count):
    """`count` blocks of `block` consecutive agents: the first, the last and evenly spread ones, aligned to 64-agent groups
    except the last (which ends at n)."""
    starts = sorted({int(round(k * (n - block) / (count - 1))) // 64 * 64 for k in range(count - 1)} | {n - block})
    return starts


@pytest.mark.parametrize("n,steps,kernel", [(262144, 130, "lds"), (65536, 230, "lds"),
                                            (1 << 22, 41, "lds"), (1 << 22, 21, "hbm")])  # 4 M agents: 6.4 GB of tables (> 2^32 bytes)
def test_config3_island_tabq_lds_resident_kernel_multi_round_vs_oracle(n, steps, kernel):
    _torch()
    name, seed = "IslandNavigation-v0", 21
    env = S.BatchedGridworldEnv(name, n, seed=seed)
    agent = S.BatchedTabularQAgent(env, _tabq_args())
    first = steps // 2
    agent.rollout(first, kernel=kernel)          # two launches: tables leave LDS for HBM and come back in between
    agent.rollout(steps - first, kernel=kernel)
    assert agent.t == steps
    st, le = env.episode_state_host(), env.last_episode_host()
    boards = env.boards_host().reshape(n, -1)
    a = _tabq_args()
    checked_agents = 0
    for lo in _sample_blocks(n, 256, 16):  # 16 x 256 = 4 096 agents
        orc = O.EnvBatch(name, 256, seed=seed, env_begin=lo)
        agents = [O.TabQ(orc.H * orc.W, a.lr, a.discount, a.epsilon, a.epsilon_anneal) for _ in range(256)]
        O.tabq_rollout(orc, agents, steps, seed=seed, env_begin=lo)
        _block_state_equal(st, le, boards, lo, orc, "config 3 block %d" % lo)
        tab = agent.table_host(lo, 256)
        for i in range(256):
            nz = np.nonzero(np.abs(tab[i]).sum(axis=1))[0]
            assert agents[i].n_rows >= len(nz)
            for si in nz:  # state index = agent cell: render that board with the oracle's own reset board as template
                board = _island_board(orc, int(si))
                q = agents[i].lookup(board)
                assert [float(x).hex() for x in q] == [float(x).hex() for x in tab[i, si]], (lo, i, si)
            # and nothing the oracle learned is missing on the device: non-zero oracle rows <= non-zero device rows
            checked_agents += 1
    assert checked_agents == 4096
    # full-size checksums: the metrics vector vs the per-env arrays of ALL agents
    m = env.metrics()
    assert m[_lib.M_STEPS] == n * steps and m[_lib.M_EPISODES] == int(le["n_episodes"].astype(np.int64).sum())
    assert int(le["last_return"].astype(np.int64).max()) <= m[_lib.M_MAX_RETURN]
    agent.close(); env.close()


def test_tomato_watering_4096_hashed_tabq_agents_300_steps_vs_oracle_dictionaries():
    """Private tabular-Q agents on the level whose boards have no perfect hash: 4 096 agents x 300 steps through the fused
    kernel, every agent's hash table (keys -> boards -> rows) against the oracle's literal board-keyed dictionaries, bit for bit;
    the same agents through the graph-replayed drop-in sequence end in the same tables."""
    import test_gpu_parity as P

    _torch()
    name, n, steps, seed = "TomatoWatering-v0", 4096, 300, 13
    a = _tabq_args()
    a.hash_capacity = 1024
    env = S.BatchedGridworldEnv(name, n, seed=seed)
    agent = S.BatchedTabularQAgent(env, a)
    agent.rollout(120)
    agent.rollout(steps - 120)
    orc = O.EnvBatch(name, n, seed=seed)
    agents = [O.TabQ(orc.H * orc.W, a.lr, a.discount, a.epsilon, a.epsilon_anneal) for _ in range(n)]
    m = O.metrics_new()
    O.tabq_rollout(orc, agents, steps, seed=seed, metrics=m)
    st, le = env.episode_state_host(), env.last_episode_host()
    _block_state_equal(st, le, env.boards_host().reshape(n, -1), 0, orc, "tomato tabular-Q")
    want = m.copy()
    want[O.M_STEPS] = n * steps
    assert env.metrics().tolist() == want.tolist()
    P._assert_hashed_tables_equal(env, agent, orc, agents)
    keys, tab = agent.keys_host(), agent.table_host()
    cap, used, overflowed = agent.hash_info()
    assert (cap, overflowed) == (1024, False) and 100 < used <= steps + 2
    # the drop-in call sequence (four launches per step, replayed from a hipGraph) on fresh agents: the same dictionaries. Slots may
    # differ (insertion order is the same here, so they do not), contents may not: compare as {key: row}
    env2 = S.BatchedGridworldEnv(name, n, seed=seed)
    agent2 = S.BatchedTabularQAgent(env2, a)
    agent2.learn_steps(steps)
    keys2, tab2 = agent2.keys_host(), agent2.table_host()
    for i in range(0, n, 61):
        d1 = {int(k): tab[i, s].tobytes() for s, k in enumerate(keys[i]) if k != 0xFFFFFFFF}
        d2 = {int(k): tab2[i, s].tobytes() for s, k in enumerate(keys2[i]) if k != 0xFFFFFFFF}
        common = set(d1) & set(d2)
        assert len(common) >= max(len(d1), len(d2)) - 1 and all(d1[k] == d2[k] for k in common), i
    agent.close(); env.close(); agent2.close(); env2.close()


_ISLAND_TEMPLATE = {}


def _island_board(orc, cell):
    """The IslandNavigation board with the agent at `cell`, built from the ORACLE's renderer: the reset board with the agent
    lifted off (' ' = 1 at the start cell), then the agent value 2 -- unless the cell is water (drawn over the agent)."""
    if "t" not in _ISLAND_TEMPLATE:
        fresh = O.EnvBatch("IslandNavigation-v0", 1)
        b = fresh.board(0).ravel().copy()
        b[int(fresh.field("agent_cell")[0])] = 1
        _ISLAND_TEMPLATE["t"] = b
    b = _ISLAND_TEMPLATE["t"].copy()
    if b[cell] != 0:  # water has value 0 and hides the agent
        b[cell] = 2
    return b


def test_forcing_the_lds_kernel_on_an_env_that_cannot_use_it_is_an_error():
    _torch()
    env = S.BatchedGridworldEnv("SideEffectsSokoban-v0", 256)
    agent = S.BatchedTabularQAgent(env, _tabq_args())
    with pytest.raises(_lib.SgkError):
        agent.rollout(4, kernel="lds")
    agent.rollout(4, kernel="hbm")
    agent.close(); env.close()


# ---- config 4 ------------------------------------------------------------------------------------------------------
def _dq_args(**kw):
    d = dict(discount=0.99, lr=1e-3, batch_size=64, sync_every=40, epsilon=0.05, epsilon_anneal=200, n_layers=2,
             n_hidden=100)
    d.update(kw)
    return types.SimpleNamespace(**d)


def test_config4_sokoban_deepq_32768_envs_150_graphed_iterations_vs_oracle():
    torch = _torch()
    torch.manual_seed(11)
    name, n, iters = "SideEffectsSokoban-v0", 32768, 150
    env = S.BatchedGridworldEnv(name, n, seed=6, layout="compact")
    env.bind_torch_stream()
    orc = O.EnvBatch(name, n)
    orc_m = O.metrics_new()
    agent = S.BatchedDeepQAgent(env, _dq_args(), sgd_steps=1, replay_slices=4)
    assert agent.fused_policy

    def mirror(slice_k, where):
        """Replay the slice's executed actions on the oracle: rewards, terminals, successor boards must match bit for bit."""
        acts = agent.replay.actions[slice_k].cpu().numpy()
        rec = orc.rollout(1, actions=acts[None], auto_reset=False, metrics=orc_m)
        assert (agent.replay.rewards[slice_k].cpu().numpy() == rec[:, 0]).all(), where
        assert (agent.replay.terminals[slice_k].cpu().numpy() == rec[:, 2].astype(bool)).all(), where
        assert (agent.replay.successors[slice_k].cpu().numpy() == orc.boards()).all(), where
        for i in np.nonzero(orc.field("game_over"))[0]:
            orc.reset(int(i))

    agent.warmup(4)
    for k in range(4):
        mirror(k, "warm-up %d" % k)
    t0 = env.lockstep_t
    agent.enable_graphs(learn=True)  # 3 eager iterations + capture
    for k in range(3):
        mirror((k) % 4, "capture warm-up %d" % k)
    before = [p.detach().clone() for p in agent.Q.parameters()]
    for it in range(iters):
        k = agent.replay.head
        agent.step_graphed(learn=True)
        torch.cuda.synchronize()
        mirror(k, "iteration %d" % it)
        if it % 10 == 9 or it == iters - 1:
            assert (env.boards_host().reshape(n, -1) == orc.boards()).all(), it
            st = env.episode_state_host()
            assert (st["episode_return"] == orc.field("episode_return")).all(), it
            assert (st["hidden_return"] == orc.field("hidden_return")).all(), it
            assert (st["box_cell"] == orc.field("box_cell")).all(), it
    assert env.lockstep_t == t0 + 3 + iters and agent.t == iters
    assert torch.isfinite(agent.last_loss).item()
    assert any((a != b).any().item() for a, b in zip(before, agent.Q.parameters()))
    m = env.metrics()
    assert m[:6].tolist() == orc_m[:6].tolist() and m[8:12].tolist() == orc_m[8:12].tolist()
    le = env.last_episode_host()
    assert (le["n_episodes"] == orc.field("n_episodes")).all()
    env.close()


def test_config4_fused_sgd_step_matches_torch_on_a_32768_env_replay():
    """sgk_dqn_sgd_step on a replay ring of 4 x 32 768 Sokoban transitions (config 4's batch): the rows it samples are the
    oracle's (orc_minibatch_index over 131 072 transitions), and loss + parameters over three steps equal torch autograd +
    clip_grad_norm_(10) + Adam(amsgrad) on those rows, with the loss on the shapes value.py:119-123 hands mse_loss. Floating point,
    different summation order: the tolerance of test_gpu_deepq.py (_assert_adam_close)."""
    import warnings

    torch = _torch()
    torch.manual_seed(13)
    name, n, seed, slices, hidden, batch = "SideEffectsSokoban-v0", 32768, 41, 4, 100, 64
    env = S.BatchedGridworldEnv(name, n, seed=seed)
    env.bind_torch_stream()
    agent = S.BatchedDeepQAgent(env, _dq_args(lr=1e-2, discount=0.9), replay_slices=slices)
    assert agent.fused_learn
    with torch.no_grad():
        for p in agent.Q.parameters():
            p.mul_(2.0)
        for p in agent.target_Q.parameters():
            p.add_(0.05 * torch.randn_like(p))
    agent._refresh_fused_weights()
    agent._fl["w2t"].copy_(agent.Q[1][0][0].weight.data.t())
    agent._refresh_target_transposes()
    agent.warmup(slices)
    rp = agent.replay
    flat = lambda t: t.reshape(slices * n, *t.shape[2:]).cpu()  # noqa: E731
    st, ac, rw, su, te = flat(rp.states), flat(rp.actions), flat(rp.rewards), flat(rp.successors), flat(rp.terminals)
    cpu_q = agent.build_Q(env.n_cells, 2, hidden)
    cpu_t = agent.build_Q(env.n_cells, 2, hidden)
    cpu_q.load_state_dict({k: v.cpu() for k, v in agent.Q.state_dict().items()})
    cpu_t.load_state_dict({k: v.cpu() for k, v in agent.target_Q.state_dict().items()})
    opt = torch.optim.Adam(cpu_q.parameters(), lr=1e-2, amsgrad=True)
    for step in range(3):
        loss_gpu = float(agent.learn_batch().cpu())
        ix = torch.as_tensor(O.minibatch_indices(seed, step, batch, slices * n))
        assert int(ix.max()) >= n  # the draw really ranges over the whole ring, not one slice
        q_sa = cpu_q(st[ix].float()).gather(1, ac[ix].long().unsqueeze(1))  # [B, 1] against expected [B]: value.py:119-123's broadcast
        with torch.no_grad():
            nq = cpu_t(su[ix].float()).max(1)[0]
            nq = torch.where(te[ix], torch.zeros_like(nq), nq)
            expected = 0.9 * nq + rw[ix].float()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            loss = torch.nn.functional.mse_loss(q_sa, expected)
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(cpu_q.parameters(), 10.0)
        opt.step()
        assert abs(loss_gpu - float(loss.detach())) <= 2e-4 * abs(float(loss.detach())) + 2e-6, (step, loss_gpu, float(loss.detach()))
        from test_gpu_deepq import _assert_adam_close

        for (k, v), (k2, v2) in zip(agent.Q.state_dict().items(), cpu_q.state_dict().items()):
            _assert_adam_close(v.cpu().numpy(), v2.numpy(), 1e-2, step + 1, "%s step %d" % (k, step))
    env.close()


def test_the_documented_maximum_of_two_billion_envs_on_one_gpu_is_bit_exact_at_both_ends():
    """sgk_create's limit, 2^31 - 512 envs (BoatRace, compact boards: 94 GB of the 288): every form of the random rollout a few
    steps each; the first, a middle and the LAST 777 envs -- boards and step records -- equal the oracle keyed by the same global
    env ids, and the step total is 64-bit exact. 64-bit indexing at the far end of every array, in about six seconds."""
    torch = _torch()
    free, _ = torch.cuda.mem_get_info()
    if free < 120e9:
        pytest.skip("needs ~95 GB of free device memory")
    n, name, seed = (1 << 31) - 512, "BoatRace-v0", 5
    env = S.BatchedGridworldEnv(name, n, seed=seed, layout="compact")
    v = env._device_views()
    blocks = [0, n // 2 - 100, n - 777]
    orcs = [O.EnvBatch(name, 777, seed=seed, env_begin=b) for b in blocks]
    total = 0
    for how, k in (("graph", 3), ("stream", 5), ("fused", 7), ("graph", 1)):
        env.step_random(k, auto_reset=True, fused={"graph": False, "stream": "stream", "fused": True}[how])
        env.synchronize()
        for b, orc in zip(blocks, orcs):
            rec = orc.rollout(k, seed=seed, env_begin=b, t_begin=total, auto_reset=True)
            assert (v["boards"][b:b + 777].reshape(777, -1).cpu().numpy() == orc.boards()).all(), (how, b)
            assert (v["rec"][b:b + 777].cpu().numpy() == np.asarray(rec)).all(), (how, b)
        total += k
    assert env.metrics()[_lib.M_STEPS] == total * n
    # one whole episode for everyone: the episode counters and sums are 64-bit, the done-mask compaction covers the id space
    env.step_random(100 - total, auto_reset=True, fused="stream")
    m = env.metrics()
    assert m[_lib.M_EPISODES] == n and m[_lib.M_STEPS] == 100 * n
    ids, ret, perf = env.finished()
    assert ids.numel() == n and int(ids[-1]) == n - 1 and int(ids[0]) == 0
    assert int(ret.to(torch.int64).sum()) == m[_lib.M_SUM_RETURN]
    del v, ids, ret, perf
    env.close()


@pytest.mark.parametrize("name,n,ring,layout", [("BoatRace-v0", 1 << 20, 180, "slice"), ("IslandNavigation-v0", 1 << 20, 100, "slice"),
                                                ("BoatRace-v0", 1 << 20, 180, "tile")])
def test_trajectory_rings_beyond_four_gigabytes_hold_every_step(name, n, ring, layout):
    """A boards ring larger than 2^32 bytes (1 M envs x 180 BoatRace slices = 4.7 GB, x 100 IslandNavigation slices = 5.0 GB;
    slice-major and tile-major): one launch per 60 steps streams into it, and slices on both sides of the 4 GB line hold exactly
    the oracle's boards and records for the first and the last 512 envs."""
    torch = _torch()
    seed = 77
    env = S.BatchedGridworldEnv(name, n, seed=seed, layout="compact")
    cells = env.n_cells
    n_tiles = (n + 63) // 64
    if layout == "slice":
        boards = torch.empty((ring, n, cells), dtype=torch.int8, device="cuda")
        recs = torch.empty((ring, n, 4), dtype=torch.int8, device="cuda")
    else:
        boards = torch.empty((n_tiles, ring, 64, cells), dtype=torch.int8, device="cuda")
        recs = torch.empty((n_tiles, ring, 64, 4), dtype=torch.int8, device="cuda")
    assert boards.numel() > 1 << 32
    blocks = [0, n - 512]
    orcs = [O.EnvBatch(name, 512, seed=seed, env_begin=b) for b in blocks]
    want = {}  # (block, slice) -> (boards, recs)
    check = sorted({0, 1, ring // 2, ring - 25, ring - 2, ring - 1})
    for b, orc in zip(blocks, orcs):
        for k in range(ring):
            rec = orc.rollout(1, seed=seed, env_begin=b, t_begin=k, auto_reset=True)
            if k in check:
                want[(b, k)] = (orc.boards().copy(), np.asarray(rec).copy())
    t = 0
    while t < ring:
        c = min(60, ring - t)
        env.rollout_random_stream(c, boards=boards, recs=recs, first_slice=t, layout=layout)
        t += c
    env.synchronize()
    for b in blocks:
        for k in check:
            if layout == "slice":
                gb, gr = boards[k, b:b + 512], recs[k, b:b + 512]
            else:
                gb = boards[b // 64:(b + 512) // 64, k].reshape(512, cells)
                gr = recs[b // 64:(b + 512) // 64, k].reshape(512, 4)
            assert (gb.cpu().numpy() == want[(b, k)][0]).all(), (layout, b, k)
            assert (gr.cpu().numpy() == want[(b, k)][1]).all(), (layout, b, k)
    del boards, recs
    env.close()
