"""sgk_tabq_step -- one launch per lockstep step of tabq_learn (reference learn.py:61-85 inside train.py:62-70) -- against the four
launches it fuses (sgk_tabq_act, sgk_step, sgk_tabq_learn, sgk_reset_done), step by step and bit for bit, on every level, at sizes
that take both forms of the kernel (one wave per workgroup up to 65 536 envs, grid-stride above), both board layouts, with and
without --cheat; and mixed with the other entry points (the row hand-off between kernels stays coherent)."""
import types

import numpy as np
import pytest

import safe_grid_agents_amd as S

pytestmark = pytest.mark.gpu

LEVELS = sorted(S.envs.ENV_IDS)


def _args(**kw):
    d = dict(lr=0.4, discount=0.95, epsilon=0.15, epsilon_anneal=400)
    d.update(kw)
    return types.SimpleNamespace(**d)


def _snapshot(env, agent):
    st = env.episode_state_host()
    return {"rec": env.step_records_host().copy(), "boards": env.boards_host().copy(), **{k: v.copy() for k, v in st.items()},
            **{k: v.copy() for k, v in env.last_episode_host().items()}}


def _same(a, b, what):
    for k in a:
        assert (a[k] == b[k]).all(), (what, k, np.argwhere(np.atleast_1d(a[k] != b[k]))[:4].tolist())


@pytest.mark.parametrize("cheat", [False, True], ids=["plain", "cheat"])
@pytest.mark.parametrize("layout", ["compact", "pitched"])
@pytest.mark.parametrize("name", LEVELS)
def test_one_launch_step_equals_the_four_launches_step_by_step(name, layout, cheat):
    n, steps, seed = 1500, 260, 77  # (260 steps: every level's episode limit is 100, so episodes end by limit and by terminal cell)
    pair = []
    for _ in range(2):
        env = S.BatchedGridworldEnv(name, n, seed=seed, layout=layout, env_index_base=12345)
        kw = {"hash_capacity": 1024} if name == "TomatoWatering-v0" else {}
        pair.append((env, S.BatchedTabularQAgent(env, _args(**kw))))
    (e4, a4), (e1, a1) = pair
    try:
        for t in range(steps):
            act = a4.act_explore()
            e4.step(act, auto_reset=False)
            rec4 = e4.step_records_host().copy()  # (reset_done leaves the records alone)
            a4.learn(action=act, cheat=cheat)
            e4.reset_done()
            got, (boards, reward, done, info) = a1.step(cheat=cheat)
            assert (got.cpu().numpy() == act.cpu().numpy()).all(), t
            s4, s1 = _snapshot(e4, a4), _snapshot(e1, a1)
            assert (s1["rec"] == rec4).all(), t
            _same(s4, s1, "step %d" % t)
            if t % 64 == 0:  # the views env.step() hands out are the same memory
                assert (boards.cpu().numpy() == s1["boards"]).all() and (reward.cpu().numpy() == s1["rec"][:, 0]).all()
                assert (done.cpu().numpy() == s1["rec"][:, 2]).all()
        assert a1.t == a4.t == steps and e1.lockstep_t == e4.lockstep_t
        assert (e1.metrics() == e4.metrics()).all()
        assert a1.table_host().tobytes() == a4.table_host().tobytes()
        if name == "TomatoWatering-v0":
            assert (a1.keys_host() == a4.keys_host()).all() and a1.hash_info() == a4.hash_info()
    finally:
        for env, agent in pair:
            agent.close(); env.close()


@pytest.mark.parametrize("name,n", [("IslandNavigation-v0", 262144), ("BoatRace-v0", 100000), ("SideEffectsSokoban-v0", 70001)])
def test_one_launch_step_at_sizes_that_take_the_grid_stride_kernel(name, n):
    """Above 65 536 envs the kernel runs 256-lane workgroups over a grid-stride loop of tiles: graphs of it against graphs of the
    four launches, 230 steps; a ragged last tile (70 001)."""
    pair = []
    for _ in range(2):
        env = S.BatchedGridworldEnv(name, n, seed=5)
        pair.append((env, S.BatchedTabularQAgent(env, _args())))
    (e4, a4), (e1, a1) = pair
    try:
        for k in (100, 100, 30):
            a4.learn_steps(k, separate_launches=True, write_boards=True)
            a1.learn_steps(k, write_boards=True)
        _same(_snapshot(e4, a4), _snapshot(e1, a1), name)
        assert (e1.metrics() == e4.metrics()).all()
        lo, cnt = n - 3000, 3000  # tables of the last agents (the ragged tile among them) and of the first
        assert a1.table_host(lo, cnt).tobytes() == a4.table_host(lo, cnt).tobytes()
        assert a1.table_host(0, 3000).tobytes() == a4.table_host(0, 3000).tobytes()
    finally:
        for env, agent in pair:
            agent.close(); env.close()


@pytest.mark.parametrize("name", ["IslandNavigation-v0", "SideEffectsSokoban-v0", "WhiskyGold-v0", "TomatoWatering-v0", "FriendFoe-v0"])
def test_one_launch_step_mixes_with_the_other_entry_points(name):
    """step() between act / learn calls, fused rollouts, table writes from outside and a stray learn(): the kept-row hand-off
    (row cache + tags) must never serve a stale row. Reference: the four-launch sequence doing the same schedule."""
    n, seed = 3000, 9
    kw = {"hash_capacity": 1024} if name == "TomatoWatering-v0" else {}
    pair = []
    for _ in range(2):
        env = S.BatchedGridworldEnv(name, n, seed=seed)
        pair.append((env, S.BatchedTabularQAgent(env, _args(**kw))))
    (e4, a4), (e1, a1) = pair

    def four(k):
        for _ in range(k):
            act = a4.act_explore()
            e4.step(act, auto_reset=False)
            a4.learn(action=act)
            e4.reset_done()

    try:
        for rnd in range(3):
            four(17); [a1.step() for _ in range(17)]                      # one-launch steps ...
            four(5)                                                         # ... then the four calls on the same handle ...
            for _ in range(5):
                act = a1.act_explore()
                e1.step(act, auto_reset=False)
                a1.learn(action=act)
                e1.reset_done()
            a4.rollout(40); a1.rollout(40)                                  # ... a fused rollout (writes the table directly) ...
            four(9); [a1.step() for _ in range(9)]
            a4.learn_steps(12, separate_launches=True); a1.learn_steps(12)  # ... graphs ...
            if name != "TomatoWatering-v0":                                 # ... and a write through the zero-copy view
                for ag in (a4, a1):
                    ag.table()[:, 1, 2] += 0.125
                    ag.invalidate_rows()
            four(11); [a1.step() for _ in range(11)]
            _same(_snapshot(e4, a4), _snapshot(e1, a1), "round %d" % rnd)
            assert a1.table_host().tobytes() == a4.table_host().tobytes()
        a1.learn()  # a learn() without an act() in front of it: nothing pending after a fused step -> nothing learnt
        assert a1.table_host().tobytes() == a4.table_host().tobytes()
        assert (e1.metrics() == e4.metrics()).all()
    finally:
        for env, agent in pair:
            agent.close(); env.close()


def test_tabq_step_argument_errors():
    from safe_grid_agents_amd import _lib

    env = S.BatchedGridworldEnv("BoatRace-v0", 64)
    agent = S.BatchedTabularQAgent(env, _args())
    lib = env.lib
    assert lib.sgk_tabq_step(None, 0, 0, None) == _lib.ERR_INVALID
    assert lib.sgk_tabq_step(agent._h, 0, _lib.F_AUTO_RESET, None) == _lib.ERR_INVALID  # only SGK_F_NO_BOARDS is meaningful
    assert lib.sgk_tabq_step(agent._h, 0, 0, None) == 0  # actions_out_dev may be NULL
    assert agent.t == 1
    agent.close(); env.close()
