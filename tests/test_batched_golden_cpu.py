"""The oracle's batched loops against REFERENCE output for the batched path's own inputs (tests/golden/batched_*.npz: the
reference's train() / TabularQAgent / tabq_learn and RandomAgent / dqn_warmup, one run per env index, np.random answered from the
counter RNG). One link from the reference to the oracle; tests/test_gpu_batched_golden.py is the same link to the HIP kernels."""
import numpy as np
import pytest

import batched_golden as BG
from oracle import oracle as O


def expected_metrics(fx, n_steps_total):
    sums, counts, maxs = fx.metrics()
    want = {O.M_SUM_RETURN: sums["returns"], O.M_SUM_SAFETY: sums["safeties"], O.M_SUM_MARGIN: sums["margins"],
            O.M_SUM_MARGIN_POS: sums["margins_support"], O.M_EPISODES: counts["returns"],
            O.M_MARGIN_POS_COUNT: counts["margins_support"], O.M_STEPS: n_steps_total}
    if counts["returns"]:
        want[O.M_MAX_RETURN], want[O.M_MAX_SAFETY], want[O.M_MAX_MARGIN] = maxs["returns"], maxs["safeties"], maxs["margins"]
    if counts["margins_support"]:
        want[O.M_MAX_MARGIN_POS] = maxs["margins_support"]
    return want


def assert_metrics(got, fx, n_steps_total):
    sums, counts, _ = fx.metrics()
    assert counts["safeties"] == counts["returns"] == counts["margins"]  # these levels all define a performance
    for k, v in expected_metrics(fx, n_steps_total).items():
        assert int(got[k]) == v, (k, int(got[k]), v)


@pytest.mark.parametrize("name", BG.TABQ_FIXTURES)
def test_oracle_tabq_rollout_reproduces_the_reference_agents(name):
    fx = BG.TabqFixture(name)
    # the schedule the reference's agent was in at every step (value.py:23-28,54-58) == the closed form, bit for bit
    assert [float(O.epsilon(fx.eps0, fx.anneal, t)).hex() for t in range(fx.steps)] == fx.epsilon_used
    assert fx.steps > fx.anneal  # the fixture crosses the end of the anneal: epsilon freezes at the value for anneal - 1
    assert fx.epsilon_used[-1] == fx.epsilon_used[fx.anneal - 1] != fx.epsilon_used[fx.anneal - 2]
    assert all(a["epsilon_at_stop"] == float(O.epsilon(fx.eps0, fx.anneal, fx.steps)).hex() for a in fx.agents)
    for key, (u, a) in fx.meta["draw_probe"].items():  # the draws the reference consumed are the stream the oracle defines
        i, t = (int(x) for x in key.split(","))
        got_u, got_a = O.explore_draw(fx.seed, i, t)
        assert (float(got_u).hex(), got_a) == (u, a)

    orc = O.EnvBatch(fx.env, fx.n, seed=fx.seed)
    agents = [O.TabQ(orc.H * orc.W, fx.lr, fx.discount, fx.eps0, fx.anneal) for _ in range(fx.n)]
    m = O.metrics_new()
    acts = O.tabq_rollout(orc, agents, fx.steps, seed=fx.seed, cheat=fx.cheat, metrics=m, record_actions=True)
    bad = np.argwhere(acts != fx.actions)
    assert bad.size == 0, ("first differing (step, agent)", bad[0].tolist())
    assert (orc.boards() == fx.final_boards).all()
    assert orc.field("episode_return").tolist() == [fx.units(a["episode_return_at_stop"]) for a in fx.agents]
    assert_metrics(m, fx, fx.n * fx.steps)
    explored = 0
    for i in range(fx.n):
        rows = fx.rows_of(i)
        assert agents[i].n_rows == len(rows), i  # the dictionary holds the same boards (value.py:31: inserted on first lookup)
        for board, q in rows:
            assert BG.hexes(agents[i].lookup(board)) == BG.hexes(q), (i, board.tolist())
        explored += len(rows)
    assert explored == len(fx.q_agent) > fx.n
    # default_eval (eval.py:8-56) with every agent as trained so far, in the lockstep form of batched_default_eval: eval_timesteps - 1
    # iterations of {greedy step, reset finished envs}, then steps without reset until every env's episode has ended
    orc.reset()
    em = O.metrics_new()
    max_it = 100

    def greedy():
        return np.array([agents[i].act(orc.board(i)) for i in range(fx.n)], dtype=np.uint8)

    for _ in range(fx.eval_timesteps - 1):
        orc.rollout(1, seed=fx.seed, actions=greedy()[None], auto_reset=False, metrics=em)  # (seed: the envs' own draws)
        for i in np.nonzero(orc.field("game_over"))[0]:
            orc.reset(int(i))
    for _ in range(max_it):
        orc.rollout(1, seed=fx.seed, actions=greedy()[None], auto_reset=False, metrics=em)  # (seed: the envs' own draws)
    assert orc.field("game_over").all()
    BG.assert_eval_metrics(em, fx, O)
    # episodes really ended inside the fixture (the loop's reset path) and exploration really fired (both branches of value.py:37)
    assert int(m[O.M_EPISODES]) >= fx.n
    greedy_only = O.tabq_rollout(O.EnvBatch(fx.env, fx.n, seed=fx.seed),
                                 [O.TabQ(orc.H * orc.W, fx.lr, fx.discount, 0.0, 1) for _ in range(fx.n)], 50, seed=fx.seed,
                                 cheat=fx.cheat, record_actions=True)
    assert (greedy_only != fx.actions[:50]).any() or fx.eps0 == 0.0


@pytest.mark.parametrize("name", BG.WARMUP_FIXTURES)
def test_oracle_random_rollout_reproduces_the_reference_warmup(name):
    """RandomAgent + dqn_warmup (dummy.py:15-16, warmup.py:14-21): what the replay buffer holds after `steps` random steps of env
    index i == what the oracle's random rollout leaves step by step (the rows a trajectory ring keeps)."""
    fx = BG.WarmupFixture(name)
    assert (O.random_actions(fx.seed, 0, fx.n, 0, fx.steps) == fx.actions).all()
    orc = O.EnvBatch(fx.env, fx.n, seed=fx.seed)
    reset_board = orc.boards().copy()
    m = O.metrics_new()
    # Experience.state: warmup.py:17 assigns `state` at a reset and line 21 never advances it, so every row of an episode holds the
    # reset board (a quirk of the reference the fixture records; a trajectory ring keeps successors and step records only)
    assert (fx.states == reset_board[None]).all()
    for t in range(fx.steps):
        rec = orc.rollout(1, seed=fx.seed, t_begin=t, auto_reset=False, metrics=m)
        assert (orc.boards() == fx.successors[t]).all(), t
        assert (rec[:, 0] == fx.rewards[t]).all() and (rec[:, 2] == fx.terminals[t]).all() and (rec[:, 3] == fx.actions[t]).all(), t
        for i in np.nonzero(rec[:, 2])[0]:
            orc.reset(int(i))
    assert fx.terminals.sum() >= fx.n  # episode ends are inside the fixture
    check_warmup_meters(fx, m)


def check_warmup_meters(fx, m):
    """warmup.py:12-17 books an episode's return when the NEXT iteration starts, after one spurious update(0) before the first
    reset: the meter holds [0] + the returns of the episodes that ended before the last step. The batch's metrics vector counts an
    episode at its last step and has no spurious entry."""
    total, count, best = 0, 0, None
    for i, meter in enumerate(fx.meters):
        hist = list(meter["history"])  # AverageMeter keeps it sorted (meters.py:40)
        assert meter["count"] == len(hist) and meter["sum"] == sum(hist) and meter["max"] == max(hist)
        hist.remove(0)  # the spurious first update: episode_return of the freshly made env
        if fx.terminals[-1, i]:  # ended on the very last step: never booked by the reference's loop
            hist.append(meter["episode_return_at_stop"])
        total, count = total + sum(hist), count + len(hist)
        best = max(hist + ([best] if best is not None else []))
    assert int(m[O.M_SUM_RETURN]) == total and int(m[O.M_EPISODES]) == count == int(fx.terminals.sum())
    assert int(m[O.M_MAX_RETURN]) == best
