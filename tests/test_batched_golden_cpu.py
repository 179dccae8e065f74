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


# ---- PPO: the reference's PPOMLPAgent through train(), rollout r = env index base + r ---------------------------

def _ppo_reward(fx, r, h):
    """What gather_rollout stores for a step (policy_base.py:147-158: float(reward), the hidden one under --cheat), as the float32
    get_discounted_returns turns it into (policy_base.py:181)."""
    v = h if fx.cheat else r
    return np.float32(v * fx.scale) if fx.scale != 1.0 else np.float32(v)


@pytest.mark.parametrize("name", BG.PPO_FIXTURES + BG.PPO_CNN_FIXTURES)
def test_oracle_and_host_agent_reproduce_the_reference_ppo_run(name, monkeypatch):
    """Gathering: the oracle's Categorical draw on the reference's logits gives the reference's actions, the oracle env steps to the
    reference's boards / rewards / episode ends (each env index reset as the batched gather resets it), the oracle's discounted
    returns are the reference's bit for bit, the oracle's minibatch rows are the ones the reference drew. Learning: the host
    PPOMLPAgent / PPOCNNAgent (CPU torch) fed the same rollout and rows ends at the reference's weights. Evaluation: the reference's greedy
    episodes replayed on the oracle env, the host agent choosing the same actions."""
    import torch

    import safe_grid_agents_amd as S

    fx = BG.PpoFixture(name)
    T, n, m = fx.horizon, fx.n, fx.meta
    envs = [O.EnvBatch(fx.env, 1, seed=fx.seed, env_begin=fx.base + i) for i in range(n)]
    cells = envs[0].H * envs[0].W
    shim = type("E", (), {"action_space": type("A", (), {"n": 4})(), "observation_space": type("Sp", (), {"shape": (1, envs[0].H, envs[0].W)})()})()
    host = (S.PPOCNNAgent if m.get("agent") == "ppo-cnn" else S.PPOMLPAgent)(shim, fx.args("cpu"))  # (policy_mlp.py / policy_cnn.py)
    host.load_state_dict({k: torch.as_tensor(v) for k, v in fx.weights(0).items()}, strict=False)
    host.sync()
    assert m["min_margin"] > 1e-5
    episodes = {(it, i): (ret, perf) for it, i, ret, perf in m["episodes"]}
    epoch = 0
    for k in range(fx.iterations):
        logits, states, actions = fx.it(k, "logits"), fx.it(k, "states"), fx.it(k, "actions")
        rewards, returns, lengths = fx.it(k, "rewards"), fx.it(k, "returns"), fx.it(k, "lengths")
        for i in range(n):
            b = envs[i]
            b.reset(0)
            for t in range(T):
                assert (b.board(0).ravel() == states[i, t]).all(), (k, i, t)
                a, margin = O.categorical_sample(logits[i, t:t + 1], fx.seed, fx.base + i, k * T + t)
                # the oracle's own margin of every draw: clear of the interval boundaries by what the generator promised (2e-5; 2e-4 once
                # the weights carry a learner's rounding), so float32 rounding in another summation order cannot flip an action
                assert margin[0] > (2e-4 if fx.learn and k > 0 else 2e-5), (k, i, t, margin[0])
                r, h, d, actual = b.step(0, int(a[0]))
                assert actions[i, t] == (actual if fx.cheat else a[0]), (k, i, t)
                assert rewards[i, t] == _ppo_reward(fx, r, h), (k, i, t)
                if d:
                    break
            L = t + 1
            assert L == lengths[i] and (d or L == T)
            assert not states[i, L:].any() and not actions[i, L:].any() and not rewards[i, L:].any()
            ret, perf = episodes[(k, i)]
            assert fx.units(ret) == int(b.field("episode_return")[0]) and fx.units(perf) == b.last_performance(0)
            want = O.discounted_returns(rewards[i, :L], m["discount"])
            assert want.tobytes() == returns[i, :L].tobytes(), (k, i)
            b.reset(0)
        if not fx.learn:
            assert all((fx.weights(k + 1)[key] == fx.weights(0)[key]).all() for key in m["weight_keys"])
            continue
        assert (lengths == T).all()
        rows = fx.it(k, "rows")
        for e in range(m["epochs"]):
            assert (O.ppo_rows(fx.seed, epoch + e, m["batch_size"], lengths, T) == rows[e]).all(), (k, e)
        # the host agent's learn on the same rollout: rollout-major rows (policy_base.py:66-72), the reference's draws
        feed = iter(rows)
        monkeypatch.setattr(torch, "randint", lambda high, size, dtype=None: torch.as_tensor([(q % n) * T + q // n for q in next(feed)]))
        w = S.RecordingWriter()
        host.learn([list(states[i].astype(np.float32).reshape(T, 1, envs[0].H, envs[0].W)) for i in range(n)],
                   [list(actions[i].astype(np.int64)) for i in range(n)], None, [list(returns[i]) for i in range(n)],
                   {"writer": w, "t": 0, "t_learn": epoch}, None)
        monkeypatch.undo()
        host.sync()
        got = np.array([float.fromhex(c[2]) for c in w.calls]).reshape(m["epochs"], 3)
        np.testing.assert_allclose(got, fx.losses(k), rtol=1e-5, atol=1e-7)
        for key, v in fx.weights(k + 1).items():
            np.testing.assert_allclose(host.state_dict()[key].numpy(), v, rtol=1e-5, atol=1e-7, err_msg=key)
        epoch += m["epochs"]
    # default_eval (eval.py:8-56) of every index: whole episodes until at least eval_timesteps steps
    for i, agent in enumerate(fx.agents):
        b, acts, t, ep = envs[i], agent["eval_actions"], 0, 0
        b.reset(0)
        while True:
            obs = b.board(0).astype(np.float32)[np.newaxis]
            assert host.act(obs) == int(acts[t]), (i, t)
            _, _, d, _ = b.step(0, int(acts[t]))
            t += 1
            if d:
                ret, perf = agent["eval_episodes"][ep]
                assert fx.units(ret) == int(b.field("episode_return")[0]) and fx.units(perf) == b.last_performance(0)
                ep += 1
                if t >= fx.eval_timesteps:
                    break
                b.reset(0)
        assert t == len(acts) and ep == len(agent["eval_episodes"])


# ---- DeepQ: tests/golden/batched_dqn_*.npz (the reference's train() with DeepQAgent + dqn_warmup + dqn_learn on one env index) --------

@pytest.mark.parametrize("name", BG.DQN_FIXTURES)
def test_host_deepq_train_reproduces_the_reference_dqn_run(name, monkeypatch):
    """This repo's host mirror -- trainer.train, loops.dqn_warmup / dqn_learn / whiler, agents.DeepQAgent / ReplayBuffer / RandomAgent --
    on the oracle's env of the fixture's env index, its three random calls answered from the oracle's counter-RNG restatements
    (orc_random_action, orc_eps_greedy, orc_minibatch_index: the functions the GPU tests check the kernels' draws against): every
    action, every Train/value_loss (same torch CPU ops in the same order: in practice bit-equal, rtol 1e-5 for another torch build) and
    the final weights equal what the reference's own classes produced. Links the oracle's draw functions and the host agent to
    reference output; the GPU test links the kernels to the same file."""
    import types
    import warnings

    import torch

    import safe_grid_agents_amd as S
    from oracle.gym_shim import OracleGridworldEnv

    fx = BG.DqnFixture(name)
    m = fx.meta
    cap, steps, seed, index = fx.capacity, fx.steps, fx.seed, fx.index
    st = {"warm": 0, "t": 0, "learn": 0}

    class Budget(Exception):
        pass

    class IndexedEnv(OracleGridworldEnv):  # the env of this index as the batch creates it
        def __init__(self, env_name):
            super().__init__(env_name)
            self._b = O.EnvBatch(self._b.env_id, 1, seed=seed, env_begin=index)

    class PhiloxDeepQ(S.DeepQAgent):
        def act_explore(self, state):  # value.py:94-96 with Categorical.sample() answered from stream 2
            if st["t"] == steps:
                raise Budget()
            probs = self.policy(state).probs.numpy()
            with torch.no_grad():
                sc = self.Q(self._lift(np.asarray(state).flatten()).reshape(1, -1)).numpy()
            a = int(O.eps_greedy(sc, self.epsilon, seed, index, st["t"])[0])
            assert probs[a] > 0 and abs(probs.sum() - 1) < 1e-6
            st["t"] += 1
            return a

    def randint(lo, hi):
        st["warm"] += 1
        return int(O.random_action(seed, index, st["warm"] - 1))

    def choice(n, size):
        assert n == cap
        step = st["learn"]
        st["learn"] += 1
        slots = O.minibatch_indices(seed, step, size, cap)
        assert (slots == fx.rows[step]).all()
        return (slots - (step + 1) % cap) % cap  # ring slot -> position in the deque (position 0 = the slot written next)

    monkeypatch.setattr(np.random, "randint", randint)
    monkeypatch.setattr(np.random, "choice", choice)
    monkeypatch.setitem(S.trainer.AGENT_MAP, "deep-q", PhiloxDeepQ)
    args = types.SimpleNamespace(seed=seed, env_alias={v: k for k, v in S.ENV_MAP.items()}[fx.env], agent_alias="deep-q", episodes=10**9,
                                 eval_every=10**9, eval_timesteps=fx.eval_timesteps, eval_visualize_episodes=0, discount=m["discount"],
                                 cheat=fx.cheat, log_dir=None, device="cpu", lr=m["lr"], epsilon=m["epsilon"],
                                 epsilon_anneal=m["epsilon_anneal"], replay_capacity=cap, sync_every=fx.sync_every, n_layers=m["n_layers"],
                                 n_hidden=m["n_hidden"], batch_size=fx.batch, log_gradients=False)
    writers, agents = [], []
    orig_init = PhiloxDeepQ.__init__

    def init(self, env, a):
        orig_init(self, env, a)
        for net, tag in ((self.Q, "init_Q"), (self.target_Q, "init_T")):
            net.load_state_dict({k: torch.as_tensor(v) for k, v in fx.weights(tag).items()})
        agents.append(self)

    monkeypatch.setattr(PhiloxDeepQ, "__init__", init)
    envs = []

    def factory(env_name):
        envs.append(IndexedEnv(env_name))
        return envs[-1]

    def wf(d):
        writers.append(S.RecordingWriter(d))
        return writers[-1]

    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            with pytest.raises(Budget):
                S.train(args, env_factory=factory, writer_factory=wf)
    finally:
        torch.set_num_threads(threads)
    agent, env = agents[0], envs[0]
    assert st == {"warm": cap, "t": steps, "learn": steps}
    assert env.actions_log[:cap] == fx.warm("actions").tolist()
    assert env.actions_log[cap:] == fx.actions.tolist()
    calls = writers[0].calls
    losses = np.array([float.fromhex(c[2]) if isinstance(c[2], str) else float(c[2]) for c in calls if c[1] == "Train/value_loss"])
    np.testing.assert_allclose(losses, fx.losses, rtol=1e-5)
    eps = [c[2] for c in calls if c[1] == "Train/epsilon"]
    assert eps == m["epsilon_written"]
    for tag, net in (("final_Q", agent.Q), ("final_T", agent.target_Q)):
        for k, v in fx.weights(tag).items():
            np.testing.assert_allclose(net.state_dict()[k].numpy(), v, rtol=1e-4, atol=1e-6, err_msg=tag + " " + k)
    assert env._obs().ravel().astype(np.int8).tolist() == m["final_board"]
