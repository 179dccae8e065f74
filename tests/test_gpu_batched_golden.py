"""The HIP kernels of the BATCHED agent path against REFERENCE output (tests/golden/batched_*.npz): the reference's own
train() / TabularQAgent / tabq_learn (value.py:15-58, learn.py:8-26,61-85) and RandomAgent / dqn_warmup (dummy.py:10-16,
warmup.py:8-23), one run per env index with np.random answered from the counter RNG the kernels draw from. One link from the
reference to the kernels -- no host restatement, no oracle in between (tests/test_batched_golden_cpu.py is the oracle's own link).

Run on the GPU box:  python -m pytest tests -m gpu -x -q
"""
import numpy as np
import pytest

import batched_golden as BG
import safe_grid_agents_amd as S
from safe_grid_agents_amd import _lib
from test_batched_golden_cpu import check_warmup_meters, expected_metrics

pytestmark = pytest.mark.gpu


def _torch():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


def _assert_tables_are_the_reference_dictionaries(env, agent, fx):
    """Every row of every reference agent's Q dictionary (keyed by the flattened board, value.py:34) == the row of the state
    index that board hashes to, as float64 bit patterns; and the device holds no other non-zero row."""
    from test_gpu_parity import _board_of_state

    if env.name == "TomatoWatering-v0":
        return _assert_hashed_tables_are_the_reference_dictionaries(env, agent, fx)
    tab = agent.table_host()
    index_of = {}
    for si in range(agent.n_states):
        try:
            index_of[_board_of_state(env, si).tobytes()] = si
        except Exception:  # state indices that name no board (e.g. the agent on a wall cell)
            pass
    for i in range(fx.n):
        claimed = set()
        for board, q in fx.rows_of(i):
            si = index_of.get(board.astype(np.int8).tobytes())
            if si is None:  # a board without a row of its own: terminal boards are never looked up before they are left...
                assert not np.abs(q).any(), (i, board.tolist())  # ...so the reference only ever holds zeros for them
                continue
            assert BG.hexes(tab[i, si]) == BG.hexes(q), (i, si, board.tolist())
            claimed.add(si)
        nz = set(np.nonzero(np.abs(tab[i]).sum(axis=1))[0].tolist())
        assert nz <= claimed, (i, sorted(nz - claimed))


def _assert_hashed_tables_are_the_reference_dictionaries(env, agent, fx):
    """TomatoWatering: per-agent hash tables. Every occupied slot's key -> the board it names (rendered from the product's level tables)
    -> that board's row in the reference agent's dictionary, bit for bit (a slot whose board the reference never looked up holds zeros);
    and every non-zero row of the reference is found in some slot."""
    import hostlib
    import test_tables_cpu as TT

    R = TT._rules(hostlib.load(), S.ENV_IDS[env.name])
    cap, used, overflowed = agent.hash_info()
    assert not overflowed and 0 < used < cap
    keys, tab = agent.keys_host(), agent.table_host()
    for i in range(fx.n):
        want = {b.astype(np.int8).tobytes(): q for b, q in fx.rows_of(i)}
        found = set()
        for slot in np.nonzero(keys[i] != 0xFFFFFFFF)[0]:
            key = int(keys[i, slot])
            cell, shown = key & 0xFF, key >> 8
            board = TT._product_board(env.name, R, cell, shown & 0x1FFF, 0).astype(np.int8).tobytes()
            q = want.get(board)
            if q is None:
                assert not np.abs(tab[i, slot]).any(), (i, hex(key))
            else:
                assert BG.hexes(tab[i, slot]) == BG.hexes(q), (i, hex(key))
                found.add(board)
        missing = [b for b, q in want.items() if b not in found and np.abs(q).any()]
        assert not missing, (i, len(missing))
        assert not np.abs(tab[i][keys[i] == 0xFFFFFFFF]).any()  # unclaimed slots are untouched


def _assert_final_state(env, agent, fx):
    assert agent.t == fx.steps
    assert float(agent.epsilon).hex() == fx.agents[0]["epsilon_at_stop"]
    assert (env.boards_host().reshape(fx.n, -1) == fx.final_boards).all()
    st = env.episode_state_host()
    assert st["episode_return"].tolist() == [fx.units(a["episode_return_at_stop"]) for a in fx.agents]
    got = env.metrics()
    for k, v in expected_metrics(fx, fx.n * fx.steps).items():
        assert int(got[k]) == v, (k, int(got[k]), v)


@pytest.mark.parametrize("name", BG.TABQ_FIXTURES)
def test_device_epsilon_schedule_is_the_reference_agents(name):
    fx = BG.TabqFixture(name)
    lib = _lib.load()
    assert [float(lib.sgk_tabq_epsilon(fx.eps0, fx.anneal, t)).hex() for t in range(fx.steps)] == fx.epsilon_used


@pytest.mark.parametrize("kernel", ["auto", "lds", "hbm"])
@pytest.mark.parametrize("name", BG.TABQ_FIXTURES)
def test_fused_tabq_rollout_kernels_reproduce_the_reference_agents(name, kernel):
    """tabq_rollout_kernel (tables in LDS) / tabq_rollout_hbm_kernel (rows in HBM): what bench.py --config 3 times."""
    _torch()
    fx = BG.TabqFixture(name)
    env = S.BatchedGridworldEnv(fx.env, fx.n, seed=fx.seed)
    agent = S.BatchedTabularQAgent(env, fx.args())
    try:
        try:
            agent.rollout(700, cheat=fx.cheat, kernel=kernel)
        except _lib.SgkError as exc:
            assert kernel == "lds" and "do not fit LDS" in str(exc)
            pytest.skip("%s: tables do not fit LDS" % fx.env)
        agent.rollout(fx.steps - 700, cheat=fx.cheat, kernel=kernel)
        _assert_final_state(env, agent, fx)
        _assert_tables_are_the_reference_dictionaries(env, agent, fx)
    finally:
        agent.close(); env.close()


@pytest.mark.parametrize("name", BG.TABQ_FIXTURES)
def test_batched_default_eval_reproduces_the_reference_default_eval(name):
    """default_eval (eval.py:8-56) per agent, greedy, after the fixture's training steps == batched_default_eval on the device tables:
    every episode's return and performance as the reference's track_metrics calls saw them, aggregated like the metrics vector."""
    from oracle import oracle as O

    _torch()
    fx = BG.TabqFixture(name)
    env = S.BatchedGridworldEnv(fx.env, fx.n, seed=fx.seed)
    agent = S.BatchedTabularQAgent(env, fx.args())
    try:
        agent.rollout(fx.steps, cheat=fx.cheat)
        bm = S.batched_default_eval(agent, env, fx.eval_timesteps)
        BG.assert_eval_metrics(bm.vec, fx, O)
        assert bm.episodes == sum(len(a["eval_episodes"]) for a in fx.agents)
    finally:
        agent.close(); env.close()


@pytest.mark.parametrize("how", ["calls", "step", "graph", "graph_separate"])
@pytest.mark.parametrize("name", BG.TABQ_FIXTURES)
def test_drop_in_call_sequence_reproduces_the_reference_agents(name, how):
    """One lockstep step of tabq_learn, four ways: "calls" = act_explore -> env.step -> learn -> reset_done as four launches made
    from Python, "step" = the same as ONE launch (sgk_tabq_step) -- every step's actions compared with the reference agents' --,
    "graph" = sgk_tabq_learn_steps (a hipGraph of the one-launch step), "graph_separate" = a hipGraph of the four launches."""
    _torch()
    fx = BG.TabqFixture(name)
    env = S.BatchedGridworldEnv(fx.env, fx.n, seed=fx.seed)
    agent = S.BatchedTabularQAgent(env, fx.args())
    try:
        if how == "calls":
            for t in range(fx.steps):
                a = agent.act_explore()
                assert a.cpu().numpy().tolist() == fx.actions[t].tolist(), t
                env.step(a, auto_reset=False, write_boards=(t == fx.steps - 1))
                agent.learn(action=a, cheat=fx.cheat)
                env.reset_done()
        elif how == "step":
            for t in range(fx.steps):
                a, _ = agent.step(cheat=fx.cheat, write_boards=(t == fx.steps - 1))
                assert a.cpu().numpy().tolist() == fx.actions[t].tolist(), t
        else:
            sep = how == "graph_separate"
            done = 0
            for k in (100, 100, 7, 1, 500):
                agent.learn_steps(k, cheat=fx.cheat, separate_launches=sep)
                done += k
            agent.learn_steps(fx.steps - done - 1, cheat=fx.cheat, separate_launches=sep)
            agent.learn_steps(1, cheat=fx.cheat, write_boards=True, separate_launches=sep)
        _assert_final_state(env, agent, fx)
        _assert_tables_are_the_reference_dictionaries(env, agent, fx)
    finally:
        agent.close(); env.close()


@pytest.mark.parametrize("path", ["stream_ring", "launches"])
@pytest.mark.parametrize("name", BG.WARMUP_FIXTURES)
def test_random_rollout_into_a_trajectory_ring_reproduces_the_reference_warmup(name, path):
    """RandomAgent + dqn_warmup per env index: slice t of the trajectory ring == what the reference's replay buffer holds for
    step t (successor board, reward, terminal, action) -- the headline kernel (rollout_random_kernel<stream>, bench.py's `value`)
    and the per-step step_kernel launches."""
    torch = _torch()
    fx = BG.WarmupFixture(name)
    env = S.BatchedGridworldEnv(fx.env, fx.n, seed=fx.seed)
    try:
        reset_board = env.boards_host().reshape(fx.n, -1).copy()
        assert (fx.states == reset_board[None]).all()  # (warmup.py:17,21 never advances `state`: see the CPU test)
        if path == "stream_ring":
            boards = torch.full((fx.steps, fx.n, env.n_cells), -7, dtype=torch.int8, device="cuda")
            recs = torch.full((fx.steps, fx.n, 4), -7, dtype=torch.int8, device="cuda")
            env.rollout_random_stream(fx.steps, boards=boards, recs=recs, first_slice=0)
            got_b, got_r = boards.cpu().numpy(), recs.cpu().numpy()
        else:
            got_b = np.empty((fx.steps, fx.n, env.n_cells), dtype=np.int8)
            got_r = np.empty((fx.steps, fx.n, 4), dtype=np.int8)
            for t in range(fx.steps):
                env.step_random(1, auto_reset=False)
                got_b[t] = env.boards_host().reshape(fx.n, -1)
                got_r[t] = env.step_records_host()
                env.reset_done()
        want_b = fx.successors
        if path == "stream_ring":
            # The one convention that differs, by design: a trajectory ring keeps, at an episode's last step, the observation the
            # agent acts on NEXT (the auto-reset board: `state` of the following row) where the reference's row keeps the terminal
            # board as `successor` -- a board DeepQAgent.learn never uses (value.py:121 zeroes next_Q where terminal). The terminal
            # boards themselves are compared on the per-step path of this same test (no auto-reset: the step leaves them in place).
            want_b = np.where((fx.terminals != 0)[:, :, None], reset_board[None], fx.successors)
        assert (got_b == want_b).all(), np.argwhere((got_b != want_b).any(axis=2))[:3]
        assert (got_r[:, :, 0] == fx.rewards).all()
        assert ((got_r[:, :, 2] != 0) == (fx.terminals != 0)).all()
        assert (got_r[:, :, 3].view(np.uint8) == fx.actions).all()
        check_warmup_meters(fx, env.metrics())
    finally:
        env.close()


# ---- PPO: tests/golden/batched_ppo_*.npz -- the reference's train() with PPOMLPAgent, rollout r = env index base + r,
# Categorical.sample() and torch.randint answered from the counter RNG (streams 3 and 5) ------------------------------------------

def _metrics_agree(vec, want, O):
    got = {"sum_return": vec[O.M_SUM_RETURN], "sum_safety": vec[O.M_SUM_SAFETY], "sum_margin": vec[O.M_SUM_MARGIN],
           "sum_margin_pos": vec[O.M_SUM_MARGIN_POS], "episodes": vec[O.M_EPISODES], "margin_pos_count": vec[O.M_MARGIN_POS_COUNT]}
    for k, v in want.items():
        assert int(got[k]) == v, (k, int(got[k]), v)


@pytest.mark.parametrize("how", ["fused", "stepwise", "torch"])
@pytest.mark.parametrize("name", BG.PPO_FIXTURES)
def test_batched_ppo_reproduces_the_reference_ppo_run(name, how):
    """BatchedPPOAgent from the reference's initial weights: every iteration's gathered rollout (sgk_policy_rollout in one launch, or
    policy draw + env.step per lockstep step) holds the reference's boards, actions, rewards and episode lengths exactly and its
    discounted returns bit for bit; the fused learner (sgk_ppo_epochs) draws the reference's minibatch rows and lands on the
    reference's losses and weights (fp32 in another summation order: rtol 2e-3 / atol 2e-5); batched_default_eval of the trained
    policy books the reference's greedy episodes. No weights are re-loaded between iterations: the second gather runs under the
    weights this path learned (the fixtures' draws keep clear of the interval boundaries by more than that drift)."""
    from oracle import oracle as O

    torch = _torch()
    fx = BG.PpoFixture(name)
    m, T, n = fx.meta, fx.horizon, fx.n
    env = S.BatchedGridworldEnv(fx.env, n, seed=fx.seed, env_index_base=fx.base)
    env.bind_torch_stream()
    agent = S.BatchedPPOAgent(env, fx.args(0))
    try:
        assert agent.fused_policy and agent.fused_learn
        agent.fused_rollout = how == "fused"
        if how == "torch":
            agent.fused_policy = agent.fused_learn = False
        agent.net.load_state_dict({k: torch.as_tensor(v).to(agent.device) for k, v in fx.weights(0).items()}, strict=False)
        agent.sync()
        used = torch.zeros((m["epochs"], m["batch_size"]), dtype=torch.int64, device=agent.device)
        for k in range(fx.iterations):
            env.metrics_reset()
            ro = agent.gather_rollout(cheat=fx.cheat)
            assert tuple(ro.actions.shape) == (T, n)
            lengths = ro.lengths.cpu().numpy()
            assert (lengths == fx.it(k, "lengths")).all(), (k, lengths, fx.it(k, "lengths"))
            got_actions = ro.actions.cpu().numpy().T
            bad = np.argwhere(got_actions != fx.it(k, "actions"))
            assert bad.size == 0, (k, bad[:4], [float(fx.it(k, "margins")[i, t]) for i, t in bad[:4]])
            assert (ro.states.cpu().numpy().transpose(1, 0, 2) == fx.it(k, "states")).all(), k
            assert (ro.rewards.cpu().numpy() == fx.it(k, "rewards")).all(), k
            assert ro.returns.cpu().numpy().tobytes() == fx.it(k, "returns").tobytes(), k
            _metrics_agree(env.metrics(), fx.gather_metrics(k), O)
            if fx.learn and how == "torch":
                w = S.RecordingWriter()
                agent.learn(ro, {"writer": w, "t": 0, "t_learn": 0}, rows=list(fx.it(k, "rows")))  # (every episode lasts T steps: flat rows)
                got = np.array([float.fromhex(c[2]) for c in w.calls]).reshape(m["epochs"], 3)
                np.testing.assert_allclose(got, fx.losses(k), rtol=2e-3, atol=2e-5)
            elif fx.learn:
                agent._learn_fused(ro, rows_out=used)
                assert (used.cpu().numpy() == fx.it(k, "rows")).all(), k
                np.testing.assert_allclose(agent._stats.cpu().numpy().astype(np.float64), fx.losses(k), rtol=2e-3, atol=2e-5)
            if fx.learn:
                sd = agent.net.state_dict()
                for key, v in fx.weights(k + 1).items():
                    np.testing.assert_allclose(sd[key].cpu().numpy(), v, rtol=2e-3, atol=2e-5, err_msg="%s after iteration %d" % (key, k))
            agent.sync()
        bm = S.batched_default_eval(agent, env, fx.eval_timesteps)
        BG.assert_eval_metrics(bm.vec, fx, O)
        assert bm.episodes == sum(len(a["eval_episodes"]) for a in fx.agents)
    finally:
        env.close()


@pytest.mark.parametrize("how", ["rollout", "fused", "graph", "torch"])
@pytest.mark.parametrize("name", BG.PPO_CNN_FIXTURES)
def test_batched_ppo_cnn_reproduces_the_reference_ppo_cnn_run(name, how):
    """The reference's PPOCNNAgent (policy_cnn.py) run by its own train(): BatchedPPOAgent(body="cnn") from the reference's initial
    weights gathers the reference's rollouts -- boards, actions, rewards, episode lengths exactly, returns bit for bit -- with the
    whole gather ONE launch (sgk_convq_rollout: `rollout`, the default), or the trunk + actor forward + Categorical draw of every lockstep
    step one launch (sgk_convq_sample: `fused` eager, `graph` the second gather replayed from a hipGraph), or through the torch module +
    sgk_categorical_sample (`torch`); learn() (torch
    autograd on the conv body, the reference's minibatch rows) lands on the reference's losses and weights to rtol 2e-3 / atol 2e-5,
    the second gather runs under those weights, and batched_default_eval (greedy: the same kernel with epsilon 0) books the reference's
    evaluation episodes. 5, 8 and 4 channels; the fixtures' draws keep clear of the interval boundaries by more than fp32 drift."""
    from oracle import oracle as O

    torch = _torch()
    fx = BG.PpoFixture(name)
    m, T, n = fx.meta, fx.horizon, fx.n
    assert m["agent"] == "ppo-cnn"
    env = S.BatchedGridworldEnv(fx.env, n, seed=fx.seed, env_index_base=fx.base)
    env.bind_torch_stream()
    agent = S.BatchedPPOAgent(env, fx.args(0), body="cnn", fused_conv=how != "torch")
    try:
        assert agent.fused_conv == (how != "torch") and not agent.fused_policy and agent.fused_rollout
        agent.fused_rollout = how == "rollout"
        agent.graph_gather = how == "graph"
        agent.net.load_state_dict({k: torch.as_tensor(v).to(agent.device) for k, v in fx.weights(0).items()}, strict=False)
        agent.sync()
        for k in range(fx.iterations):
            env.metrics_reset()
            ro = agent.gather_rollout(cheat=fx.cheat)
            lengths = ro.lengths.cpu().numpy()
            assert (lengths == fx.it(k, "lengths")).all(), (k, lengths, fx.it(k, "lengths"))
            got_actions = ro.actions.cpu().numpy().T
            bad = np.argwhere(got_actions != fx.it(k, "actions"))
            assert bad.size == 0, (k, bad[:4], [float(fx.it(k, "margins")[i, t]) for i, t in bad[:4]])
            assert (ro.states.cpu().numpy().transpose(1, 0, 2) == fx.it(k, "states")).all(), k
            assert (ro.rewards.cpu().numpy() == fx.it(k, "rewards")).all(), k
            assert ro.returns.cpu().numpy().tobytes() == fx.it(k, "returns").tobytes(), k
            _metrics_agree(env.metrics(), fx.gather_metrics(k), O)
            if fx.learn:
                w = S.RecordingWriter()
                agent.learn(ro, {"writer": w, "t": 0, "t_learn": 0}, rows=list(fx.it(k, "rows")))
                got = np.array([float.fromhex(c[2]) for c in w.calls]).reshape(m["epochs"], 3)
                np.testing.assert_allclose(got, fx.losses(k), rtol=2e-3, atol=2e-5)
                sd = agent.net.state_dict()
                for key, v in fx.weights(k + 1).items():
                    np.testing.assert_allclose(sd[key].cpu().numpy(), v, rtol=2e-3, atol=2e-5, err_msg="%s after iteration %d" % (key, k))
            agent.sync()
        if how == "graph":
            assert agent._gather_graph is not None
        # the logits the kernel computes at the first step of a fresh episode == the reference's (fp32, another summation order)
        env.reset()
        logits = torch.empty((n, 4), dtype=torch.float32, device=agent.device)
        if how != "torch":
            agent.net.old_policy.load_state_dict({k: torch.as_tensor(v).to(agent.device) for k, v in fx.weights(0).items()}, strict=False)
            env.convq_sample(agent._cw_old, 0, agent.n_channels, logits_out=logits)
            np.testing.assert_allclose(logits.cpu().numpy(), fx.it(0, "logits")[:, 0], rtol=1e-4, atol=2e-5)
            agent.sync()
        bm = S.batched_default_eval(agent, env, fx.eval_timesteps)
        BG.assert_eval_metrics(bm.vec, fx, O)
        assert bm.episodes == sum(len(a["eval_episodes"]) for a in fx.agents)
    finally:
        env.close()


# ---- DeepQ (A12): tests/golden/batched_dqn_*.npz -- the reference's train() with DeepQAgent + dqn_warmup + dqn_learn on one env index,
# its random calls answered from the counter RNG (streams 0 / 2 / 4) -------------------------------------------------------------------

def _load_q(torch, net, sd, device):
    net.load_state_dict({k: torch.as_tensor(v).to(device) for k, v in sd.items()})


@pytest.mark.parametrize("name", BG.DQN_FIXTURES)
def test_batched_deepq_n1_reproduces_the_reference_dqn_run(name):
    """An N = 1 BatchedDeepQAgent (sgd_steps = 1, replay ring = replay_capacity slices) from the reference's initial weights:
    warmup() (the streamed random rollout, one launch) leaves the replay the reference's dqn_warmup left (including warmup.py:17-21's
    never-advanced `state`); then every lockstep step = sgk_replay_store + sgk_policy_act + sgk_step + sgk_dqn_sgd_step (loss_mode =
    the reference's [B,1]-vs-[B] broadcast, value.py:119-123) takes the reference's action exactly, draws the reference's minibatch,
    logs the reference's Train/value_loss and, across three sync_target_Q, lands on the reference's weights (fp32 in another summation
    order: see the tolerances below); batched_default_eval of the result books the reference's greedy evaluation episodes."""
    from oracle import oracle as O

    torch = _torch()
    fx = BG.DqnFixture(name)
    env = S.BatchedGridworldEnv(fx.env, 1, seed=fx.seed, env_index_base=fx.index)
    env.bind_torch_stream()
    agent = S.BatchedDeepQAgent(env, fx.args(), sgd_steps=1, replay_slices=fx.capacity)
    try:
        assert agent.fused_policy and agent.fused_learn and agent.reference_loss_broadcast
        _load_q(torch, agent.Q, fx.weights("init_Q"), agent.device)
        _load_q(torch, agent.target_Q, fx.weights("init_T"), agent.device)
        agent._refresh_fused_weights()
        agent._fl["w2t"].copy_(agent.Q[1][0][0].weight.data.t())
        agent._refresh_target_transposes()
        reset_board = env.boards_host().reshape(-1).copy()
        # ---- dqn_warmup ----
        agent.warmup(fx.capacity)
        rp = agent.replay
        assert rp.filled == fx.capacity and rp.head == 0
        term = fx.warm("terminals") != 0
        assert (rp.states[:, 0].cpu().numpy() == fx.warm("states")).all()
        want_succ = np.where(term[:, None], reset_board[None], fx.warm("successors"))  # (the ring's convention at an episode's last step)
        assert (rp.successors[:, 0].cpu().numpy() == want_succ).all()
        assert (rp.actions[:, 0].cpu().numpy() == fx.warm("actions")).all()
        assert (rp.rewards[:, 0].cpu().numpy().astype(np.int32) == fx.warm("rewards")).all()
        assert (rp.terminals[:, 0].cpu().numpy() == term).all()
        # ---- dqn_learn, step by step (train.py:62-70: every episode starts from a reset) ----
        env.reset()
        env.metrics_reset()
        got_actions, got_losses, drift = [], [], 0.0
        syncs = 0
        for t in range(fx.steps):
            assert agent.epsilon == fx.epsilon_used[t], (t, agent.epsilon, fx.epsilon_used[t])
            sc = agent.scores().cpu().numpy()[0]
            drift = max(drift, float(np.abs(sc - fx.scores[t]).max()))
            a = agent.step(learn=True, cheat=fx.cheat)
            got_actions.append(int(a[0]))
            got_losses.append(float(agent.last_loss))
            assert got_actions[-1] == int(fx.actions[t]), (t, got_actions[-1], int(fx.actions[t]), sc, fx.scores[t], float(fx.gaps[t]),
                                                           bool(fx.explored[t]), drift)
            if t % fx.sync_every == fx.sync_every - 1:  # the reference's Q at this sync_target_Q
                for key, v in fx.weights("sync%d_Q" % syncs).items():
                    np.testing.assert_allclose(agent.target_Q.state_dict()[key].cpu().numpy(), v, rtol=DQN_RTOL, atol=DQN_ATOL,
                                               err_msg="%s at sync %d" % (key, syncs))
                syncs += 1
        assert syncs == len(fx.meta["syncs_at"]) >= 1
        print("%s: max |Q - reference Q| over %d steps = %.3g (smallest greedy gap of the fixture %.3g)" % (name, fx.steps, drift,
                                                                                                        fx.meta["min_greedy_gap"]))
        np.testing.assert_allclose(np.array(got_losses), fx.losses, rtol=DQN_LOSS_RTOL, atol=1e-6)
        for tag, net in (("final_Q", agent.Q), ("final_T", agent.target_Q)):
            for key, v in fx.weights(tag).items():
                np.testing.assert_allclose(net.state_dict()[key].cpu().numpy(), v, rtol=DQN_RTOL, atol=DQN_ATOL, err_msg="%s %s" % (tag, key))
        assert int(agent._fl["step"].cpu()) == fx.steps and agent.t == fx.steps
        assert (env.boards_host().reshape(-1) == np.array(fx.meta["final_board"], dtype=np.int8)).all()
        assert int(env.episode_state_host()["episode_return"][0]) == fx.units(fx.meta["episode_return_at_stop"])
        m = env.metrics()
        want = fx.episode_metrics()
        assert int(m[O.M_EPISODES]) == want["episodes"] and int(m[O.M_SUM_RETURN]) == want["sum_return"]
        assert int(m[O.M_SUM_SAFETY]) == want["sum_safety"]
        # ---- default_eval, greedy ----
        bm = S.batched_default_eval(agent, env, fx.eval_timesteps)
        BG.assert_eval_metrics(bm.vec, fx, O)
    finally:
        env.close()


# weights after 300+ Adam(amsgrad) steps in float32 with another summation order than torch's CPU kernels
DQN_RTOL, DQN_ATOL, DQN_LOSS_RTOL = 2e-4, 2e-5, 2e-4
