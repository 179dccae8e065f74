"""BASELINE config 4 pieces on the GPU: Sokoban boards -> float32 obs kernel -> shared MLP -> batched epsilon-greedy,
device replay, SGD. Floating point: forward parity vs a CPU float32 evaluation of the same weights, rtol 1e-4 /
atol 1e-4 (GEMM accumulation order differs between rocBLAS and the CPU); everything integer stays bit-exact."""
import types

import numpy as np
import pytest

import safe_grid_agents_amd as S
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _assert_adam_close(got, want, lr, n_steps, what):
    """Parameters after n_steps Adam steps, float32 in two summation orders: rtol 2e-4 / atol 2e-6 -- except that Adam's first steps
    divide every gradient element by its own magnitude, so the few elements whose gradient is a sum cancelling to rounding level
    (relative error of the sum ~1e-3) move by lr x that error per step: at most 0.5 % of a tensor's elements may lie outside the
    tolerance, and none by more than 1e-2 x lr x n_steps."""
    bad = np.abs(got - want) > 2e-6 + 2e-4 * np.abs(want)
    assert bad.mean() <= 5e-3, (what, int(bad.sum()), bad.size)
    assert np.abs(got - want).max() <= 1e-2 * lr * n_steps + 2e-6 + 2e-4 * np.abs(want).max(), (what, float(np.abs(got - want).max()))


def _args(**kw):
    d = dict(discount=0.99, lr=1e-3, batch_size=64, sync_every=20, epsilon=0.05, epsilon_anneal=200, n_layers=2,
             n_hidden=100)
    d.update(kw)
    return types.SimpleNamespace(**d)


def test_forward_and_greedy_actions_match_cpu_fp32():
    import torch

    torch.manual_seed(3)
    env = S.BatchedGridworldEnv("SideEffectsSokoban-v0", 4096, seed=9, layout="compact")
    env.bind_torch_stream()
    env.step_random(23, auto_reset=True)
    agent = S.BatchedDeepQAgent(env, _args())
    scores = agent.scores().cpu().numpy()
    cpu_net = agent.build_Q(36, 2, 100)
    cpu_net.load_state_dict({k: v.cpu() for k, v in agent.Q.state_dict().items()})
    obs = torch.as_tensor(env.boards_host().reshape(4096, -1).astype(np.float32))
    with torch.no_grad():
        want = cpu_net(obs).numpy()
    np.testing.assert_allclose(scores, want, rtol=1e-4, atol=1e-4)
    # argmax agrees wherever the top-2 gap is above the tolerance
    srt = np.sort(want, axis=1)
    clear = (srt[:, -1] - srt[:, -2]) > 1e-3
    assert clear.mean() > 0.9
    assert (agent.act().cpu().numpy()[clear] == want.argmax(1)[clear]).all()
    env.close()


def test_epsilon_schedule_and_exploration_mix():
    import torch

    torch.manual_seed(0)
    env = S.BatchedGridworldEnv("SideEffectsSokoban-v0", 1 << 15, seed=1)
    env.bind_torch_stream()
    agent = S.BatchedDeepQAgent(env, _args(epsilon=0.1, epsilon_anneal=50))
    assert agent.epsilon == 1.0  # DeepQAgent keeps future_eps[0] (value.py:76)
    greedy = agent.act().cpu().numpy()
    a = agent.act_explore().cpu().numpy()
    assert abs((a != greedy).mean() - 0.75) < 0.02 and set(a.tolist()) <= {0, 1, 2, 3}  # eps = 1: uniform over 4
    for _ in range(60):
        agent.update_epsilon()
    assert agent.epsilon == 1.0 - (1 - 0.1) * 49 / 50  # frozen at t = anneal - 1
    a = agent.act_explore().cpu().numpy()
    assert abs((a != greedy).mean() - agent.epsilon * 0.75) < 0.02
    env.close()


def test_pipeline_steps_learns_and_env_stays_bit_exact_vs_oracle():
    import torch

    torch.manual_seed(5)
    n = 2048
    env = S.BatchedGridworldEnv("SideEffectsSokoban-v0", n, seed=4, layout="compact")
    env.bind_torch_stream()
    orc = O.EnvBatch("SideEffectsSokoban-v0", n)
    agent = S.BatchedDeepQAgent(env, _args(), sgd_steps=2, replay_slices=4)
    agent.warmup(3)
    assert len(agent.replay) == 3 * n
    orc_m = O.metrics_new()
    # replay the warm-up actions on the oracle from the stored slices
    for k in range(3):
        acts = agent.replay.actions[k].cpu().numpy()
        rec = orc.rollout(1, actions=acts[None], auto_reset=False, metrics=orc_m)
        assert (agent.replay.rewards[k].cpu().numpy() == rec[:, 0]).all()
        assert (agent.replay.terminals[k].cpu().numpy() == rec[:, 2].astype(bool)).all()
        assert (agent.replay.successors[k].cpu().numpy() == orc.boards()).all()
        for i in np.nonzero(orc.field("game_over"))[0]:
            orc.reset(int(i))
    before = [p.detach().clone() for p in agent.Q.parameters()]
    for t in range(40):
        prev = env.boards_host().reshape(n, -1)
        acts = agent.step(learn=True).cpu().numpy()
        k = (agent.replay.head - 1) % agent.replay.slices
        assert (agent.replay.states[k].cpu().numpy() == prev).all()
        rec = orc.rollout(1, actions=acts[None], auto_reset=False, metrics=orc_m)
        assert (agent.replay.rewards[k].cpu().numpy() == rec[:, 0]).all()
        assert (agent.replay.successors[k].cpu().numpy() == orc.boards()).all()
        for i in np.nonzero(orc.field("game_over"))[0]:
            orc.reset(int(i))
        assert (env.boards_host().reshape(n, -1) == orc.boards()).all()
    assert torch.isfinite(agent.last_loss).item()
    assert any((a != b).any().item() for a, b in zip(before, agent.Q.parameters()))
    assert agent.t == 40
    # target sync happened at t = 19 and t = 39
    for p, q in zip(agent.Q.parameters(), agent.target_Q.parameters()):
        assert torch.equal(p, q)
    m = env.metrics()
    assert m[:6].tolist() == orc_m[:6].tolist() and m[8:12].tolist() == orc_m[8:12].tolist()
    env.close()


def test_graph_captured_iteration_matches_oracle_env_and_learns():
    import torch

    torch.manual_seed(7)
    n = 4096
    env = S.BatchedGridworldEnv("SideEffectsSokoban-v0", n, seed=2, layout="compact")
    env.bind_torch_stream()
    orc = O.EnvBatch("SideEffectsSokoban-v0", n)
    orc_m = O.metrics_new()
    agent = S.BatchedDeepQAgent(env, _args(sync_every=7), sgd_steps=1, replay_slices=3)

    def mirror(slice_k):
        acts = agent.replay.actions[slice_k].cpu().numpy()
        rec = orc.rollout(1, actions=acts[None], auto_reset=False, metrics=orc_m)
        assert (agent.replay.rewards[slice_k].cpu().numpy() == rec[:, 0]).all()
        assert (agent.replay.successors[slice_k].cpu().numpy() == orc.boards()).all()
        for i in np.nonzero(orc.field("game_over"))[0]:
            orc.reset(int(i))

    agent.warmup(3)
    for k in range(3):
        mirror(k)
    t_before = env.lockstep_t
    agent.enable_graphs(learn=True)   # 3 eager warm-up iterations on a side stream + capture
    assert env.lockstep_t == t_before + 3
    for k in range(3):                # the warm-up iterations wrote slices 0, 1, 2 again
        mirror(k)
    before = [p.detach().clone() for p in agent.Q.parameters()]
    for it in range(25):
        k = agent.replay.head
        agent.step_graphed(learn=True)
        torch.cuda.synchronize()
        mirror(k)
        assert (env.boards_host().reshape(n, -1) == orc.boards()).all(), it
    assert env.lockstep_t == t_before + 3 + 25 and agent.t == 25
    assert torch.isfinite(agent.last_loss).item()
    assert any((a != b).any().item() for a, b in zip(before, agent.Q.parameters()))
    m = env.metrics()
    assert m[:6].tolist() == orc_m[:6].tolist() and m[_lib_M_STEPS()] == (3 + 3 + 25) * n
    # the no-learning graph
    agent.enable_graphs(learn=False)
    for _ in range(5):
        agent.step_graphed(learn=False)
    torch.cuda.synchronize()
    assert env.lockstep_t == t_before + 3 + 25 + 3 + 5
    env.close()


def _lib_M_STEPS():
    from safe_grid_agents_amd import _lib

    return _lib.M_STEPS


def test_epsilon_greedy_kernel_bit_exact_vs_oracle_and_eager_equals_graphed_stream():
    import torch

    torch.manual_seed(1)
    n, seed, base = 5000, 77, 1 << 33
    env = S.BatchedGridworldEnv("SideEffectsSokoban-v0", n, seed=seed, env_index_base=base)
    scores = torch.randn(n, 4, device="cuda")
    scores[::7, 1] = scores[::7, 0]  # ties: the first maximum wins
    sc = scores.cpu().numpy()
    for eps, draw in [(0.0, 0), (0.3, 5), (1.0, 123456), (0.05, 2**31 + 3)]:
        got = env.epsilon_greedy(scores, eps, draw).cpu().numpy()
        want = O.eps_greedy(sc, eps, seed, base, draw)
        assert (got == want).all(), (eps, draw)
        # the device-scalar form reads the same values
        e = torch.tensor([eps], dtype=torch.float64, device="cuda")
        d = torch.tensor([draw], dtype=torch.int64, device="cuda")
        assert (env.epsilon_greedy(scores, e, d).cpu().numpy() == want).all()
    assert (env.epsilon_greedy(scores, 0.0, 9).cpu().numpy() == sc.argmax(1)).all()
    env.close()


@pytest.mark.parametrize("name", ["SideEffectsSokoban-v0", "BoatRace-v0", "IslandNavigation-v0", "DistributionalShift-v0",
                                  "ConveyorBelt-v0", "FriendFoe-v0"])
@pytest.mark.parametrize("layout", ["compact", "pitched"])
def test_fused_policy_kernel_matches_torch_forward(name, layout):
    """sgk_policy_act (boards -> MLP -> argmax / eps-greedy in one launch) vs the same network evaluated by torch on the CPU
    in fp32. Floating point: rtol 1e-4 / atol 1e-4 on the scores (different summation order); actions equal wherever the
    top-2 gap is above that tolerance; the exploration draws are the counter RNG's, bit-exact vs the oracle."""
    import torch

    torch.manual_seed(4)
    n, seed = 3001, 12
    env = S.BatchedGridworldEnv(name, n, seed=seed, layout=layout)
    env.bind_torch_stream()
    env.step_random(17, auto_reset=True)
    agent = S.BatchedDeepQAgent(env, _args())
    assert agent.fused_policy
    with torch.no_grad():  # make the weights less tiny than the default init so that scores spread
        for p in agent.Q.parameters():
            p.mul_(3.0)
    agent._fw_stale = True
    agent._refresh_fused_weights()
    scores = torch.zeros(n, 4, device="cuda")
    greedy = env.policy_act(agent._fw, 0.0, 0, scores_out=scores).cpu().numpy()
    cpu_net = agent.build_Q(env.n_cells, 2, 100)
    cpu_net.load_state_dict({k: v.cpu() for k, v in agent.Q.state_dict().items()})
    obs = torch.as_tensor(env.boards_host().reshape(n, -1).astype(np.float32))
    with torch.no_grad():
        want = cpu_net(obs).numpy()
    got = scores.cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-4)
    srt = np.sort(want, axis=1)
    clear = (srt[:, -1] - srt[:, -2]) > 1e-3
    assert clear.mean() > 0.9 and (greedy[clear] == want.argmax(1)[clear]).all()
    # exploration: same draws as the stand-alone eps-greedy kernel / the oracle, applied to the kernel's own scores
    a = env.policy_act(agent._fw, 0.4, 77).cpu().numpy()
    assert (a == O.eps_greedy(got, 0.4, seed, 0, 77)).all()
    # agent-level: act() is the greedy path, act_explore() at eps = 1 is ~uniform
    assert (agent.act().cpu().numpy() == greedy).all()
    agent.t = 0
    assert abs((agent.act_explore().cpu().numpy() != greedy).mean() - 0.75) < 0.04
    env.close()


@pytest.mark.parametrize("name,hidden", [("SideEffectsSokoban-v0", 64), ("DistributionalShift-v0", 128), ("BoatRace-v0", 128),
                                         ("IslandNavigation-v0", 64)])
def test_fused_policy_kernel_other_hidden_widths(name, hidden):
    """The MFMA policy kernel is instantiated for 64 and 128 hidden units besides the reference's 100: scores vs the torch
    network on the CPU in fp32 (rtol 1e-4 / atol 1e-4), draws bit-exact vs the oracle on the kernel's own scores."""
    import torch

    torch.manual_seed(9)
    n, seed = 1777, 5
    env = S.BatchedGridworldEnv(name, n, seed=seed)
    env.bind_torch_stream()
    env.step_random(11, auto_reset=True)
    agent = S.BatchedDeepQAgent(env, _args(n_hidden=hidden))
    assert agent.fused_policy
    with torch.no_grad():
        for p in agent.Q.parameters():
            p.mul_(3.0)
    agent._fw_stale = True
    agent._refresh_fused_weights()
    scores = torch.zeros(n, 4, device="cuda")
    a = env.policy_act(agent._fw, 0.3, 21, scores_out=scores).cpu().numpy()
    cpu_net = agent.build_Q(env.n_cells, 2, hidden)
    cpu_net.load_state_dict({k: v.cpu() for k, v in agent.Q.state_dict().items()})
    obs = torch.as_tensor(env.boards_host().reshape(n, -1).astype(np.float32))
    with torch.no_grad():
        want = cpu_net(obs).numpy()
    got = scores.cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-4)
    assert (a == O.eps_greedy(got, 0.3, seed, 0, 21)).all()
    sampled = env.policy_sample(agent._fw, 8).cpu().numpy()
    want_s, margin = O.categorical_sample(got, seed, 0, 8)
    clear = margin > 1e-6
    assert clear.mean() > 0.999 and (sampled[clear] == want_s[clear]).all()
    bad = types.SimpleNamespace(**{**vars(_args()), "n_hidden": 72})
    assert not S.BatchedDeepQAgent(env, bad).fused_policy  # other widths take the torch forward + sgk_epsilon_greedy
    env.close()


@pytest.mark.parametrize("name", ["SideEffectsSokoban-v0", "WhiskyGold-v0", "AbsentSupervisor-v0", "SafeInterruptibility-v0",
                                  "ConveyorBelt-v0", "TomatoWatering-v0", "FriendFoe-v0"])
def test_fused_greedy_eval_and_act_rollout_equal_the_stepwise_paths(name):
    """batched_default_eval through sgk_policy_rollout (two launches) == the per-step loop (act, step, reset_done) on the same
    agent; act_rollout(n, eps) == n calls of policy_act + step with the same draw indices. Integer results: exact."""
    import torch

    torch.manual_seed(5)
    n, seed = 777, 9
    outs = []
    for fused in (True, False):
        env = S.BatchedGridworldEnv(name, n, seed=seed)
        env.bind_torch_stream()
        torch.manual_seed(5)
        agent = S.BatchedDeepQAgent(env, _args())
        with torch.no_grad():
            for p in agent.Q.parameters():
                p.mul_(3.0)
        agent._fw_stale = True
        if not fused:
            agent.greedy_weights = lambda: None  # force the per-step loop
        bm = S.batched_default_eval(agent, env, 57)
        outs.append((np.asarray(bm.vec).copy(), env.boards_host().copy(), {k: v.copy() for k, v in env.episode_state_host().items()}))
        # epsilon-greedy acting with frozen weights: one launch vs the per-step calls
        env.reset()
        env.metrics_reset()
        agent.t = 1000
        if fused:
            agent.act_rollout(40, epsilon=0.3, auto_reset=True)
        else:
            agent._refresh_fused_weights()
            for k in range(40):
                a = env.policy_act(agent._fw, 0.3, 1000 + k)
                env.step(a, auto_reset=True)
        outs[-1] += (np.asarray(env.metrics()).copy(), env.boards_host().copy(), env.episode_state_host()["episode_return"].copy())
        env.close()
    f, s = outs
    assert (f[0] == s[0]).all() and f[0][S.metering.M_EPISODES] >= n
    assert (f[1] == s[1]).all() and all((f[2][k] == s[2][k]).all() for k in f[2])
    assert (f[3] == s[3]).all() and (f[4] == s[4]).all() and (f[5] == s[5]).all()


@pytest.mark.parametrize("name,hidden,batch", [("SideEffectsSokoban-v0", 100, 64), ("BoatRace-v0", 64, 32), ("DistributionalShift-v0", 100, 48),
                                               ("TomatoWatering-v0", 100, 64), ("ConveyorBelt-v0", 64, 48), ("FriendFoe-v0", 100, 32)])
@pytest.mark.parametrize("broadcast", [True, False], ids=["reference_broadcast_loss", "per_sample_loss"])
def test_fused_dqn_sgd_step_equals_torch_autograd_adam(name, hidden, batch, broadcast):
    """sgk_dqn_sgd_step (sampling, both forwards, TD target, MSE, backward, clip_grad_norm_(10), Adam amsgrad in ONE kernel)
    vs the same update by torch on the CPU in fp32 with the minibatch the kernel's counter RNG selects (indices restated by
    the oracle). broadcast: F.mse_loss on Qs [B,1] against expected_Qs [B], the shapes value.py:119-123 hands it (the kernel's
    SGK_DQN_LOSS_REFERENCE, the default); else the squeezed per-sample loss (SGK_DQN_LOSS_PER_SAMPLE). Floating point, different
    summation order: parameters (see _assert_adam_close) and loss (rtol 2e-4) agree over three consecutive steps; the transposed copies
    the kernel maintains equal the updated weights exactly. (The reference's own learn() output is pinned by
    test_fused_dqn_sgd_step_reproduces_the_reference_learn_steps and tests/test_gpu_batched_golden.py.)"""
    import warnings

    import torch

    torch.manual_seed(13)
    n, seed, slices = 600, 41, 4
    env = S.BatchedGridworldEnv(name, n, seed=seed)
    env.bind_torch_stream()
    agent = S.BatchedDeepQAgent(env, _args(n_hidden=hidden, batch_size=batch, lr=1e-2, discount=0.9), replay_slices=slices,
                                reference_loss_broadcast=broadcast)
    assert agent.fused_learn
    with torch.no_grad():
        for p in agent.Q.parameters():
            p.mul_(2.0)
        for p in agent.target_Q.parameters():
            p.add_(0.05 * torch.randn_like(p))
    agent._refresh_fused_weights()
    agent._fl["w2t"].copy_(agent.Q[1][0][0].weight.data.t())
    agent._refresh_target_transposes()
    agent.warmup(slices)  # random-action slices fill the ring
    rp = agent.replay
    flat = lambda t: t.reshape(slices * n, *t.shape[2:]).cpu()  # noqa: E731
    st, ac, rw, su, te = flat(rp.states), flat(rp.actions), flat(rp.rewards), flat(rp.successors), flat(rp.terminals)
    # the reference update on the CPU
    cpu_q = agent.build_Q(env.n_cells, 2, hidden)
    cpu_t = agent.build_Q(env.n_cells, 2, hidden)
    cpu_q.load_state_dict({k: v.cpu() for k, v in agent.Q.state_dict().items()})
    cpu_t.load_state_dict({k: v.cpu() for k, v in agent.target_Q.state_dict().items()})
    opt = torch.optim.Adam(cpu_q.parameters(), lr=1e-2, amsgrad=True)
    for step in range(3):
        loss_gpu = float(agent.learn_batch().cpu())
        ix = torch.as_tensor(O.minibatch_indices(seed, step, batch, slices * n))
        q_sa = cpu_q(st[ix].float()).gather(1, ac[ix].long().unsqueeze(1))  # [B, 1], as value.py:119
        with torch.no_grad():
            nq = cpu_t(su[ix].float()).max(1)[0]
            nq = torch.where(te[ix], torch.zeros_like(nq), nq)
            expected = 0.9 * nq + (rw[ix].double() * float(env.reward_scale)).float()  # the replay holds integer rewards (tomato: counts)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")  # torch warns about exactly this broadcast
            loss = torch.nn.functional.mse_loss(q_sa if broadcast else q_sa.squeeze(1), expected)
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(cpu_q.parameters(), 10.0)
        opt.step()
        assert abs(loss_gpu - float(loss.detach())) <= 2e-4 * abs(float(loss.detach())) + 2e-6, (step, loss_gpu, float(loss.detach()))
        for (k, v), (k2, v2) in zip(agent.Q.state_dict().items(), cpu_q.state_dict().items()):
            _assert_adam_close(v.cpu().numpy(), v2.numpy(), 1e-2, step + 1, "%s step %d" % (k, step))
    assert int(agent._fl["step"].cpu()) == 3
    assert (agent._fw["w1t"].cpu() == agent.Q[0][0].weight.data.t().cpu()).all()
    assert (agent._fl["w2t"].cpu() == agent.Q[1][0][0].weight.data.t().cpu()).all()
    assert (agent._fw["w3t"].cpu() == agent.Q[2].weight.data.t().cpu()).all()
    env.close()


def test_fused_dqn_sgd_step_reproduces_the_reference_learn_steps():
    """tests/golden/deepq_learn.npz -- the reference's own DeepQAgent.learn (value.py:113-136) called 14 times, its replay positions
    recorded -- through sgk_dqn_sgd_step: the same transitions written into a 10-slot device ring (slot = step % capacity, as the
    deque evicts), the same minibatch handed in as `rows`, a target sync after step 7. Every step's loss and the final weights of
    the Q-network equal the reference's to fp32 tolerance (another summation order): this is the kernel against reference output,
    including the [B,1]-vs-[B] mse_loss broadcast -- the losses of the squeezed form differ from step 2 on by far more."""
    import json
    import os

    import torch

    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "deepq_learn.npz"))
    meta = json.loads(str(z["meta"]))
    cap, B, steps = meta["replay_capacity"], meta["batch_size"], meta["steps"]
    assert (meta["H"], meta["W"]) == (6, 6) and meta["n_hidden"] in (64, 100)
    outs = {}
    for broadcast in (True, False):
        env = S.BatchedGridworldEnv("SideEffectsSokoban-v0", 1, seed=1)  # a 6 x 6 level: the handle gives the kernel its board size
        env.bind_torch_stream()
        agent = S.BatchedDeepQAgent(env, _args(n_hidden=meta["n_hidden"], batch_size=B, lr=meta["lr"], discount=meta["discount"]),
                                    replay_slices=cap, reference_loss_broadcast=broadcast)
        assert agent.fused_learn
        dev = agent.device
        for net, tag in ((agent.Q, "init_Q_"), (agent.target_Q, "init_T_")):
            net.load_state_dict({k: torch.as_tensor(z[tag + k.replace(".", "_")]).to(dev) for k in net.state_dict().keys()})
        agent._refresh_fused_weights()
        agent._fl["w2t"].copy_(agent.Q[1][0][0].weight.data.t())
        agent._refresh_target_transposes()
        rp = agent.replay
        losses = []
        used = torch.zeros(B, dtype=torch.int64, device=dev)
        for k in range(steps):
            slot = k % cap  # ReplayBuffer.add (contain.py:15-17): the deque evicts its oldest entry = the ring overwrites slot k % capacity
            rp.states[slot, 0] = torch.as_tensor(z["states"][k].reshape(-1).astype(np.int8)).to(dev)
            rp.successors[slot, 0] = torch.as_tensor(z["successors"][k].reshape(-1).astype(np.int8)).to(dev)
            rp.actions[slot, 0] = int(z["actions"][k])
            rp.rewards[slot, 0] = int(z["rewards"][k])
            rp.terminals[slot, 0] = bool(z["terminals"][k])
            rp.filled = min(k + 1, cap)
            oldest = (k + 1) % cap if k + 1 >= cap else 0  # the ring slot of deque position 0
            rows = torch.as_tensor((oldest + z["sample_ix"][k]) % rp.filled, dtype=torch.int64).to(dev)
            losses.append(float(agent._learn_batch_fused(rows=rows, rows_out=used).cpu()))
            assert (used.cpu() == rows.cpu()).all()
            if k + 1 == meta["sync_after_step"]:
                agent.sync_target_Q()
        outs[broadcast] = (np.array(losses), {k: v.cpu().numpy() for k, v in agent.Q.state_dict().items()})
        env.close()
    losses, final = outs[True]
    np.testing.assert_allclose(losses, z["losses"], rtol=2e-4)
    for k, v in final.items():
        np.testing.assert_allclose(v, z["final_Q_" + k.replace(".", "_")], rtol=2e-4, atol=2e-6, err_msg=k)
    assert not np.allclose(outs[False][0][1:], z["losses"][1:], rtol=1e-2)  # the per-sample loss is another computation


def test_train_batched_cli_deepq():
    """`python -m safe_grid_agents_amd -N 512 sokoban deep-q ...`: warm-up fills the slice replay, every lockstep step learns
    (the fused SGD kernel), the evaluation is greedy through the fused rollout."""
    args = S.prepare_parser().parse_args(["-S", "2", "-E", "2", "-EE", "2", "-V", "110", "-N", "512", "sokoban", "deep-q", "-l", "0.001",
                                          "-r", "2000", "-s", "50", "-b", "32", "-dl", "300"])
    writers = []

    def wf(d):
        writers.append(S.RecordingWriter(d))
        return writers[-1]

    agent, env = S.train_batched(args, writer_factory=wf)
    assert isinstance(agent, S.BatchedDeepQAgent) and agent.fused_learn and agent.replay.slices == 4
    tags = [c[1] for c in writers[0].calls]
    assert tags.count("Train/returns") == 2 and "Train/value_loss" in tags and "Evaluation/returns" in tags
    assert agent.t == 200 and int(agent._fl["step"].cpu()) == 200  # one SGD step per lockstep step
    env.close()


@pytest.mark.parametrize("how", ["step", "graph"])
@pytest.mark.parametrize("name,n,hidden", [("SideEffectsSokoban-v0", 4133, 100), ("BoatRace-v0", 600, 64), ("TomatoWatering-v0", 500, 100),
                                           ("FriendFoe-v0", 70000, 100), ("IslandNavigation-v0", 1, 100)])
def test_sgd_step_with_the_reset_in_its_adam_launch_equals_the_two_calls(name, n, hidden, how):
    """sgk_dqn_sgd_step_reset_store (the SGD kernel, then ONE launch for Adam + reset_done + the next transitions' states) against
    sgk_dqn_sgd_step followed by sgk_reset_done_store, on twin agents over 40 learning steps (eager calls and the captured iteration):
    weights, Adam state, the whole replay ring, boards, records, episode arrays and metrics are identical bit for bit."""
    import torch

    agents = []
    for fuse in (True, False):
        torch.manual_seed(13)
        env = S.BatchedGridworldEnv(name, n, seed=21)
        env.bind_torch_stream()
        agent = S.BatchedDeepQAgent(env, _args(n_hidden=hidden, epsilon_anneal=30, sync_every=7), replay_slices=6)
        assert agent.fused_learn
        agent.fuse_reset = fuse
        agent.warmup(6)
        if how == "graph":
            agent.enable_graphs(learn=True)
        for k in range(40):
            if how == "graph":
                agent.step_graphed(learn=True)
            else:
                agent.step(learn=True)
        torch.cuda.synchronize()
        agents.append((env, agent))
    (ea, a), (eb, b) = agents
    for pa, pb in zip(a.Q.parameters(), b.Q.parameters()):
        assert torch.equal(pa, pb)
    for key in ("m", "v", "vmax"):
        for x, y in zip(a._fl[key], b._fl[key]):
            assert torch.equal(x, y), key
    for key in ("states", "successors", "actions", "rewards", "terminals"):
        assert torch.equal(getattr(a.replay, key), getattr(b.replay, key)), key
    assert a.replay.head == b.replay.head and a.t == b.t == 40
    assert torch.equal(ea.boards(), eb.boards())
    va, vb = ea._device_views(), eb._device_views()
    for key in ("rec", "last_return", "last_performance", "n_episodes"):
        assert torch.equal(va[key], vb[key]), key
    assert (np.asarray(ea.metrics()) == np.asarray(eb.metrics())).all()
    ea.close()
    eb.close()


def test_conv_q_body_option_shapes_env_parity_and_learning_on_boat_race():
    """`q_body="cnn"`: the NON-PARITY conv Q-body (the reference's DeepQAgent is an MLP, value.py:148-158; BASELINE config 4 says
    "conv policy") -- policy_cnn.py's trunk with a Q head, through PyTorch-ROCm. Shapes, the env under it stays bit-exact against the
    oracle, and it learns: on BoatRace the mean TD loss falls and the greedy policy's observed return beats the random walk's."""
    import torch

    torch.manual_seed(11)
    n = 1024
    env = S.BatchedGridworldEnv("BoatRace-v0", n, seed=9, layout="compact")
    env.bind_torch_stream()
    orc = O.EnvBatch("BoatRace-v0", n)
    agent = S.BatchedDeepQAgent(env, _args(lr=2e-3, epsilon=0.1, epsilon_anneal=150, sync_every=25, n_channels=4), sgd_steps=2,
                                replay_slices=16, q_body="cnn", reference_loss_broadcast=False)  # (the textbook loss: the test asks for learning)
    assert agent.q_body == "cnn" and not agent.fused_policy and not agent.fused_learn
    obs = env.obs_f32()
    q = agent.scores(obs)
    assert tuple(q.shape) == (n, 4) and q.dtype == torch.float32
    n_params = sum(p.numel() for p in agent.Q.parameters())
    assert n_params == (4 * 9 + 4) + (4 * 4 * 9 + 4) + (4 + 4) + (4 * 4 * 9 + 4) + (4 * 25 * 4 + 4)  # conv3, conv3, 1x1, head conv3, linear
    agent.warmup(8)
    for k in range(8):  # the env under the agent is the same env: replay the stored actions through the oracle
        acts = agent.replay.actions[k].cpu().numpy()
        orc.rollout(1, actions=acts[None], auto_reset=False)
        assert (agent.replay.successors[k].cpu().numpy() == orc.boards()).all()
    losses = []
    for t in range(300):
        agent.step(learn=True)
        losses.append(float(agent.last_loss))
    assert np.isfinite(losses).all() and np.mean(losses[-50:]) < np.mean(losses[:50])
    env.reset()
    env.metrics_reset()
    for _ in range(100):
        agent.step(learn=False, explore=False)
    greedy = S.BatchMetrics(env.metrics()).meter("returns")["avg"]
    assert greedy > -40.0, greedy  # a random walk averages about -62 observed return per 100-step episode
    env.close()


@pytest.mark.parametrize("cheat", [False, True], ids=["plain", "cheat"])
@pytest.mark.parametrize("layout", ["compact", "pitched"])
@pytest.mark.parametrize("name,n", [("SideEffectsSokoban-v0", 1), ("BoatRace-v0", 600), ("WhiskyGold-v0", 4133), ("IslandNavigation-v0", 70000),
                                    ("TomatoWatering-v0", 1000), ("FriendFoe-v0", 777)])
def test_fused_step_store_and_reset_store_equal_the_separate_launches(name, n, layout, cheat):
    """sgk_step_store (= sgk_step + sgk_replay_store phase 1) and sgk_reset_done_store (= sgk_reset_done + phase 0 for the next step)
    against the launches they fuse, slice by slice over a wrapping 5-slice ring: the rings, the env's own boards / records / state words
    and the metrics are identical; with the slice read from device memory (graph form) too. Sizes: one env, ragged last tiles, both
    forms of the step kernel (one wave per workgroup up to 65 536 envs, grid-stride above)."""
    import torch

    S_, steps, seed = 5, 23, 19
    dev = "cuda"
    envs = [S.BatchedGridworldEnv(name, n, seed=seed, layout=layout) for _ in range(3)]
    reps = [S.DeviceReplay(n, e.n_cells, S_, dev) for e in envs]
    for r in reps:
        for t in (r.states, r.successors, r.actions, r.rewards):
            t.fill_(-7 if t.dtype == torch.int8 else 9)
        r.terminals.fill_(False)
    sep, fus, cap = zip(envs, reps)
    g = torch.Generator(device=dev).manual_seed(3)
    try:
        for k in range(steps):
            acts = torch.randint(0, 4, (n,), device=dev, generator=g).to(torch.uint8)
            # the separate launches
            e, r = sep
            r.store(e, 0)
            e.step(acts, auto_reset=False)
            r.store(e, 1, acts, cheat)
            e.reset_done()
            # fused, slice index from the host
            e, r = fus
            if not r.states_ready(e):
                assert k == 0
                r.store(e, 0)
            r.step_store(e, acts, cheat)
            r.reset_store(e)
            # fused, slice index read by the launch from device memory
            e, r = cap
            if k == 0:
                r.head_dev.fill_(r.head)
                r.store(e, 0, captured=True)
            r.step_store(e, acts, cheat, captured=True)
            r.head_dev.add_(1).remainder_(S_)
            r.note_replayed_add()
            r.reset_store(e, captured=True)
        ref_e, ref_r = sep
        # (the fused forms have already written the NEXT transition's states into slice `head`: the separate form has not yet)
        ref_r.store(ref_e, 0)
        for e, r in (fus, cap):
            for what in ("states", "successors", "actions", "rewards", "terminals"):
                a, b = getattr(ref_r, what).cpu().numpy(), getattr(r, what).cpu().numpy()
                assert (a == b).all(), (what, np.argwhere(a != b)[:3].tolist())
            assert r.head == ref_r.head and r.filled == ref_r.filled == S_
            assert (e.boards_host() == ref_e.boards_host()).all() and (e.step_records_host() == ref_e.step_records_host()).all()
            st, st0 = e.episode_state_host(), ref_e.episode_state_host()
            assert all((st[k2] == st0[k2]).all() for k2 in st)
            assert (e.metrics() == ref_e.metrics()).all() and e.lockstep_t == ref_e.lockstep_t
    finally:
        for e in envs:
            e.close()
