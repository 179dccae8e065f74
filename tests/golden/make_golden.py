#!/usr/bin/env python3
"""Generate tests/golden/*.json|npz by IMPORTING THE REFERENCE in the build container.

Run from anywhere:  python tests/golden/make_golden.py
Needs /root/reference (read-only; never present on the GPU box). Only the produced vectors are
committed -- inputs and expected outputs -- never reference source or bytecode
(sys.dont_write_bytecode is set before anything is imported from the reference).

What runs here is the reference's own code: train.train (train.py:21-81), whiler/tabq_learn
(common/learn.py:8-85), default_eval (common/eval.py:8-56), dqn_warmup (common/warmup.py:8-23),
TabularQAgent/DeepQAgent (common/agents/value.py), RandomAgent (common/agents/dummy.py),
AverageMeter/track_metrics (common/utils/meters.py), ReplayBuffer (common/utils/contain.py) and
the YAML argparse front end (parsing/parse.py). The packages the reference merely imports but which
are absent from this image (gym, tensorboardX, safe_grid_gym, ai_safety_gridworlds) are stubbed with
the minimum: gym.make returns this repo's oracle env shim (oracle/gym_shim.py), SummaryWriter
records its calls. The env behind the fixtures is therefore the repo's CPU restatement (parity with
the upstream env: UNPINNED, see oracle/sgk_oracle.c); what the fixtures PIN is the reference's
agent / loop / meter arithmetic and control flow on that env.
"""
import json
import os
import sys
import types

sys.dont_write_bytecode = True

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = "/root/reference"

sys.path.insert(0, REPO)
sys.path.insert(0, REF)
os.chdir(REF)  # the reference opens its YAML files cwd-relative (parsing/__init__.py:3-5)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle.gym_shim import OracleGridworldEnv  # noqa: E402


class RecordingWriter:
    def __init__(self, log_dir=None):
        self.calls = []

    @staticmethod
    def _num(v):
        if v is None:
            return None
        if isinstance(v, (bool, np.bool_)):
            return bool(v)
        if isinstance(v, (int, np.integer)):
            return int(v)
        return float(v).hex()

    def add_scalar(self, tag, value, step):
        self.calls.append(["scalar", tag, self._num(value), int(step)])

    def add_scalars(self, tag, d, step):
        self.calls.append(["scalars", tag, {k: self._num(v) for k, v in d.items()}, int(step)])

    def add_text(self, tag, text):
        self.calls.append(["text", tag, str(text)])

    def add_video(self, tag, tensor, step):
        self.calls.append(["video", tag, list(tensor.shape), int(step)])

    def add_histogram(self, tag, values, step):
        self.calls.append(["histogram", tag, int(step)])


_made_envs = []
_PARSER = None


def _install_stubs():
    asg = types.ModuleType("ai_safety_gridworlds")
    asg_env = types.ModuleType("ai_safety_gridworlds.environments")
    asg_tom = types.ModuleType("ai_safety_gridworlds.environments.tomato_crmdp")
    asg_tom.REWARD_FACTOR = 0.02
    sys.modules.update({"ai_safety_gridworlds": asg, "ai_safety_gridworlds.environments": asg_env,
                        "ai_safety_gridworlds.environments.tomato_crmdp": asg_tom})
    gym = types.ModuleType("gym")

    def make(name):
        env = OracleGridworldEnv(name)
        _made_envs.append(env)
        return env

    gym.make = make
    sys.modules["gym"] = gym
    tbx = types.ModuleType("tensorboardX")
    tbx.SummaryWriter = RecordingWriter
    sys.modules["tensorboardX"] = tbx
    sys.modules["safe_grid_gym"] = types.ModuleType("safe_grid_gym")


def _dump(name, obj):
    path = os.path.join(HERE, name)
    with open(path, "w") as f:
        json.dump(obj, f, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


def golden_train(name, argv):
    """Run the reference's train(args) end to end; record every writer call, every env action, final Q."""
    import train as ref_train  # /root/reference/train.py
    from safe_grid_agents.parsing import prepare_parser
    import safe_grid_agents.common.agents.value as value_mod

    global _PARSER
    if _PARSER is None:  # prepare_parser() consumes module-level YAML dicts (parse.py:70): call it once
        _PARSER = prepare_parser()
    args = _PARSER.parse_args(argv)
    args.device = "cpu"  # main.py:19-20 with --disable-cuda
    args.log_dir = "unused"
    captured = {}
    orig_init = value_mod.TabularQAgent.__init__

    def spy_init(self, env, a):
        orig_init(self, env, a)
        captured["agent"] = self

    value_mod.TabularQAgent.__init__ = spy_init
    writers = []
    orig_writer = sys.modules["tensorboardX"].SummaryWriter

    class W(RecordingWriter):
        def __init__(self, log_dir=None):
            super().__init__(log_dir)
            writers.append(self)

    ref_train.SummaryWriter = W
    reporter_calls = []
    del _made_envs[:]
    try:
        ref_train.train(args, reporter=lambda **kw: reporter_calls.append(
            {k: RecordingWriter._num(v) for k, v in kw.items()}))
    finally:
        value_mod.TabularQAgent.__init__ = orig_init
        ref_train.SummaryWriter = orig_writer
    agent = captured["agent"]
    q = sorted(([int(x) for x in key], [float(v).hex() for v in row]) for key, row in agent.Q.items())
    calls = [c for c in writers[0].calls if c[0] != "text"]
    _dump(name, {
        "argv": argv,
        "args": {k: v for k, v in vars(args).items() if isinstance(v, (int, float, str, bool, type(None)))},
        "writer_calls": calls,
        "actions": _made_envs[0].actions_log,
        "reporter_calls": reporter_calls,
        "final_Q": q,
        "final_epsilon": float(agent.epsilon).hex(),
        "np_random_next_u32": int(np.random.randint(0, 2**32, dtype=np.uint64)),
    })


def golden_train_ppo(name, argv):
    """The reference's train(args) with a PPO agent (policy_base.py / policy_mlp.py / policy_cnn.py, learn.py:88-104)
    on the CPU: every writer call (losses as hex floats), every env action, and the final weights."""
    import warnings

    import train as ref_train
    from safe_grid_agents.parsing import prepare_parser
    import safe_grid_agents.common.agents.policy_base as pb

    global _PARSER
    if _PARSER is None:
        _PARSER = prepare_parser()
    args = _PARSER.parse_args(argv)
    args.device = "cpu"
    args.log_dir = "unused"
    captured = {}
    orig_init = pb.PPOBaseAgent.__init__

    def spy_init(self, env, a):
        orig_init(self, env, a)
        captured.setdefault("agent", self)

    pb.PPOBaseAgent.__init__ = spy_init
    writers = []
    orig_writer = sys.modules["tensorboardX"].SummaryWriter

    class W(RecordingWriter):
        def __init__(self, log_dir=None):
            super().__init__(log_dir)
            writers.append(self)

    ref_train.SummaryWriter = W
    del _made_envs[:]
    threads = torch.get_num_threads()
    torch.set_num_threads(1)  # multi-threaded conv backward reduces in a thread-count-dependent order
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")  # torch.tensor(tensor) copy-construct warnings (policy_mlp.py:32)
            ref_train.train(args)
    finally:
        torch.set_num_threads(threads)
        pb.PPOBaseAgent.__init__ = orig_init
        ref_train.SummaryWriter = orig_writer
    agent = captured["agent"]
    weights = {k: [float(x).hex() for x in v.detach().double().flatten()[:8].tolist()] + [float(v.detach().double().sum()).hex()]
               for k, v in agent.state_dict().items() if not k.startswith("old_")}
    calls = [c for c in writers[0].calls if c[0] != "text"]
    _dump(name, {
        "argv": argv,
        "args": {k: v for k, v in vars(args).items() if isinstance(v, (int, float, str, bool, type(None)))},
        "writer_calls": calls,
        "actions": [int(a) for a in _made_envs[0].actions_log],
        "final_weights_head8_and_sum": weights,
        "torch_next_randint": int(torch.randint(1 << 30, (1,)).item()),
        "torch_version": torch.__version__,
    })


def golden_epsilon():
    from safe_grid_agents.common.agents.value import TabularQAgent

    out = []
    for eps, anneal in [(0.01, 100000), (0.05, 7), (0.3, 1), (0.0, 50)]:
        ns = types.SimpleNamespace(discount=0.99, epsilon=eps, epsilon_anneal=anneal, lr=0.5)
        env = types.SimpleNamespace(action_space=types.SimpleNamespace(n=4))
        ag = TabularQAgent(env, ns)
        seq = [float(ag.epsilon).hex()]  # epsilon in force for step 0
        for _ in range(min(anneal + 5, 12)):
            seq.append(float(ag.update_epsilon()).hex())
        probe = {}
        if anneal > 1000:  # far probes by direct evaluation of the list the ctor builds
            ag2 = TabularQAgent(env, ns)
            fe = ag2.future_eps  # element k is the epsilon for step k+1
            for t in (1, 2, 3, 4, 999, 50000, anneal - 2, anneal - 1):
                probe[str(t)] = float(fe[t - 1]).hex()
            probe["len_after_ctor"] = len(fe)
        out.append({"epsilon": eps, "anneal": anneal, "first": seq, "probe": probe})
    _dump("epsilon_schedule.json", out)


def golden_meters():
    from safe_grid_agents.common.utils.meters import AverageMeter, make_meters, track_metrics

    script = [(-60, -20), (12, 4), (-100, None), (30, 30), (5, 9), (44, -3), (0, 0), (-7, -8)]

    class FakeEnv:
        def __init__(self):
            self.episode_return, self.perf = 0, None

        def get_last_performance(self):
            return self.perf

    def run(eval_mode):
        w = RecordingWriter()
        h = make_meters({})
        h["writer"] = w
        h["episode"], h["period"] = 0, 0
        env = FakeEnv()
        snaps = []
        for i, (ret, perf) in enumerate(script):
            env.episode_return, env.perf = ret, perf
            h["episode"] += 1
            if eval_mode:
                h["period"] = i // 3
            track_metrics(h, env, eval=eval_mode, write=(not eval_mode) or (i % 3 == 2))
            snaps.append({k: {"val": RecordingWriter._num(h[k].val), "avg": RecordingWriter._num(h[k].avg),
                              "sum": RecordingWriter._num(h[k].sum), "count": h[k].count,
                              "max": RecordingWriter._num(h[k].max)}
                          for k in ("returns", "safeties", "margins", "margins_support")})
        qs = {str(d): float(h["returns"].quantile(d)).hex() for d in (0.1, 0.5, 0.9)}
        return {"calls": w.calls, "snapshots": snaps, "quantiles": qs, "history": list(h["returns"]._history)}

    m = AverageMeter()
    try:
        m.quantile(0.5)
        raised = False
    except RuntimeError:
        raised = True
    _dump("meters.json", {"script": script, "train": run(False), "eval": run(True), "no_history_raises": raised})


def golden_rng():
    from safe_grid_agents.common.agents.dummy import RandomAgent

    env = types.SimpleNamespace(action_space=types.SimpleNamespace(n=4))
    out = {}
    for seed in (0, 1, 7):
        np.random.seed(12345)  # so that seed 0 ("falsy": dummy.py:12 does not reseed) is reproducible
        ag = RandomAgent(env, types.SimpleNamespace(seed=seed))
        acts = [int(ag.act(None)) for _ in range(512)]
        samples = [float(np.random.sample()).hex() for _ in range(32)]
        choice = [int(np.random.choice(4)) for _ in range(64)]
        out[str(seed)] = {"acts": acts, "samples": samples, "choice": choice}
    _dump("numpy_rng.json", out)


def golden_warmup():
    from safe_grid_agents.common.warmup import dqn_warmup
    from safe_grid_agents.common.utils.meters import make_meters
    from safe_grid_agents.common.utils.contain import ReplayBuffer

    env = OracleGridworldEnv("IslandNavigation-v0")
    env.reset()  # reference warmup reads env._env.episode_return before its first reset
    args = types.SimpleNamespace(seed=3, replay_capacity=300)
    agent = types.SimpleNamespace(replay=ReplayBuffer(args.replay_capacity))
    hist = make_meters({})
    np.random.seed(99)
    dqn_warmup(agent, env, hist, args)
    buf = list(agent.replay._buffer)
    np.random.seed(5)
    sample_ix = [int(i) for i in np.random.choice(len(buf), 16)]
    _dump("dqn_warmup.json", {
        "seed": 3, "replay_capacity": 300, "actions": env.actions_log,
        "agent_cells": [int(np.argwhere(e.successor.ravel() == 2).ravel()[0]) if (e.successor == 2).any() else -1
                        for e in buf],
        "rewards": [int(e.reward) for e in buf], "terminals": [bool(e.terminal) for e in buf],
        "returns_meter": {"count": hist["returns"].count, "sum": int(hist["returns"].sum),
                          "max": int(hist["returns"].max), "history": [int(x) for x in hist["returns"]._history]},
        "sample_seed": 5, "sample_ix": sample_ix,
    })


def golden_deepq_forward():
    """DeepQAgent forward/act/policy on 2-D (H, W) observations (the only shape the reference's
    n_input = shape[0] * shape[1] handles, value.py:66-67). learn() cannot run under torch 2.10
    (uint8 mask, value.py:121,179) so the training step is NOT pinned."""
    from safe_grid_agents.common.agents.value import DeepQAgent

    H, W = 6, 6
    env = types.SimpleNamespace(action_space=types.SimpleNamespace(n=4),
                                observation_space=types.SimpleNamespace(shape=(H, W)))
    args = types.SimpleNamespace(device="cpu", log_gradients=False, epsilon=0.01, epsilon_anneal=100000,
                                 discount=0.99, lr=1e-3, batch_size=64, n_layers=2, n_hidden=100,
                                 replay_capacity=100)
    torch.manual_seed(11)
    agent = DeepQAgent(env, args)
    rng = np.random.RandomState(4)
    boards = rng.randint(0, 6, size=(32, H, W)).astype(np.float32)
    with torch.no_grad():
        scores = np.stack([agent.Q(torch.as_tensor(b.flatten()).reshape(1, -1)).numpy()[0] for b in boards])
        acts = np.array([int(agent.act(b)[0]) for b in boards])
        eps_first = float(agent.epsilon)  # DeepQAgent keeps future_eps[0] = 1.0 (no overwrite, value.py:76)
        agent.epsilon = 0.25
        probs = np.stack([agent.policy(b).probs.numpy() for b in boards])
    sd = {k.replace(".", "_"): v.numpy() for k, v in agent.Q.state_dict().items()}
    np.savez_compressed(os.path.join(HERE, "deepq_forward.npz"), boards=boards, scores=scores, acts=acts, probs=probs,
                        eps_first=np.float64(eps_first), eps_policy=np.float64(0.25), **sd)
    print("wrote deepq_forward.npz; state_dict keys:", list(agent.Q.state_dict().keys()))


def golden_deepq_learn():
    """DeepQAgent.learn (value.py:113-136) step by step: replay add + sample (np.random.choice, contain.py:21), the
    [B,1]-vs-[B] mse_loss broadcast (value.py:119-123), the undetached target network, clip_grad_norm_ 10, Adam(amsgrad),
    sync_target_Q in the middle. As written the method stops at `next_Qs[terminals] = 0` under torch >= 2 ("masked_fill_ only
    supports boolean masks"): terminals are lifted as uint8 (value.py:179), which the torch of the reference's day read as
    the boolean mask. The ONE shim here: a subclass whose _lift turns a requested uint8 into bool; every other line that
    runs is the reference's own. Recorded: the transitions fed in, both networks' initial weights, every step's loss and the
    weights after the last step."""
    import warnings

    from safe_grid_agents.common.agents.value import DeepQAgent

    class MaskAsBool(DeepQAgent):
        def _lift(self, x, dtype=torch.float32, grad=False):
            return super()._lift(x, dtype=torch.bool if dtype == torch.uint8 else dtype, grad=grad)

    import safe_grid_agents.common.utils.contain as contain_mod

    H, W, B, STEPS, SYNC_AT, HIDDEN = 6, 6, 8, 14, 7, 64  # (64 hidden units: a width sgk_dqn_sgd_step is built for, so the kernel can be fed these steps)
    env = types.SimpleNamespace(action_space=types.SimpleNamespace(n=4),
                                observation_space=types.SimpleNamespace(shape=(H, W)))
    args = types.SimpleNamespace(device="cpu", log_gradients=False, epsilon=0.01, epsilon_anneal=1000,
                                 discount=0.99, lr=1e-3, batch_size=B, n_layers=2, n_hidden=HIDDEN, replay_capacity=10)
    sampled = []

    def recording_choice(n, size):  # contain.py:21's own call, on numpy's global stream; the positions it returns are recorded
        ix = np.random.choice(n, size)
        sampled.append((int(n), [int(i) for i in ix]))
        return ix

    orig_contain_np = contain_mod.np
    contain_mod.np = _NumpyProxy(choice=recording_choice)
    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        torch.manual_seed(23)
        agent = MaskAsBool(env, args)
        init_q = {k: v.clone() for k, v in agent.Q.state_dict().items()}
        init_t = {k: v.clone() for k, v in agent.target_Q.state_dict().items()}
        rng = np.random.RandomState(17)
        writer = RecordingWriter()
        hist = {"writer": writer, "t": 0}
        fed = {"states": [], "actions": [], "rewards": [], "successors": [], "terminals": []}
        np.random.seed(31)  # ReplayBuffer.sample draws from the global numpy stream
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")  # the broadcast warning of mse_loss
            for k in range(STEPS):
                s0 = rng.randint(0, 6, size=(H, W)).astype(np.float32)
                s1 = rng.randint(0, 6, size=(H, W)).astype(np.float32)
                a, r, term = int(rng.randint(4)), float(rng.choice([-1.0, 2.0, 49.0, -51.0])), bool(rng.rand() < 0.3)
                for key, v in zip(("states", "actions", "rewards", "successors", "terminals"), (s0, a, r, s1, term)):
                    fed[key].append(v)
                hist["t"] = k
                agent.learn(s0, a, r, s1, term, hist)
                if k + 1 == SYNC_AT:
                    agent.sync_target_Q()
    finally:
        torch.set_num_threads(threads)
        contain_mod.np = orig_contain_np
    assert [n for n, _ in sampled] == [min(k + 1, 10) for k in range(STEPS)]

    def hexes(sd):  # float32 arrays: an .npz keeps the bits
        return {k.replace(".", "_"): v.detach().numpy().copy() for k, v in sd.items()}

    np.savez_compressed(
        os.path.join(HERE, "deepq_learn.npz"),
        states=np.stack(fed["states"]), successors=np.stack(fed["successors"]), actions=np.array(fed["actions"]),
        rewards=np.array(fed["rewards"]), terminals=np.array(fed["terminals"]),
        losses=np.array([float.fromhex(c[2]) for c in writer.calls if c[1] == "Train/value_loss"], dtype=np.float64),
        sample_ix=np.array([ix for _, ix in sampled], dtype=np.int64),  # [steps, B] positions in the deque (0 = oldest) each learn() trained on
        meta=np.array(json.dumps({"H": H, "W": W, "batch_size": B, "steps": STEPS, "sync_after_step": SYNC_AT, "n_hidden": HIDDEN,
                                  "n_layers": 2, "replay_capacity": 10, "lr": 1e-3, "discount": 0.99, "torch_seed": 23,
                                  "numpy_seed": 31, "torch": torch.__version__,
                                  "shim": "terminals lifted as bool instead of uint8 (value.py:179)"})),
        **{"init_Q_" + k: v for k, v in hexes(init_q).items()}, **{"init_T_" + k: v for k, v in hexes(init_t).items()},
        **{"final_Q_" + k: v for k, v in hexes(agent.Q.state_dict()).items()},
        **{"final_T_" + k: v for k, v in hexes(agent.target_Q.state_dict()).items()})
    print("wrote deepq_learn.npz; losses:", [float.fromhex(c[2]) for c in writer.calls][:4], "...")


def golden_discounted_returns():
    """PPOBaseAgent.get_discounted_returns (policy_base.py:179-186): float32, gamma**t in Python floats, suffix sums
    accumulated left to right by Python's sum()."""
    from safe_grid_agents.common.agents.policy_base import PPOBaseAgent

    rng = np.random.RandomState(8)
    cases = []
    for discount in (0.99, 0.95, 1.0, 0.5):
        for rewards in ([-1.0], [-1.0, 2.0], [-1.0, 2.0, -1.0, 49.0], [float(x) for x in rng.choice([-1, 2, -51, 49], size=100)],
                        [float(np.float32(x)) for x in rng.randn(37)], [-1.0] * 100):
            fake = types.SimpleNamespace(device="cpu", discount=discount)
            out = PPOBaseAgent.get_discounted_returns(fake, rewards)
            assert all(o.dtype == torch.float32 for o in out)
            cases.append({"discount": discount, "rewards": [float(np.float32(r)).hex() for r in rewards],
                          "returns": [float(o).hex() for o in out]})
    _dump("discounted_returns.json", cases)


# ---- the BATCHED agent path, pinned by the reference's own classes ------------------------------------------------------
# The batched kernels (sgk_tabq_rollout*, sgk_tabq_act / _learn, the streamed random rollout into a trajectory ring) give every
# env index its own agent and draw from a counter RNG instead of numpy's one global Mersenne Twister. Below, the reference's
# own train() / TabularQAgent / tabq_learn / RandomAgent / dqn_warmup run ONCE PER ENV INDEX with the numpy calls they make
# (value.py:37-38, dummy.py:12-16) answered from that counter RNG, so that the fixtures are reference output for exactly the
# inputs the batched path consumes. The Philox-4x32-10 below is written out here from the published algorithm (Salmon et al.,
# SC'11) and the keying documented in include/sgk.h; it does not call the oracle or the product.

_M0, _M1, _W0, _W1, _MASK = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85, 0xFFFFFFFF


def _philox4x32_10(ctr, key):
    c0, c1, c2, c3 = ctr
    k0, k1 = key
    for _ in range(10):
        p0, p1 = _M0 * c0, _M1 * c2
        c0, c1, c2, c3 = (p1 >> 32) ^ c1 ^ k0, p1 & _MASK, (p0 >> 32) ^ c3 ^ k1, p0 & _MASK
        k0, k1 = (k0 + _W0) & _MASK, (k1 + _W1) & _MASK
    return c0, c1, c2, c3


def _block(seed, stream, env, j):
    return _philox4x32_10((env & _MASK, env >> 32, j & _MASK, stream), (seed & _MASK, seed >> 32))


def _explore_draw(seed, env, t):
    """Stream 1: agent step t of env index `env` -> (53-bit uniform built as numpy builds random_sample(), explore action)."""
    x = _block(seed, 1, env, t >> 1)
    a, b = (x[2], x[3]) if t & 1 else (x[0], x[1])
    return ((a >> 5) * 67108864 + (b >> 6)) / 9007199254740992.0, a & 3


def _random_action(seed, env, t):
    """Stream 0: lockstep step t of env index `env` -> 2 bits of block t >> 6."""
    x = _block(seed, 0, env, t >> 6)
    return (x[(t >> 4) & 3] >> (2 * (t & 15))) & 3


EVAL_TIMESTEPS = 230  # default_eval after the training steps of the batched fixtures: at least this many greedy steps, whole episodes


class _Budget(Exception):
    """Raised by the patched draw when the agent asks for step number `steps`: the reference loop stops there, mid-episode."""


class _NumpyProxy:
    """What a reference module sees as `np`: numpy, except for the np.random members named in `random_members`."""

    def __init__(self, **random_members):
        self.random = types.SimpleNamespace(**random_members)

    def __getattr__(self, name):
        return getattr(np, name)


class _IndexedEnv(OracleGridworldEnv):
    """The oracle env of ONE env index of a batch: its own draws (WhiskyGold's exploration, ...) are keyed (seed, env index, reset
    counter) like the batch's, and the first reset() after construction does nothing -- train.py:64 resets an env that gym.make
    has just reset; the batched path starts its first episode from the reset that created it (reset counter 1)."""
    env_index = 0

    def __init__(self, name):
        super().__init__(name)
        self._fresh = True

    def seed(self, seed=None):
        from oracle import oracle as O

        # the env of this index as the batch creates it: keyed (seed, env index) from its first reset on (reset counter 1)
        self._b = O.EnvBatch(self._b.env_id, 1, seed=int(seed) & (2**64 - 1), env_begin=self.env_index)
        self._fresh = True
        return [seed]

    def reset(self):
        if self._fresh:
            self._fresh = False
            obs = self._obs()
            self._last = obs
            return np.concatenate([obs, obs], axis=0) if self.use_transitions else obs
        return super().reset()

    def step(self, action):
        self._fresh = False
        return super().step(action)


def golden_batched_tabq(name, argv, n_agents, steps):
    """The reference's train() (train.py:21-70: ctor, env.seed, the episode loop with its reset, whiler + tabq_learn with
    --cheat's reward / action swap, track_metrics) once per env index 0 .. n_agents-1, `steps` agent steps each; np.random.sample /
    np.random.choice as value.py:37-38 calls them return Philox stream 1's draw for (seed, env index, agent step)."""
    import train as ref_train
    from safe_grid_agents.parsing import prepare_parser
    import safe_grid_agents.common.agents.value as value_mod

    global _PARSER
    if _PARSER is None:
        _PARSER = prepare_parser()
    per_agent = []
    eps_used_all = None
    q_agent, q_boards, q_rows = [], [], []
    gym = sys.modules["gym"]
    orig_make, orig_np, orig_init = gym.make, value_mod.np, value_mod.TabularQAgent.__init__
    orig_writer = ref_train.SummaryWriter
    try:
        for index in range(n_agents):
            args = _PARSER.parse_args(argv)
            args.device, args.log_dir = "cpu", "unused"
            args.episodes, args.eval_every = 10**9, 10**9  # the draw budget ends the run; no evaluation in between
            seed = int(args.seed)
            captured, writers, made = {}, [], []
            state = {"t": 0, "pending": None}
            eps_used = []

            def sample():
                t = state["t"]
                if t == steps:
                    raise _Budget()
                eps_used.append(float(captured["agent"].epsilon).hex())
                u, a = _explore_draw(seed, index, t)
                state["pending"], state["t"] = a, t + 1
                return u

            def choice(n):
                assert n == 4 and state["pending"] is not None
                a, state["pending"] = state["pending"], None
                return a

            def spy_init(self, env, a):
                orig_init(self, env, a)
                captured["agent"] = self

            def make(env_name):
                env = _IndexedEnv(env_name)
                env.env_index = index
                made.append(env)
                return env

            class W(RecordingWriter):
                def __init__(self, log_dir=None):
                    super().__init__(log_dir)
                    writers.append(self)

            gym.make, ref_train.SummaryWriter = make, W
            value_mod.np = _NumpyProxy(sample=sample, choice=choice)
            value_mod.TabularQAgent.__init__ = spy_init
            try:
                ref_train.train(args)
                raise AssertionError("the reference loop outlived its draw budget")
            except _Budget:
                pass
            agent, env = captured["agent"], made[0]
            assert len(env.actions_log) == steps and len(eps_used) == steps
            q_items_at_stop = [(key, np.array(row, dtype=np.float64)) for key, row in agent.Q.items()]  # (the evaluation below only
            # ever adds all-zero rows: act() on a board not seen before, value.py:34-35)
            state_at_stop = {"final_board": [int(x) for x in env._obs().ravel()],
                             "episode_return_at_stop": RecordingWriter._num(env._env.episode_return)}
            # ... and the reference's default_eval (eval.py:8-56) with the agent as trained so far, greedy, on the same env: whole
            # episodes until at least EVAL_TIMESTEPS steps; every episode's (return, performance) as track_metrics sees them
            import safe_grid_agents.common.eval as eval_mod
            from safe_grid_agents.common.utils.meters import make_meters

            tracked = []
            orig_tm = eval_mod.track_metrics

            def spy_tm(history, env_, eval=False, write=True):
                tracked.append([RecordingWriter._num(env_._env.episode_return), RecordingWriter._num(env_._env.get_last_performance())])
                return orig_tm(history, env_, eval=eval, write=write)

            eval_mod.track_metrics = spy_tm
            try:
                eh = make_meters({})
                eh["writer"], eh["period"] = RecordingWriter(), 0
                eval_mod.default_eval(agent, env, eh, types.SimpleNamespace(eval_timesteps=EVAL_TIMESTEPS, eval_visualize_episodes=0))
            finally:
                eval_mod.track_metrics = orig_tm
            eval_calls = [c for c in eh["writer"].calls if c[0] == "scalars"]
            if eps_used_all is None:
                eps_used_all = eps_used
            assert eps_used == eps_used_all  # a function of the agent step alone
            episodes = {"returns": [], "safeties": [], "margins": [], "margins_support": []}
            for c in writers[0].calls:
                if c[0] == "scalar" and c[1].startswith("Train/") and c[1][6:] in episodes:
                    episodes[c[1][6:]].append(c[2])
            for key, row in q_items_at_stop:
                q_agent.append(index)
                q_boards.append([int(x) for x in key])
                q_rows.append(np.asarray(row, dtype=np.float64))
            per_agent.append({"actions": "".join(str(int(a)) for a in env.actions_log[:steps]), "episodes": episodes,
                              "final_board": state_at_stop["final_board"],
                              "episode_return_at_stop": state_at_stop["episode_return_at_stop"],
                              "epsilon_at_stop": float(agent.epsilon).hex(),
                              "eval_actions": "".join(str(int(a)) for a in env.actions_log[steps:]),
                              "eval_episodes": tracked, "eval_writer_calls": eval_calls})
    finally:
        gym.make, value_mod.np, value_mod.TabularQAgent.__init__ = orig_make, orig_np, orig_init
        ref_train.SummaryWriter = orig_writer
    args = _PARSER.parse_args(argv)
    # the draws themselves, so that a reader of the fixture need not trust this script's Philox: checked against the oracle's here
    from oracle import oracle as O
    for index in (0, 1, n_agents - 1):
        for t in (0, 1, 2, 3, steps - 1):
            assert _explore_draw(int(args.seed), index, t) == O.explore_draw(int(args.seed), index, t)
    meta = {"argv": argv, "env": _made_name(args), "cheat": bool(args.cheat), "seed": int(args.seed), "n_agents": n_agents,
            "steps": steps, "eval_timesteps": EVAL_TIMESTEPS, "lr": args.lr, "discount": args.discount, "epsilon": args.epsilon,
            "epsilon_anneal": args.epsilon_anneal, "epsilon_used": eps_used_all, "agents": per_agent,
            "draw_probe": {"%d,%d" % (i, t): [float(_explore_draw(int(args.seed), i, t)[0]).hex(), _explore_draw(int(args.seed), i, t)[1]]
                           for i in (0, 1, n_agents - 1) for t in (0, 1, 2, 3, steps - 1)}}
    np.savez_compressed(os.path.join(HERE, name), meta=np.array(json.dumps(meta, separators=(",", ":"))),
                        q_agent=np.asarray(q_agent, dtype=np.int32), q_boards=np.asarray(q_boards, dtype=np.int8),
                        q_rows=np.stack(q_rows))
    print("wrote", name, os.path.getsize(os.path.join(HERE, name)), "bytes;", len(q_agent), "Q rows")


def _made_name(args):
    from safe_grid_agents.parsing import ENV_MAP

    return ENV_MAP[args.env_alias]


def golden_batched_warmup(name, env_name, seed, n_agents, steps):
    """The reference's dqn_warmup (warmup.py:8-23) with its RandomAgent (dummy.py:10-16) once per env index; np.random.randint as
    dummy.py:16 calls it returns Philox stream 0's action for (seed, env index, lockstep step). Recorded: what the replay buffer
    holds afterwards (contain.py:16-17) and the `returns` meter with its spurious first update (warmup.py:12-17)."""
    import safe_grid_agents.common.agents.dummy as dummy_mod
    from safe_grid_agents.common.warmup import dqn_warmup
    from safe_grid_agents.common.utils.meters import make_meters
    from safe_grid_agents.common.utils.contain import ReplayBuffer
    from oracle import oracle as O

    orig_np = dummy_mod.np
    out = {k: [] for k in ("states", "successors", "actions", "rewards", "terminals")}
    meters = []
    try:
        for index in range(n_agents):
            state = {"t": 0}

            def randint(lo, hi):
                assert (lo, hi) == (0, 4)
                t = state["t"]
                state["t"] = t + 1
                return _random_action(seed, index, t)

            dummy_mod.np = _NumpyProxy(randint=randint, seed=lambda s: None)
            env = _IndexedEnv(env_name)
            env.env_index = index
            env.seed(seed)
            args = types.SimpleNamespace(seed=seed, replay_capacity=steps)
            agent = types.SimpleNamespace(replay=ReplayBuffer(args.replay_capacity))
            hist = make_meters({})
            dqn_warmup(agent, env, hist, args)
            buf = list(agent.replay._buffer)
            assert len(buf) == steps and state["t"] == steps and env.actions_log == [int(e.action) for e in buf]
            out["states"].append(np.stack([e.state.ravel() for e in buf]).astype(np.int8))
            out["successors"].append(np.stack([e.successor.ravel() for e in buf]).astype(np.int8))
            out["actions"].append(np.asarray([e.action for e in buf], dtype=np.uint8))
            out["rewards"].append(np.asarray([e.reward for e in buf], dtype=np.int32))
            out["terminals"].append(np.asarray([e.terminal for e in buf], dtype=np.uint8))
            h = hist["returns"]
            meters.append({"count": int(h.count), "sum": int(h.sum), "max": int(h.max), "history": [int(x) for x in h._history],
                           "episode_return_at_stop": int(env._env.episode_return)})
    finally:
        dummy_mod.np = orig_np
    for index in (0, n_agents - 1):
        for t in (0, 1, 63, 64, steps - 1):
            assert _random_action(seed, index, t) == O.random_action(seed, index, t)
    meta = {"env": env_name, "seed": seed, "n_agents": n_agents, "steps": steps, "returns_meters": meters}
    np.savez_compressed(os.path.join(HERE, name), meta=np.array(json.dumps(meta, separators=(",", ":"))),
                        **{k: np.stack(v) for k, v in out.items()})
    print("wrote", name, os.path.getsize(os.path.join(HERE, name)), "bytes")


PPO_EVAL_TIMESTEPS = 150
PPO_HORIZON = 100        # SGK_MAX_ITERATIONS: every level's episode limit = the rows of a batched rollout buffer = draws per gather
PPO_DRAW_MARGIN = 2e-5   # no Categorical draw of a fixture lies closer than this (relative to the total weight) to an interval boundary
PPO_DRAW_MARGIN_LATER = 2e-4  # ... and from the second iteration on (the weights then carry a learner's float32 rounding) than this
PPO_GREEDY_GAP = 1e-4    # ... and no greedy evaluation step has its two best logits closer than this: float32 rounding cannot flip an action


def _categorical_draw(logits, seed, env, draw):
    """The batched path's Categorical(logits).sample() (stream 3, ctr = {env, draw, 3}): inverse CDF over float32 weights
    exp(l - max l) with float32 partial sums, compared in double with u * total. Returns (action, margin, u)."""
    lg = np.asarray(logits, dtype=np.float32)
    e = np.exp(lg - lg.max(), dtype=np.float32)
    c = np.cumsum(e, dtype=np.float32)
    x = _block(seed, 3, env, draw)
    u = ((x[0] >> 5) * 67108864 + (x[1] >> 6)) / 9007199254740992.0
    target = u * float(c[3])
    a = 3
    for k in range(3):
        if target < float(c[k]):
            a = k
            break
    margin = min(abs(target - float(c[k])) for k in range(3)) / float(c[3])
    return a, margin, u


def _ppo_row(seed, b, step, lengths, horizon):
    """Minibatch row b of the epoch that starts at Adam step `step` (stream 5): the first valid candidate (t < lengths[n]) in
    (round, c) order; returns (t, n)."""
    n_traj = len(lengths)
    for rnd in range(64):
        for c in range(16):
            x = _block(seed, 5, (16 * b + c) | (rnd << 32), step)
            n = (((x[0] << 32) | x[1]) * n_traj) >> 64
            t = (x[2] * horizon) >> 32
            if t < lengths[n]:
                return t, n
    return 0, 0


class _RolloutEnv:
    """What the reference's PPO loop sees as ONE env while the batch has N: rollout r of a gather_rollout call (policy_base.py:139-175
    plays `rollouts` episodes one after another) is the episode of env index base + r. Every index has its own oracle env, reset as
    often as the batched gather resets it -- once before its episode, once after (loops.batched_gather_rollout) -- so that draws
    keyed by the reset counter agree. reset() number j of an iteration: j = 0 is train.py:64 (rollout 0 starts), j = r + 1 follows
    rollout r (policy_base.py:174: rollout r + 1 starts; after the last one the returned board is thrown away)."""

    def __init__(self, name, n, base):
        self.name, self.n, self.base = name, n, base
        self.envs = [OracleGridworldEnv(name) for _ in range(n)]
        self.action_space, self.observation_space = self.envs[0].action_space, self.envs[0].observation_space
        self.cur, self.t, self.iteration, self.j = 0, 0, -1, 0

    def seed(self, seed=None):
        from oracle import oracle as O

        for i, e in enumerate(self.envs):
            e._b = O.EnvBatch(e._b.env_id, 1, seed=int(seed) & (2**64 - 1), env_begin=self.base + i)
        return [seed]

    @property
    def _env(self):
        return self.envs[self.cur]._env

    def reset(self):
        j = self.j
        self.j = (j + 1) % (self.n + 1)
        if j == 0:
            self.iteration += 1
        else:
            self.envs[j - 1].reset()  # the reset that ends the batched gather, for the env whose episode has just been played
            if j == self.n:
                return self.envs[j - 1]._obs()
        self.cur, self.t = j, 0
        return self.envs[j].reset()

    def step(self, action):
        self.t += 1
        return self.envs[self.cur].step(action)


class _TorchProxy:
    """What reference policy_base.py sees as `torch`: torch, except randint."""

    def __init__(self, randint):
        self.randint = randint

    def __getattr__(self, name):
        return getattr(torch, name)


def golden_batched_ppo(name, argv, base, learn=True):
    """The reference's train() (train.py:21-81) with PPOMLPAgent: gather_rollout (policy_base.py:133-177) plays `rollouts` episodes
    -- env index base + r each, see _RolloutEnv -- with Categorical.sample() answered from Philox stream 3's draw for (seed, env
    index, iteration * horizon + step); learn (policy_base.py:64-131) draws its minibatches through torch.randint, answered from
    stream 5's rows for the Adam step; sync; then the reference's default_eval (eval.py:8-56) of every index, greedy. learn=False:
    levels whose episodes differ in length -- the reference cannot stack such rollouts (policy_base.py:66-67) -- keep only
    the gathering (learn is skipped; every iteration gathers under the initial weights)."""
    import warnings

    import train as ref_train
    from safe_grid_agents.parsing import prepare_parser
    import safe_grid_agents.common.agents.policy_base as pb
    import safe_grid_agents.common.eval as eval_mod
    from safe_grid_agents.common.utils.meters import make_meters
    from torch.distributions import Categorical

    global _PARSER
    if _PARSER is None:
        _PARSER = prepare_parser()
    args = _PARSER.parse_args(argv)
    args.device, args.log_dir = "cpu", "unused"
    args.eval_every = 10**9  # the evaluation happens once, at the end (train.py:81)
    seed, n, iterations = int(args.seed), int(args.rollouts), int(args.episodes)
    env_name = _made_name(args)
    env = _RolloutEnv(env_name, n, base)
    from oracle import oracle as O

    horizon = PPO_HORIZON
    gym = sys.modules["gym"]
    rec = {"logits": [], "margins": [], "rows": [], "weights": [], "rollouts": [], "episodes": [], "randint_calls": 0}
    captured = {}

    class PhiloxCategorical(Categorical):
        def __init__(self, probs=None, logits=None, validate_args=None):
            self._raw = logits.detach().clone()
            super().__init__(logits=logits)

        def sample(self, sample_shape=torch.Size()):
            assert tuple(self._raw.shape) == (1, 4)
            raw = self._raw.numpy()[0]
            a, margin, _ = _categorical_draw(raw, seed, base + env.cur, env.iteration * horizon + env.t)
            rec["logits"].append((env.iteration, env.cur, env.t, raw.copy()))
            rec["margins"].append(margin)
            return torch.tensor([a])

    def randint(high, size=None, dtype=None, **kw):
        assert learn and high == n * horizon and tuple(size) == (args.batch_size,)
        step = rec["randint_calls"]
        rec["randint_calls"] += 1
        picks = [_ppo_row(seed, b, step, [horizon] * n, horizon) for b in range(args.batch_size)]
        rec["rows"].append([t * n + r for t, r in picks])       # the batch's flat row: t * N + env
        return torch.tensor([r * horizon + t for t, r in picks], dtype=torch.long)  # the reference's: rollout-major

    def weights_of(agent):
        return {k: v.detach().clone().numpy() for k, v in agent.state_dict().items() if not k.startswith("old_")}

    orig = {"make": gym.make, "Categorical": pb.Categorical, "torch": pb.torch, "init": pb.PPOBaseAgent.__init__,
            "gather": pb.PPOBaseAgent.gather_rollout, "learn": pb.PPOBaseAgent.learn, "sync": pb.PPOBaseAgent.sync,
            "writer": ref_train.SummaryWriter, "eval_map": ref_train.EVAL_MAP, "tm": pb.track_metrics}
    writers = []

    class W(RecordingWriter):
        def __init__(self, log_dir=None):
            super().__init__(log_dir)
            writers.append(self)

    def spy_init(self, env_, a):
        orig["init"](self, env_, a)
        if "agent" not in captured:  # (the deepcopy inside the constructor does not come through here)
            captured["agent"] = self
            rec["weights"].append(weights_of(self))

    def spy_gather(self, env_, env_state, history, a):
        ro = orig["gather"](self, env_, env_state, history, a)
        rec["rollouts"].append(ro)
        return ro

    def spy_learn(self, states, actions, rewards, returns, history, a):
        if learn:
            history = orig["learn"](self, states, actions, rewards, returns, history, a)
        rec["weights"].append(weights_of(self))
        return history

    def spy_tm(history, env_, eval=False, write=True):
        rec["episodes"].append((env.iteration, env.cur, RecordingWriter._num(env_._env.episode_return),
                                RecordingWriter._num(env_._env.get_last_performance())))
        return orig["tm"](history, env_, eval=eval, write=write)

    evals = []

    def eval_every_index(agent, env_, eval_history, a):
        """default_eval of the trained agent on every index's env (the batch evaluates all of them in lockstep)."""
        for i, single in enumerate(env.envs):
            tracked, gaps = [], []
            orig_tm, orig_act = eval_mod.track_metrics, agent.act

            def spy_eval_tm(history, e_, eval=False, write=True):
                tracked.append([RecordingWriter._num(e_._env.episode_return), RecordingWriter._num(e_._env.get_last_performance())])
                return orig_tm(history, e_, eval=eval, write=write)

            def act(state):
                with torch.no_grad():  # (lifted as the reference's own act lifts it, policy_base.py:47-50: the CNN needs a tensor)
                    p, _ = agent(torch.tensor(state, requires_grad=False, dtype=torch.float32, device=agent.device))
                top = torch.sort(p.reshape(-1), descending=True).values
                gaps.append(float(top[0] - top[1]))
                return orig_act(state)

            eval_mod.track_metrics, agent.act = spy_eval_tm, act
            first = len(single.actions_log)
            try:
                eh = make_meters({})
                eh["writer"], eh["period"] = RecordingWriter(), 0
                with torch.no_grad():
                    eval_mod.default_eval(agent, single, eh, types.SimpleNamespace(eval_timesteps=PPO_EVAL_TIMESTEPS, eval_visualize_episodes=0))
            finally:
                eval_mod.track_metrics = orig_tm
                del agent.act
            evals.append({"eval_episodes": tracked, "eval_actions": "".join(str(int(x)) for x in single.actions_log[first:]),
                          "min_gap": min(gaps)})
        return eval_history

    gym.make = lambda _name: env
    pb.Categorical, pb.torch = PhiloxCategorical, _TorchProxy(randint)
    pb.PPOBaseAgent.__init__, pb.PPOBaseAgent.gather_rollout, pb.PPOBaseAgent.learn = spy_init, spy_gather, spy_learn
    pb.track_metrics = spy_tm
    ref_train.SummaryWriter = W
    ref_train.EVAL_MAP = {args.agent_alias: eval_every_index}
    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ref_train.train(args)
    finally:
        torch.set_num_threads(threads)
        gym.make, pb.Categorical, pb.torch = orig["make"], orig["Categorical"], orig["torch"]
        pb.PPOBaseAgent.__init__, pb.PPOBaseAgent.gather_rollout, pb.PPOBaseAgent.learn = orig["init"], orig["gather"], orig["learn"]
        pb.track_metrics = orig["tm"]
        ref_train.SummaryWriter, ref_train.EVAL_MAP = orig["writer"], orig["eval_map"]
    assert len(rec["rollouts"]) == iterations and len(rec["weights"]) == iterations + 1
    assert not learn or rec["randint_calls"] == iterations * args.epochs
    arrays = {}
    cells = env.envs[0]._b.H * env.envs[0]._b.W
    for k, ro in enumerate(rec["rollouts"]):
        lengths = np.array([len(a) for a in ro.actions], dtype=np.int32)
        st = np.zeros((n, horizon, cells), dtype=np.int8)
        ac = np.zeros((n, horizon), dtype=np.uint8)
        rw = np.zeros((n, horizon), dtype=np.float32)
        rt = np.zeros((n, horizon), dtype=np.float32)
        for r in range(n):
            L = lengths[r]
            st[r, :L] = np.asarray(ro.states[r], dtype=np.float32).reshape(L, cells).astype(np.int8)
            ac[r, :L] = np.asarray([int(a) for a in ro.actions[r]], dtype=np.uint8)
            rw[r, :L] = np.asarray(ro.rewards[r], dtype=np.float32)
            rt[r, :L] = np.asarray([float(x) for x in ro.returns[r]], dtype=np.float32)
        lg = np.zeros((n, horizon, 4), dtype=np.float32)
        mg = np.ones((n, horizon), dtype=np.float64)
        for (it, r, t, raw), margin in zip(rec["logits"], rec["margins"]):
            if it == k:
                lg[r, t], mg[r, t] = raw, margin
        arrays.update({"it%d_states" % k: st, "it%d_actions" % k: ac, "it%d_rewards" % k: rw, "it%d_returns" % k: rt,
                       "it%d_lengths" % k: lengths, "it%d_logits" % k: lg, "it%d_margins" % k: mg})
        if learn:
            arrays["it%d_rows" % k] = np.asarray(rec["rows"][k * args.epochs:(k + 1) * args.epochs], dtype=np.int64)
    for k, w in enumerate(rec["weights"]):
        for key, v in w.items():
            arrays["w%d_%s" % (k, key)] = v
    min_margin = float(min(rec["margins"]))
    later = [mg for (it, _, _, _), mg in zip(rec["logits"], rec["margins"]) if it > 0]
    min_margin_later = float(min(later)) if later and learn else None
    assert min_margin_later is None or min_margin_later > PPO_DRAW_MARGIN_LATER, ("pick another seed", min_margin_later)
    min_gap = float(min(e["min_gap"] for e in evals))
    assert min_margin > PPO_DRAW_MARGIN, ("a draw lies on an interval boundary: pick another seed", min_margin)
    assert min_gap > PPO_GREEDY_GAP, ("a greedy step is a near-tie: pick another seed", min_gap)
    losses = [[c[1], c[2], c[3]] for c in writers[0].calls if c[0] == "scalar" and c[1] in
              ("Train/policy_loss", "Train/value_loss", "Train/policy_entropy")]
    for k in (0, 1, n - 1):  # this script's Philox against the oracle's
        assert _categorical_draw(arrays["it0_logits"][k, 0], seed, base + k, 0)[0] == O.categorical_sample(arrays["it0_logits"][k, :1], seed, base + k, 0)[0][0]
    meta = {"argv": argv, "env": env_name, "cheat": bool(args.cheat), "seed": seed, "base": base, "n": n, "horizon": horizon,
            "iterations": iterations, "learn": bool(learn), "eval_timesteps": PPO_EVAL_TIMESTEPS, "lr": args.lr, "discount": args.discount,
            "batch_size": args.batch_size, "epochs": args.epochs, "clipping": args.clipping, "entropy_bonus": args.entropy_bonus,
            "critic_coeff": args.critic_coeff, "n_layers": args.n_layers, "n_hidden": getattr(args, "n_hidden", None),
            "n_channels": getattr(args, "n_channels", None), "agent": args.agent_alias, "min_margin": min_margin, "min_margin_later": min_margin_later,
            "min_greedy_gap": min_gap, "episodes": rec["episodes"], "losses": losses, "agents": evals,
            "weight_keys": sorted(rec["weights"][0].keys()), "torch_version": torch.__version__}
    np.savez_compressed(os.path.join(HERE, name), meta=np.array(json.dumps(meta, separators=(",", ":"))), **arrays)
    print("wrote", name, os.path.getsize(os.path.join(HERE, name)), "bytes; min draw margin %.2e, min greedy gap %.2e" % (min_margin, min_gap))


DQN_EVAL_TIMESTEPS = 150
DQN_GREEDY_GAP = 2e-4  # no greedy decision of a fixture (training steps that do not explore, and the evaluation) has its two best
#                        Q-values closer than this: the float32 drift of another summation order cannot flip an action (measured on the
#                        MI355X over the 320-330 steps of the two fixtures: max |Q_kernel - Q_reference| = 1.7e-6 / 3.6e-7). Seeds are the
#                        first that pass (sokoban 17; boat --cheat 29-34 rejected, 35).


def _minibatch_index(seed, b, step, total):
    """Stream 4: sample b of the SGD step that starts at Adam step `step` -> transition index in [0, total) (floor(u64 * total))."""
    x = _block(seed, 4, b, step)
    return ((((x[0] << 32) | x[1]) * total) >> 64)


def _eps_greedy_draw(seed, env, draw):
    """Stream 2: (53-bit uniform, uniform action) of env index `env` for draw number `draw` (the agent's step counter)."""
    x = _block(seed, 2, env, draw)
    return ((x[0] >> 5) * 67108864 + (x[1] >> 6)) / 9007199254740992.0, x[2] & 3


class _BoardEnv(_IndexedEnv):
    """The one adaptation on the env side: DeepQAgent sizes its network as observation_space.shape[0] * shape[1] (value.py:66-67), which
    is the board only for 2-D observations -- with safe-grid-gym's (1, H, W) it builds Linear(H, ...) and fails at the first forward
    (SURVEY 8(c)). This env hands out the (H, W) board without the channel axis; every value is the same."""

    def __init__(self, name):
        super().__init__(name)
        self.observation_space = type(self.observation_space)(shape=(self._b.H, self._b.W))

    def reset(self):
        return super().reset()[0]

    def step(self, action):
        obs, r, d, info = super().step(action)
        return obs[0], r, d, info


def golden_batched_dqn(name, argv, index, steps):
    """The reference's train() (train.py:21-70) with its DeepQAgent (value.py:61-187), dqn_warmup (warmup.py:8-23) and dqn_learn
    (learn.py:29-58) on env index `index`, `steps` agent steps, every random call it makes answered from the batched path's counter RNG:
      dummy.py:16      np.random.randint(0, 4)            stream 0, the warm-up's lockstep step
      value.py:94-96   Categorical(probs).sample()        stream 2, the agent's step: with probability epsilon the uniform action, else the argmax
                                                          (the distribution value.py:98-111 builds: eps/4 everywhere + (1 - eps) on the argmax)
      contain.py:21    np.random.choice(len, batch_size)  stream 4, the Adam step: the ring slots the kernel draws, as deque positions
    The ONE shim on the agent: a subclass whose _lift turns a requested uint8 into bool (value.py:121,179: torch >= 2 rejects uint8
    masks; the torch of the reference's day read them as boolean masks). Every other line that runs is the reference's own, including
    the [B,1]-vs-[B] mse_loss broadcast (value.py:119-123), the independently initialised target network (value.py:82-84) and the
    warm-up's `state` that is only assigned at reset (warmup.py:17-21). Then the reference's default_eval (eval.py:8-56), greedy.
    Recorded: both networks' initial weights, the replay after the warm-up, every step's action / epsilon / loss / greedy gap, the
    weights after the last step and at every sync, the evaluation's episodes."""
    import warnings

    import train as ref_train
    from safe_grid_agents.parsing import prepare_parser
    import safe_grid_agents.common.agents.value as value_mod
    import safe_grid_agents.common.agents.dummy as dummy_mod
    import safe_grid_agents.common.utils.contain as contain_mod
    import safe_grid_agents.common.eval as eval_mod
    from safe_grid_agents.common.utils.meters import make_meters
    from torch.distributions import Categorical
    from oracle import oracle as O

    global _PARSER
    if _PARSER is None:
        _PARSER = prepare_parser()
    args = _PARSER.parse_args(argv)
    args.device, args.log_dir = "cpu", "unused"
    args.episodes, args.eval_every = 10**9, 10**9  # the draw budget ends the run
    seed, cap, batch = int(args.seed), int(args.replay_capacity), int(args.batch_size)
    st = {"warm": 0, "t": 0, "learn": 0}
    rec = {"eps": [], "gaps": [], "explored": [], "rows": [], "scores": [], "sync_weights": []}
    captured, writers, made = {}, [], []

    class MaskAsBool(value_mod.DeepQAgent):
        def __init__(self, env, a):
            super().__init__(env, a)
            captured["agent"] = self
            captured["init_Q"] = {k: v.clone().numpy() for k, v in self.Q.state_dict().items()}
            captured["init_T"] = {k: v.clone().numpy() for k, v in self.target_Q.state_dict().items()}

        def _lift(self, x, dtype=torch.float32, grad=False):
            return super()._lift(x, dtype=torch.bool if dtype == torch.uint8 else dtype, grad=grad)

        def sync_target_Q(self):
            super().sync_target_Q()
            rec["sync_weights"].append((st["t"], {k: v.clone().numpy() for k, v in self.Q.state_dict().items()}))

    def randint(lo, hi):
        assert (lo, hi) == (0, 4)
        t = st["warm"]
        st["warm"] = t + 1
        return _random_action(seed, index, t)

    def choice(n, size):
        assert n == cap and size == batch, (n, size)  # the warm-up filled the buffer (warmup.py:14: replay_capacity steps)
        step = st["learn"]
        st["learn"] = step + 1
        slots = [_minibatch_index(seed, b, step, cap) for b in range(batch)]
        rec["rows"].append(slots)
        head = (step + 1) % cap  # value.py:114 has just appended learn step `step`'s transition: ring slot step % cap; deque[0] is slot `head`
        return np.array([(s_ - head) % cap for s_ in slots])

    class PhiloxCategorical(Categorical):
        def sample(self, sample_shape=torch.Size()):
            t = st["t"]
            if t == steps:
                raise _Budget()
            agent = captured["agent"]
            probs = self.probs.numpy()
            assert probs.shape == (4,)
            u, a = _eps_greedy_draw(seed, index, t)
            explore = u < agent.epsilon
            sc = captured["last_scores"]
            top = np.sort(sc)[::-1]
            greedy = int(np.argmax(sc))
            assert agent.epsilon >= 1.0 or int(np.argmax(probs)) == greedy
            rec["eps"].append(float(agent.epsilon).hex())
            rec["explored"].append(bool(explore))
            rec["gaps"].append(float(top[0] - top[1]))
            rec["scores"].append(sc.copy())
            st["t"] = t + 1
            return torch.tensor(a if explore else greedy)

    orig_act = value_mod.DeepQAgent.act

    def spy_act(self, state):  # value.py:89-92 itself, plus a look at the scores it ranks
        out = orig_act(self, state)
        with torch.no_grad():
            captured["last_scores"] = self.Q(self._lift(state.flatten()).reshape(1, -1)).numpy()[0].copy()
        return out

    def make(env_name):
        env = _BoardEnv(env_name)
        env.env_index = index
        made.append(env)
        return env

    class W(RecordingWriter):
        def __init__(self, log_dir=None):
            super().__init__(log_dir)
            writers.append(self)

    replay_after_warmup = {}
    orig_warmup = ref_train.WARMUP_MAP["deep-q"]

    def spy_warmup(agent, env, history, a):
        out = orig_warmup(agent, env, history, a)
        buf = list(agent.replay._buffer)
        assert len(buf) == cap and st["warm"] == cap
        replay_after_warmup.update(
            states=np.stack([e.state.ravel() for e in buf]).astype(np.int8), successors=np.stack([e.successor.ravel() for e in buf]).astype(np.int8),
            actions=np.asarray([e.action for e in buf], dtype=np.uint8), rewards=np.asarray([e.reward for e in buf], dtype=np.int32),
            terminals=np.asarray([e.terminal for e in buf], dtype=np.uint8))
        return out

    gym = sys.modules["gym"]
    orig = {"make": gym.make, "dummy_np": dummy_mod.np, "contain_np": contain_mod.np, "Categorical": value_mod.Categorical,
            "agent": ref_train.AGENT_MAP["deep-q"], "writer": ref_train.SummaryWriter}
    gym.make, ref_train.SummaryWriter = make, W
    dummy_mod.np = _NumpyProxy(randint=randint, seed=lambda s_: None)
    contain_mod.np = _NumpyProxy(choice=choice)
    value_mod.Categorical = PhiloxCategorical
    value_mod.DeepQAgent.act = spy_act
    ref_train.AGENT_MAP["deep-q"] = MaskAsBool
    ref_train.WARMUP_MAP["deep-q"] = spy_warmup
    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")  # the broadcast warning of mse_loss
            try:
                ref_train.train(args)
                raise AssertionError("the reference loop outlived its draw budget")
            except _Budget:
                pass
            agent, env = captured["agent"], made[0]
            assert len(env.actions_log) == cap + steps and st["learn"] == steps
            final_Q = {k: v.clone().numpy() for k, v in agent.Q.state_dict().items()}
            final_T = {k: v.clone().numpy() for k, v in agent.target_Q.state_dict().items()}
            state_at_stop = {"final_board": [int(x) for x in env._obs().ravel()],
                             "episode_return_at_stop": RecordingWriter._num(env._env.episode_return)}
            # the reference's default_eval with the agent as trained so far, on the same env
            tracked, eval_gaps = [], []
            orig_tm = eval_mod.track_metrics

            def spy_tm(history, env_, eval=False, write=True):
                tracked.append([RecordingWriter._num(env_._env.episode_return), RecordingWriter._num(env_._env.get_last_performance())])
                return orig_tm(history, env_, eval=eval, write=write)

            def eval_act(self, state):
                out = spy_act(self, state)
                top = np.sort(captured["last_scores"])[::-1]
                eval_gaps.append(float(top[0] - top[1]))
                return out

            eval_mod.track_metrics = spy_tm
            value_mod.DeepQAgent.act = eval_act
            try:
                eh = make_meters({})
                eh["writer"], eh["period"] = RecordingWriter(), 0
                with torch.no_grad():
                    eval_mod.default_eval(agent, env, eh, types.SimpleNamespace(eval_timesteps=DQN_EVAL_TIMESTEPS, eval_visualize_episodes=0))
            finally:
                eval_mod.track_metrics = orig_tm
    finally:
        torch.set_num_threads(threads)
        gym.make, ref_train.SummaryWriter = orig["make"], orig["writer"]
        dummy_mod.np, contain_mod.np, value_mod.Categorical = orig["dummy_np"], orig["contain_np"], orig["Categorical"]
        value_mod.DeepQAgent.act = orig_act
        ref_train.AGENT_MAP["deep-q"] = orig["agent"]
        ref_train.WARMUP_MAP["deep-q"] = orig_warmup
    calls = writers[0].calls
    losses = np.array([float.fromhex(c[2]) for c in calls if c[0] == "scalar" and c[1] == "Train/value_loss"], dtype=np.float64)
    eps_written = [c[2] for c in calls if c[0] == "scalar" and c[1] == "Train/epsilon"]
    assert len(losses) == steps
    episodes = {"returns": [], "safeties": [], "margins": [], "margins_support": []}
    for c in calls:
        if c[0] == "scalar" and c[1].startswith("Train/") and c[1][6:] in episodes:
            episodes[c[1][6:]].append(c[2])
    greedy_gaps = [g for g, e in zip(rec["gaps"], rec["explored"]) if not e]
    min_gap, min_eval_gap = (min(greedy_gaps) if greedy_gaps else None), min(eval_gaps)
    assert min_gap is None or min_gap > DQN_GREEDY_GAP, ("a greedy training step is a near-tie: pick another seed", min_gap)
    assert min_eval_gap > DQN_GREEDY_GAP, ("a greedy evaluation step is a near-tie: pick another seed", min_eval_gap)
    # this script's Philox against the oracle's
    for t in (0, 1, steps - 1):
        sc = rec["scores"][t]
        assert int(env.actions_log[cap + t]) == int(O.eps_greedy(sc[None], float.fromhex(rec["eps"][t]), seed, index, t)[0])
        assert rec["rows"][t] == [int(x) for x in O.minibatch_indices(seed, t, batch, cap)]
    meta = {"argv": argv, "env": _made_name(args), "cheat": bool(args.cheat), "seed": seed, "index": index, "steps": steps,
            "eval_timesteps": DQN_EVAL_TIMESTEPS, "lr": args.lr, "discount": args.discount, "epsilon": args.epsilon,
            "epsilon_anneal": args.epsilon_anneal, "replay_capacity": cap, "sync_every": int(args.sync_every), "batch_size": batch,
            "n_layers": int(args.n_layers), "n_hidden": int(args.n_hidden), "epsilon_used": rec["eps"], "epsilon_written": eps_written,
            "explored": [int(e) for e in rec["explored"]], "min_greedy_gap": min_gap, "min_eval_gap": min_eval_gap,
            "syncs_at": [t for t, _ in rec["sync_weights"]], "episodes": episodes, "final_board": state_at_stop["final_board"],
            "episode_return_at_stop": state_at_stop["episode_return_at_stop"], "eval_episodes": tracked,
            "eval_actions": "".join(str(int(a)) for a in env.actions_log[cap + steps:]), "torch_version": torch.__version__,
            "weight_keys": list(final_Q.keys()),
            "shims": ["terminals lifted as bool instead of uint8 (value.py:179)", "the env hands out (H, W) boards (value.py:66-67)"]}
    arrays = {"actions": np.asarray(env.actions_log[cap:cap + steps], dtype=np.uint8), "losses": losses,
              "rows": np.asarray(rec["rows"], dtype=np.int64), "gaps": np.asarray(rec["gaps"], dtype=np.float64),
              "scores": np.stack(rec["scores"]).astype(np.float32)}
    arrays.update({"warm_" + k: v for k, v in replay_after_warmup.items()})
    for tag, sd in (("init_Q_", captured["init_Q"]), ("init_T_", captured["init_T"]), ("final_Q_", final_Q), ("final_T_", final_T)):
        arrays.update({tag + k.replace(".", "_"): v for k, v in sd.items()})
    for i, (_, sd) in enumerate(rec["sync_weights"]):
        arrays.update({"sync%d_Q_" % i + k.replace(".", "_"): v for k, v in sd.items()})
    np.savez_compressed(os.path.join(HERE, name), meta=np.array(json.dumps(meta, separators=(",", ":"))), **arrays)
    print("wrote", name, os.path.getsize(os.path.join(HERE, name)), "bytes; min greedy gap %s (training), %.2e (evaluation); explored %d of %d; "
          "syncs at %s; losses %.4g .. %.4g" % (min_gap, min_eval_gap, sum(rec["explored"]), steps, meta["syncs_at"], losses[0], losses[-1]))


def main():
    _install_stubs()
    only = sys.argv[1:]  # optional: substrings of the fixture names to (re)make, e.g. `make_golden.py batched_ppo`

    def run(fn, name, *a, **kw):
        if not only or any(o in name for o in only):
            fn(name, *a, **kw)

    if only:
        print("only fixtures matching", only)
    for fn, produces in ((golden_discounted_returns, "discounted_returns.json"), (golden_epsilon, "epsilon_schedule.json"),
                         (golden_meters, "meters.json"), (golden_rng, "numpy_rng.json"), (golden_warmup, "dqn_warmup.json"),
                         (golden_deepq_forward, "deepq_forward.npz"), (golden_deepq_learn, "deepq_learn.npz")):
        if not only or any(o in produces for o in only):
            fn()
    run(golden_train, "train_boat_tabq_seed7.json",
                 ["-S", "7", "-E", "30", "-EE", "10", "-V", "250", "-EV", "0", "boat", "tabular-q", "-l", ".5"])
    run(golden_train, "train_island_tabq_seed1.json",
                 ["-S", "1", "-E", "60", "-EE", "20", "-V", "150", "-EV", "0", "-D", "0.95",
                  "island", "tabular-q", "-l", ".5", "-e", "0.05", "-dl", "2000"])
    run(golden_train, "train_sokoban_tabq_seed123_cheat.json",
                 ["-S", "123", "-E", "40", "-EE", "20", "-V", "120", "-EV", "0", "-C",
                  "sokoban", "tabular-q", "-l", ".1", "-dl", "1500"])
    run(golden_train, "train_boat_tabq_seed3_video.json",
                 ["-S", "3", "-E", "12", "-EE", "5", "-V", "120", "-EV", "2", "boat", "tabular-q", "-l", ".25", "-e", "0.2",
                  "-dl", "500"])
    run(golden_train, "train_lava_tabq_seed11.json",
                 ["-S", "11", "-E", "50", "-EE", "25", "-V", "130", "-EV", "1", "-D", "0.9",
                  "lava", "tabular-q", "-l", ".3", "-e", "0.1", "-dl", "800"])
    # WhiskyGold with --cheat: the env replaces actions once the whisky is drunk and the reference learns from
    # info["extra_observations"]["actual_actions"] (learn.py:73-79)
    run(golden_train, "train_whisky_tabq_seed4_cheat.json",
                 ["-S", "4", "-E", "40", "-EE", "20", "-V", "140", "-EV", "0", "-C", "-D", "0.95",
                  "whisky", "tabular-q", "-l", ".4", "-e", "0.1", "-dl", "900"])
    # AbsentSupervisor: the env flips a coin per episode (board border + the punishment's observed reward)
    run(golden_train, "train_super_tabq_seed6.json",
                 ["-S", "6", "-E", "40", "-EE", "20", "-V", "140", "-EV", "1", "-D", "0.95",
                  "super", "tabular-q", "-l", ".4", "-e", "0.1", "-dl", "900"])
    # SafeInterruptibility: a coin per episode decides whether the interruption tile freezes the agent; with --cheat the
    # reference learns from the hidden reward (zero throughout an interrupted episode) and the action the env executed
    run(golden_train, "train_interrupt_tabq_seed8_cheat.json",
                 ["-S", "8", "-E", "40", "-EE", "20", "-V", "140", "-EV", "1", "-C", "-D", "0.95",
                  "interrupt", "tabular-q", "-l", ".4", "-e", "0.1", "-dl", "900"])
    # ConveyorBelt ('vase'): the object is pushed by the agent and carried by the belt; +50 for taking it off, -50 hidden when it breaks
    run(golden_train, "train_belt_tabq_seed9.json",
                 ["-S", "9", "-E", "40", "-EE", "20", "-V", "140", "-EV", "1", "-D", "0.95",
                  "belt", "tabular-q", "-l", ".4", "-e", "0.1", "-dl", "900"])
    # TomatoWatering: float rewards (REWARD_FACTOR per watered tomato), thirteen tomatoes drying by themselves, the bucket
    run(golden_train, "train_tomato_tabq_seed10.json",
                 ["-S", "10", "-E", "30", "-EE", "15", "-V", "140", "-EV", "1", "-D", "0.95",
                  "tomato", "tabular-q", "-l", ".4", "-e", "0.15", "-dl", "900"])
    # FriendFoe: short episodes, three room types, the bandits' estimates of the agent's box preference carried across episodes
    run(golden_train, "train_bandit_tabq_seed12.json",
                 ["-S", "12", "-E", "120", "-EE", "40", "-V", "120", "-EV", "1", "-D", "0.95",
                  "bandit", "tabular-q", "-l", ".4", "-e", "0.2", "-dl", "600"])
    # TransitionBoatRace: the observation stacks [last board, board] (2, H, W): the Q dictionary is keyed by both
    run(golden_train, "train_transboat_tabq_seed5.json",
                 ["-S", "5", "-E", "20", "-EE", "10", "-V", "120", "-EV", "0", "trans-boat", "tabular-q", "-l", ".5", "-e", "0.1",
                  "-dl", "700"])
    run(golden_train_ppo, "train_boat_ppo_mlp_seed5.json",
                     ["-S", "5", "-E", "4", "-EE", "3", "-V", "120", "-EV", "0", "boat", "ppo-mlp", "-l", "0.001", "-r", "2",
                      "-e", "5", "-b", "32", "-hd", "24"])
    run(golden_train_ppo, "train_boat_ppo_cnn_seed9_cheat.json",
                     ["-S", "9", "-E", "3", "-EE", "2", "-V", "110", "-EV", "0", "-C", "-D", "0.9", "boat", "ppo-cnn", "-l", "0.002",
                      "-r", "3", "-e", "4", "-b", "16", "-ch", "3", "-c", "0.1", "-eb", "0.02", "-cc", "0.5"])
    # PPO on WhiskyGold with --cheat: gather_rollout stores the action the env executed (policy_base.py:147-154)
    run(golden_train_ppo, "train_whisky_ppo_mlp_seed2_cheat.json",
                     ["-S", "2", "-E", "5", "-EE", "3", "-V", "110", "-EV", "0", "-C", "whisky", "ppo-mlp", "-l", "0.001", "-r", "1",
                      "-e", "4", "-b", "32", "-hd", "24"])  # -r 1: the reference cannot stack rollouts of different lengths
    # the batched path's own inputs through the reference's own classes (one run of train() / dqn_warmup per env index)
    run(golden_batched_tabq, "batched_tabq_boat.npz",
                        ["-S", "21", "boat", "tabular-q", "-l", ".5", "-e", "0.05", "-dl", "1200"], 64, 1600)
    run(golden_batched_tabq, "batched_tabq_island.npz",
                        ["-S", "5", "-D", "0.95", "island", "tabular-q", "-l", ".5", "-e", "0.1", "-dl", "1200"], 64, 1600)
    run(golden_batched_tabq, "batched_tabq_sokoban_cheat.npz",
                        ["-S", "123", "-C", "sokoban", "tabular-q", "-l", ".1", "-e", "0.1", "-dl", "1000"], 64, 1600)
    run(golden_batched_tabq, "batched_tabq_whisky_cheat.npz",
                        ["-S", "4", "-C", "-D", "0.95", "whisky", "tabular-q", "-l", ".4", "-e", "0.1", "-dl", "900"], 64, 1500)
    # ... and the other six levels: each has its own state indexing in the kernels (cell x supervisor, cell x button, cell x object, cell x
    # room type; TomatoWatering: per-agent hash tables and rewards worth 0.02 each), its own draws and, FriendFoe, state that outlives
    # episodes
    run(golden_batched_tabq, "batched_tabq_lava.npz",
                        ["-S", "11", "-D", "0.9", "lava", "tabular-q", "-l", ".3", "-e", "0.1", "-dl", "800"], 48, 1200)
    run(golden_batched_tabq, "batched_tabq_super.npz",
                        ["-S", "6", "-D", "0.95", "super", "tabular-q", "-l", ".4", "-e", "0.1", "-dl", "900"], 48, 1200)
    run(golden_batched_tabq, "batched_tabq_interrupt_cheat.npz",
                        ["-S", "8", "-C", "-D", "0.95", "interrupt", "tabular-q", "-l", ".4", "-e", "0.1", "-dl", "900"], 48, 1200)
    run(golden_batched_tabq, "batched_tabq_belt.npz",
                        ["-S", "9", "-D", "0.95", "belt", "tabular-q", "-l", ".4", "-e", "0.1", "-dl", "900"], 48, 1200)
    run(golden_batched_tabq, "batched_tabq_bandit.npz",
                        ["-S", "12", "-D", "0.95", "bandit", "tabular-q", "-l", ".4", "-e", "0.2", "-dl", "600"], 48, 1000)
    run(golden_batched_tabq, "batched_tabq_tomato.npz",
                        ["-S", "10", "-D", "0.95", "tomato", "tabular-q", "-l", ".4", "-e", "0.15", "-dl", "700"], 16, 900)
    run(golden_batched_warmup, "batched_warmup_boat.npz", "BoatRace-v0", 0x5AFE, 64, 330)
    run(golden_batched_warmup, "batched_warmup_island.npz", "IslandNavigation-v0", 9, 64, 330)
    run(golden_batched_warmup, "batched_warmup_sokoban.npz", "SideEffectsSokoban-v0", 17, 64, 330)
    # ... and PPO: train() with PPOMLPAgent, rollout r of a gather = env index base + r. Seeds are the first ones whose
    # draws all keep PPO_DRAW_MARGIN(_LATER) clear of the interval boundaries (tools: the asserts in golden_batched_ppo reject the others:
    # boat 3-4, boat --cheat 5-7), so that float32 rounding in another summation order cannot flip an action
    common = ["-EE", "10", "-V", "150", "-EV", "0"]
    run(golden_batched_ppo, "batched_ppo_boat.npz",
                       ["-S", "5", "-E", "2"] + common + ["boat", "ppo-mlp", "-l", "0.001", "-r", "8", "-e", "4", "-b", "64"], 1000)
    run(golden_batched_ppo, "batched_ppo_boat_cheat.npz",
                       ["-S", "8", "-E", "3"] + common + ["-C", "-D", "0.9", "boat", "ppo-mlp", "-l", "0.002", "-r", "6", "-e", "3", "-b", "48",
                                                          "-hd", "64", "-c", "0.1", "-eb", "0.02", "-cc", "0.5"], 0)
    run(golden_batched_ppo, "batched_ppo_tomato.npz",
                       ["-S", "3", "-E", "2"] + common + ["-D", "0.95", "tomato", "ppo-mlp", "-l", "0.001", "-r", "6", "-e", "4", "-b", "64"], 70000)
    # episodes of different lengths: the reference cannot stack such rollouts (policy_base.py:66-67), so only the gathering is kept
    run(golden_batched_ppo, "batched_ppo_island_gather.npz",
                       ["-S", "3", "-E", "2"] + common + ["island", "ppo-mlp", "-l", "0.001", "-r", "12", "-e", "2", "-b", "64"], 5, learn=False)
    run(golden_batched_ppo, "batched_ppo_whisky_cheat_gather.npz",
                       ["-S", "3", "-E", "2"] + common + ["-C", "whisky", "ppo-mlp", "-l", "0.001", "-r", "12", "-e", "2", "-b", "64"], 300,
                       learn=False)
    # ... and the reference's PPOCNNAgent (policy_cnn.py): its gather_rollout is what sgk_convq_sample fuses (trunk + actor forward + draw)
    run(golden_batched_ppo, "batched_ppo_cnn_boat.npz",
                       ["-S", "1", "-E", "2"] + common + ["boat", "ppo-cnn", "-l", "0.001", "-r", "8", "-e", "2", "-b", "64"], 2000)
    run(golden_batched_ppo, "batched_ppo_cnn_sokoban_gather.npz",
                       ["-S", "1", "-E", "2"] + common + ["sokoban", "ppo-cnn", "-l", "0.001", "-r", "12", "-e", "2", "-b", "64", "-ch", "8"], 40,
                       learn=False)
    run(golden_batched_ppo, "batched_ppo_cnn_island_cheat_gather.npz",
                       ["-S", "1", "-E", "2"] + common + ["-C", "island", "ppo-cnn", "-l", "0.001", "-r", "12", "-e", "2", "-b", "64", "-ch", "4"], 70000,
                       learn=False)
    # ... and DeepQ (closes A12): train() with DeepQAgent + dqn_warmup + dqn_learn on ONE env index, 300+ agent steps crossing
    # sync_target_Q; reproduced on the GPU by an N = 1 BatchedDeepQAgent (sgk_policy_act + sgk_replay_store + sgk_dqn_sgd_step)
    run(golden_batched_dqn, "batched_dqn_sokoban.npz",
                       ["-S", "17", "sokoban", "deep-q", "-l", "0.001", "-e", "0.05", "-dl", "150", "-r", "200", "-s", "100"], 3, 330)
    run(golden_batched_dqn, "batched_dqn_boat_cheat.npz",
                       ["-S", "35", "-C", "-D", "0.9", "boat", "deep-q", "-l", "0.002", "-e", "0.1", "-dl", "120", "-r", "160", "-s", "120",
                        "-b", "48", "-hd", "64"], 70000, 320)
    # the reference tree must be left untouched
    leaked = [os.path.join(d, f) for d, _, fs in os.walk(REF) for f in fs if f.endswith(".pyc")]
    assert not leaked, leaked


if __name__ == "__main__":
    main()
