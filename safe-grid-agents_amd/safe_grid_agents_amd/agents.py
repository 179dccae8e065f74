"""Agents with the reference's constructor and method surface (`Agent(env, args)`, act / act_explore / learn /
update_epsilon / sync_target_Q / replay.add), reference safe_grid_agents/common/agents/{base,dummy,value}.py and
common/utils/contain.py.

Two families:
  * single-env host agents (RandomAgent, SingleActionAgent, TabularQAgent, DeepQAgent): drop-ins for the
    reference classes, same numpy-global-RNG draw order, so a seeded run reproduces the reference's trajectory
    (tests/golden/train_*.json);
  * BatchedTabularQAgent: N private tabular agents whose tables live in HBM and whose act / learn / epsilon
    schedule run in the HIP kernels (sgk_tabq_*), bit-exact in float64 against the same update rule.
"""
import collections
import ctypes
import os
from typing import NamedTuple

import numpy as np

from . import _lib


# ---- mixins (reference base.py:5-47) ---------------------------------------------------------------
class BaseActor:
    def act(self, state, *args, **kwargs):
        raise NotImplementedError


class BaseExplorer:
    def act_explore(self, state, *args, **kwargs):
        raise NotImplementedError


class BaseLearner:
    def learn(self, *args, **kwargs):
        raise NotImplementedError


# ---- records (reference types.py:18-38) ------------------------------------------------------------
class Experience(NamedTuple):
    state: np.ndarray
    action: int
    reward: float
    successor: np.ndarray
    terminal: bool


ExperienceBatch = collections.namedtuple("ExperienceBatch", ["states", "actions", "rewards", "successors", "terminals"])
Rollout = collections.namedtuple("Rollout", ["states", "actions", "rewards", "returns"])


class ReplayBuffer:
    """Bounded FIFO of Experience; sampling is uniform WITH replacement from numpy's global RNG (contain.py:19-22)."""

    def __init__(self, capacity):
        self.capacity = capacity
        self._buffer = collections.deque(maxlen=capacity)

    def add(self, state, action, reward, successor, terminal):
        self._buffer.append(Experience(state, action, reward, successor, terminal))

    def sample(self, sample_size):
        picks = np.random.choice(len(self._buffer), sample_size)
        return [self._buffer[i] for i in picks]

    def __len__(self):
        return len(self._buffer)


# ---- dummy agents (reference dummy.py:7-30) --------------------------------------------------------
class RandomAgent(BaseActor):
    def __init__(self, env, args):
        self.action_n = env.action_space.n
        if args.seed:  # a falsy seed (None or 0) leaves the global RNG untouched, as dummy.py:12
            np.random.seed(args.seed)

    def act(self, state):
        return np.random.randint(0, self.action_n)


class SingleActionAgent(BaseActor):
    def __init__(self, env, args):
        self.action = args.action
        assert self.action < env.action_space.n, "Not a valid action."

    def act(self, state):
        return self.action


def _epsilon_schedule(epsilon, anneal):
    # element t is 1 - (1 - eps) * t / anneal in Python's evaluation order (value.py:23-26)
    return collections.deque(1.0 - (1 - epsilon) * t / anneal for t in range(anneal))


class _EpsilonMixin:
    def update_epsilon(self):
        if self.future_eps:
            self.epsilon = self.future_eps.popleft()  # the reference pops a list head: O(n) per step
        return self.epsilon


# ---- tabular Q (reference value.py:15-58) -----------------------------------------------------------
class TabularQAgent(_EpsilonMixin, BaseActor, BaseLearner, BaseExplorer):
    """Dictionary Q-table keyed by the flattened board; float64 rows of action values."""

    def __init__(self, env, args):
        self.action_n = env.action_space.n
        self.discount = args.discount
        self.lr = args.lr
        self.future_eps = _epsilon_schedule(args.epsilon, args.epsilon_anneal)
        self.update_epsilon()
        self.epsilon = 0.0  # value.py:28: the very first action is always greedy
        self.Q = collections.defaultdict(lambda: np.zeros(self.action_n))

    @staticmethod
    def _key(state):
        return tuple(state.flatten())

    def act(self, state):
        return np.argmax(self.Q[self._key(state)])

    def act_explore(self, state):
        # draw order matters for seed parity: one uniform always, one integer only when exploring
        if np.random.sample() < self.epsilon:
            return np.random.choice(self.action_n)
        return self.act(state)

    def learn(self, state, action, reward, successor):
        s, s2 = self._key(state), self._key(successor)
        best_next = np.argmax(self.Q[s2])
        target = reward + self.discount * self.Q[s2][best_next]  # no terminal masking (value.py:48-50)
        row = self.Q[s]
        row[action] += self.lr * (target - row[action])


# ---- deep Q (reference value.py:61-187) --------------------------------------------------------------
class DeepQAgent(_EpsilonMixin, BaseActor, BaseLearner, BaseExplorer):
    """MLP Q-network + target network, replay, Adam(amsgrad), MSE, grad-clip 10 -- PyTorch-ROCm.

    Differences from the reference, all forced by it not running as written:
      * n_input = prod(observation shape) (the reference multiplies only the first two dims, value.py:66-67, which
        only works for 2-D observations; for those the two agree);
      * terminal mask is bool (uint8 masks are rejected by torch >= 2, value.py:121,179);
      * `reference_loss_broadcast=True` keeps the reference's [B,1]-vs-[B] mse_loss broadcast (value.py:119-123).
    """

    def __init__(self, env, args, reference_loss_broadcast=True):
        import torch

        self.torch = torch
        self.action_n = env.action_space.n
        shape = env.observation_space.shape
        self.n_input = int(np.prod(shape))
        self.device = args.device
        if isinstance(self.device, int):
            self.device = "cuda:%d" % self.device
        self.log_gradients = getattr(args, "log_gradients", False)
        self.reference_loss_broadcast = reference_loss_broadcast
        self.future_eps = _epsilon_schedule(args.epsilon, args.epsilon_anneal)
        self.update_epsilon()  # unlike TabularQAgent there is no overwrite: the first epsilon is 1.0
        self.discount = args.discount
        self.lr = args.lr
        self.batch_size = args.batch_size
        self.Q = self.build_Q(self.n_input, args.n_layers, args.n_hidden).to(self.device).eval()
        self.target_Q = self.build_Q(self.n_input, args.n_layers, args.n_hidden).to(self.device).eval()
        self.replay = ReplayBuffer(args.replay_capacity)
        self.optim = torch.optim.Adam(self.Q.parameters(), lr=args.lr, amsgrad=True)

    def build_Q(self, n_input, n_layers, n_hidden):
        nn = self.torch.nn
        first = nn.Sequential(nn.Linear(n_input, n_hidden), nn.ReLU())
        hidden = nn.Sequential(*[nn.Sequential(nn.Linear(n_hidden, n_hidden), nn.ReLU()) for _ in range(n_layers - 1)])
        return nn.Sequential(first, hidden, nn.Linear(n_hidden, int(self.action_n)))

    def _lift(self, x, dtype=None, grad=False):
        t = self.torch.as_tensor(x, dtype=dtype or self.torch.float32, device=self.device)
        return t.requires_grad_() if grad else t

    def act(self, state):
        board = self._lift(np.asarray(state).flatten()).reshape(1, -1)
        return self.Q(board).argmax(1)  # a 1-element tensor, as value.py:92 (env.step accepts it)

    def policy(self, state):
        greedy = self.act(state)
        probs = self.torch.full((self.action_n,), self.epsilon / self.action_n, dtype=self.torch.float32, device=self.device)
        probs[greedy] += 1 - self.epsilon
        return self.torch.distributions.Categorical(probs=probs)

    def act_explore(self, state):
        return self.policy(state).sample().item()

    def process(self, experiences):
        boards = np.concatenate([e.state.flatten() for e in experiences], axis=0)
        successors = np.concatenate([e.successor.flatten() for e in experiences], axis=0)
        return ExperienceBatch(
            self._lift(boards, grad=True).reshape(-1, self.n_input),
            self._lift([e.action for e in experiences], dtype=self.torch.long).reshape(-1, 1),
            self._lift([e.reward for e in experiences]),
            self._lift(successors, grad=True).reshape(-1, self.n_input),
            self._lift([e.terminal for e in experiences], dtype=self.torch.bool),
        )

    def learn(self, state, action, reward, successor, terminal, history):
        torch = self.torch
        self.replay.add(state, action, reward, successor, terminal)
        states, actions, rewards, successors, terminals = self.process(self.replay.sample(self.batch_size))
        self.Q.train()
        q_sa = self.Q(states).gather(1, actions)
        next_q = self.target_Q(successors).max(1)[0]
        next_q[terminals] = 0
        expected = self.discount * next_q + rewards
        if not self.reference_loss_broadcast:
            q_sa = q_sa.squeeze(1)
        loss = torch.nn.functional.mse_loss(q_sa, expected)
        history["writer"].add_scalar("Train/value_loss", loss.item(), history["t"])
        self.optim.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(self.Q.parameters(), 10.0)
        if self.log_gradients:
            for name, param in self.Q.named_parameters():
                history["writer"].add_histogram(name, param.grad.clone().cpu().data.numpy(), history["t"])
        self.optim.step()
        self.Q.eval()
        return history

    def sync_target_Q(self):
        self.target_Q.load_state_dict(self.Q.state_dict())


# ---- batched tabular Q on the GPU -------------------------------------------------------------------
class BatchedTabularQAgent(BaseActor, BaseLearner, BaseExplorer):
    """N private TabularQAgents (one per env of a BatchedGridworldEnv), tables float64 in HBM, state-major: [n_states][N][4]
    (table() / table_host() present them agent by agent, [N][n_states][4]).

    Same hyper-parameter names as the reference (`args.lr`, `.discount`, `.epsilon`, `.epsilon_anneal`). The
    dictionary key (flattened board) becomes a perfect-hash state index (agent cell, or agent cell x box cell);
    exploration draws come from the counter RNG (Philox stream 1: one 53-bit uniform + one action per step), so
    parity with the CPU restatement is stated on that stream, not on numpy's Mersenne Twister.
    """

    def __init__(self, env, args):
        import torch

        self.env = env
        self.lib = env.lib
        self.action_n = env.action_space.n
        self.lr, self.discount = float(args.lr), float(args.discount)
        self.epsilon0, self.epsilon_anneal = float(args.epsilon), int(args.epsilon_anneal)
        h = ctypes.c_void_p()
        # levels without a perfect hash of their boards (TomatoWatering): slots of each agent's hash table, 0 = the library's default
        # (args.hash_capacity, else SGK_TABQ_HASH_CAPACITY: the reference's flag grammar has no such option to extend)
        self.hash_capacity = int(getattr(args, "hash_capacity", 0) or os.environ.get("SGK_TABQ_HASH_CAPACITY", 0) or 0)
        env._follow()  # the tables are zeroed on the stream the handle enqueues on
        _lib.check(self.lib.sgk_tabq_create_ex(env.handle, self.lr, self.discount, self.epsilon0, self.epsilon_anneal,
                                               self.hash_capacity, ctypes.byref(h)))
        self._h = h
        p, ns = ctypes.c_void_p(), ctypes.c_int64()
        _lib.check(self.lib.sgk_tabq_table_dev(h, ctypes.byref(p), ctypes.byref(ns), None))
        self.n_states = int(ns.value)  # rows per agent: the level's state count, or the hash capacity
        self._actions = torch.empty(env.n_envs, dtype=torch.uint8, device="cuda:%d" % env.device)

    def close(self):
        if getattr(self, "_h", None):
            self.lib.sgk_tabq_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def t(self):
        t = ctypes.c_int64()
        _lib.check(self.lib.sgk_tabq_global_step(self._h, ctypes.byref(t)))
        return t.value

    @property
    def epsilon(self):
        return self.lib.sgk_tabq_epsilon(self.epsilon0, self.epsilon_anneal, self.t)

    def update_epsilon(self):
        return self.epsilon  # the schedule advances inside learn(), as learn.py:79-82 pairs them

    def _act(self, explore):
        self.env._sync_torch_to_lib()
        _lib.check(self.lib.sgk_tabq_act(self._h, int(explore), ctypes.c_void_p(self._actions.data_ptr())))
        self.env._sync_lib_to_torch()
        return self._actions

    def act(self, state=None):
        return self._act(False)

    def act_explore(self, state=None):
        return self._act(True)

    def learn(self, state=None, action=None, reward=None, successor=None, cheat=False):
        """Uses the env's last step records and current state words; `action` defaults to the last act*() output."""
        actions = self._actions if action is None else self.env._actions_arg(action)
        self.env._sync_torch_to_lib()
        _lib.check(self.lib.sgk_tabq_learn(self._h, ctypes.c_void_p(actions.data_ptr()), int(cheat)))

    def step(self, cheat=False, write_boards=True):
        """ONE lockstep step of tabq_learn for every (env, agent) pair in ONE launch (sgk_tabq_step): act_explore -> env.step ->
        learn -> update_epsilon -> reset of the finished envs (learn.py:61-85 inside train.py:62-70). Returns
        (actions, (boards, reward, done, info)): the chosen actions (uint8 [N], valid until the next call) and env.step's tuple --
        views of the step records as sgk_step writes them and of the boards (an env whose episode ended shows its next
        episode's first board, as after reset_done). Same results as act_explore / env.step / learn / reset_done."""
        flags = 0 if write_boards else _lib.F_NO_BOARDS
        self.env._sync_torch_to_lib()
        self.env._version += 1
        _lib.check(self.lib.sgk_tabq_step(self._h, int(cheat), flags, ctypes.c_void_p(self._actions.data_ptr())))
        self.env._sync_lib_to_torch()
        return self._actions, self.env._step_outputs()

    def learn_steps(self, n_steps, cheat=False, write_boards=False, separate_launches=False):
        """n_steps lockstep steps of tabq_learn replayed from one hipGraph (sgk_tabq_learn_steps): no Python, no host round trip
        between the launches. One launch per step (the kernel of step()); separate_launches=True records the drop-in call
        sequence act_explore -> env.step -> learn -> reset_done instead (four launches per step: rounds 2-5's form)."""
        flags = (0 if write_boards else _lib.F_NO_BOARDS) | (_lib.F_SEPARATE_LAUNCHES if separate_launches else 0)
        self.env._follow()
        self.env._version += 1
        _lib.check(self.lib.sgk_tabq_learn_steps(self._h, int(n_steps), int(cheat), flags))

    def rollout(self, n_steps, cheat=False, kernel="auto"):
        """n_steps of {act_explore, env.step, learn, update_epsilon, reset on done} fused on the GPU. `kernel`: "auto", or
        "lds" / "hbm" to name the kernel (tables resident in LDS / rows in HBM; same results)."""
        k = {"auto": _lib.TABQ_KERNEL_AUTO, "lds": _lib.TABQ_KERNEL_LDS, "hbm": _lib.TABQ_KERNEL_HBM}[kernel]
        self.env._follow()
        self.env._version += 1
        _lib.check(self.lib.sgk_tabq_rollout_ex(self._h, int(n_steps), int(cheat), k))

    def table(self):
        """The Q tables where they live: a float64 torch view [N, n_states, 4] over HBM (zero copy, sgk_tabq_table_dev). The memory
        is state-major ([n_states][N][4]: a wave's 64 agents side by side in every state's plane), so the view is a PERMUTED one
        (not contiguous; element-wise reads and writes go through as usual). After WRITING through a view that was obtained
        earlier, call invalidate_rows() before the next act / learn / learn_steps."""
        from .envs import _view

        p, ns, na = ctypes.c_void_p(), ctypes.c_int64(), ctypes.c_int64()
        _lib.check(self.lib.sgk_tabq_table_dev(self._h, ctypes.byref(p), ctypes.byref(ns), ctypes.byref(na)))
        return _view(self, self.env.device, p.value, (ns.value, self.env.n_envs, na.value), "float64").permute(1, 0, 2)

    def invalidate_rows(self):
        """The table was written from outside (through table()): the per-step kernels forget the one row per env they keep."""
        _lib.check(self.lib.sgk_tabq_invalidate_rows(self._h))

    def table_host(self, env_begin=0, env_count=None):
        env_count = self.env.n_envs - env_begin if env_count is None else env_count
        out = np.empty((env_count, self.n_states, self.action_n), dtype=np.float64)
        self.env._follow()
        _lib.check(self.lib.sgk_tabq_copy_table(self._h, env_begin, env_count, out.ctypes.data))
        return out

    def keys_host(self, env_begin=0, env_count=None):
        """Hashed levels (TomatoWatering): uint32 [env_count, capacity], the board held by each slot of each agent's table
        (0xffffffff = empty; agent cell | shown watered set << 8, 0x2000 = the bucket's delusion board); rows as in table_host()."""
        env_count = self.env.n_envs - env_begin if env_count is None else env_count
        out = np.empty((env_count, self.n_states), dtype=np.uint32)
        self.env._follow()
        _lib.check(self.lib.sgk_tabq_copy_keys(self._h, env_begin, env_count, out.ctypes.data))
        return out

    def check_hash_tables(self):
        """Raise when some agent's hash table filled up (hashed levels): its Q-values are undefined from then on."""
        cap, used, overflowed = self.hash_info()
        if overflowed:
            raise RuntimeError("a tabular-Q hash table of %d slots per agent overflowed on %s: re-run with a larger table "
                               "(args.hash_capacity / SGK_TABQ_HASH_CAPACITY, a power of two; 36 bytes per slot and agent)"
                               % (cap, self.env.name))
        return cap, used

    def hash_info(self):
        """(capacity, slots used by the fullest agent, overflowed) -- capacity 0 for perfect-hash levels."""
        c, u, o = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        self.env._follow()
        _lib.check(self.lib.sgk_tabq_hash_info(self._h, ctypes.byref(c), ctypes.byref(u), ctypes.byref(o)))
        return c.value, u.value, bool(o.value)

    def check_hash_overflow(self):
        """The cheap form of check_hash_tables (a 4-byte copy, no slot count): raise when some board found its agent's table full.
        The batched trainer calls it at every period's synchronisation point."""
        c, o = ctypes.c_int32(), ctypes.c_int32()
        self.env._follow()
        _lib.check(self.lib.sgk_tabq_hash_info(self._h, ctypes.byref(c), None, ctypes.byref(o)))
        if o.value:
            self.check_hash_tables()


from .ppo import PPOCNNAgent, PPOMLPAgent  # noqa: E402  (ppo.py needs the mixins defined above)

AGENT_MAP = {  # reference parsing/parse.py:39-48, restricted to the hot-path scope and its "next" rows
    "random": RandomAgent,
    "single": SingleActionAgent,
    "tabular-q": TabularQAgent,
    "deep-q": DeepQAgent,
    "ppo-mlp": PPOMLPAgent,
    "ppo-cnn": PPOCNNAgent,
}
