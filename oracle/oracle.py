"""ctypes wrapper around oracle/liboracle_sgk.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (safe-grid-agents_amd/) never does; it fails loudly without its HIP library.

See oracle/sgk_oracle.c for what is restated and what pins it (env part: PARITY UNPINNED).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SGK_ORACLE_SO: a variant built elsewhere under an alternative reading of the upstream rules (tests/test_switch_variants.py)
_SO = os.environ.get("SGK_ORACLE_SO") or os.path.join(_HERE, "liboracle_sgk.so")

ENV_IDS = {"BoatRace-v0": 0, "IslandNavigation-v0": 1, "SideEffectsSokoban-v0": 2, "DistributionalShift-v0": 3,
           "WhiskyGold-v0": 4, "AbsentSupervisor-v0": 5, "SafeInterruptibility-v0": 6, "ConveyorBelt-v0": 7, "TomatoWatering-v0": 8, "FriendFoe-v0": 9}
M_LEN = 16
(M_SUM_RETURN, M_SUM_SAFETY, M_SUM_MARGIN, M_SUM_MARGIN_POS, M_EPISODES, M_MARGIN_POS_COUNT, M_STEPS, M_RESERVED,
 M_MAX_RETURN, M_MAX_SAFETY, M_MAX_MARGIN, M_MAX_MARGIN_POS) = range(12)


def _ok(rc):
    """A C call that must succeed (not `assert call() == 0`: python -O would drop the call with the assert)."""
    if rc != 0:
        raise RuntimeError("oracle call failed: %r" % (rc,))


def build(force=False):
    """Compile the C restatement with gcc (oracle/Makefile)."""
    if os.environ.get("SGK_ORACLE_SO"):
        return _SO  # a prebuilt variant: its builder owns it
    src = os.path.join(_HERE, "sgk_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "sgk_levels.h")
    stale = (not os.path.exists(_SO)) or any(
        os.path.exists(p) and os.path.getmtime(p) > os.path.getmtime(_SO) for p in (src, hdr))
    if (force or stale) and os.environ.get("SGK_NO_BUILD") == "1":  # profiler runs: never spawn a compiler from here
        raise RuntimeError("liboracle_sgk.so is missing or stale and SGK_NO_BUILD=1 forbids building it here")
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle_sgk.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        c_int_p = ctypes.POINTER(ctypes.c_int)
        L.orc_sizeof.restype = ctypes.c_size_t
        L.orc_init.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.orc_reset.argtypes = [ctypes.c_void_p]
        L.orc_step.argtypes = [ctypes.c_void_p, ctypes.c_int, c_int_p, c_int_p, c_int_p, c_int_p]
        L.orc_board.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.orc_shape.argtypes = [ctypes.c_int, c_int_p, c_int_p]
        L.orc_render_rgb.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.orc_render_rgb.restype = ctypes.c_int
        for name in ("orc_episode_return", "orc_hidden_return", "orc_n_episodes", "orc_last_episode_return",
                     "orc_safety", "orc_frame", "orc_game_over", "orc_agent_cell", "orc_box_cell", "orc_exploring",
                     "orc_supervisor", "orc_coin", "orc_n_resets", "orc_tomato_mask", "orc_ext"):
            getattr(L, name).argtypes = [ctypes.c_void_p]
            getattr(L, name).restype = ctypes.c_int
        L.orc_foe_policy.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.orc_reward_scale.argtypes = [ctypes.c_int]
        L.orc_reward_scale.restype = ctypes.c_double
        L.orc_init_batch.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int]
        L.orc_export.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
        L.orc_set_rng.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64]
        L.orc_last_performance.argtypes = [ctypes.c_void_p, c_int_p]
        L.orc_last_performance.restype = ctypes.c_int
        L.orc_philox4x32_10.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.orc_random_action.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64]
        L.orc_random_action.restype = ctypes.c_int
        L.orc_explore_draw.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64,
                                       ctypes.POINTER(ctypes.c_double), c_int_p]
        L.orc_metrics_init.argtypes = [ctypes.c_void_p]
        L.orc_rollout.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64,
                                  ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.orc_rollout_mt.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64,
                                     ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
        L.orc_rollout_mt.restype = ctypes.c_int
        L.orc_eps_greedy.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64]
        L.orc_eps_greedy.restype = ctypes.c_int
        L.orc_minibatch_index.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64]
        L.orc_minibatch_index.restype = ctypes.c_int64
        L.orc_ppo_row.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64]
        L.orc_ppo_row.restype = ctypes.c_int64
        L.orc_categorical_sample.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p]
        L.orc_categorical_sample.restype = ctypes.c_int
        L.orc_discounted_returns.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_void_p]
        L.orc_tabq_new.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int64]
        L.orc_tabq_new.restype = ctypes.c_void_p
        L.orc_tabq_free.argtypes = [ctypes.c_void_p]
        L.orc_tabq_lookup.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.orc_tabq_lookup.restype = ctypes.c_int
        L.orc_tabq_n_rows.argtypes = [ctypes.c_void_p]
        L.orc_tabq_n_rows.restype = ctypes.c_int
        L.orc_epsilon.argtypes = [ctypes.c_double, ctypes.c_int64, ctypes.c_int64]
        L.orc_epsilon.restype = ctypes.c_double
        L.orc_tabq_act.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.orc_tabq_act.restype = ctypes.c_int
        L.orc_tabq_learn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_void_p]
        L.orc_tabq_rollout.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64,
                                       ctypes.c_uint64, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        _lib = L
    return _lib


def _env_id(name_or_id):
    return ENV_IDS[name_or_id] if isinstance(name_or_id, str) else int(name_or_id)


def shape(env):
    H, W = ctypes.c_int(), ctypes.c_int()
    _ok(lib().orc_shape(_env_id(env), ctypes.byref(H), ctypes.byref(W)))
    return H.value, W.value


def philox4x32_10(ctr, key):
    c = np.asarray(ctr, dtype=np.uint32)
    k = np.asarray(key, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    lib().orc_philox4x32_10(c.ctypes.data, k.ctypes.data, out.ctypes.data)
    return out


def random_action(seed, env, t):
    return lib().orc_random_action(seed, env, t)


def random_actions(seed, env_begin, n, t_begin, n_steps):
    """[n_steps, n] uint8 matrix of the Philox stream-0 actions."""
    L = lib()
    out = np.empty((n_steps, n), dtype=np.uint8)
    for s in range(n_steps):
        for i in range(n):
            out[s, i] = L.orc_random_action(seed, env_begin + i, t_begin + s)
    return out


def explore_draw(seed, env, t):
    u, a = ctypes.c_double(), ctypes.c_int()
    lib().orc_explore_draw(seed, env, t, ctypes.byref(u), ctypes.byref(a))
    return u.value, a.value


def epsilon(eps0, anneal, t):
    return lib().orc_epsilon(eps0, anneal, t)


def metrics_new():
    m = np.zeros(M_LEN, dtype=np.int64)
    lib().orc_metrics_init(m.ctypes.data)
    return m


class EnvBatch:
    """n independent oracle envs (array of orc_env records)."""

    def __init__(self, env, n, reset=True, seed=0, env_begin=0):
        """`seed` / `env_begin` key the envs' own draws from the start (AbsentSupervisor flips its coin at every reset, the
        first one included): pass what the product's batch was created with."""
        L = lib()
        self.env_id = _env_id(env)
        self.n = int(n)
        self.H, self.W = shape(self.env_id)
        self.rec = L.orc_sizeof()
        self.buf = ctypes.create_string_buffer(self.rec * max(self.n, 1))
        self.base = ctypes.addressof(self.buf)
        _ok(L.orc_init_batch(self.base, self.n, self.env_id, int(seed), int(env_begin), int(bool(reset))))

    def ptr(self, i=0):
        return self.base + i * self.rec

    def set_rng(self, seed, env_begin=0):
        """Key of the envs' own draws (WhiskyGold): the batch seed and the global index of env 0. The rollout functions set it
        from their seed / env_begin arguments; direct step() users set it here."""
        L = lib()
        for i in range(self.n):
            L.orc_set_rng(self.ptr(i), int(seed), int(env_begin) + i)

    def reset(self, i=None):
        L = lib()
        for k in (range(self.n) if i is None else [i]):
            L.orc_reset(self.ptr(k))

    def step(self, i, action):
        r, h, d, a = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        lib().orc_step(self.ptr(i), int(action), ctypes.byref(r), ctypes.byref(h), ctypes.byref(d), ctypes.byref(a))
        return r.value, h.value, d.value, a.value

    def board(self, i):
        out = np.empty(self.H * self.W, dtype=np.int8)
        lib().orc_board(self.ptr(i), out.ctypes.data)
        return out.reshape(self.H, self.W)

    def render_rgb(self, i):
        out = np.empty((3, self.H, self.W), dtype=np.uint8)
        _ok(lib().orc_render_rgb(self.ptr(i), out.ctypes.data))
        return out

    def boards(self):
        out = np.empty((self.n, self.H * self.W), dtype=np.int8)
        L = lib()
        for i in range(self.n):
            L.orc_board(self.ptr(i), out[i].ctypes.data)
        return out

    EXPORT_FIELDS = ("episode_return", "hidden_return", "frame", "game_over", "agent_cell", "box_cell", "n_episodes",
                     "last_episode_return", "last_performance", "coin")

    def export(self, boards=True):
        """(boards int8 [n, H*W] or None, {field: int32 [n]}) of every env in one C call (orc_export): what a parity test
        compares at a million envs."""
        b = np.empty((self.n, self.H * self.W), dtype=np.int8) if boards else None
        f = np.empty((self.n, 10), dtype=np.int32)
        lib().orc_export(self.base, self.n, None if b is None else b.ctypes.data, f.ctypes.data)
        return b, {name: f[:, k] for k, name in enumerate(self.EXPORT_FIELDS)}

    def field(self, name):
        f = getattr(lib(), "orc_" + name)
        return np.array([f(self.ptr(i)) for i in range(self.n)], dtype=np.int32)

    def foe_policy(self, i=None):
        """FriendFoe: the three bandits' estimates of the agent's box preference, float64 [n, 3, 2] (or [3, 2] for env i)."""
        out = np.empty((self.n, 6), dtype=np.float64)
        for k in range(self.n):
            lib().orc_foe_policy(self.ptr(k), out[k].ctypes.data)
        out = out.reshape(self.n, 3, 2)
        return out if i is None else out[i]

    def last_performance(self, i):
        has = ctypes.c_int()
        v = lib().orc_last_performance(self.ptr(i), ctypes.byref(has))
        return v if has.value else None

    def rollout(self, n_steps, seed=0, env_begin=0, t_begin=0, auto_reset=True, actions=None, metrics=None):
        """Run n_steps lockstep steps; returns rec [n,4] int8 (reward, hidden, done, actual) of the last step."""
        rec = np.zeros((self.n, 4), dtype=np.int8)
        a_ptr = None
        if actions is not None:
            actions = np.ascontiguousarray(actions, dtype=np.uint8)
            if actions.shape != (n_steps, self.n):
                raise ValueError("actions must be [n_steps, n]")
            a_ptr = actions.ctypes.data
        lib().orc_rollout(self.base, self.n, env_begin, seed, t_begin, n_steps, int(auto_reset), a_ptr,
                          rec.ctypes.data, None if metrics is None else metrics.ctypes.data)
        return rec


def rollout_mt(envs, n_steps, n_threads, seed=0, env_begin=0, t_begin=0, auto_reset=True, metrics=None):
    """EnvBatch.rollout with the env range split over n_threads POSIX threads; returns the thread count used."""
    used = lib().orc_rollout_mt(envs.base, envs.n, env_begin, seed, t_begin, n_steps, int(auto_reset),
                                None if metrics is None else metrics.ctypes.data, int(n_threads))
    assert used > 0, "thread creation failed"
    return used


def eps_greedy(scores, eps, seed, env_begin, draw):
    sc = np.ascontiguousarray(scores, dtype=np.float32)
    L = lib()
    return np.array([L.orc_eps_greedy(sc[i].ctypes.data, float(eps), seed, env_begin + i, draw) for i in range(sc.shape[0])],
                    dtype=np.uint8)


def minibatch_indices(seed, step, batch, total):
    """Flat replay indices the fused DeepQ learner samples for the SGD step that starts at Adam step `step`."""
    L = lib()
    return np.array([L.orc_minibatch_index(seed, b, step, total) for b in range(batch)], dtype=np.int64)


def ppo_rows(seed, step, batch, lengths, horizon):
    """Flat rollout rows t * N + env the fused PPO learner draws for the epoch that starts at Adam step `step`."""
    L = lib()
    ln = np.ascontiguousarray(lengths, dtype=np.int32)
    return np.array([L.orc_ppo_row(seed, b, step, ln.ctypes.data, horizon, ln.size) for b in range(batch)], dtype=np.int64)


def categorical_sample(logits, seed, env_begin, draw):
    """(actions uint8 [n], margins float64 [n]) of orc_categorical_sample for envs env_begin .. env_begin + n - 1."""
    lg = np.ascontiguousarray(logits, dtype=np.float32)
    L = lib()
    margin = ctypes.c_double()
    acts = np.empty(lg.shape[0], dtype=np.uint8)
    margins = np.empty(lg.shape[0], dtype=np.float64)
    for i in range(lg.shape[0]):
        acts[i] = L.orc_categorical_sample(lg[i].ctypes.data, seed, env_begin + i, draw, ctypes.byref(margin))
        margins[i] = margin.value
    return acts, margins


def has_hidden_reward(env):
    """False for envs that define no hidden reward (their step record mirrors the observed reward; performance = return)."""
    return bool(lib().orc_has_hidden_reward(_env_id(env)))


def reward_scale(env):
    """What one unit of the integer rewards is worth (TomatoWatering: REWARD_FACTOR per watered tomato; 1.0 elsewhere)."""
    return float(lib().orc_reward_scale(_env_id(env)))


def discounted_returns(rewards, discount):
    r = np.ascontiguousarray(rewards, dtype=np.float32)
    out = np.empty_like(r)
    lib().orc_discounted_returns(r.ctypes.data, r.shape[0], float(discount), out.ctypes.data)
    return out


class TabQ:
    """One literal dict-keyed tabular Q agent (reference value.py:15-58)."""

    def __init__(self, ncell, lr, discount, eps0, anneal):
        self.ncell = ncell
        self.h = lib().orc_tabq_new(ncell, lr, discount, eps0, anneal)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_tabq_free(self.h)
            self.h = None

    def act(self, board):
        b = np.ascontiguousarray(board, dtype=np.int8).ravel()
        return lib().orc_tabq_act(self.h, b.ctypes.data)

    def learn(self, state, action, reward, successor):
        s = np.ascontiguousarray(state, dtype=np.int8).ravel()
        s2 = np.ascontiguousarray(successor, dtype=np.int8).ravel()
        lib().orc_tabq_learn(self.h, s.ctypes.data, int(action), float(reward), s2.ctypes.data)

    def lookup(self, board):
        b = np.ascontiguousarray(board, dtype=np.int8).ravel()
        q = np.zeros(4, dtype=np.float64)
        lib().orc_tabq_lookup(self.h, b.ctypes.data, q.ctypes.data)
        return q

    @property
    def n_rows(self):
        return lib().orc_tabq_n_rows(self.h)


def tabq_rollout(envs, agents, n_steps, seed=0, env_begin=0, cheat=False, metrics=None, record_actions=False):
    n = envs.n
    arr = (ctypes.c_void_p * n)(*[a.h for a in agents])
    acts = np.zeros((n_steps, n), dtype=np.uint8) if record_actions else None
    lib().orc_tabq_rollout(envs.base, arr, n, env_begin, seed, n_steps, int(cheat),
                           None if metrics is None else metrics.ctypes.data,
                           None if acts is None else acts.ctypes.data)
    return acts
