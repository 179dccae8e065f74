"""Gridworld envs behind the safe-grid-gym Env API, backed by the HIP library (no CPU path).

`GridworldEnv`         one env, the exact gym duck type the reference touches:
                       reset / step / seed / render, action_space.n, observation_space.shape,
                       `_env.episode_return`, `_env.get_last_performance()`
                       (reference train.py:51-52,64; learn.py:38,69; eval.py:13-42; warmup.py:16-20;
                       meters.py:67-80).
`BatchedGridworldEnv`  N envs in lockstep on one GPU, same method names, arrays with a leading N that
                       stay in HBM (torch views over the library's buffers, zero copy).
`make(name)`           the reference's `gym.make(ENV_MAP[alias])` (train.py:51) without gym.
"""
import ctypes

import numpy as np

from . import _lib

ENV_IDS = {
    "BoatRace-v0": _lib.BOAT_RACE,
    "IslandNavigation-v0": _lib.ISLAND_NAVIGATION,
    "SideEffectsSokoban-v0": _lib.SIDE_EFFECTS_SOKOBAN,
    "DistributionalShift-v0": _lib.DISTRIBUTIONAL_SHIFT,
    "WhiskyGold-v0": _lib.WHISKY_GOLD,
    "AbsentSupervisor-v0": _lib.ABSENT_SUPERVISOR,
    "SafeInterruptibility-v0": _lib.SAFE_INTERRUPTIBILITY,
    "ConveyorBelt-v0": _lib.CONVEYOR_BELT,
    "TomatoWatering-v0": _lib.TOMATO_WATERING,
    "FriendFoe-v0": _lib.FRIEND_FOE,
}
# safe-grid-gym registers some envs a second time with use_transitions=True: the observation stacks the PREVIOUS board and the
# current one, (2, H, W) (consistent with reference spiky/agents.py:43-44, which indexes channel 0 / 1 of such observations).
# name -> the env whose rules it shares
TRANSITION_ENVS = {"TransitionBoatRace-v0": "BoatRace-v0"}

# ENV_MAP ids of levels that only the reference's fork of ai-safety-gridworlds has (reference parse.py:32-35, spiky/agents.py:11,35)
FORK_ONLY_ENVS = frozenset({"TomatoCrmdp-v0", "ToyGridworldCorners-v0", "ToyGridworldOnTheWay-v0"})

# envs that define no hidden reward upstream: performance = episode return, info["hidden_reward"] is None in the
# single-env wrapper (the batched integer record mirrors the observed reward instead, include/sgk_levels.h)
NO_HIDDEN_REWARD = frozenset({"DistributionalShift-v0", "FriendFoe-v0"})

# reference parsing/parse.py:22-37; the three envs of the hot-path scope plus DistributionalShift, WhiskyGold, AbsentSupervisor
# and SafeInterruptibility
ENV_MAP = {
    "bandit": "FriendFoe-v0",
    "belt": "ConveyorBelt-v0",
    "boat": "BoatRace-v0",
    "interrupt": "SafeInterruptibility-v0",
    "island": "IslandNavigation-v0",
    "lava": "DistributionalShift-v0",
    "sokoban": "SideEffectsSokoban-v0",
    "super": "AbsentSupervisor-v0",
    "tomato": "TomatoWatering-v0",
    "tomato-crmdp": "TomatoCrmdp-v0",
    "whisky": "WhiskyGold-v0",
    "corners": "ToyGridworldCorners-v0",
    "way": "ToyGridworldOnTheWay-v0",
    "trans-boat": "TransitionBoatRace-v0",
}

_TORCH = [None]   # the torch module, once imported (envs.py itself imports without it)
_DTYPES = {}      # tuple of dtype names -> frozenset of torch dtypes (BatchedGridworldEnv._check)
_TYPESTR = {"int8": "|i1", "uint8": "|u1", "int32": "<i4", "int64": "<i8", "float64": "<f8", "uint32": "<u4"}


class _DeviceBuffer:
    """Exposes a library-owned HBM range through __cuda_array_interface__ so torch can view it zero-copy."""

    def __init__(self, owner, ptr, shape, dtype, strides=None):
        self._owner = owner  # keeps the handle (and thus the memory) alive
        self.__cuda_array_interface__ = {
            "shape": tuple(int(x) for x in shape),
            "typestr": _TYPESTR[dtype],
            "data": (int(ptr), False),
            "version": 2,
            "strides": None if strides is None else tuple(int(x) for x in strides),
        }


def _view(owner, device, ptr, shape, dtype, strides=None):
    import torch

    return torch.as_tensor(_DeviceBuffer(owner, ptr, shape, dtype, strides), device="cuda:%d" % device)


class _Space:
    """gym.spaces stand-in: the reference reads only `.n` (dummy.py:11, value.py:19) and `.shape` (value.py:66)."""

    def __init__(self, n=None, shape=None):
        self.n = n
        self.shape = shape

    def sample(self):
        return int(np.random.randint(0, self.n))


class _Handle:
    """Owns one sgk_env*."""

    def __init__(self, env_id, n_envs, device, seed, env_index_base, layout):
        self.lib = _lib.load()
        h = ctypes.c_void_p()
        _lib.check(self.lib.sgk_create_ex(env_id, n_envs, device, seed & (2**64 - 1), env_index_base, layout,
                                          ctypes.byref(h)))
        self.ptr = h

    def close(self):
        if getattr(self, "ptr", None):
            self.lib.sgk_destroy(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _RingMemory:
    """Device memory from sgk_ring_alloc as a __cuda_array_interface__ object: torch.as_tensor(this) is a zero-copy int8 view that
    keeps this object alive; the memory goes back to the library when the last such tensor is gone."""

    def __init__(self, lib, ptr, shape):
        self._lib, self._ptr = lib, ptr
        self.__cuda_array_interface__ = {"shape": shape, "typestr": "|i1", "data": (ptr, False), "version": 2, "strides": None}

    def __del__(self):
        try:
            if self._ptr:
                self._lib.sgk_ring_free(ctypes.c_void_p(self._ptr))
                self._ptr = None
        except Exception:  # interpreter shutdown: the process's memory goes with it
            pass


class BatchedGridworldEnv:
    """N independent grid instances stepped in lockstep by hand-written HIP kernels on one MI355X.

    step()/reset() keep the reference's return shape `(state, reward, done, info)` with a leading env axis;
    everything returned is a torch tensor VIEW of library memory in HBM (valid until the next call that
    writes it; `.clone()` to keep).

    Streams (`stream=`): "torch" (default) -- the library enqueues its kernels on torch's CURRENT stream of the device, looked up
    again at every call (a `with torch.cuda.stream(...)` block or a graph capture is followed), so env steps and torch ops are
    ordered by the stream itself: the per-step drop-in sequence act_explore / step / learn / reset_done costs 16-28 us per lockstep
    step from Python. "own" -- the handle keeps a private stream and every call that takes or returns tensors orders itself against
    torch's current stream with an event record + wait each way (~12 us a pair on ROCm: 88-103 us for the same four calls,
    EXPERIMENTS R5.13); for callers that want the env to run beside torch work, and for the single-env step server.
    `bind_torch_stream(stream)` pins the handle to one given torch stream; `bind_torch_stream()` returns to following;
    `use_own_stream()` to the private stream. The fused entry points (agent.rollout, step_random, rollout_random_stream,
    policy_rollout) make one call per rollout and do not care.

    Arguments are checked with exceptions, not asserts (`python -O` keeps them): a tensor of the wrong size, dtype or device is a
    ValueError, never a pointer handed to a kernel.
    """

    def __init__(self, name, n_envs, device=0, seed=0, env_index_base=0, layout="compact", host_visible=False, stream="torch"):
        if name not in ENV_IDS:
            raise KeyError("unknown or out-of-scope env %r; available: %s" % (name, sorted(ENV_IDS)))
        self.name = name
        self.n_envs = int(n_envs)
        self._h = _Handle(ENV_IDS[name], self.n_envs, int(device), int(seed or 0), int(env_index_base),
                          {"pitched": _lib.LAYOUT_PITCHED, "compact": _lib.LAYOUT_COMPACT}[layout]
                          | (_lib.MEM_HOST_VISIBLE if host_visible else 0))
        self.lib = self._h.lib
        info = _lib.SgkInfo()
        _lib.check(self.lib.sgk_get_info(self._h.ptr, ctypes.byref(info)))
        self.info = info
        self.device = info.device
        self.H, self.W, self.n_cells, self.pitch = info.height, info.width, info.n_cells, info.board_pitch
        self.n_states = info.n_states
        scale = ctypes.c_double(1.0)
        _lib.check(self.lib.sgk_reward_scale(self._h.ptr, ctypes.byref(scale)))
        # what one unit of the INTEGER rewards (step records, episode sums, metrics) is worth: 1.0, TomatoWatering 0.02 per tomato
        self.reward_scale = scale.value
        self.action_space = _Space(n=info.n_actions)
        self.observation_space = _Space(shape=(1, self.H, self.W))
        self._env = self  # track_metrics looks for env._env (reference meters.py:67-70)
        self._views = None
        self._finished_bufs = None
        self._tstream = None
        self._events = None
        self._outputs = None
        if stream not in ("torch", "own"):
            raise ValueError("stream must be 'torch' (enqueue on torch's current stream) or 'own' (a private stream), not %r" % (stream,))
        # "follow": torch's current stream, re-read at every call; "pinned": one torch stream (bind_torch_stream(s)); "own": private
        self._mode = "follow" if stream == "torch" else "own"
        self._version = 0  # bumped by every call that changes env states: what a consumer (the replay's fused stores) compares
        self._bound_ptr = None  # the raw stream the library currently enqueues on (follow / pinned); None = its own stream
        self._raw_current = None

    # ---- plumbing -------------------------------------------------------------------------------
    @property
    def handle(self):
        return self._h.ptr

    @property
    def stream_ptr(self):
        """The HIP stream the library enqueues on, as an integer (0 = the device's NULL stream)."""
        return self.lib.sgk_get_stream(self._h.ptr) or 0

    @property
    def _bound(self):
        """True when the library enqueues on a torch stream (no cross-stream events needed)."""
        return self._mode != "own"

    def _current_raw(self):
        """torch's current stream of this device as a raw hipStream_t value (0 = the NULL stream)."""
        fn = self._raw_current
        if fn is None:
            import torch

            raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)  # an int straight from the C extension: ~0.2 us
            if raw is not None:
                fn = lambda d=self.device: raw(d)  # noqa: E731
            else:
                fn = lambda d=self.device: torch.cuda.current_stream(d).cuda_stream  # noqa: E731
            self._raw_current = fn
        return fn()

    def _set_raw(self, ptr):
        if ptr == self._bound_ptr:
            return
        if ptr == 0:  # torch's default stream IS the NULL stream; a NULL argument to sgk_set_stream would mean "own stream"
            _lib.check(self.lib.sgk_use_default_stream(self._h.ptr))
        else:
            _lib.check(self.lib.sgk_set_stream(self._h.ptr, ctypes.c_void_p(ptr)))
        self._bound_ptr = ptr
        self._tstream = None

    def _follow(self):
        """In "torch" mode: make the library enqueue on whatever torch's current stream is NOW. Every method that launches or waits
        calls this first (one C call and an integer compare when nothing changed)."""
        if self._mode == "follow":
            self._set_raw(self._current_raw())

    def torch_stream(self):
        """The stream the library enqueues on, as a torch stream: torch's current stream when following it, the pinned stream, or
        the library's own HIP stream as a torch.cuda.ExternalStream. torch does not own the latter; the library never destroys it
        either (streams of closed envs wait in a pool for the next env of the device), so torch-side objects that remember it --
        pinned host tensors copied on it record an event there when they are freed -- stay valid after close()."""
        import torch

        if self._mode == "follow":
            self._follow()
            return torch.cuda.current_stream(self.device)
        if self._tstream is None:
            self._tstream = torch.cuda.ExternalStream(self.stream_ptr, device="cuda:%d" % self.device)
        return self._tstream

    def bind_torch_stream(self, stream=None):
        """Enqueue the library's kernels on a torch stream so that env steps and torch ops (the Q-network) are ordered by the
        stream itself, with no cross-stream events. With a stream: pinned to it until told otherwise. Without: follow torch's
        current stream from now on (the constructor's default)."""
        if stream is None:
            self._mode = "follow"
            self._follow()
            return
        self._mode = "pinned"
        self._set_raw(int(stream.cuda_stream))
        self._tstream = stream

    def use_own_stream(self):
        """Back to the handle's private stream; calls that take or return tensors order themselves against torch's current stream
        with events."""
        _lib.check(self.lib.sgk_set_stream(self._h.ptr, None))
        self._mode, self._bound_ptr, self._tstream = "own", None, None

    def synchronize(self):
        self._follow()
        _lib.check(self.lib.sgk_synchronize(self._h.ptr))

    def close(self):
        self._views = None
        self._outputs = None
        self._h.close()

    def _device_views(self):
        if self._views is None:
            p, pitch = ctypes.c_void_p(), ctypes.c_int64()
            _lib.check(self.lib.sgk_boards_dev(self._h.ptr, ctypes.byref(p), ctypes.byref(pitch)))
            boards = _view(self._h, self.device, p.value, (self.n_envs, 1, self.H, self.W), "int8",
                           (pitch.value, pitch.value, self.W, 1))
            r = ctypes.c_void_p()
            _lib.check(self.lib.sgk_step_records_dev(self._h.ptr, ctypes.byref(r)))
            rec = _view(self._h, self.device, r.value, (self.n_envs, 4), "int8")
            m = ctypes.c_void_p()
            _lib.check(self.lib.sgk_metrics_dev(self._h.ptr, ctypes.byref(m)))
            metrics = _view(self._h, self.device, m.value, (_lib.METRICS_LEN,), "int64")
            a, b, c = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
            _lib.check(self.lib.sgk_episode_arrays_dev(self._h.ptr, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
            self._views = {
                "boards": boards, "rec": rec, "metrics": metrics,
                "last_return": _view(self._h, self.device, a.value, (self.n_envs,), "int32"),
                "last_performance": _view(self._h, self.device, b.value, (self.n_envs,), "int32"),
                "n_episodes": _view(self._h, self.device, c.value, (self.n_envs,), "int32"),
            }
        return self._views

    def _sync_torch_to_lib(self):
        """Make the library's stream wait for work queued on torch's current stream (e.g. the policy net): nothing to do when the
        library enqueues on that very stream."""
        if self._mode == "follow":
            self._set_raw(self._current_raw())
            return
        if self._mode == "pinned":
            return
        import torch

        _lib.check(self.lib.sgk_stream_wait(self._h.ptr, ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))

    def _sync_lib_to_torch(self):
        if self._mode != "own":
            return
        import torch

        _lib.check(self.lib.sgk_stream_signal(self._h.ptr, ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))

    def _check(self, t, what, numel=None, shape=None, dtypes=None, contiguous=True):
        """A tensor argument that a kernel will read or write through its raw pointer: on THIS device, of the expected dtype and
        size, dense. ValueError otherwise (not assert: `python -O` must not turn a wrong tensor into a wild pointer). On the per-step
        path this runs several times per lockstep step: the passing case is a handful of attribute reads (dtype names are resolved
        to torch dtypes once)."""
        torch = _TORCH[0]
        if torch is None:
            import torch

            _TORCH[0] = torch
        if not isinstance(t, torch.Tensor):
            raise ValueError("%s must be a torch tensor on cuda:%d, not %s" % (what, self.device, type(t).__name__))
        dev = t.device
        if dev.type != "cuda" or dev.index != self.device:
            raise ValueError("%s lives on %s; this env's kernels run on cuda:%d" % (what, t.device, self.device))
        if dtypes is not None:
            allowed = _DTYPES.get(dtypes)
            if allowed is None:
                allowed = _DTYPES[dtypes] = frozenset(getattr(torch, d) for d in dtypes)
            if t.dtype not in allowed:
                raise ValueError("%s has dtype %s, expected %s" % (what, t.dtype, " or ".join(dtypes)))
        if shape is not None and t.shape != shape and tuple(t.shape) != tuple(shape):
            raise ValueError("%s has shape %s, expected %s" % (what, tuple(t.shape), tuple(shape)))
        if numel is not None and t.numel() != numel:
            raise ValueError("%s has %d elements, expected %d" % (what, t.numel(), numel))
        if contiguous and not t.is_contiguous():
            raise ValueError("%s must be contiguous" % what)
        return t

    def _actions_arg(self, actions):
        """uint8 [N] on this device from what the caller passed: a host array / list is uploaded, an integer tensor of another
        width is narrowed (argmax gives int64), everything else is a ValueError."""
        torch = _TORCH[0]
        if torch is None:
            import torch

            _TORCH[0] = torch
        if not isinstance(actions, torch.Tensor):
            actions = torch.as_tensor(np.asarray(actions, dtype=np.uint8), device="cuda:%d" % self.device)
        if actions.dtype != torch.uint8:
            if actions.dtype.is_floating_point or actions.dtype.is_complex or actions.dtype == torch.bool:
                raise ValueError("actions have dtype %s, expected uint8 (or another integer type)" % actions.dtype)
            actions = actions.to(torch.uint8)
        if not actions.is_contiguous():
            actions = actions.contiguous()
        return self._check(actions, "actions", numel=self.n_envs, dtypes=("uint8",), contiguous=False)

    def _weights_arg(self, weights):
        """sgk_mlp_weights from the dict of float32 device tensors w1t [cells, H], b1 [H], w2 [H, H], b2 [H], w3t [H, 4], b3 [4]."""
        try:
            h = int(weights["b1"].numel())
        except (KeyError, TypeError, AttributeError):
            raise ValueError("weights must be a dict of float32 device tensors w1t, b1, w2, b2, w3t, b3") from None
        shapes = {"w1t": (self.n_cells, h), "b1": (h,), "w2": (h, h), "b2": (h,), "w3t": (h, 4), "b3": (4,)}
        for k, shape in shapes.items():
            if k not in weights:
                raise ValueError("weights lack %r" % k)
            self._check(weights[k], "weights[%r]" % k, shape=shape, dtypes=("float32",))
        return _lib.SgkMlpWeights(*(ctypes.c_void_p(weights[k].data_ptr()) for k in ("w1t", "b1", "w2", "b2", "w3t", "b3")), h)

    def _scalar_arg(self, v, what, dtype):
        """(pointer or None, scalar): a 1-element device tensor is read by the launch itself (graph replays), a number is passed."""
        import torch

        if isinstance(v, torch.Tensor):
            self._check(v, what, numel=1, dtypes=(dtype,))
            return ctypes.c_void_p(v.data_ptr()), 0
        return None, v

    def _out_actions(self, out):
        import torch

        if out is None:
            return torch.empty(self.n_envs, dtype=torch.uint8, device="cuda:%d" % self.device)
        return self._check(out, "out", numel=self.n_envs, dtypes=("uint8",))

    # ---- gym-shaped API -------------------------------------------------------------------------
    def seed(self, seed=None):
        """env.seed(seed) (reference train.py:52): the envs are deterministic; this re-keys the counter RNG that drives
        step_random / exploration draws from now on."""
        self._follow()
        if seed is not None:
            _lib.check(self.lib.sgk_set_seed(self._h.ptr, int(seed) & (2**64 - 1)))
        return [seed]

    def boards(self):
        """int8 observation cells, torch view [N, 1, H, W] over HBM (strided when rows are padded)."""
        return self._device_views()["boards"]

    def reset(self, mask=None):
        self._version += 1
        if mask is None:
            self._follow()
            _lib.check(self.lib.sgk_reset(self._h.ptr, None))
        else:
            self._check(mask, "mask", numel=self.n_envs, dtypes=("uint8", "bool"))
            self._sync_torch_to_lib()
            _lib.check(self.lib.sgk_reset(self._h.ptr, ctypes.c_void_p(mask.data_ptr())))
        self._sync_lib_to_torch()
        return self.boards()

    def reset_done(self):
        self._follow()
        self._version += 1
        _lib.check(self.lib.sgk_reset_done(self._h.ptr))
        self._sync_lib_to_torch()
        return self.boards()

    def _step_outputs(self):
        if self._outputs is None:  # the views alias fixed library buffers: build them once
            v = self._device_views()
            rec = v["rec"]
            reward, hidden, done, actual = rec[:, 0], rec[:, 1], rec[:, 2], rec[:, 3]
            info = {"hidden_reward": hidden, "observed_reward": reward,
                    "extra_observations": {"actual_actions": actual}}
            self._outputs = (v["boards"], reward, done, info)
        return self._outputs

    def step(self, actions, auto_reset=False, write_boards=True):
        """actions: torch uint8 tensor [N] on this GPU (another integer dtype is narrowed; a host array is uploaded)."""
        actions = self._actions_arg(actions)
        flags = (_lib.F_AUTO_RESET if auto_reset else 0) | (0 if write_boards else _lib.F_NO_BOARDS)
        self._sync_torch_to_lib()
        self._version += 1
        _lib.check(self.lib.sgk_step(self._h.ptr, ctypes.c_void_p(actions.data_ptr()), flags))
        self._sync_lib_to_torch()
        return self._step_outputs()

    def step_repeat(self, actions, n_steps, auto_reset=True, write_boards=True):
        """SingleActionAgent over the batch (reference dummy.py:19-30): env i repeats actions[i] for n_steps steps."""
        actions = self._actions_arg(actions)
        flags = (_lib.F_AUTO_RESET if auto_reset else 0) | (0 if write_boards else _lib.F_NO_BOARDS)
        self._sync_torch_to_lib()
        self._version += 1
        _lib.check(self.lib.sgk_step_repeat(self._h.ptr, ctypes.c_void_p(actions.data_ptr()), int(n_steps), flags))
        self._sync_lib_to_torch()
        return self._step_outputs()

    def step_store(self, actions, slice_index, rings, cheat=False, slice_dev=None, write_boards=True):
        """env.step(actions) (no auto-reset) and ReplayBuffer.add's second half in ONE launch (sgk_step_store): the successor boards,
        action, reward (hidden + executed action under cheat) and terminal flag of every env go into slice `slice_index` of `rings` =
        (successors int8 [S, N, cells], actions uint8 [S, N], rewards int8 [S, N], terminals bool / uint8 [S, N]). slice_dev: a
        1-element int64 device tensor added to slice_index modulo S by the launch itself (graph replays)."""
        actions = self._actions_arg(actions)
        succ, ract, rrew, rterm = rings
        S_ = int(succ.shape[0])
        self._check(succ, "successors ring", shape=(S_, self.n_envs, self.n_cells), dtypes=("int8",))
        self._check(ract, "actions ring", shape=(S_, self.n_envs), dtypes=("uint8",))
        self._check(rrew, "rewards ring", shape=(S_, self.n_envs), dtypes=("int8",))
        self._check(rterm, "terminals ring", shape=(S_, self.n_envs), dtypes=("bool", "uint8"))
        sd = None
        if slice_dev is not None:
            sd = ctypes.c_void_p(self._check(slice_dev, "slice_dev", numel=1, dtypes=("int64",)).data_ptr())
        ptr = lambda x: ctypes.c_void_p(x.data_ptr())  # noqa: E731
        self._version += 1
        self._sync_torch_to_lib()
        _lib.check(self.lib.sgk_step_store(self._h.ptr, ptr(actions), 0 if write_boards else _lib.F_NO_BOARDS, int(bool(cheat)),
                                           int(slice_index), sd, S_, ptr(succ), ptr(ract), ptr(rrew), ptr(rterm)))
        self._sync_lib_to_torch()
        return self._step_outputs()

    def reset_done_store(self, states_ring, slice_index, slice_dev=None, write_boards=True):
        """reset_done() and ReplayBuffer.add's first half for the NEXT step in ONE launch (sgk_reset_done_store): after the reset every
        env's board goes into slice `slice_index` (+ *slice_dev, modulo the ring) of states_ring int8 [S, N, cells]."""
        S_ = int(states_ring.shape[0])
        self._check(states_ring, "states ring", shape=(S_, self.n_envs, self.n_cells), dtypes=("int8",))
        sd = None
        if slice_dev is not None:
            sd = ctypes.c_void_p(self._check(slice_dev, "slice_dev", numel=1, dtypes=("int64",)).data_ptr())
        self._version += 1
        self._sync_torch_to_lib()
        _lib.check(self.lib.sgk_reset_done_store(self._h.ptr, 0 if write_boards else _lib.F_NO_BOARDS, int(slice_index), sd, S_,
                                                 ctypes.c_void_p(states_ring.data_ptr())))
        self._sync_lib_to_torch()
        return self.boards()

    def step_random(self, n_steps=1, auto_reset=True, fused=False, write_boards=True):
        """n_steps lockstep steps with RandomAgent-style actions from the counter RNG (no torch sync: pure library work).
        fused=False: one launch per step (hipGraph replay); fused=True: ONE launch, outputs materialised after the last step
        only; fused="stream": ONE launch with every step's boards and records materialised (rollout_random_stream)."""
        self._follow()
        flags = (_lib.F_AUTO_RESET if auto_reset else 0) | (0 if write_boards else _lib.F_NO_BOARDS)
        self._version += 1
        if fused == "stream":
            _lib.check(self.lib.sgk_rollout_random_stream(self._h.ptr, int(n_steps), flags, None, None, 1, 0))
            return self._step_outputs()
        fn = self.lib.sgk_rollout_random if fused else self.lib.sgk_step_random
        _lib.check(fn(self._h.ptr, int(n_steps), flags))
        return self._step_outputs()

    def rollout_random_stream(self, n_steps, boards=None, recs=None, first_slice=0, auto_reset=True, layout="slice"):
        """n_steps random-action lockstep steps in ONE launch, every step's successor boards / step records kept: into the
        trajectory rings `boards` int8 [ring, N, n_cells] and / or `recs` int8 [ring, N, 4] (device tensors; step k goes to
        slice (first_slice + k) % ring) -- the batched dqn_warmup (reference warmup.py:14-21) --, or, with neither, into the
        env's own buffers. layout="tile": the rings are tile-major, boards [n_tiles, ring, 64, n_cells] / recs
        [n_tiles, ring, 64, 4] with n_tiles = ceil(N / 64) (`ring_slices` re-orders one to [ring, N, ...]): one contiguous
        run per wave and launch -- the HBM write rate of the env's own buffers instead of the slice-major layout's."""
        if layout not in ("slice", "tile"):
            raise ValueError("layout must be 'slice' or 'tile'")
        ring = 1
        n_tiles = (self.n_envs + 63) // 64
        for t, tail, what in ((boards, self.n_cells, "boards ring"), (recs, 4, "recs ring")):
            if t is not None:
                if t.dim() != (4 if layout == "tile" else 3):
                    raise ValueError("%s must have %d dimensions" % (what, 4 if layout == "tile" else 3))
                want = (n_tiles, int(t.shape[1]), 64, tail) if layout == "tile" else (int(t.shape[0]), self.n_envs, tail)
                self._check(t, what, shape=want, dtypes=("int8",))
                ring = int(t.shape[1] if layout == "tile" else t.shape[0])
        if boards is not None and recs is not None and boards.shape[1 if layout == "tile" else 0] != recs.shape[1 if layout == "tile" else 0]:
            raise ValueError("the boards ring and the recs ring must have the same number of slices")
        ptr = lambda x: None if x is None else ctypes.c_void_p(x.data_ptr())  # noqa: E731
        flags = (_lib.F_AUTO_RESET if auto_reset else 0) | (_lib.F_RING_TILE_MAJOR if layout == "tile" else 0)
        self._sync_torch_to_lib()
        self._version += 1
        _lib.check(self.lib.sgk_rollout_random_stream(self._h.ptr, int(n_steps), flags, ptr(boards), ptr(recs), ring, int(first_slice)))
        self._sync_lib_to_torch()
        return self._step_outputs()

    def probe_trajectory_ring(self, boards=None, recs=None, layout="slice"):
        """Microseconds per slice a store-bound streamed rollout needs into THESE rings (sgk_ring_probe: the streamed kernel's
        stores and nothing else over every slice; the rings hold zeros afterwards)."""
        if layout not in ("slice", "tile") or (boards is None and recs is None):
            raise ValueError("probe_trajectory_ring needs layout 'slice' or 'tile' and at least one ring")
        n_tiles = (self.n_envs + 63) // 64
        for t, tail, what in ((boards, self.n_cells, "boards ring"), (recs, 4, "recs ring")):
            if t is not None:
                if t.dim() != (4 if layout == "tile" else 3):
                    raise ValueError("%s must have %d dimensions" % (what, 4 if layout == "tile" else 3))
                self._check(t, what, dtypes=("int8",),
                            shape=(n_tiles, int(t.shape[1]), 64, tail) if layout == "tile" else (int(t.shape[0]), self.n_envs, tail))
        ring = int((boards if boards is not None else recs).shape[1 if layout == "tile" else 0])
        ptr = lambda x: None if x is None else ctypes.c_void_p(x.data_ptr())  # noqa: E731
        us = ctypes.c_double(0.0)
        self._sync_torch_to_lib()
        _lib.check(self.lib.sgk_ring_probe(self._h.ptr, ptr(boards), ptr(recs), ring, _lib.F_RING_TILE_MAJOR if layout == "tile" else 0,
                                           ctypes.byref(us)))
        return float(us.value)

    def _ring_tensor(self, shape, backing):
        """An int8 device tensor of `shape`: backing="ring" -> memory from sgk_ring_alloc (HIP virtual memory management, 256 MiB
        physical chunks), handed to torch through __cuda_array_interface__ and returned to the library when the tensor dies;
        backing="torch" -> torch.empty."""
        import torch

        dev = "cuda:%d" % self.device
        nbytes = 1
        for d in shape:
            nbytes *= int(d)
        if backing == "torch" or nbytes == 0:
            return torch.empty(shape, dtype=torch.int8, device=dev)
        ptr = ctypes.c_void_p()
        _lib.check(self.lib.sgk_ring_alloc(self.device, nbytes, ctypes.byref(ptr)))
        t = torch.as_tensor(_RingMemory(self.lib, ptr.value, tuple(int(d) for d in shape)), device=dev)
        if t.data_ptr() != ptr.value or t.dtype != torch.int8:
            raise RuntimeError("torch did not adopt the ring memory in place")
        return t

    def alloc_trajectory_ring(self, slices, candidates=1, layout="slice", with_boards=True, with_recs=True, min_bytes=1 << 30,
                              spread_gib=16, backing="ring"):
        """Trajectory rings for rollout_random_stream -- boards int8 [slices, N, n_cells], recs int8 [slices, N, 4] (tile-major:
        [n_tiles, slices, 64, ...]) -- placed where they can be written fast. The rate at which a persistent kernel writes a
        multi-GB ring depends on how the ring's physical memory is made up: hipMalloc blocks (torch.empty) of one
        process measure 4.6-4.9 us per step at 1 M BoatRace envs / 3 GB or 5.6-6.1, for the block's lifetime; memory mapped from
        256 MiB physical chunks through HIP's virtual-memory management (sgk_ring_alloc) 4.5-4.8 every time. backing="ring"
        (default) uses the latter, backing="torch" plain torch.empty.
        `candidates` > 1 additionally allocates that many pairs side by side (a `spread_gib` spacer between them while the device has
        room), times each with the store-only probe (sgk_ring_probe) and keeps the fastest -- what one can do about torch.empty
        blocks; rings below `min_bytes` take one candidate.
        Returns (boards, recs, info); info["candidates_us"] lists every candidate's probe time in allocation order (one entry,
        not probed = nan, when nothing had to be chosen)."""
        import torch

        if layout not in ("slice", "tile") or not (with_boards or with_recs) or slices < 1 or backing not in ("ring", "torch"):
            raise ValueError("alloc_trajectory_ring: layout 'slice'|'tile', backing 'ring'|'torch', slices >= 1, boards and / or recs")
        n_tiles = (self.n_envs + 63) // 64
        dev = "cuda:%d" % self.device
        total = slices * self.n_envs * ((self.n_cells if with_boards else 0) + (4 if with_recs else 0))
        probe_ok = self.n_envs >= 64 and (layout == "tile" or (self.n_envs * self.n_cells) % 16 == 0 or not with_boards)
        if total < min_bytes or not probe_ok:
            candidates = 1

        def make():
            shape = (lambda tail: (n_tiles, slices, 64, tail)) if layout == "tile" else (lambda tail: (slices, self.n_envs, tail))
            b = self._ring_tensor(shape(self.n_cells), backing) if with_boards else None
            r = self._ring_tensor(shape(4), backing) if with_recs else None
            return b, r

        if candidates <= 1:
            b, r = make()
            return b, r, {"candidates_us": [float("nan")], "chosen": 0, "layout": layout, "bytes": total, "backing": backing}
        # fast and slow placements come in runs of 10-20 GB of consecutively allocated memory (profiles/r03/ring_alloc_map.log):
        # candidates allocated back to back would share their fate, so a spacer block is allocated between them -- held until
        # the choice is made -- while the device has the room
        held, times, spacers = [], [], []
        for i in range(int(candidates)):
            if i and spread_gib > 0:
                free_b, _ = torch.cuda.mem_get_info(self.device)
                if free_b > (int(spread_gib) << 30) + 2 * total + (8 << 30):
                    spacers.append(torch.empty(int(spread_gib) << 30, dtype=torch.int8, device=dev))
            b, r = make()
            held.append((b, r))
            times.append(self.probe_trajectory_ring(b, r, layout))
        del spacers
        best = min(range(len(held)), key=lambda i: times[i])
        boards, recs = held[best]
        del held, b, r
        torch.cuda.empty_cache()  # the losing candidates go back to the driver, not into torch's cache
        return boards, recs, {"candidates_us": times, "chosen": best, "layout": layout, "bytes": total, "backing": backing}

    def ring_slices(self, ring_tensor):
        """A tile-major trajectory ring [n_tiles, ring, 64, X] re-ordered to the slice-major form [ring, N, X] (a copy)."""
        n_tiles, ring, _, x = ring_tensor.shape
        return ring_tensor.permute(1, 0, 2, 3).reshape(ring, n_tiles * 64, x)[:, : self.n_envs]

    def prepare_step_random(self, n_steps, auto_reset=True, write_boards=True):
        """Build the hipGraph step_random(n_steps, ...) replays without stepping (sgk_step_random_prepare): callers that time
        a region prepare every chunk size they will use first."""
        self._follow()
        flags = (_lib.F_AUTO_RESET if auto_reset else 0) | (0 if write_boards else _lib.F_NO_BOARDS)
        _lib.check(self.lib.sgk_step_random_prepare(self._h.ptr, int(n_steps), flags))

    def epsilon_greedy(self, scores, epsilon, draw_index, out=None):
        """DeepQAgent.act_explore for every env (reference value.py:94-111): scores float32 [N, 4] -> uint8 actions [N].
        `epsilon` / `draw_index` may be 1-element device tensors (float64 / int64): the launch then reads them from HBM."""
        import torch

        scores = self._check(scores.contiguous() if isinstance(scores, torch.Tensor) else scores, "scores", shape=(self.n_envs, 4),
                             dtypes=("float32",))
        out = self._out_actions(out)
        eps_p, epsilon = self._scalar_arg(epsilon, "epsilon", "float64")
        draw_p, draw_index = self._scalar_arg(draw_index, "draw_index", "int64")
        self._sync_torch_to_lib()
        _lib.check(self.lib.sgk_epsilon_greedy_ex(self._h.ptr, ctypes.c_void_p(scores.data_ptr()), float(epsilon),
                                                  int(draw_index), eps_p, draw_p, ctypes.c_void_p(out.data_ptr())))
        self._sync_lib_to_torch()
        return out

    def policy_act(self, weights, epsilon, draw_index, out=None, scores_out=None):
        """Q-network forward (Linear-ReLU-Linear-ReLU-Linear, n_hidden 100) + act_explore for every env in one HIP launch,
        straight from the int8 boards. `weights`: dict of contiguous float32 device tensors w1t [cells,100], b1, w2
        [100,100], b2, w3t [100,4], b3. epsilon / draw_index: scalars or 1-element device tensors (float64 / int64)."""
        out = self._out_actions(out)
        w = self._weights_arg(weights)
        eps_p, epsilon = self._scalar_arg(epsilon, "epsilon", "float64")
        draw_p, draw_index = self._scalar_arg(draw_index, "draw_index", "int64")
        sp = None if scores_out is None else ctypes.c_void_p(self._check(scores_out, "scores_out", shape=(self.n_envs, 4),
                                                                         dtypes=("float32",)).data_ptr())
        self._sync_torch_to_lib()
        _lib.check(self.lib.sgk_policy_act(self._h.ptr, ctypes.byref(w), float(epsilon), int(draw_index), eps_p, draw_p,
                                           ctypes.c_void_p(out.data_ptr()), sp))
        self._sync_lib_to_torch()
        return out

    def _convq_weights(self, weights, n_channels, n_layers):
        C = int(n_channels)
        shapes = {"w1": (C, 1, 3, 3), "b1": (C,), "w2": (C, C, 3, 3), "b2": (C,), "wb": (C, 1, 1, 1), "bb": (C,), "wh": (C, C, 3, 3),
                  "bh": (C,), "wl": (4, C * self.n_cells), "bl": (4,)}
        for k, shape in shapes.items():
            if k not in weights:
                raise ValueError("weights lack %r" % k)
            self._check(weights[k], "weights[%r]" % k, shape=shape, dtypes=("float32",))
        return _lib.SgkConvQWeights(*(ctypes.c_void_p(weights[k].data_ptr()) for k in ("w1", "b1", "w2", "b2", "wb", "bb", "wh", "bh", "wl", "bl")),
                                    C, int(n_layers))

    def convq_act(self, weights, epsilon, draw_index, n_channels, n_layers=2, out=None, scores_out=None):
        """A conv Q-body's forward on every env's board + DeepQAgent.act_explore in ONE launch (sgk_convq_act; a labelled non-parity
        option: the reference's DeepQAgent is an MLP). `weights`: dict of contiguous float32 device tensors in torch's layouts -- w1
        [C,1,3,3], b1, w2 [C,C,3,3], b2, wb [C,1,1,1], bb, wh [C,C,3,3], bh, wl [4, C * cells], bl. epsilon / draw_index: scalars or
        1-element device tensors (float64 / int64)."""
        w = self._convq_weights(weights, n_channels, n_layers)
        out = self._out_actions(out)
        eps_p, epsilon = self._scalar_arg(epsilon, "epsilon", "float64")
        draw_p, draw_index = self._scalar_arg(draw_index, "draw_index", "int64")
        sp = None if scores_out is None else ctypes.c_void_p(self._check(scores_out, "scores_out", shape=(self.n_envs, 4),
                                                                         dtypes=("float32",)).data_ptr())
        self._sync_torch_to_lib()
        _lib.check(self.lib.sgk_convq_act(self._h.ptr, ctypes.byref(w), float(epsilon), int(draw_index), eps_p, draw_p,
                                          ctypes.c_void_p(out.data_ptr()), sp))
        self._sync_lib_to_torch()
        return out

    def convq_sample(self, weights, draw_index, n_channels, n_layers=2, out=None, logits_out=None):
        """PPOCNNAgent's trunk + actor forward (policy_cnn.py:66-74) on every env's board + PPOBaseAgent.act_explore
        (Categorical(logits).sample(), policy_base.py:54-64) in ONE launch (sgk_convq_sample). `weights` as for convq_act with wh / bh
        = actor_cnn and wl / bl = actor_linear; draw_index: scalar or 1-element int64 device tensor."""
        w = self._convq_weights(weights, n_channels, n_layers)
        out = self._out_actions(out)
        draw_p, draw_index = self._scalar_arg(draw_index, "draw_index", "int64")
        lp = None if logits_out is None else ctypes.c_void_p(self._check(logits_out, "logits_out", shape=(self.n_envs, 4),
                                                                         dtypes=("float32",)).data_ptr())
        self._sync_torch_to_lib()
        _lib.check(self.lib.sgk_convq_sample(self._h.ptr, ctypes.byref(w), int(draw_index), draw_p, ctypes.c_void_p(out.data_ptr()), lp))
        self._sync_lib_to_torch()
        return out

    def convq_rollout(self, weights, n_steps, n_channels, mode="sample", epsilon=0.0, draw_index0=0, auto_reset=False, states=None,
                      actions=None, recs=None, mask_finished=False, n_layers=2):
        """n_steps of {conv body forward, action draw, env.step} in ONE HIP launch (sgk_convq_rollout): `weights` as for convq_sample
        / convq_act; mode "sample" = Categorical(logits) (ppo-cnn's gather_rollout), "greedy" = epsilon-greedy with a fixed epsilon.
        Outputs as for policy_rollout."""
        w = self._convq_weights(weights, n_channels, n_layers)
        if mode not in ("greedy", "sample"):
            raise ValueError("mode must be 'greedy' or 'sample'")

        def ptr(t, shape, what, dtype):
            if t is None:
                return None
            return ctypes.c_void_p(self._check(t, what, shape=shape, dtypes=(dtype,)).data_ptr())

        n = self.n_envs
        self._version += 1
        self._sync_torch_to_lib()
        _lib.check(self.lib.sgk_convq_rollout(
            self._h.ptr, ctypes.byref(w), {"greedy": 0, "sample": 1}[mode], float(epsilon), int(draw_index0), int(n_steps),
            (_lib.F_AUTO_RESET if auto_reset else 0) | (_lib.F_MASK_FINISHED if mask_finished else 0),
            ptr(states, (n_steps, n, self.n_cells), "states", "int8"),
            ptr(actions, (n_steps, n), "actions", "uint8"), ptr(recs, (n_steps, n, 4), "recs", "int8")))
        self._sync_lib_to_torch()

    def categorical_sample(self, logits, draw_index, out=None):
        """PPOBaseAgent.act_explore for every env (reference policy_base.py:54-64): logits float32 [N, 4] -> uint8 actions
        [N] drawn from Categorical(logits) with the counter RNG. `draw_index`: int or a 1-element int64 device tensor."""
        import torch

        logits = self._check(logits.contiguous() if isinstance(logits, torch.Tensor) else logits, "logits", shape=(self.n_envs, 4),
                             dtypes=("float32",))
        out = self._out_actions(out)
        draw_p, draw_index = self._scalar_arg(draw_index, "draw_index", "int64")
        self._sync_torch_to_lib()
        _lib.check(self.lib.sgk_categorical_sample(self._h.ptr, ctypes.c_void_p(logits.data_ptr()), int(draw_index), draw_p,
                                                   ctypes.c_void_p(out.data_ptr())))
        self._sync_lib_to_torch()
        return out

    def policy_sample(self, weights, draw_index, out=None, logits_out=None):
        """PPOMLPAgent (default topology) trunk + actor forward and the Categorical draw for every env in one HIP launch,
        straight from the int8 boards. `weights` as for policy_act (w3t / b3 = the actor head)."""
        out = self._out_actions(out)
        w = self._weights_arg(weights)
        draw_p, draw_index = self._scalar_arg(draw_index, "draw_index", "int64")
        lp = None if logits_out is None else ctypes.c_void_p(self._check(logits_out, "logits_out", shape=(self.n_envs, 4),
                                                                         dtypes=("float32",)).data_ptr())
        self._sync_torch_to_lib()
        _lib.check(self.lib.sgk_policy_sample(self._h.ptr, ctypes.byref(w), int(draw_index), draw_p,
                                              ctypes.c_void_p(out.data_ptr()), lp))
        self._sync_lib_to_torch()
        return out

    def policy_rollout(self, weights, n_steps, mode="sample", epsilon=0.0, draw_index0=0, auto_reset=False, states=None,
                       actions=None, recs=None, mask_finished=False):
        """n_steps of {MLP forward, action draw, env.step} in ONE HIP launch (sgk_policy_rollout): `weights` as for
        policy_act / policy_sample; mode "sample" = Categorical(logits) (PPO), "greedy" = epsilon-greedy with a fixed
        epsilon (DeepQ acting with frozen weights). Optional device outputs: states int8 [n_steps, N, cells] (the board
        each action was chosen on), actions uint8 [n_steps, N], recs int8 [n_steps, N, 4]; with mask_finished the states /
        actions entries of an env whose episode is over are zeros (it idles when auto_reset is off)."""
        w = self._weights_arg(weights)
        if mode not in ("greedy", "sample"):
            raise ValueError("mode must be 'greedy' or 'sample'")

        def ptr(t, shape, what, dtype):
            if t is None:
                return None
            return ctypes.c_void_p(self._check(t, what, shape=shape, dtypes=(dtype,)).data_ptr())

        n = self.n_envs
        self._version += 1
        self._sync_torch_to_lib()
        _lib.check(self.lib.sgk_policy_rollout(
            self._h.ptr, ctypes.byref(w), {"greedy": 0, "sample": 1}[mode], float(epsilon), int(draw_index0), int(n_steps),
            (_lib.F_AUTO_RESET if auto_reset else 0) | (_lib.F_MASK_FINISHED if mask_finished else 0),
            ptr(states, (n_steps, n, self.n_cells), "states", "int8"),
            ptr(actions, (n_steps, n), "actions", "uint8"), ptr(recs, (n_steps, n, 4), "recs", "int8")))
        self._sync_lib_to_torch()

    def ppo_epochs(self, learner):
        """All epochs of one PPO learn() call in ONE HIP launch (sgk_ppo_epochs); `learner` is a filled _lib.SgkPpoLearner
        whose device pointers the caller keeps alive."""
        self._sync_torch_to_lib()
        _lib.check(self.lib.sgk_ppo_epochs(self._h.ptr, ctypes.byref(learner)))
        self._sync_lib_to_torch()

    def discounted_returns(self, rewards, discount, lengths=None, out=None):
        """PPOBaseAgent.get_discounted_returns (reference policy_base.py:179-186) for a batch: rewards float32
        [n_trajectories, T] on this GPU (lengths int32 [n_trajectories] optional) -> returns of the same shape, with the
        reference's float32 rounding order (bit-exact)."""
        import torch

        rewards = self._check(rewards.contiguous() if isinstance(rewards, torch.Tensor) else rewards, "rewards", dtypes=("float32",))
        if rewards.dim() != 2:
            raise ValueError("rewards must be [n_trajectories, T]")
        if out is None:
            out = torch.zeros_like(rewards)
        self._check(out, "out", shape=tuple(rewards.shape), dtypes=("float32",))
        lp = None
        if lengths is not None:
            lengths = self._check(lengths, "lengths", numel=rewards.shape[0], contiguous=False).to(torch.int32).contiguous()
            lp = ctypes.c_void_p(lengths.data_ptr())
        self._sync_torch_to_lib()
        _lib.check(self.lib.sgk_discounted_returns(self._h.ptr, ctypes.c_void_p(rewards.data_ptr()), lp,
                                                   ctypes.c_void_p(out.data_ptr()), rewards.shape[0], rewards.shape[1],
                                                   float(discount)))
        self._sync_lib_to_torch()
        return out

    def account_steps(self, n_steps):
        """After replaying an external graph that contains step() launches: advance the host-side counters."""
        self._version += 1
        _lib.check(self.lib.sgk_account_steps(self._h.ptr, int(n_steps)))

    def obs_f32(self, out=None):
        """float32 [N, n_cells] observation for the Q-network (what the reference builds per sample, value.py:161-164)."""
        import torch

        if out is None:
            out = torch.empty((self.n_envs, self.n_cells), dtype=torch.float32, device="cuda:%d" % self.device)
        self._check(out, "out", shape=(self.n_envs, self.n_cells), dtypes=("float32",))
        self._sync_torch_to_lib()
        _lib.check(self.lib.sgk_obs_f32(self._h.ptr, ctypes.c_void_p(out.data_ptr())))
        self._sync_lib_to_torch()
        return out

    def render(self, mode="rgb_array", out=None):
        """uint8 [N, 3, H, W] device tensor: the level's colour map applied to every env's board (HIP kernel)."""
        import torch

        if out is None:
            shape = (self.n_envs, self.H, self.W, 3) if self.info.render_hwc else (self.n_envs, 3, self.H, self.W)
            out = torch.empty(shape, dtype=torch.uint8, device="cuda:%d" % self.device)
        self._check(out, "out", numel=self.n_envs * 3 * self.H * self.W, dtypes=("uint8",))
        self._sync_torch_to_lib()
        _lib.check(self.lib.sgk_render_rgb(self._h.ptr, ctypes.c_void_p(out.data_ptr())))
        self._sync_lib_to_torch()
        return out

    # ---- host copies (synchronising) --------------------------------------------------------------
    def boards_host(self):
        self._follow()
        out = np.empty((self.n_envs, 1, self.H, self.W), dtype=np.int8)
        _lib.check(self.lib.sgk_copy_boards(self._h.ptr, out.ctypes.data))
        return out

    def step_records_host(self):
        self._follow()
        out = np.empty((self.n_envs, 4), dtype=np.int8)
        _lib.check(self.lib.sgk_copy_step_records(self._h.ptr, out.ctypes.data))
        return out

    def episode_state_host(self):
        self._follow()
        n = self.n_envs
        ret, hid, frame = (np.empty(n, dtype=np.int32) for _ in range(3))
        over, cell, box = (np.empty(n, dtype=np.uint8) for _ in range(3))
        _lib.check(self.lib.sgk_copy_episode_state(self._h.ptr, ret.ctypes.data, hid.ctypes.data, frame.ctypes.data,
                                                   over.ctypes.data, cell.ctypes.data, box.ctypes.data))
        return {"episode_return": ret, "hidden_return": hid, "frame": frame, "over": over, "agent_cell": cell,
                "box_cell": box}

    def last_performance_host(self):
        """last_episode_host()["last_performance"] alone: what get_last_performance() reads (on a host-visible handle a plain read
        of pinned host memory that leaves the resident step server where it is)."""
        self._follow()
        perf = np.empty(self.n_envs, dtype=np.int32)
        _lib.check(self.lib.sgk_copy_last_episode(self._h.ptr, None, perf.ctypes.data, None))
        return perf

    def last_episode_host(self):
        self._follow()
        n = self.n_envs
        a, b, c = (np.empty(n, dtype=np.int32) for _ in range(3))
        _lib.check(self.lib.sgk_copy_last_episode(self._h.ptr, a.ctypes.data, b.ctypes.data, c.ctypes.data))
        return {"last_return": a, "last_performance": b, "n_episodes": c}

    # what track_metrics reads (reference meters.py:76-77), batched
    @property
    def episode_return(self):
        return self.episode_state_host()["episode_return"]

    def get_last_performance(self):
        le = self.last_episode_host()
        return np.where(le["n_episodes"] > 0, le["last_performance"], 0), le["n_episodes"] > 0

    def bandit_policy(self):
        """FriendFoe: environment_data['bandit'] of every env -- float64 [n_envs, 3 bandit types, 2 boxes], the exponentially
        smoothed probability that the agent opens box 0 / box 1 in an episode of that type (kept across episodes)."""
        self._follow()
        out = np.empty((self.n_envs, 3, 2), dtype=np.float64)
        _lib.check(self.lib.sgk_copy_bandit_policy(self._h.ptr, out.ctypes.data))
        return out

    def metrics(self):
        self._follow()
        out = np.zeros(_lib.METRICS_LEN, dtype=np.int64)
        _lib.check(self.lib.sgk_metrics(self._h.ptr, out.ctypes.data))
        return out

    def metrics_device(self):
        return self._device_views()["metrics"]

    def metrics_reset(self):
        self._follow()
        _lib.check(self.lib.sgk_metrics_reset(self._h.ptr))

    def finished(self):
        """(ids, episode_return, performance) of the envs whose last step ended an episode, ascending ids (device tensors)."""
        import torch

        if self._finished_bufs is None:
            dev = "cuda:%d" % self.device
            self._finished_bufs = tuple(torch.empty(self.n_envs, dtype=torch.int32, device=dev) for _ in range(3))
        ids, ret, perf = self._finished_bufs
        n = ctypes.c_int64(0)
        self._sync_torch_to_lib()
        _lib.check(self.lib.sgk_finished(self._h.ptr, ctypes.c_void_p(ids.data_ptr()), ctypes.c_void_p(ret.data_ptr()),
                                         ctypes.c_void_p(perf.data_ptr()), ctypes.byref(n)))
        k = n.value
        return ids[:k], ret[:k], perf[:k]

    @property
    def lockstep_t(self):
        info = _lib.SgkInfo()
        _lib.check(self.lib.sgk_get_info(self._h.ptr, ctypes.byref(info)))
        return info.lockstep_t


class _SafetyEnvView:
    """`env._env` of the single-env wrapper: SafetyEnvironment members the reference reads (meters.py:76-77, warmup.py:16)."""

    def __init__(self, owner):
        self._o = owner

    @property
    def episode_return(self):
        return self._o._episode_return

    def get_last_performance(self):
        return self._o._last_performance


class GridworldEnv:
    """Single env with the safe_grid_gym.GridworldEnv surface, stepped on the GPU through the same kernels (N = 1).

    Observations are fresh float32 numpy arrays of shape (1, H, W) the caller may keep (replay buffers and
    Q-table keys do: reference contain.py:17, value.py:34).
    """

    def __init__(self, name, device=0):
        self.use_transitions = name in TRANSITION_ENVS  # observation = [last board, board], shape (2, H, W)
        # (a stream of its own: the step server is a kernel that stays resident on the handle's stream between steps)
        self._b = BatchedGridworldEnv(TRANSITION_ENVS.get(name, name), 1, device=device, host_visible=True, layout="pitched", stream="own")
        self.name = name
        self.action_space = self._b.action_space
        self.observation_space = self._b.observation_space
        if self.use_transitions:
            self.observation_space = _Space(shape=(2, self._b.H, self._b.W))
        self._last_board = None
        self._env = _SafetyEnvView(self)
        self._episode_return = 0
        self._last_performance = None
        self._over = False
        self._act = np.zeros(1, dtype=np.uint8)
        self._rec = np.zeros((1, 4), dtype=np.int8)
        self._board = np.zeros((1, self._b.n_cells), dtype=np.int8)
        self._ret = np.zeros(1, dtype=np.int32)
        # TomatoWatering pays REWARD_FACTOR per watered tomato: the kernels carry the counts; the floats the reference consumes are
        # made here with upstream's own expression (count * REWARD_FACTOR) and summed step by step, as SafetyEnvironment does
        self._scale = self._b.reward_scale
        self._hidden_return = 0.0
        self._water_dist = None
        if name == "IslandNavigation-v0":  # the side information "safety": Manhattan distance from each cell to the nearest water cell
            first = self._b.boards_host()[0, 0]
            water = np.argwhere(first == 0)  # value_mapping: water = 0
            cells = np.argwhere(np.ones_like(first, dtype=bool))
            self._water_dist = np.abs(cells[:, None, :] - water[None, :, :]).sum(axis=2).min(axis=1).astype(np.int64)
        # everything env.step touches on every call, looked up once (numpy's .ctypes and the property chain cost microseconds)
        self._no_hidden = name in NO_HIDDEN_REWARD
        self._step_fn = self._b.lib.sgk_step_host
        # (not the handle: it is read per call -- one attribute lookup -- so that step() after close(), or after the wrapped batched
        # env was closed, passes NULL and gets the library's "handle is NULL" instead of a freed pointer)
        self._step_tail = (self._act.ctypes.data, 0, self._rec.ctypes.data, self._board.ctypes.data, self._ret.ctypes.data)
        self._handle_owner = self._b._h
        self._rec_row, self._rec_u8 = self._rec[0], self._rec.view(np.uint8)[0]
        self._board_hw = self._board.reshape(1, self._b.H, self._b.W)
        self._board_flat = self._board[0]
        self._n_actions = self.action_space.n

    def seed(self, seed=None):
        """env.seed(seed) (reference train.py:52): re-keys the counter RNG -- the env's own draws (WhiskyGold's replaced
        actions) come from it."""
        return self._b.seed(seed)

    def close(self):
        self._b.close()

    def reset(self):
        _lib.check(self._b.lib.sgk_reset(self._b.handle, None))  # host-visible memory: no torch views involved
        self._episode_return = 0 if self._scale == 1.0 else 0.0
        self._hidden_return = 0.0
        self._over = False
        board = self._b.boards_host()[0].astype(np.float32)
        if self.use_transitions:  # at reset the "last board" is the board itself
            self._last_board = board
            return np.concatenate([board, board], axis=0)
        return board

    def step(self, action):
        if hasattr(action, "item"):  # np.int64 (value.py:35) or a 1-element tensor (value.py:92 via eval.py:35-36)
            action = action.item()
        action = int(action)
        if not 0 <= action < self._n_actions:  # (the reference's convention for a bad action, dummy.py:27 -- raised, so that -O keeps it)
            raise AssertionError("Not a valid action.")
        self._act[0] = action
        rc = self._step_fn(self._handle_owner.ptr, *self._step_tail)
        if rc:
            _lib.check(rc)
        reward, hidden, done, _ = self._rec_row.tolist()
        done = bool(done)
        actual = int(self._rec_u8[3])
        if self._scale != 1.0:
            if not self._over:
                reward, hidden = reward * self._scale, hidden * self._scale
                self._episode_return += reward
                self._hidden_return += hidden
                if done:
                    self._last_performance = self._hidden_return
            else:
                reward, hidden = 0.0, 0.0
        else:
            self._episode_return = int(self._ret[0])
            if done and not self._over:  # the episode just ended: get_last_performance() now has a value
                self._last_performance = int(self._b.last_performance_host()[0])
        self._over = done
        info = {
            "hidden_reward": None if self._no_hidden else hidden,  # (None: as safe_grid_gym reports an env without a hidden reward)
            "observed_reward": reward,
            "discount": 0.0 if done else 1.0,
            "extra_observations": {"actual_actions": actual},
        }
        state = self._board_hw.astype(np.float32)
        if self.use_transitions:
            state, self._last_board = np.concatenate([self._last_board, state], axis=0), state
        if self._water_dist is not None:  # IslandNavigation's side information: distance of the agent (value 2) to the nearest water cell
            at = np.flatnonzero(self._board_flat == 2)
            info["extra_observations"]["safety"] = 0 if at.size == 0 else int(self._water_dist[at[0]])
        return state, reward, done, info

    def render(self, mode="rgb_array"):
        """(3, H, W) uint8 frame, as the reference stacks them (eval.py:16-31)."""
        return self._b.render()[0].cpu().numpy()


def make(name, n_envs=None, **kwargs):
    """gym.make(name) (reference train.py:51). n_envs=None -> the single-env drop-in; an int -> the batched env."""
    if name in ENV_MAP:
        name = ENV_MAP[name]
    if name in FORK_ONLY_ENVS:
        raise KeyError("env %r exists only in the fork of ai-safety-gridworlds the reference was developed against; nothing in the "
                       "reference, the paper or the public repository describes its rules, so it is not restated here" % name)
    if name not in ENV_IDS and name not in TRANSITION_ENVS:
        raise KeyError("env %r is not implemented (hot-path scope: %s)" % (name, sorted(ENV_IDS) + sorted(TRANSITION_ENVS)))
    if n_envs is None:
        return GridworldEnv(name, **kwargs)
    if name in TRANSITION_ENVS:
        # batched: the same kernels; the two-board observation is what a 2-slice trajectory ring holds
        # (rollout_random_stream(..., boards=ring) with ring.shape[0] == 2: slice k % 2 = board after step k)
        name = TRANSITION_ENVS[name]
    return BatchedGridworldEnv(name, n_envs, **kwargs)
