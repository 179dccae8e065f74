"""Episode meters: host mirror of the reference's AverageMeter / make_meters / track_metrics
(reference safe_grid_agents/common/utils/meters.py:9-108) plus the batched form that is filled from the
GPU metrics vector (include/sgk.h SGK_M_*), which is also what the multi-GPU all-reduce carries.
"""
import bisect
import math

import numpy as np

METER_NAMES = ("returns", "safeties", "margins", "margins_support")


class AverageMeter:
    """Running value / sum / count / mean / max, optionally with a sorted history for quantiles.

    Field names and update rules follow reference meters.py:16-49 (val, avg, sum, count, _max, _history).
    """

    def __init__(self, include_history=False):
        self.include_history = include_history
        self._history = None
        self.reset(reset_history=True)

    def reset(self, reset_history=False):
        self.val = 0
        self.avg = 0
        self.sum = 0
        self.count = 0
        self._max = -math.inf
        if reset_history:
            self._history = [] if self.include_history else None

    def update(self, val, n=1):
        self.val = val
        if val > self._max:
            self._max = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count
        if self._history is not None:
            for _ in range(n):
                bisect.insort(self._history, val)

    def quantile(self, delta):
        if self._history is None:
            raise RuntimeError("Meter instantiated without history.")
        return np.percentile(self._history, q=(1 - delta) * 100)

    @property
    def max(self):
        return self._max


def make_meters(history):
    """Fresh meter set; an existing `returns` meter is carried over (reference meters.py:52-63)."""
    returns = history["returns"] if "returns" in history else AverageMeter(include_history=True)
    return {"returns": returns, "safeties": AverageMeter(), "margins": AverageMeter(), "margins_support": AverageMeter()}


def track_metrics(history, env, eval=False, write=True):
    """Record the episode that just ended (reference meters.py:66-108).

    Reads `env._env.episode_return` and `env._env.get_last_performance()` (or the env itself when it has no `_env`),
    updates returns / safeties / margins (= return - safety) / margins_support (margin only when > 0), and writes
    the same tensorboard tags in the same order as the reference.
    """
    inner = env._env if hasattr(env, "_env") else env
    step_id = history["period"] if eval else history["episode"]
    episode_return = inner.episode_return
    history["returns"].update(episode_return)
    safety = inner.get_last_performance()
    margin = None
    if safety is not None:
        margin = episode_return - safety
        history["safeties"].update(safety)
        history["margins"].update(margin)
        if margin > 0:
            history["margins_support"].update(margin)
    if not write:
        return history
    writer = history["writer"]
    if eval:
        for name in METER_NAMES:
            if name != "returns" and safety is None:
                continue
            meter = history[name]
            writer.add_scalars("Evaluation/" + name, {"avg": meter.avg, "max": meter.max}, step_id)
    else:
        writer.add_scalar("Train/returns", history["returns"].val, step_id)
        if safety is not None:
            writer.add_scalar("Train/safeties", safety, step_id)
            writer.add_scalar("Train/margins", margin, step_id)
            if margin > 0:
                writer.add_scalar("Train/margins_support", margin, step_id)
    return history


class NullWriter:
    """Stand-in for tensorboardX.SummaryWriter (absent from this image): accepts and drops every call the
    reference makes (add_scalar / add_scalars / add_text / add_video / add_histogram)."""

    def __init__(self, log_dir=None):
        self.log_dir = log_dir

    def add_scalar(self, *a, **k):
        pass

    add_scalars = add_text = add_video = add_histogram = add_scalar


class RecordingWriter(NullWriter):
    """Keeps (kind, tag, value, step) tuples; floats as hex so that comparisons are bit-exact."""

    def __init__(self, log_dir=None):
        super().__init__(log_dir)
        self.calls = []

    @staticmethod
    def _num(v):
        if v is None:
            return None
        if isinstance(v, (bool, np.bool_)):
            return bool(v)
        if isinstance(v, (int, np.integer)):
            return int(v)
        if hasattr(v, "detach"):  # a 0-d torch tensor (PPO logs its entropy as one, policy_base.py:117-119)
            v = v.detach()
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


# ---- batched form ---------------------------------------------------------------------------------

# indices of include/sgk.h
M_SUM_RETURN, M_SUM_SAFETY, M_SUM_MARGIN, M_SUM_MARGIN_POS, M_EPISODES, M_MARGIN_POS_COUNT, M_STEPS = range(7)
M_MAX_RETURN, M_MAX_SAFETY, M_MAX_MARGIN, M_MAX_MARGIN_POS = 8, 9, 10, 11
METRICS_LEN = 16
_INT64_MIN = -(2 ** 63)


class BatchMetrics:
    """The four meters' aggregate fields for ALL episodes a batch (or a whole multi-GPU job) finished.

    Built from the int64 metrics vector the kernels accumulate: integer sums and maxima, so the result does not
    depend on how envs are sharded or in which order episodes ended. `meter(name)` gives val-less AverageMeter-like
    records: sum, count, avg (true division, as meters.py:33) and max. `scale` = what one unit of the integer rewards is worth
    (env.reward_scale: 1.0, or TomatoWatering's 0.02 per watered tomato): sums and maxima are reported times it.
    """

    def __init__(self, vec, scale=1.0):
        v = [int(x) for x in np.asarray(vec).reshape(-1)[:METRICS_LEN]]
        self.vec = v
        self.scale = float(scale)
        self.episodes = v[M_EPISODES]
        self.steps = v[M_STEPS]

    def meter(self, name):
        v = self.vec
        total, count, mx = {
            "returns": (v[M_SUM_RETURN], v[M_EPISODES], v[M_MAX_RETURN]),
            "safeties": (v[M_SUM_SAFETY], v[M_EPISODES], v[M_MAX_SAFETY]),
            "margins": (v[M_SUM_MARGIN], v[M_EPISODES], v[M_MAX_MARGIN]),
            "margins_support": (v[M_SUM_MARGIN_POS], v[M_MARGIN_POS_COUNT], v[M_MAX_MARGIN_POS]),
        }[name]
        has_max = bool(count) and mx != _INT64_MIN
        if self.scale != 1.0:
            total, mx = total * self.scale, mx * self.scale
        return {
            "sum": total,
            "count": count,
            "avg": (total / count) if count else 0,
            "max": mx if has_max else -math.inf,
        }

    def as_dict(self):
        return {name: self.meter(name) for name in METER_NAMES}

    def write(self, writer, step, prefix="Evaluation/"):
        """Same tags as track_metrics' eval branch (meters.py:96-106)."""
        for name in METER_NAMES:
            m = self.meter(name)
            if m["count"]:
                writer.add_scalars(prefix + name, {"avg": m["avg"], "max": m["max"]}, step)
