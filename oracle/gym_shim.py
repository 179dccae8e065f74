"""safe_grid_gym.GridworldEnv-shaped single env over the C oracle -- TEST INFRASTRUCTURE ONLY.

Presents exactly the members the reference touches: reset/step/seed/render,
action_space.n, observation_space.shape, and `_env.episode_return` / `_env.get_last_performance()`
(reference meters.py:67-80, warmup.py:16). Used by tests/golden/make_golden.py to drive the
reference's own train() loop, and by the CPU tests as the env behind this repo's host logic.
"""
import numpy as np

from . import oracle as O


class _Space:
    def __init__(self, n=None, shape=None):
        self.n = n
        self.shape = shape


class _SafetyEnvView:
    """What `env._env` exposes (ai_safety_gridworlds SafetyEnvironment members used by the reference)."""

    def __init__(self, owner):
        self._o = owner

    @property
    def episode_return(self):
        if self._o._scale != 1.0:  # float rewards: SafetyEnvironment adds them up step by step
            return self._o._ret
        return int(self._o._b.field("episode_return")[0])

    def get_last_performance(self):
        if self._o._scale != 1.0:
            return self._o._last_perf
        return self._o._b.last_performance(0)


TRANSITION_ENVS = {"TransitionBoatRace-v0": "BoatRace-v0"}  # use_transitions=True: observation = [last board, board]


class OracleGridworldEnv:
    def __init__(self, name):
        self.name = name
        self.use_transitions = name in TRANSITION_ENVS
        self._b = O.EnvBatch(TRANSITION_ENVS.get(name, name), 1)  # reset once at construction, as sgk_create leaves the product's env (reset counter 1)
        self.action_space = _Space(n=4)
        self.observation_space = _Space(shape=(2 if self.use_transitions else 1, self._b.H, self._b.W))
        self._last = None
        # TomatoWatering pays REWARD_FACTOR per watered tomato: the integer engine carries the counts, the floats are made here
        # with upstream's own expression (count * REWARD_FACTOR) and accumulated the way SafetyEnvironment / the_plot do
        self._scale = O.reward_scale(self._b.env_id)
        self._ret, self._hid, self._last_perf = 0.0, 0.0, None
        self._env = _SafetyEnvView(self)
        self.actions_log = []

    def seed(self, seed=None):
        if seed is not None:  # keys the env's own draws (WhiskyGold), as sgk_set_seed does for the product's env
            self._b.set_rng(int(seed) & (2**64 - 1), 0)
        return [seed]

    def _obs(self):
        return self._b.board(0).astype(np.float32)[np.newaxis]

    def reset(self):
        self._b.reset(0)
        self._ret, self._hid = 0.0, 0.0
        obs = self._obs()
        if self.use_transitions:
            self._last = obs
            return np.concatenate([obs, obs], axis=0)
        return obs

    def step(self, action):
        if hasattr(action, "item"):
            action = action.item()
        action = int(action)
        self.actions_log.append(action)
        r, h, d, actual = self._b.step(0, action)
        if self._scale != 1.0:
            r, h = r * self._scale, h * self._scale
            self._ret += r
            self._hid += h
            if d:
                self._last_perf = self._hid
        info = {
            "hidden_reward": h,
            "observed_reward": r,
            "discount": 0.0 if d else 1.0,
            "extra_observations": {"actual_actions": actual},
        }
        if self._b.env_id == 1:
            info["extra_observations"]["safety"] = int(self._b.field("safety")[0])
        if not O.has_hidden_reward(self._b.env_id):
            info["hidden_reward"] = None  # safe_grid_gym reports None for envs that define no hidden reward
        obs = self._obs()
        if self.use_transitions:
            obs, self._last = np.concatenate([self._last, obs], axis=0), obs
        return obs, r, bool(d), info

    def render(self, mode="rgb_array"):
        return self._b.render_rgb(0)
