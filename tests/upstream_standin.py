"""Stand-ins for an upstream checkout, for tests/test_check_upstream.py only: the oracle's own gym shim (a) keyed differently from
the harness's oracle, so that every env-side draw has to be found by the harness's search, and (b) with one rule changed the way
a switch of include/sgk_levels.h would change it, so that the harness has a mismatch to report and to reconcile."""
import importlib
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _default_build_shim():
    """The gym shim over the DEFAULT oracle build, whatever SGK_ORACLE_SO says: when the harness re-runs a level against an
    oracle variant (its reconciliation pass) the stand-in must stay what it was, as a real upstream checkout would. A private
    copy of the `oracle` package, imported while the variable is unset."""
    if "oracle_default" not in sys.modules:
        saved = os.environ.pop("SGK_ORACLE_SO", None)
        try:
            d = os.path.join(_ROOT, "oracle")
            spec = importlib.util.spec_from_file_location("oracle_default", os.path.join(d, "__init__.py"), submodule_search_locations=[d])
            pkg = importlib.util.module_from_spec(spec)
            sys.modules["oracle_default"] = pkg
            spec.loader.exec_module(pkg)
            importlib.import_module("oracle_default.gym_shim").O.lib()  # dlopen now
        finally:
            if saved is not None:
                os.environ["SGK_ORACLE_SO"] = saved
    return sys.modules["oracle_default.gym_shim"].OracleGridworldEnv


OracleGridworldEnv = _default_build_shim()


def rekeyed(name):
    env = OracleGridworldEnv(name)
    env.seed(0xC0FFEE)  # upstream draws from a stream of its own: so does this stand-in
    return env


class _MovementInHidden:
    """BoatRace as SGK_BOAT_MOVEMENT_IN_HIDDEN=1 reads it: the -1 per step is part of the hidden reward as well."""

    def __init__(self, name):
        self._e = OracleGridworldEnv(name)
        self.action_space, self.observation_space = self._e.action_space, self._e.observation_space
        self._extra, self._last_perf = 0, None
        self._env = self

    def reset(self):
        self._extra = 0
        return self._e.reset()

    def step(self, action):
        obs, r, d, info = self._e.step(action)
        info["hidden_reward"] -= 1
        self._extra -= 1
        if d:
            self._last_perf = self._e._env.get_last_performance() + self._extra
        return obs, r, d, info

    @property
    def episode_return(self):
        return self._e._env.episode_return

    def get_last_performance(self):
        return self._last_perf


def boat_movement_in_hidden(name):
    return _MovementInHidden(name) if name == "BoatRace-v0" else OracleGridworldEnv(name)
