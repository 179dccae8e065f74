#!/usr/bin/env python3
"""Upstream pin harness for the env rules (the env transition is PARITY UNPINNED because `safe_grid_gym`,
`ai_safety_gridworlds` and `pycolab` are absent from /root/reference and from this image).

    python tests/golden/check_upstream.py --path <dir with the upstream checkouts> [--levels BoatRace-v0,...]

Build container only; nothing here runs on the GPU box. With a checkout at hand a later session pins SURVEY row A17 in one
command instead of reading 21 switches by hand:

 * imports `gym` + `safe_grid_gym` (or `safe_grid_gym.envs.gridworlds_env.GridworldEnv` directly) from --path, builds each level
   the way the reference does (`gym.make(ENV_MAP[...])`, reference train.py:51-52, then `env.reset()`, train.py:64);
 * replays the action sequences of tests/golden/env_traces.json (the sequences this repo's own reading is frozen on) through
   `env.step(action)` (reference learn.py:69) with `reset()` after `done`;
 * diffs, per step, what the reference consumes -- `reward`, `info["hidden_reward"]`, `done`, the board
   (`state[-1]`), and at episode ends `env._env.episode_return` / `get_last_performance()` (reference meters.py:67-80) --
   against this repo's CPU restatement (oracle/) stepped through the same actions;
 * for every level with mismatches re-runs it against the oracle built under each alternative reading (`-D<switch>=<alt>`,
   the switches of include/sgk_levels.h listed in tests/test_switch_variants.py) and prints the switches that reconcile it.

Levels whose rules draw random numbers (WhiskyGold's replaced actions, the per-episode coins of AbsentSupervisor /
SafeInterruptibility / FriendFoe, TomatoWatering's drying tomatoes) cannot share a random stream with upstream (numpy's global
MT19937 there, a counter RNG here). The harness instead SEARCHES the oracle's draws: at every reset and step it re-keys clones
of the oracle env until one reproduces what upstream shows of its own draw (the board, the executed action) and then compares
the rewards and `done` of that clone; several hidden states that all fit are carried along as candidates (the interruption coin
does not show on the board). A step no draw reproduces is reported as `unsynced`, not as a rule mismatch.

Exit status: 0 = upstream absent (says so) or everything matched; 1 = mismatches; 2 = usage.
"""
import argparse
import ctypes
import importlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))

# the `env_name` strings safe_grid_gym's GridworldEnv takes, for a checkout without gym registration [UPSTREAM -- UNVERIFIED]
DIRECT_NAMES = {"BoatRace-v0": "boat_race", "IslandNavigation-v0": "island_navigation", "SideEffectsSokoban-v0": "side_effects_sokoban",
                "DistributionalShift-v0": "distributional_shift", "WhiskyGold-v0": "whisky_gold", "AbsentSupervisor-v0": "absent_supervisor",
                "SafeInterruptibility-v0": "safe_interruptibility", "ConveyorBelt-v0": "conveyor_belt",
                "TomatoWatering-v0": "tomato_watering", "FriendFoe-v0": "friend_foe"}
# how many re-keyed clones a reset / a step may try before it is called `unsynced` (1 = the level draws nothing there)
RESET_TRIES = {"AbsentSupervisor-v0": 64, "SafeInterruptibility-v0": 64, "FriendFoe-v0": 256}
STEP_TRIES = {"WhiskyGold-v0": 256, "TomatoWatering-v0": 20000}


def upstream_factory(paths, spec=None):
    """name -> env object with the gym duck type the reference uses, or None when upstream cannot be imported."""
    if spec:  # "module:callable" -- a stand-in (the unit test passes the oracle's own gym shim)
        mod, _, fn = spec.partition(":")
        return getattr(importlib.import_module(mod), fn)
    for p in paths:
        for q in [p] + [os.path.join(p, d) for d in sorted(os.listdir(p)) if os.path.isdir(os.path.join(p, d))]:
            if q not in sys.path:
                sys.path.insert(0, q)
    try:
        import gym  # noqa: F401
        import safe_grid_gym  # noqa: F401  (registers the ids of reference parse.py:22-37)

        return lambda name: importlib.import_module("gym").make(name)
    except Exception:
        pass
    try:
        from safe_grid_gym.envs.gridworlds_env import GridworldEnv

        return lambda name: GridworldEnv(env_name=DIRECT_NAMES[name])
    except Exception:
        return None


class OracleSide:
    """The oracle env(s) consistent with everything upstream has shown so far (more than one only while part of the state stays
    hidden: SafeInterruptibility's coin does not show on the board)."""

    def __init__(self, O, name):
        self.O, self.name = O, name
        self.scale = O.reward_scale(O.ENV_IDS[name])
        self.hidden = O.has_hidden_reward(O.ENV_IDS[name])
        self.cands = [O.EnvBatch(name, 1)]
        self.key = 1

    def clone(self, env, rekey=False):
        c = self.O.EnvBatch(self.name, 1, reset=False)
        ctypes.memmove(c.base, env.base, env.rec)
        if rekey:  # a fresh key for the env's own draws: another outcome of the same rules
            self.key += 1
            c.set_rng((0x9E3779B97F4A7C15 * self.key) & (2**64 - 1), 0)
        return c

    def reset(self, up_board):
        """-> (found, board the un-rekeyed oracle shows). Exhausting the tries is evidence, not bad luck: 2^-64 for a coin."""
        tries = RESET_TRIES.get(self.name, 1)
        found, first = [], None
        # part of what a reset draws does not show on the board -- SafeInterruptibility's coin, FriendFoe's box with the reward
        # (drawn only by the neutral bandit) --: every outcome that fits is carried along until a step tells them apart
        want = 2 if self.name in ("SafeInterruptibility-v0", "FriendFoe-v0") else 1
        hidden_state = lambda e: (int(e.field("coin")[0]), int(e.field("ext")[0]))  # noqa: E731
        for base in self.cands:
            for k in range(tries):
                c = self.clone(base, rekey=k > 0)
                c.reset(0)
                if first is None:
                    first = c
                if np.array_equal(c.board(0), up_board) and hidden_state(c) not in {hidden_state(f) for f in found}:
                    found.append(c)
                    if len(found) >= want:
                        break
            if len(found) >= want:
                break
        self.cands = found or [first]
        return bool(found), first.board(0)

    def step(self, action, up):
        """up: what upstream's step showed (board, actual, reward, hidden, done). -> ("full" | "partial" | "none", clone, result):
        full = some draw of the oracle reproduces all of it; partial = board and executed action only (the rewards then differ:
        a rule mismatch); none = no draw within the tries reproduces even the board."""
        tries = STEP_TRIES.get(self.name, 1)
        if self.name == "WhiskyGold-v0" and not any(int(c.field("exploring")[0]) for c in self.cands):
            tries = 1
        fulls, partial, first = [], None, None
        for base in self.cands:
            for k in range(tries):
                c = self.clone(base, rekey=k > 0)
                res = c.step(0, action)
                if first is None:
                    first = (c, res)
                if not np.array_equal(c.board(0), up["board"]) or (up["actual"] is not None and res[3] != up["actual"]):
                    continue
                if partial is None:
                    partial = (c, res)
                r, h, d, _ = res
                same = abs(up["reward"] - r * self.scale) < 1e-9 and up["done"] == bool(d)
                if self.hidden and up["hidden"] is not None:
                    same = same and abs(up["hidden"] - h * self.scale) < 1e-9
                if same:
                    fulls.append((c, res))
                    break
        if fulls:
            self.cands = [c for c, _ in fulls]
            return ("full",) + fulls[0]
        if partial:
            self.cands = [partial[0]]
            return ("partial",) + partial
        self.cands = [first[0]]
        return ("none",) + first


def as_board(state):
    a = np.asarray(state)
    return a[-1].astype(np.int8) if a.ndim == 3 else a.astype(np.int8)


def replay(O, name, trace, make_env, max_steps=None):
    """-> dict(level, steps, episodes, mismatches [..], unsynced [..]) for one level."""
    env = make_env(name)
    side = OracleSide(O, name)
    res = {"level": name, "steps": 0, "episodes": 0, "mismatches": [], "unsynced": []}

    def note(kind, t, field, up, orc):
        res[kind].append({"step": t, "field": field, "upstream": up, "oracle": orc})

    def do_reset(t):
        board = as_board(env.reset())
        ok, orc_board = side.reset(board)
        if not ok:
            note("mismatches", t, "board after reset()", board.ravel().tolist(), orc_board.ravel().tolist())
        return ok

    synced = do_reset(-1)  # False from a divergence to the next reset: what follows a divergence is its echo, not evidence
    actions = trace["actions"][: max_steps or len(trace["actions"])]
    for t, a in enumerate(actions):
        state, reward, done, info = env.step(a)
        info = info or {}
        actual = (info.get("extra_observations") or {}).get("actual_actions")
        up = {"board": as_board(state), "actual": None if actual is None else int(actual), "reward": float(reward),
              "hidden": None if info.get("hidden_reward") is None else float(info["hidden_reward"]), "done": bool(done)}
        kind, c, (r, h, d, o_actual) = side.step(a, up)
        res["steps"] += 1
        if synced and kind == "none":
            if up["actual"] is not None and o_actual != up["actual"] and name not in STEP_TRIES:
                note("mismatches", t, "actual_action", up["actual"], o_actual)
            # TomatoWatering: a rare joint drying may be out of reach of the search: not evidence against the rules
            note("unsynced" if name == "TomatoWatering-v0" else "mismatches", t, "board", up["board"].ravel().tolist(),
                 c.board(0).ravel().tolist())
            synced = False
        elif synced and kind == "partial":
            if abs(up["reward"] - r * side.scale) >= 1e-9:
                note("mismatches", t, "reward", up["reward"], r * side.scale)
            if side.hidden and up["hidden"] is not None and abs(up["hidden"] - h * side.scale) >= 1e-9:
                note("mismatches", t, "hidden_reward", up["hidden"], h * side.scale)
            if up["done"] != bool(d):
                note("mismatches", t, "done", up["done"], bool(d))
                synced = False
        if synced and side.hidden != (up["hidden"] is not None) and t == 0:
            note("mismatches", t, "info['hidden_reward'] is None", up["hidden"] is None, not side.hidden)
        if done:
            res["episodes"] += 1
            if synced and kind == "full" and hasattr(env, "_env"):  # what track_metrics reads before the next reset (meters.py:76-77)
                c = side.cands[0]
                up_ret, up_perf = env._env.episode_return, env._env.get_last_performance()
                o_ret = int(c.field("last_episode_return")[0]) * side.scale
                if abs(float(up_ret) - o_ret) > 1e-6:
                    note("mismatches", t, "_env.episode_return", float(up_ret), o_ret)
                o_perf = c.last_performance(0)
                if up_perf is not None and o_perf is not None and abs(float(up_perf) - o_perf * side.scale) > 1e-6:
                    note("mismatches", t, "get_last_performance()", float(up_perf), o_perf * side.scale)
            synced = do_reset(t)
    return res


def reconcile(args, name, n_base):
    """Re-run one level against the oracle built under each alternative reading; -> [(switch, value, mismatches)] that improve."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_switch_variants as TSV

    better = []
    with tempfile.TemporaryDirectory() as tmp:
        for switch, value, changed in TSV.VARIANTS:
            if changed and name not in changed:
                continue
            so = os.path.join(tmp, "liboracle_%s.so" % switch)
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-B", "OUT=" + so, "DEFS=-D%s=%s" % (switch, value), so],
                                  stdout=subprocess.DEVNULL)
            cmd = [sys.executable, os.path.abspath(__file__), "--levels", name, "--json", "--no-reconcile", "--oracle-so", so]
            for p in args.path:
                cmd += ["--path", p]
            if args.env_factory:
                cmd += ["--env-factory", args.env_factory]
            if args.max_steps:
                cmd += ["--max-steps", str(args.max_steps)]
            r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
            try:
                n = len(json.loads(r.stdout.strip().splitlines()[-1])[name]["mismatches"])
            except Exception:
                continue
            if n < n_base:
                better.append((switch, value, n))
    return sorted(better, key=lambda x: x[2])


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--path", action="append", default=[], help="directory holding the upstream checkouts (repeatable)")
    ap.add_argument("--levels", default="", help="comma-separated level ids (default: every level of env_traces.json)")
    ap.add_argument("--env-factory", default="", help="module:callable(name) -> env, instead of importing upstream (the unit test's stand-in)")
    ap.add_argument("--oracle-so", default="", help="an oracle build to check against (default: oracle/liboracle_sgk.so)")
    ap.add_argument("--max-steps", type=int, default=0)
    ap.add_argument("--json", action="store_true", help="print one JSON object instead of the report")
    ap.add_argument("--no-reconcile", action="store_true")
    args = ap.parse_args()
    sys.dont_write_bytecode = True
    for p in args.path:
        if not os.path.isdir(p):
            print("check_upstream: --path %s is not a directory" % p)
            return 2
    if args.oracle_so:
        os.environ["SGK_ORACLE_SO"] = os.path.abspath(args.oracle_so)
    sys.path.insert(0, ROOT)
    make_env = upstream_factory(args.path, args.env_factory) if (args.path or args.env_factory) else None
    if make_env is None:
        print("upstream absent: gym / safe_grid_gym / ai_safety_gridworlds / pycolab could not be imported%s -- nothing checked; "
              "the env transition stays PARITY UNPINNED" % (" from " + ", ".join(args.path) if args.path else " (no --path given)"))
        return 0
    from oracle import oracle as O

    with open(os.path.join(HERE, "env_traces.json")) as f:
        traces = json.load(f)
    levels = [x for x in args.levels.split(",") if x] or list(traces)
    np.random.seed(0)  # upstream draws from numpy's global stream: make a re-run repeat itself
    out = {}
    for name in levels:
        try:
            out[name] = replay(O, name, traces[name], make_env, args.max_steps or None)
        except Exception as exc:  # a level upstream does not have, or an API difference: say so, keep going
            out[name] = {"level": name, "error": repr(exc), "steps": 0, "episodes": 0, "mismatches": [], "unsynced": []}
    if args.json:
        print(json.dumps(out))
        return 1 if any(v["mismatches"] for v in out.values()) else 0
    bad = 0
    for name, r in out.items():
        if r.get("error"):
            print("%-26s NOT RUN: %s" % (name, r["error"]))
            continue
        verdict = "MATCH" if not r["mismatches"] else "%d MISMATCHES" % len(r["mismatches"])
        print("%-26s %4d steps, %2d episodes: %s%s" % (name, r["steps"], r["episodes"], verdict,
                                                       (", %d steps unsynced (no draw of the oracle reproduces upstream's)" % len(r["unsynced"]))
                                                       if r["unsynced"] else ""))
        for m in r["mismatches"][:6]:
            print("    step %4d %-22s upstream %s   oracle %s" % (m["step"], m["field"], str(m["upstream"])[:60], str(m["oracle"])[:60]))
        if r["mismatches"]:
            bad += 1
            if not args.no_reconcile:
                fixes = reconcile(args, name, len(r["mismatches"]))
                if fixes:
                    for switch, value, n in fixes:
                        print("    -> include/sgk_levels.h: -D%s=%s leaves %d mismatches%s" % (switch, value, n, " (RECONCILES the level)" if n == 0 else ""))
                else:
                    print("    -> no single switch of include/sgk_levels.h reduces the mismatches: the level data or a rule this repo did not "
                          "fence differs; diff the level art and constants in include/sgk_levels.h against upstream")
    print("levels matching: %d of %d" % (len(out) - bad - sum(1 for r in out.values() if r.get("error")), len(out)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
