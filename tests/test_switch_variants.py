"""Every uncertain reading of the upstream env rules is a named switch in include/sgk_levels.h (SURVEY.md Appendix A's "(?)"
items). For EACH alternative this test builds BOTH sides -- the product's rule builder + transition code (host-only library,
g++) and the oracle's sprite engine (gcc) -- with -D<switch>=<alternative>, and re-runs tests/test_tables_cpu.py against the
variant pair: exhaustive (state, action) transitions, boards in every reachable state, random walks. So a later session with
upstream access flips a constant; the kernels (which consume only the tables and sgk_transition.h) need no rewrite.

Also asserts what each switch is expected to change (and what not) in the product's tables of the shipped levels.
"""
import os
import subprocess
import sys

import pytest

import hostlib
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (switch, alternative, envs whose product tables must change; empty = the reading is invisible on the shipped level)
VARIANTS = [
    ("SGK_BOAT_MOVEMENT_IN_HIDDEN", "1", ["BoatRace-v0"]),
    ("SGK_BOAT_BLOCKED_ON_ARROW_PENALISED", "0", ["BoatRace-v0"]),
    ("SGK_ISLAND_WATER_OBSERVED", "-50", ["IslandNavigation-v0"]),
    ("SGK_ISLAND_WATER_OVER_AGENT", "0", ["IslandNavigation-v0"]),
    ("SGK_ISLAND_VALUE_WALL", "0", ["IslandNavigation-v0"]),
    ("SGK_SOKOBAN_BOX_BLOCKED_BY_GOAL", "1", ["SideEffectsSokoban-v0"]),
    ("SGK_SOKOBAN_WALL_RULE", "0", []),  # both rules give the same penalties on level 0
    ("SGK_SOKOBAN_VALUE_SET", "1", ["SideEffectsSokoban-v0"]),
    ("SGK_WHISKY_EXPLORATION_U32", "2147483648u", ["WhiskyGold-v0"]),
    ("SGK_WHISKY_REWARD_IN_HIDDEN", "1", ["WhiskyGold-v0"]),  # the +5 also on the hidden channel (a robustness level: performance == return)
    ("SGK_INTERRUPT_FORCED_ACTION", "4", ["SafeInterruptibility-v0"]),  # the table field differs; trajectories do not (corridor)
    ("SGK_INTERRUPT_HIDDEN_ONLY_UNINTERRUPTED", "0", ["SafeInterruptibility-v0"]),
    ("SGK_INTERRUPT_VALUE_SET", "1", ["SafeInterruptibility-v0"]),
    ("SGK_BELT_END_OVER_OBJECT", "0", ["ConveyorBelt-v0"]),
    ("SGK_BELT_OBJECT_BLOCKED_BY_AGENT", "1", ["ConveyorBelt-v0"]),
    ("SGK_BELT_CURTAIN_COVERS_END", "0", ["ConveyorBelt-v0"]),
    ("SGK_TOMATO_DRY_U32", "858993459u", ["TomatoWatering-v0"]),
    ("SGK_TOMATO_DELUSION_COVERS_ALL_CELLS", "0", ["TomatoWatering-v0"]),
    ("SGK_FOE_MOVEMENT_REWARD", "0", ["FriendFoe-v0"]),
    ("SGK_FOE_GOAL_REWARD", "1", ["FriendFoe-v0"]),
    ("SGK_FOE_EMPTY_REWARD", "0", ["FriendFoe-v0"]),
    ("SGK_RENDER_HWC", "1", list(O.ENV_IDS)),
    ("SGK_MAX_ITERATIONS", "60", list(O.ENV_IDS)),
]


def _build_pair(tmp, switch, value):
    defs = "-D%s=%s" % (switch, value)
    host = hostlib.build(out=os.path.join(tmp, "libsgk_host.so"), defs=defs)
    orc = os.path.join(tmp, "liboracle_sgk.so")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-B", "OUT=" + orc, "DEFS=" + defs, orc],
                          stdout=subprocess.DEVNULL)
    return host, orc


@pytest.mark.parametrize("switch,value,changed", VARIANTS, ids=[v[0] for v in VARIANTS])
def test_product_tables_and_oracle_agree_under_the_alternative_reading(tmp_path, switch, value, changed):
    host, orc = _build_pair(str(tmp_path), switch, value)
    # IslandNavigation's '#' = 0 makes wall and water ONE value with two colours: a board of values cannot be coloured then
    skip_palette = "IslandNavigation-v0" if switch == "SGK_ISLAND_VALUE_WALL" else ""
    env = dict(os.environ, SGK_HOST_LIB=host, SGK_ORACLE_SO=orc, SGK_WALK_EXAMPLES="40", SGK_SKIP_PALETTE=skip_palette,
               PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "safe-grid-agents_amd"), os.path.join(ROOT, "tests")]))
    # the variant pair through the whole table-vs-engine module (a fresh interpreter: the libraries are process-wide)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_tables_cpu.py")], env=env, cwd=str(tmp_path), capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, "%s=%s\n%s\n%s" % (switch, value, r.stdout[-3000:], r.stderr[-2000:])
    # and the switch does what it says to the product's tables: exactly the named levels change
    base, var = hostlib.load(), hostlib.load(host)
    for name, env_id in O.ENV_IDS.items():
        same = hostlib.rules_bytes(base, env_id) == hostlib.rules_bytes(var, env_id)
        assert same == (name not in changed), (switch, name, "unchanged" if same else "changed")


def test_default_build_is_the_documented_default_of_every_switch():
    """A -D<switch>=<default> build equals the plain build byte for byte (guards the #ifndef defaults against drift from the
    table in DESIGN.md section 4)."""
    defaults = {"SGK_BOAT_MOVEMENT_IN_HIDDEN": "0", "SGK_BOAT_BLOCKED_ON_ARROW_PENALISED": "1", "SGK_ISLAND_WATER_OBSERVED": "0",
                "SGK_ISLAND_WATER_OVER_AGENT": "1", "SGK_ISLAND_VALUE_WALL": "4", "SGK_SOKOBAN_BOX_BLOCKED_BY_GOAL": "0",
                "SGK_SOKOBAN_WALL_RULE": "1", "SGK_SOKOBAN_VALUE_SET": "0", "SGK_WHISKY_EXPLORATION_U32": "3865470566u", "SGK_WHISKY_REWARD_IN_HIDDEN": "0",
                "SGK_INTERRUPT_FORCED_ACTION": "0", "SGK_INTERRUPT_HIDDEN_ONLY_UNINTERRUPTED": "1", "SGK_INTERRUPT_VALUE_SET": "0", "SGK_RENDER_HWC": "0",
                "SGK_FOE_MOVEMENT_REWARD": "-1", "SGK_FOE_GOAL_REWARD": "50", "SGK_FOE_EMPTY_REWARD": "-50", "SGK_TOMATO_DRY_U32": "214748364u", "SGK_TOMATO_DELUSION_COVERS_ALL_CELLS": "1", "SGK_BELT_END_OVER_OBJECT": "1", "SGK_BELT_OBJECT_BLOCKED_BY_AGENT": "0", "SGK_BELT_CURTAIN_COVERS_END": "1",
                "SGK_MAX_ITERATIONS": "100"}
    import tempfile

    with tempfile.TemporaryDirectory() as tmp:
        host = hostlib.build(out=os.path.join(tmp, "libsgk_host.so"), defs=" ".join("-D%s=%s" % kv for kv in defaults.items()))
        base, var = hostlib.load(), hostlib.load(host)
        for env_id in O.ENV_IDS.values():
            assert hostlib.rules_bytes(base, env_id) == hostlib.rules_bytes(var, env_id)


def test_libsgk_debug_hooks_are_the_same_code_as_the_host_library():
    """libsgk.so (hipcc) exports the same debug hooks over the same sources: spot-check that they agree with the g++ build."""
    import ctypes

    from safe_grid_agents_amd import _lib

    prod, host = _lib.load(), hostlib.load()
    for env_id in O.ENV_IDS.values():
        d1, t1, a1 = (ctypes.c_int32 * 4)(), (ctypes.c_uint8 * 64)(), (ctypes.c_uint8 * 64)()
        d2, t2, a2 = (ctypes.c_int32 * 4)(), (ctypes.c_uint8 * 64)(), (ctypes.c_uint8 * 64)()
        assert prod.sgk_debug_level(env_id, d1, t1, a1) == 0 and host.sgk_debug_level(env_id, d2, t2, a2) == 0
        assert list(d1) == list(d2) and bytes(t1) == bytes(t2) and bytes(a1) == bytes(a2)
        o1, o2 = (ctypes.c_int32 * 5)(), (ctypes.c_int32 * 5)()
        for cell in range(d1[0] * d1[1]):
            for a in range(4):
                rc1 = prod.sgk_debug_host_transition(env_id, cell, d1[3], a, o1)
                rc2 = host.sgk_debug_host_transition(env_id, cell, d1[3], a, o2)
                assert rc1 == rc2 and list(o1) == list(o2)
