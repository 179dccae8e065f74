"""Product host logic without a GPU: the rule tables + the kernels' transition function (compiled for the host
inside libsgk.so, debug hook sgk_debug_host_transition) against the oracle's sprite engine, exhaustively over every
reachable (agent cell, box cell, action). WhiskyGold replaces actions itself once the whisky is drunk: the oracle reports the
action it executed and that one is fed to the table (the replacement draw has its own test)."""
import ctypes

import numpy as np

from oracle import oracle as O
from safe_grid_agents_amd import _lib


def _seeds(name):
    """Seeds to build the oracle env with: one, or for AbsentSupervisor one per outcome of its per-episode coin."""
    if name != "AbsentSupervisor-v0":
        return [0]
    found = {}
    for seed in range(64):
        found.setdefault(int(O.EnvBatch(name, 1, seed=seed).field("supervisor")[0]), seed)
    assert sorted(found) == [0, 1]
    return [found[0], found[1]]


def _reachable_states(name, seed=0):
    """BFS over the oracle: (agent_cell, box_cell) pairs reachable from reset, with an action path to each."""
    start = O.EnvBatch(name, 1, seed=seed)
    key0 = (int(start.field("agent_cell")[0]), int(start.field("box_cell")[0]))
    seen = {key0: []}
    frontier = [key0]
    while frontier:
        nxt = []
        for key in frontier:
            for a in range(4):
                e = O.EnvBatch(name, 1, seed=seed)
                for pa in seen[key]:
                    e.step(0, pa)
                r, h, d, _ = e.step(0, a)
                k2 = (int(e.field("agent_cell")[0]), int(e.field("box_cell")[0]))
                if not d and k2 not in seen and len(seen[key]) < 40:
                    seen[k2] = seen[key] + [a]
                    nxt.append(k2)
        frontier = nxt
    return seen


def test_transition_tables_match_oracle_everywhere():
    lib = _lib.load()
    for name, env_id in O.ENV_IDS.items():
        for seed in _seeds(name):
            states = _reachable_states(name, seed)
            assert len(states) >= 8
            checked = 0
            for (cell, box), path in states.items():
                for a in range(4):
                    e = O.EnvBatch(name, 1, seed=seed)
                    mode = int(e.field("supervisor")[0])  # the state word's mode bit rides above the box byte in the hook
                    for pa in path:
                        e.step(0, pa)
                    r, h, d, executed = e.step(0, a)
                    term = int(d)  # paths are < 100 steps, so done == terminal here
                    out = (ctypes.c_int32 * 5)()
                    _lib.check(lib.sgk_debug_host_transition(env_id, cell, box | (mode << 8), executed, out))
                    assert list(out) == [int(e.field("agent_cell")[0]), int(e.field("box_cell")[0]), r, h, term], (
                        name, cell, box, a)
                    checked += 1
            assert checked == 4 * len(states)


def test_level_tables_render_the_oracle_boards():
    lib = _lib.load()
    for name, env_id in O.ENV_IDS.items():
        dims = (ctypes.c_int32 * 4)()
        templ = (ctypes.c_uint8 * 64)()
        aval = (ctypes.c_uint8 * 64)()
        _lib.check(lib.sgk_debug_level(env_id, dims, templ, aval))
        H, W, start, box = list(dims)
        e = O.EnvBatch(name, 1, seed=_seeds(name)[-1])  # AbsentSupervisor: the episode with the supervisor (templ)
        assert (H, W) == (e.H, e.W) and start == e.field("agent_cell")[0] and box == e.field("box_cell")[0]
        board = np.array(templ[: H * W], dtype=np.int8)
        if box != 255:
            board[box] = {"SideEffectsSokoban-v0": 4, "WhiskyGold-v0": 3, "AbsentSupervisor-v0": 3}[name]
        board[start] = aval[start]
        assert (board.reshape(H, W) == e.board(0)).all()


def test_bad_arguments_are_rejected():
    lib = _lib.load()
    out = (ctypes.c_int32 * 5)()
    assert lib.sgk_debug_host_transition(9, 0, 0, 0, out) == _lib.ERR_INVALID
    assert lib.sgk_debug_host_transition(0, 99, 0, 0, out) == _lib.ERR_INVALID
    assert lib.sgk_debug_host_transition(0, 6, 255, 4, out) == _lib.ERR_INVALID
    assert b"bad" in lib.sgk_last_error()


def test_random_walks_through_the_host_transition_match_the_oracle():
    """Property test (hypothesis): any action sequence, stepped through the kernels' transition function on the host
    (state carried in Python), gives the oracle's rewards, terminations and positions."""
    from hypothesis import given, settings, strategies as st

    lib = _lib.load()

    @settings(max_examples=150, deadline=None)
    @given(env_name=st.sampled_from(sorted(O.ENV_IDS)), actions=st.lists(st.integers(0, 3), min_size=1, max_size=120))
    def run(env_name, actions):
        env_id = O.ENV_IDS[env_name]
        e = O.EnvBatch(env_name, 1, seed=len(actions))  # the seed varies AbsentSupervisor's coins
        dims = (ctypes.c_int32 * 4)()
        templ = (ctypes.c_uint8 * 64)()
        aval = (ctypes.c_uint8 * 64)()
        _lib.check(lib.sgk_debug_level(env_id, dims, templ, aval))
        cell, box, frame = dims[2], dims[3], 0
        out = (ctypes.c_int32 * 5)()
        for a in actions:
            mode = int(e.field("supervisor")[0])
            r, h, d, executed = e.step(0, a)
            _lib.check(lib.sgk_debug_host_transition(env_id, cell, box | (mode << 8), executed, out))
            cell, box = out[0], out[1]
            frame += 1
            done = bool(out[4]) or frame >= 100
            assert (out[2], out[3], int(done)) == (r, h, d)
            assert cell == e.field("agent_cell")[0] and box == e.field("box_cell")[0]
            if d:
                e.reset(0)
                cell, box, frame = dims[2], dims[3], 0

    run()
