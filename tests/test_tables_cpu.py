"""Product host logic without a GPU: the rule tables + the kernels' transition function (sgk_transition.h, the very code the
kernels compile, built for the host by g++: tests/hostlib.py) against the oracle's sprite engine, exhaustively over every
reachable (agent cell, second sprite cell, mode bit, action) -- the mode bit being the per-episode coin of AbsentSupervisor /
SafeInterruptibility and the conveyor belt's "the object has arrived" flag. WhiskyGold replaces actions itself once the whisky is
drunk: the oracle reports the action it executed and that one is fed to the table (the replacement draw has its own test);
SafeInterruptibility's substitution is deterministic and part of the product code under test, so there the agent's own action
is fed.

This module also runs, unchanged, under every alternative reading of an uncertain upstream detail: test_switch_variants.py
builds both sides with -DSGK_...=... and points SGK_HOST_LIB / SGK_ORACLE_SO at the variant libraries."""
import ctypes
import os

import numpy as np

import hostlib
from oracle import oracle as O

ERR_INVALID = hostlib.ERR_INVALID
COIN_ENVS = ("AbsentSupervisor-v0", "SafeInterruptibility-v0")
# levels whose transition draws by itself every step (tomatoes dry): no exhaustive (state, action) table -- they are checked along
# seeded walks through the full host step (state word in / state word out), like every other level
STOCHASTIC_ENVS = ("TomatoWatering-v0", "FriendFoe-v0")  # (FriendFoe: which box pays depends on estimates kept across episodes)
DETERMINISTIC = {k: v for k, v in O.ENV_IDS.items() if k not in STOCHASTIC_ENVS}


def check(rc):
    assert rc == 0, rc


def _seeds(name):
    """Seeds to build the oracle env with: one, or for the envs with a per-episode coin one per outcome."""
    if name not in COIN_ENVS:
        return [0]
    found = {}
    for seed in range(64):
        found.setdefault(int(O.EnvBatch(name, 1, seed=seed).field("coin")[0]), seed)
    assert sorted(found) == [0, 1]
    return [found[0], found[1]]


def _hook_action(name, chosen, executed):
    return chosen if name == "SafeInterruptibility-v0" else executed


def _key(e):
    return int(e.field("agent_cell")[0]), int(e.field("box_cell")[0]), int(e.field("coin")[0])


def _reachable_states(name, seed=0):
    """BFS over the oracle: (agent_cell, box_cell, mode bit) triples reachable from reset, with an action path to each."""
    start = O.EnvBatch(name, 1, seed=seed)
    key0 = _key(start)
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
                k2 = _key(e)
                if not d and k2 not in seen and len(seen[key]) < 40:
                    seen[k2] = seen[key] + [a]
                    nxt.append(k2)
        frontier = nxt
    return seen


def test_transition_tables_match_oracle_everywhere():
    lib = hostlib.load()
    for name, env_id in DETERMINISTIC.items():
        for seed in _seeds(name):
            states = _reachable_states(name, seed)
            assert len(states) >= 8
            checked = 0
            for (cell, box, mode), path in states.items():  # the state word's mode bit rides above the box byte in the hook
                for a in range(4):
                    e = O.EnvBatch(name, 1, seed=seed)
                    for pa in path:
                        e.step(0, pa)
                    r, h, d, executed = e.step(0, a)
                    term = int(d)  # paths are < 100 steps, so done == terminal here
                    out = (ctypes.c_int32 * 5)()
                    check(lib.sgk_debug_host_transition(env_id, cell, box | (mode << 8), _hook_action(name, a, executed), out))
                    cell2, box2, mode2 = _key(e)
                    assert list(out) == [cell2, box2, r, h, term | (mode2 << 1)], (name, cell, box, mode, a)
                    checked += 1
            assert checked == 4 * len(states)


def _product_board(name, R, cell, box, coin):
    """The board the product's writers materialise for a state: backdrop (one of two), second sprite, agent on top."""
    templ, templ_alt, aval, nc = R.templ, R.templ_alt, R.agent_value, R.n_cells
    alt = (name == "AbsentSupervisor-v0" and not coin) or (name == "SafeInterruptibility-v0" and box == 255)
    board = np.array((templ_alt if alt else templ)[:nc], dtype=np.int8)
    if name == "FriendFoe-v0":  # `coin` carries the ext bits here: the room type picks one of three backdrops; the level does not show
        board = np.array((templ, templ_alt, R.templ_alt2)[coin & 3][:nc], dtype=np.int8)
    elif name == "TomatoWatering-v0":  # `box` carries the whole watered mask here; on the bucket the second backdrop shows it all
        board = np.array((templ_alt if cell == R.aux_cell else templ)[:nc], dtype=np.int8)
        if cell != R.aux_cell:
            for k in range(R.n_tomatoes):
                if (box >> k) & 1:
                    board[R.tomato_cell[k]] = R.value_box
    elif box != 255:
        board[box] = R.value_box_alt if (name == "ConveyorBelt-v0" and coin) else R.value_box
    board[cell] = aval[cell]
    return board


class _Rules(ctypes.Structure):
    """Prefix of SgkRules (safe-grid-agents_amd/csrc/sgk_rules.h) up to the fields this module reads."""
    _fields_ = [("env_id", ctypes.c_int32), ("height", ctypes.c_int32), ("width", ctypes.c_int32), ("n_cells", ctypes.c_int32),
                ("start_agent", ctypes.c_int32), ("start_box", ctypes.c_int32), ("max_iterations", ctypes.c_int32),
                ("n_states", ctypes.c_int32), ("stay_obs", ctypes.c_int32), ("stay_hid", ctypes.c_int32),
                ("value_box", ctypes.c_int32), ("aux_reward", ctypes.c_int32), ("dcell", ctypes.c_int32 * 4),
                ("trans", ctypes.c_uint32 * 256), ("templ", ctypes.c_uint8 * 64), ("agent_value", ctypes.c_uint8 * 64),
                ("box_penalty", ctypes.c_int8 * 64), ("box_blocked", ctypes.c_uint8 * 64), ("safety", ctypes.c_uint8 * 64),
                ("state_slot", ctypes.c_uint8 * 64), ("slot_cell", ctypes.c_uint8 * 64), ("n_slots", ctypes.c_int32),
                ("n_live_slots", ctypes.c_int32), ("aux_cell", ctypes.c_int32), ("forced_action", ctypes.c_int32),
                ("palette", (ctypes.c_uint8 * 4) * 8), ("draw_threshold", ctypes.c_uint32), ("render_hwc", ctypes.c_int32),
                ("value_box_alt", ctypes.c_int32), ("env_flags", ctypes.c_int32), ("templ_alt", ctypes.c_uint8 * 64),
                ("reward_scale", ctypes.c_double), ("tomato_cell", ctypes.c_uint8 * 16), ("tomato_index", ctypes.c_uint8 * 64),
                ("templ_alt2", ctypes.c_uint8 * 64), ("aux_cell2", ctypes.c_int32), ("pad3", ctypes.c_int32),
                ("start_ext", ctypes.c_int32), ("n_tomatoes", ctypes.c_int32)]


def _rules(lib, env_id):
    assert lib.sgk_debug_rules_size() == ctypes.sizeof(_Rules), "tests/test_tables_cpu.py: _Rules is out of date with sgk_rules.h"
    R = _Rules()
    check(lib.sgk_debug_rules(env_id, ctypes.byref(R)))
    return R


def test_level_tables_render_the_oracle_boards_in_every_reachable_state():
    """Backdrop(s) + second sprite + agent value, composed the way the board writers do, equal the oracle's rendered board in
    EVERY reachable state (both outcomes of a per-episode coin), not just after reset."""
    lib = hostlib.load()
    for name, env_id in DETERMINISTIC.items():
        R = _rules(lib, env_id)
        nc = R.n_cells
        for seed in _seeds(name):
            e0 = O.EnvBatch(name, 1, seed=seed)
            assert (R.height, R.width) == (e0.H, e0.W)
            assert R.start_agent == e0.field("agent_cell")[0] and R.start_box == e0.field("box_cell")[0]
            for (cell, box, coin), path in _reachable_states(name, seed).items():
                e = O.EnvBatch(name, 1, seed=seed)
                for pa in path:
                    e.step(0, pa)
                got = _product_board(name, R, cell, box, coin)
                assert (got.reshape(e.H, e.W) == e.board(0)).all(), (name, cell, box, coin)


def test_palette_renders_the_oracle_frame():
    """render("rgb_array") in the product is palette[value] per cell, laid out per SgkRules.render_hwc: equal to the oracle's
    character-colour rendering in every reachable state. (An alternative value map that gives two differently coloured
    characters ONE value cannot be rendered from values: such levels are named in SGK_SKIP_PALETTE by the variant test.)"""
    lib = hostlib.load()
    skip = os.environ.get("SGK_SKIP_PALETTE", "").split(",")
    for name, env_id in DETERMINISTIC.items():
        if name in skip:
            continue
        R = _rules(lib, env_id)
        nc = R.n_cells
        pal = np.array([[R.palette[v][k] for k in range(3)] for v in range(8)], dtype=np.uint8)
        for seed in _seeds(name):
            for (cell, box, coin), path in list(_reachable_states(name, seed).items())[::3]:
                e = O.EnvBatch(name, 1, seed=seed)
                for pa in path:
                    e.step(0, pa)
                board = _product_board(name, R, cell, box, coin)
                frame = pal[board & 7]  # [cell][3]
                got = frame.reshape(-1) if R.render_hwc else frame.T.reshape(-1)
                assert (got == e.render_rgb(0).reshape(-1)).all(), (name, cell, box)


def test_bad_arguments_are_rejected():
    lib = hostlib.load()
    out = (ctypes.c_int32 * 5)()
    assert lib.sgk_debug_host_transition(99, 0, 0, 0, out) == ERR_INVALID
    assert lib.sgk_debug_host_transition(0, 99, 0, 0, out) == ERR_INVALID
    assert lib.sgk_debug_host_transition(0, 6, 255, 4, out) == ERR_INVALID


def test_random_action_stream_and_episode_coins_match_the_oracle():
    lib = hostlib.load()
    for seed, env, t in [(0, 0, 0), (0x5AFE, 7, 63), (0x5AFE, 7, 64), (99, (1 << 40) + 5, 12345), (2**63 + 1, 3, 2**33 + 17)]:
        assert lib.sgk_random_action(seed, env, t) == O.random_action(seed, env, t)
    for name in COIN_ENVS:
        e = O.EnvBatch(name, 40, seed=17, env_begin=1000)
        for k in range(1, 6):  # reset number k (the create-time reset is number 1)
            want = e.field("coin")
            got = [lib.sgk_debug_episode_coin(O.ENV_IDS[name], 17, 1000 + i, k) for i in range(40)]
            assert got == want.tolist(), (name, k)
            e.reset()
        assert 0 < sum(got) < 40


def test_random_walks_through_the_host_transition_match_the_oracle():
    """Property test (hypothesis): any action sequence, stepped through the kernels' transition function on the host
    (state carried in Python), gives the oracle's rewards, terminations and positions."""
    from hypothesis import given, settings, strategies as st

    lib = hostlib.load()

    @settings(max_examples=int(os.environ.get("SGK_WALK_EXAMPLES", "150")), deadline=None)
    @given(env_name=st.sampled_from(sorted(DETERMINISTIC)), actions=st.lists(st.integers(0, 3), min_size=1, max_size=120))
    def run(env_name, actions):
        env_id = O.ENV_IDS[env_name]
        e = O.EnvBatch(env_name, 1, seed=len(actions))  # the seed varies the per-episode coins
        dims = (ctypes.c_int32 * 4)()
        templ = (ctypes.c_uint8 * 64)()
        aval = (ctypes.c_uint8 * 64)()
        check(lib.sgk_debug_level(env_id, dims, templ, aval))
        cell, box, frame = dims[2], dims[3], 0
        mode = int(e.field("coin")[0])  # the episode's coin; from then on the product's own mode bit is carried
        horizon = _rules(lib, env_id).max_iterations
        out = (ctypes.c_int32 * 5)()
        for a in actions:
            r, h, d, executed = e.step(0, a)
            check(lib.sgk_debug_host_transition(env_id, cell, box | (mode << 8), _hook_action(env_name, a, executed), out))
            cell, box, mode = out[0], out[1], out[4] >> 1
            frame += 1
            done = bool(out[4] & 1) or frame >= horizon
            assert (out[2], out[3], int(done)) == (r, h, d)
            assert (cell, box, mode) == _key(e)
            if d:
                e.reset(0)
                cell, box, frame = dims[2], dims[3], 0
                mode = int(e.field("coin")[0])

    run()


def _unpack(word):
    """The packed env state word (sgk_transition.h: pack_state)."""
    lo, hi = word & 0xffffffff, word >> 32
    ret, hid = hi & 0xffff, hi >> 16
    return {"pos": lo & 0xff, "box": (lo >> 8) & 0xff, "frame": (lo >> 16) & 0xff, "over": (lo >> 24) & 1, "mode": (lo >> 25) & 1,
            "ext": (lo >> 26) & 0x3f, "ret": ret - 65536 if ret >= 32768 else ret, "hid": hid - 65536 if hid >= 32768 else hid}


def test_state_word_walks_through_the_full_host_step_match_the_oracle():
    """EVERY level, the stochastic ones included, through sgk_debug_host_step -- env_actual_action + transition + the episode
    bookkeeping on the packed state word, the envs' own draws keyed (seed, env index, reset counter, frame) as on the device --
    along seeded random walks with resets: rewards, done, executed action, every field of the word and the board the product's
    writers would materialise from it, against the oracle step by step."""
    lib = hostlib.load()
    skip_palette = os.environ.get("SGK_SKIP_PALETTE", "").split(",")
    n_walks = max(2, int(os.environ.get("SGK_WALK_EXAMPLES", "150")) // 25)
    for name, env_id in O.ENV_IDS.items():
        R = _rules(lib, env_id)
        pal = np.array([[R.palette[v][k] for k in range(3)] for v in range(8)], dtype=np.uint8)
        on_bucket = 0
        for walk in range(n_walks):
            seed, genv = 1000 + walk, (walk << 33) + 7 * walk
            rng = np.random.RandomState(walk)
            e = O.EnvBatch(name, 1, seed=seed, env_begin=genv)
            n_resets = 1
            aux = (ctypes.c_double * 6)(*([0.5] * 6))  # the env's float64 side state (FriendFoe's bandit estimates)
            word = lib.sgk_debug_reset_word(env_id, seed, genv, n_resets, aux)
            out, wout = (ctypes.c_int32 * 4)(), ctypes.c_uint64()
            for t in range(260):
                # bias the tomato walks towards the bucket now and then (up / right), so that the second backdrop is exercised
                a = int(rng.choice([0, 3])) if (name == "TomatoWatering-v0" and (t // 40) % 2) else int(rng.randint(0, 4))
                r, h, d, executed = e.step(0, a)
                check(lib.sgk_debug_host_step(env_id, word, n_resets, a, seed, genv, ctypes.byref(wout), out, aux))
                word = wout.value
                assert list(out) == [r, h, d, executed], (name, walk, t)
                s = _unpack(word)
                mask = int(e.field("tomato_mask")[0])
                want = {"pos": int(e.field("agent_cell")[0]), "box": int(e.field("box_cell")[0]), "frame": int(e.field("frame")[0]),
                        "over": d, "mode": 0 if name == "FriendFoe-v0" else int(e.field("coin")[0]), "ext": int(e.field("ext")[0]),
                        "ret": int(e.field("episode_return")[0]), "hid": int(e.field("hidden_return")[0])}
                assert s == want, (name, walk, t)
                on_bucket += int(name == "TomatoWatering-v0" and s["pos"] == R.aux_cell)
                box = (s["box"] | s["ext"] << 8) if name == "TomatoWatering-v0" else s["box"]
                board = _product_board(name, R, s["pos"], box, s["ext"] if name == "FriendFoe-v0" else s["mode"])
                if name == "FriendFoe-v0":
                    assert list(aux) == e.foe_policy(0).ravel().tolist(), (walk, t)
                assert (board.reshape(e.H, e.W) == e.board(0)).all(), (name, walk, t)
                if name not in skip_palette and t % 16 == 0:
                    frame = pal[board & 7]
                    got = frame.reshape(-1) if R.render_hwc else frame.T.reshape(-1)
                    assert (got == e.render_rgb(0).reshape(-1)).all(), (name, walk, t)
                if d:
                    e.reset(0)
                    n_resets += 1
                    word = lib.sgk_debug_reset_word(env_id, seed, genv, n_resets, aux)
                    assert _unpack(word)["ext"] == int(e.field("ext")[0])
                    assert _unpack(word)["mode"] == (0 if name == "FriendFoe-v0" else int(e.field("coin")[0]))
        assert on_bucket > 0 or name != "TomatoWatering-v0"  # the delusion backdrop and its observed reward were exercised


def test_level_art_of_the_header_equals_the_surveys_transcription():
    """include/sgk_levels.h is shared by the product and the oracle: a mistyped map cell there is common-mode and no parity
    test can see it. For the three levels SURVEY.md's Appendix A draws -- the survey session's own transcription of the
    upstream maps, typed independently of the header -- the two must agree character for character."""
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    survey = open(os.path.join(root, "SURVEY.md"), encoding="utf-8").read()
    header = open(os.path.join(root, "include", "sgk_levels.h"), encoding="utf-8").read()
    for title, symbol in (("BoatRace-v0", "SGK_BOAT_ART"), ("IslandNavigation-v0", "SGK_ISLAND_ART"),
                          ("SideEffectsSokoban-v0", "SGK_SOKOBAN_ART")):
        at = survey.index("**%s**" % title)
        lo = survey.index("```", at)
        art = survey[lo + 3:survey.index("```", lo + 3)].strip("\n").split("\n")
        body = header[header.index(symbol + "["):]
        rows = re.findall(r'"([^"]*)"', body[:body.index("};")])
        assert rows == art, (title, rows, art)


def test_debug_hooks_refuse_null_outputs():
    """The host-only library's debug entry points (the same code libsgk.so links) return an error for a NULL output instead of
    writing through it, for every level."""
    H = hostlib
    L = H.load()
    for env_id in range(10):
        assert L.sgk_debug_host_transition(env_id, 0, 0, 0, None) == H.ERR_INVALID
        assert L.sgk_debug_level(env_id, None, None, None) == H.ERR_INVALID
        assert L.sgk_debug_rules(env_id, None) == H.ERR_INVALID
        assert L.sgk_debug_host_step(env_id, 0, 1, 0, 0, 0, None, None, None) == H.ERR_INVALID
        word = L.sgk_debug_reset_word(env_id, 1, 2, 1, None)  # (friend or foe: no estimates given -> the prior's level)
        assert word != 2**64 - 1
    assert L.sgk_debug_level(99, None, None, None) == H.ERR_INVALID
