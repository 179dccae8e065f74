"""Loader for tests/golden/batched_*.npz: the reference's own TabularQAgent / tabq_learn / RandomAgent / dqn_warmup run once per
env index with np.random answered from the batched path's counter RNG (tests/golden/make_golden.py, golden_batched_*). Shared by
the CPU test (the oracle reproduces them) and the GPU tests (the HIP kernels reproduce them)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

TABQ_FIXTURES = ["batched_tabq_boat.npz", "batched_tabq_island.npz", "batched_tabq_sokoban_cheat.npz",
                 "batched_tabq_whisky_cheat.npz"]
WARMUP_FIXTURES = ["batched_warmup_boat.npz", "batched_warmup_island.npz", "batched_warmup_sokoban.npz"]


def _num(v):
    return float.fromhex(v) if isinstance(v, str) else v


class TabqFixture:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name))
        m = json.loads(str(z["meta"]))
        self.meta = m
        self.env, self.cheat, self.seed = m["env"], m["cheat"], m["seed"]
        self.n, self.steps = m["n_agents"], m["steps"]
        self.lr, self.discount, self.eps0, self.anneal = m["lr"], m["discount"], m["epsilon"], m["epsilon_anneal"]
        self.epsilon_used = m["epsilon_used"]  # hex f64, one per agent step (the same for every agent)
        self.agents = m["agents"]
        # [steps, n] like the oracle's / the product's action matrices
        self.actions = np.array([[int(c) for c in a["actions"]] for a in self.agents], dtype=np.uint8).T.copy()
        self.q_agent, self.q_boards, self.q_rows = z["q_agent"], z["q_boards"], z["q_rows"]
        self.final_boards = np.array([a["final_board"] for a in self.agents], dtype=np.int8)

    def args(self):
        import types

        return types.SimpleNamespace(lr=self.lr, discount=self.discount, epsilon=self.eps0, epsilon_anneal=self.anneal)

    def rows_of(self, i):
        sel = np.nonzero(self.q_agent == i)[0]
        return [(self.q_boards[k], self.q_rows[k]) for k in sel]

    def metrics(self):
        """The 12 used words of the batch's metrics vector from the reference's per-episode TensorBoard scalars
        (meters.py:86-96: Train/returns, /safeties, /margins, /margins_support) of all agents. Integer-reward levels only."""
        sums = {k: 0 for k in ("returns", "safeties", "margins", "margins_support")}
        maxs = {k: None for k in sums}
        counts = {k: 0 for k in sums}
        for a in self.agents:
            for k in sums:
                for v in a["episodes"][k]:
                    v = _num(v)
                    assert float(v).is_integer()
                    v = int(v)
                    sums[k] += v
                    counts[k] += 1
                    maxs[k] = v if maxs[k] is None else max(maxs[k], v)
        return sums, counts, maxs


class WarmupFixture:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name))
        m = json.loads(str(z["meta"]))
        self.meta = m
        self.env, self.seed, self.n, self.steps = m["env"], m["seed"], m["n_agents"], m["steps"]
        self.meters = m["returns_meters"]
        # [n, steps, ...] as recorded -> [steps, n, ...] like a trajectory ring
        self.states = z["states"].transpose(1, 0, 2).copy()
        self.successors = z["successors"].transpose(1, 0, 2).copy()
        self.actions = z["actions"].T.copy()
        self.rewards = z["rewards"].T.copy()
        self.terminals = z["terminals"].T.copy()


def hexes(row):
    return [float(x).hex() for x in row]
