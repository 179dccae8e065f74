#!/usr/bin/env python3
"""Freeze the CPU restatement's env semantics as data: tests/golden/env_traces.json.

NOT a reference-derived fixture (the upstream env is absent: PARITY UNPINNED, oracle/sgk_oracle.c); it pins THIS repo's
reading of the published rules so that the oracle and the kernels cannot drift together unnoticed. Each trace: a seeded
action sequence (with reset after `done`) and, per step, [reward, hidden_reward, done, agent_cell, box_cell]; plus the
board after selected steps."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
from oracle import oracle as O  # noqa: E402

out = {}
for name in O.ENV_IDS:
    rng = np.random.RandomState(2024)
    e = O.EnvBatch(name, 1)
    e.reset(0)  # gym usage: make(), then reset() -- the env's own draws are keyed by its reset counter
    initial = e.board(0).ravel().tolist()
    actions = rng.randint(0, 4, size=400).tolist()
    steps, boards = [], {}
    for t, a in enumerate(actions):
        r, h, d, _ = e.step(0, a)
        steps.append([r, h, d, int(e.field("agent_cell")[0]), int(e.field("box_cell")[0])])
        if t % 50 == 0 or d:
            boards[str(t)] = e.board(0).ravel().tolist()
        if d:
            e.reset(0)
    out[name] = {"actions": actions, "steps": steps, "boards": boards, "initial_board": initial}
with open(os.path.join(HERE, "env_traces.json"), "w") as f:
    json.dump(out, f, separators=(",", ":"))
print("episodes:", {k: sum(s[2] for s in v["steps"]) for k, v in out.items()})
