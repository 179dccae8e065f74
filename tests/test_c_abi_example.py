"""The C-ABI used from plain C (examples/c_abi_rollout.c): compiles and links against libsgk.so with gcc alone (CPU check),
and on a GPU its output equals the oracle's for the same rollout."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "safe-grid-agents_amd", "lib")


def _build(tmp_path):
    exe = str(tmp_path / "c_abi_rollout")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "c_abi_rollout.c"), "-o", exe, "-L" + LIBDIR, "-lsgk",
                           "-Wl,-rpath," + LIBDIR])
    return exe


def test_c_example_compiles_links_and_fails_loudly_without_gpu(tmp_path):
    from safe_grid_agents_amd import _lib

    exe = _build(tmp_path)
    if _lib.device_count() > 0:
        pytest.skip("GPU present: covered by the gpu test")
    p = subprocess.run([exe, "0", "16", "10", "1"], capture_output=True, text=True)
    assert p.returncode == 1 and "no CPU fallback" in p.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("env_id,name", [(0, "BoatRace-v0"), (1, "IslandNavigation-v0"), (2, "SideEffectsSokoban-v0"), (3, "DistributionalShift-v0"),
                                         (6, "SafeInterruptibility-v0"),
                                         (7, "ConveyorBelt-v0"), (8, "TomatoWatering-v0"), (9, "FriendFoe-v0")])
def test_c_example_matches_oracle_on_gpu(tmp_path, env_id, name):
    from oracle import oracle as O

    exe = _build(tmp_path)
    n, steps, seed = 3000, 140, 77
    out = json.loads(subprocess.check_output([exe, str(env_id), str(n), str(steps), str(seed)], text=True))
    # two and three contiguous env-id shards (env_index_base), metrics through the library's RCCL all-reduce: the same line
    for shards in (2, 3):
        text = subprocess.check_output([exe, str(env_id), str(n), str(steps), str(seed), str(shards)], text=True)
        assert json.loads(text.strip().splitlines()[-1]) == out  # (RCCL prints a version banner on stdout when it is first used)
    orc = O.EnvBatch(name, n, seed=seed)
    m = O.metrics_new()
    orc.rollout(3 * steps, seed=seed, auto_reset=True, metrics=m)  # step kernel + fused rollout + streamed rollout
    assert out["episodes"] == m[O.M_EPISODES] and out["sum_return"] == m[O.M_SUM_RETURN]
    assert out["sum_safety"] == m[O.M_SUM_SAFETY] and out["max_return"] == m[O.M_MAX_RETURN]
    assert out["steps"] == 3 * steps * n and (out["height"], out["width"]) == (orc.H, orc.W)
    h = 1469598103934665603
    for b in orc.boards().astype(np.uint8).ravel().tolist():
        h = ((h ^ b) * 1099511628211) & (2**64 - 1)
    assert out["boards_fnv1a"] == h
