"""tests/golden/check_upstream.py -- the harness that pins the env rules against a real upstream checkout when one is at hand
(absent here) -- exercised end to end against stand-in envs: the all-match path with every env-side draw found
by search, one injected rule difference that a switch of include/sgk_levels.h reconciles, and the "upstream absent" path."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "tests", "golden", "check_upstream.py")


def run(*args):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "tests")]), PYTHONDONTWRITEBYTECODE="1")
    return subprocess.run([sys.executable, SCRIPT] + list(args), capture_output=True, text=True, env=env, timeout=900)


def test_without_upstream_the_harness_says_so_and_exits_zero(tmp_path):
    for args in ([], ["--path", str(tmp_path)]):
        r = run(*args)
        assert r.returncode == 0 and "upstream absent" in r.stdout and "PARITY UNPINNED" in r.stdout, r.stdout + r.stderr


def test_every_level_matches_a_stand_in_whose_draws_come_from_another_stream():
    r = run("--env-factory", "upstream_standin:rekeyed")
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if " steps, " in ln]
    assert len(lines) == 10 and all("MATCH" in ln and "MISMATCH" not in ln for ln in lines), r.stdout
    assert "levels matching: 10 of 10" in r.stdout
    # the drawing levels really were searched: none of them may have fallen out of sync for more than a few rare joint draws
    for ln in lines:
        if "unsynced" in ln:
            assert ln.startswith("TomatoWatering-v0") and int(ln.split(", ")[-1].split()[0]) <= 3, ln


def test_an_injected_rule_difference_is_reported_and_reconciled_by_its_switch():
    r = run("--env-factory", "upstream_standin:boat_movement_in_hidden", "--levels", "BoatRace-v0,IslandNavigation-v0")
    assert r.returncode == 1, r.stdout + r.stderr
    assert "BoatRace-v0" in r.stdout and "MISMATCHES" in r.stdout and "hidden_reward" in r.stdout, r.stdout
    assert "-DSGK_BOAT_MOVEMENT_IN_HIDDEN=1 leaves 0 mismatches (RECONCILES the level)" in r.stdout, r.stdout
    assert [ln for ln in r.stdout.splitlines() if ln.startswith("IslandNavigation-v0")][0].endswith("MATCH")
    assert "levels matching: 1 of 2" in r.stdout
