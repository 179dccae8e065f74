"""Arguments that reach a kernel as raw pointers are checked with exceptions (ValueError), not asserts: a tensor of the wrong size,
dtype or device never becomes a wild pointer, also under `python -O`. And the wrapper's default stream handling: the library enqueues
on torch's CURRENT stream, whatever that is at the time of the call."""
import os
import subprocess
import sys
import types

import numpy as np
import pytest

import safe_grid_agents_amd as S
from oracle import oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHECKS = r'''
import types, torch, numpy as np
import safe_grid_agents_amd as S

n = 256
env = S.BatchedGridworldEnv("SideEffectsSokoban-v0", n, seed=3)
dev = "cuda:0"
def bad(fn, *a, **kw):
    try:
        fn(*a, **kw)
    except ValueError as e:
        return str(e)
    raise SystemExit("no ValueError from %s%r" % (getattr(fn, "__name__", fn), tuple(type(x).__name__ for x in a)))

ok_actions = torch.zeros(n, dtype=torch.uint8, device=dev)
for wrong in (torch.zeros(n - 1, dtype=torch.uint8, device=dev), torch.zeros(n + 1, dtype=torch.uint8, device=dev),
              torch.zeros(n, dtype=torch.float32, device=dev), torch.zeros(n, dtype=torch.bool, device=dev),
              torch.zeros(n, dtype=torch.uint8), torch.zeros((n, 2), dtype=torch.uint8, device=dev)[:, 0][: n - 3]):
    bad(env.step, wrong)
    bad(env.step_repeat, wrong, 3)
if torch.cuda.device_count() >= 2:
    assert "cuda:1" in bad(env.step, torch.zeros(n, dtype=torch.uint8, device="cuda:1"))
env.step(torch.zeros(n, dtype=torch.int64, device=dev))       # another integer width is narrowed (argmax gives int64)
env.step(np.zeros(n, dtype=np.int64))                           # a host array is uploaded
env.step(torch.zeros((n, 2), dtype=torch.uint8, device=dev)[:, 0])  # a strided view is made dense
bad(env.reset, torch.zeros(n - 1, dtype=torch.uint8, device=dev))
bad(env.reset, torch.zeros(n, dtype=torch.float32, device=dev))
# scores / logits
bad(env.epsilon_greedy, torch.zeros((n, 3), device=dev), 0.1, 0)
bad(env.epsilon_greedy, torch.zeros((n, 4), dtype=torch.float64, device=dev), 0.1, 0)
bad(env.epsilon_greedy, torch.zeros((n, 4)), 0.1, 0)
bad(env.epsilon_greedy, torch.zeros((n, 4), device=dev), torch.zeros(1, dtype=torch.float32, device=dev), 0)   # epsilon scalar tensor: float64
bad(env.epsilon_greedy, torch.zeros((n, 4), device=dev), 0.1, torch.zeros(1, dtype=torch.int32, device=dev))   # draw index: int64
bad(env.epsilon_greedy, torch.zeros((n, 4), device=dev), 0.1, 0, out=torch.zeros(n - 1, dtype=torch.uint8, device=dev))
bad(env.categorical_sample, torch.zeros((n + 1, 4), device=dev), 0)
# fused policy weights
h = 100
w = {"w1t": torch.zeros((env.n_cells, h), device=dev), "b1": torch.zeros(h, device=dev), "w2": torch.zeros((h, h), device=dev),
     "b2": torch.zeros(h, device=dev), "w3t": torch.zeros((h, 4), device=dev), "b3": torch.zeros(4, device=dev)}
env.policy_act(w, 0.0, 0)
for k, wrong in (("w1t", torch.zeros((env.n_cells + 1, h), device=dev)), ("w2", torch.zeros((h, h), dtype=torch.float64, device=dev)),
                 ("w3t", torch.zeros((4, h), device=dev).t()), ("b3", torch.zeros(4))):
    bad(env.policy_act, dict(w, **{k: wrong}), 0.0, 0)
    bad(env.policy_sample, dict(w, **{k: wrong}), 0)
    bad(env.policy_rollout, dict(w, **{k: wrong}), 5)
bad(env.policy_act, {k: v for k, v in w.items() if k != "b2"}, 0.0, 0)
bad(env.policy_rollout, w, 5, states=torch.zeros((5, n, env.n_cells + 1), dtype=torch.int8, device=dev))
bad(env.policy_rollout, w, 5, actions=torch.zeros((5, n), dtype=torch.int8, device=dev))
bad(env.policy_rollout, w, 5, mode="argmax")
# trajectory rings
bad(env.rollout_random_stream, 4, boards=torch.zeros((4, n, env.n_cells + 1), dtype=torch.int8, device=dev))
bad(env.rollout_random_stream, 4, boards=torch.zeros((4, n, env.n_cells), dtype=torch.uint8, device=dev))
bad(env.rollout_random_stream, 4, boards=torch.zeros((4, n, env.n_cells), dtype=torch.int8))
bad(env.rollout_random_stream, 4, boards=torch.zeros((4, n, env.n_cells), dtype=torch.int8, device=dev),
    recs=torch.zeros((5, n, 4), dtype=torch.int8, device=dev))
bad(env.rollout_random_stream, 4, recs=torch.zeros((4, n, 4), dtype=torch.int8, device=dev), layout="tile")
bad(env.obs_f32, torch.zeros((n, env.n_cells), dtype=torch.float64, device=dev))
bad(env.discounted_returns, torch.zeros((3, 7), dtype=torch.float64, device=dev), 0.9)
# the tabular agent's action argument
agent = S.BatchedTabularQAgent(env, types.SimpleNamespace(lr=0.5, discount=0.9, epsilon=0.1, epsilon_anneal=10))
agent.act_explore()
bad(agent.learn, action=torch.zeros(n - 1, dtype=torch.uint8, device=dev))
bad(agent.learn, action=torch.zeros(n, dtype=torch.float32, device=dev))
# ... and the handle still works after every refusal
env.step(ok_actions)
try:
    S.BatchedGridworldEnv("BoatRace-v0", 8, stream="mine")
    raise SystemExit("no ValueError for an unknown stream mode")
except ValueError:
    pass
single = S.make("BoatRace-v0")
single.reset()
try:
    single.step(7)
    raise SystemExit("no AssertionError for an invalid action")
except AssertionError as e:
    assert "Not a valid action" in str(e)
print("CHECKS_OK optimised=%d" % (not __debug__))
'''


@pytest.mark.parametrize("flag", [None, "-O"], ids=["python", "python-O"])
def test_wrong_tensors_raise_value_error_also_without_asserts(flag):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "safe-grid-agents_amd")]))
    cmd = [sys.executable] + ([flag] if flag else []) + ["-c", CHECKS]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert ("CHECKS_OK optimised=%d" % (1 if flag else 0)) in r.stdout


def test_default_stream_mode_follows_torchs_current_stream():
    """No bind call: the library enqueues on torch's current stream of the device -- the default stream first, then a side stream
    inside `with torch.cuda.stream(...)`, then the default stream again -- and work handed from one to the other stays ordered (the
    library orders a stream switch with an event). The trajectory is the oracle's."""
    import torch

    n, seed = 20000, 11
    env = S.BatchedGridworldEnv("BoatRace-v0", n, seed=seed)
    assert env._mode == "follow" and env._bound
    orc = O.EnvBatch("BoatRace-v0", n)
    acts = torch.as_tensor(O.random_actions(seed, 0, 64, 0, 1)[0].repeat(n // 64 + 1)[:n].copy(), device="cuda")
    side = torch.cuda.Stream()
    host_acts = acts.cpu().numpy()
    for k in range(30):
        a = ((acts + k) % 4).to(torch.uint8)                     # made on the stream the step will be enqueued on
        if k % 3 == 1:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                a2 = (a + 0).contiguous()
                env.step(a2, auto_reset=True)
                assert env.stream_ptr == side.cuda_stream
            torch.cuda.current_stream().wait_stream(side)
        else:
            env.step(a, auto_reset=True)
            assert env.stream_ptr == torch.cuda.current_stream().cuda_stream
        orc.rollout(1, actions=((host_acts + k) % 4).astype(np.uint8)[None], auto_reset=True)
    assert (env.boards_host().reshape(n, -1) == orc.boards()).all()
    assert (env.episode_state_host()["episode_return"] == orc.field("episode_return")).all()
    # a private stream on request, and back
    env.use_own_stream()
    assert env._mode == "own" and not env._bound and env.stream_ptr not in (0, side.cuda_stream)
    env.step(acts.to(torch.uint8), auto_reset=True)
    env.bind_torch_stream(side)
    assert env._mode == "pinned" and env.stream_ptr == side.cuda_stream
    env.bind_torch_stream()
    assert env._mode == "follow"
    env.step(acts.to(torch.uint8), auto_reset=True)
    orc.rollout(2, actions=np.stack([host_acts.astype(np.uint8)] * 2), auto_reset=True)
    assert (env.boards_host().reshape(n, -1) == orc.boards()).all()
    env.close()


def test_tables_zeroed_on_one_stream_are_ready_on_the_stream_the_handle_moves_to():
    """sgk_tabq_create zeroes the tables on the stream the handle has then; a launch on the stream it is moved to afterwards waits
    for that (the switch is ordered by an event): 1.6 GB of tables, the rollout enqueued immediately behind the move."""
    n = 1 << 20
    env = S.BatchedGridworldEnv("IslandNavigation-v0", n, seed=21, stream="own")
    agent = S.BatchedTabularQAgent(env, types.SimpleNamespace(lr=0.5, discount=0.99, epsilon=0.01, epsilon_anneal=100000))
    env.bind_torch_stream()  # the NULL stream from here on
    agent.rollout(21)
    tab = agent.table_host(0, 256)
    orc = O.EnvBatch("IslandNavigation-v0", 256, seed=21)
    agents = [O.TabQ(orc.H * orc.W, 0.5, 0.99, 0.01, 100000) for _ in range(256)]
    O.tabq_rollout(orc, agents, 21, seed=21)
    assert (env.boards_host().reshape(n, -1)[:256] == orc.boards()).all()
    assert np.isfinite(tab).all() and np.abs(tab).max() < 1e3
    agent.close(); env.close()
