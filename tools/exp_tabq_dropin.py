"""What a caller of the per-step tabular-Q API pays per lockstep step, at several agent counts (us):
  graph/1   sgk_tabq_learn_steps: a hipGraph of the ONE-launch step (sgk_tabq_step's kernel), device time per step (SGK_F_NO_BOARDS; and
            with every step's boards written)
  graph/4   the same with SGK_F_SEPARATE_LAUNCHES: act_explore -> env.step -> learn -> reset_done, four launches per step (rounds 2-5)
  fused     the fused rollout (agent.rollout), device time per step
  py/4      the four calls made from Python, host clock: with the wrapper's DEFAULT stream handling (torch's current stream, no extra
            call), pinned to a side torch stream, and on the handle's own stream (event record + wait each way)
  py/1      agent.step() -- the one-launch step -- called from Python, default stream handling"""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "safe-grid-agents_amd")):
    sys.path.insert(0, p)
import torch
import safe_grid_agents_amd as S


def timed(env, fn, reps):
    st = env.torch_stream()
    with torch.cuda.stream(st):
        fn(); env.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps):
            fn()
        e1.record(st)
        env.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


args = types.SimpleNamespace(lr=0.5, discount=0.99, epsilon=0.01, epsilon_anneal=100000)
for name in (sys.argv[1:] or ["IslandNavigation-v0", "BoatRace-v0"]):
    for n in (1024, 65536, 262144):
        env = S.BatchedGridworldEnv(name, n, seed=0x5AFE)  # default: follows torch's current stream
        agent = S.BatchedTabularQAgent(env, args)
        g1 = timed(env, lambda: agent.learn_steps(100), 10) / 100
        g1b = timed(env, lambda: agent.learn_steps(100, write_boards=True), 10) / 100
        g4 = timed(env, lambda: agent.learn_steps(100, separate_launches=True), 10) / 100
        f = timed(env, lambda: agent.rollout(1000), 3) / 1000

        def calls():
            a = agent.act_explore()
            env.step(a, auto_reset=False, write_boards=False)
            agent.learn(action=a)
            env.reset_done()

        def one():
            agent.step(write_boards=False)

        def wall(fn, reps=300):
            for _ in range(30):
                fn()
            env.synchronize(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            env.synchronize(); torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps * 1e6

        default4, default1 = wall(calls), wall(one)
        side = torch.cuda.Stream()
        env.bind_torch_stream(side)
        with torch.cuda.stream(side):
            pinned4 = wall(calls)
        env.use_own_stream()
        own4 = wall(calls)
        print("%-20s n=%7d  graph/1 %6.2f (with boards %6.2f) | graph/4 %6.2f | fused %6.3f || py/4 default %6.1f, pinned %6.1f, own stream %6.1f | py/1 default %6.1f"
              % (name, n, g1, g1b, g4, f, default4, pinned4, own4, default1), flush=True)
        agent.close(); env.close()
