"""The drop-in tabular-Q call sequence (act_explore -> env.step -> learn -> reset_done: four launches per lockstep step) replayed from the
library's hipGraph (sgk_tabq_learn_steps), us per lockstep step at several agent counts; the fused rollout beside it; and the four calls
made from Python (host clock), with the handle on its own stream (every call then orders itself against torch's current stream: an event
record + wait each way) and bound to torch's stream (bind_torch_stream: no ordering calls)."""
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
        env = S.BatchedGridworldEnv(name, n, seed=0x5AFE)
        agent = S.BatchedTabularQAgent(env, args)
        g = timed(env, lambda: agent.learn_steps(100), 10) / 100
        f = timed(env, lambda: agent.rollout(1000), 3) / 1000

        def calls():
            a = agent.act_explore()
            env.step(a, auto_reset=False, write_boards=False)
            agent.learn(action=a)
            env.reset_done()

        def wall(reps=300):
            for _ in range(30):
                calls()
            env.synchronize(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                calls()
            env.synchronize(); torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps * 1e6

        own = wall()
        env.bind_torch_stream()
        bound = wall()
        print("%-22s n=%7d  four launches per step (hipGraph) %6.2f us | fused rollout %6.3f us per lockstep step | four calls from Python: "
              "%6.1f us (own stream), %6.1f us (bound to torch's stream)" % (name, n, g, f, own, bound), flush=True)
        agent.close(); env.close()
