"""Fused tabular-Q rollout per env and agent count. SGK_BENCH_KERNEL = auto | lds | hbm names the kernel (the library's own choice,
tables resident in LDS, rows in HBM); SGK_BENCH_SIZES=a,b,c the agent counts."""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "safe-grid-agents_amd")):
    sys.path.insert(0, p)
import safe_grid_agents_amd as S

tag = os.environ.get("SGK_BENCH_KERNEL", "auto")
names = [a for a in sys.argv[1:] if a.endswith("-v0")] or ["IslandNavigation-v0", "BoatRace-v0", "DistributionalShift-v0",
                                                           "WhiskyGold-v0", "AbsentSupervisor-v0"]  # the last two: HBM-resident rows
for name in names:
    for n in ([int(x) for x in os.environ["SGK_BENCH_SIZES"].split(",")] if os.environ.get("SGK_BENCH_SIZES") else (4096, 16384, 65536, 262144, 1048576)):
        args = types.SimpleNamespace(lr=0.5, discount=0.99, epsilon=0.01, epsilon_anneal=100000)
        env = S.BatchedGridworldEnv(name, n, seed=0x5AFE)
        agent = S.BatchedTabularQAgent(env, args)
        try:
            agent.rollout(200, kernel=tag)
        except S._lib.SgkError as e:  # (the LDS-resident kernel cannot serve this level)
            print(f"{tag} {name} n={n}: {e}", flush=True)
            agent.close(); env.close()
            break
        env.synchronize()
        t0 = time.perf_counter()
        agent.rollout(1000, kernel=tag)
        env.synchronize()
        dt = (time.perf_counter() - t0) / 1000
        print(f"{tag} {name} n={n}: {dt * 1e6:.2f} us/step = {n / dt:.3e} agent-steps/s", flush=True)
        agent.close(); env.close()
