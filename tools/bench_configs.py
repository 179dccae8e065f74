"""Throughput of the BASELINE.json configs 2-5 on ONE MI355X (runs on the GPU box). One JSON line per config.
Config 1 (single env, CPU reference path) and the bench.py headline are measured elsewhere."""
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "safe-grid-agents_amd")):
    sys.path.insert(0, p)
import torch

import safe_grid_agents_amd as S


def ev_time(env, fn, reps=1):
    s = env.torch_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    env.synchronize(); torch.cuda.synchronize()
    with torch.cuda.stream(s):  # (no cross-stream event hops between back-to-back library launches)
        e0.record(s)
        for _ in range(reps):
            fn()
        e1.record(s)
        env.synchronize(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 1e3 / reps


def wall_time(fn, reps=1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


out = []

# config 1: BoatRace + tabular-q, single env, the reference-shaped Python loop (`main.py boat tabular-q --lr .5`)
# (the CPU leg of this config lives in bench.py's cpu_baseline: only tests/, smoke() and that leg may touch oracle/)

def single_env_rate(env_factory, episodes=400):  # long enough that creating the env (~0.3 s) does not show
    args = S.prepare_parser().parse_args(["-S", "7", "-E", str(episodes), "-EE", "1000", "-V", "100", "-EV", "0",
                                          "boat", "tabular-q", "-l", ".5"])
    args.log_dir = None
    t0 = time.perf_counter()
    agent, hist, _ = S.train(args, env_factory=env_factory, writer_factory=lambda d: S.NullWriter(d))
    dt = time.perf_counter() - t0
    return (hist["t"] + 100) / dt  # + the final eval's 100 steps


r_gpu = single_env_rate(lambda name: S.make(name))
out.append({"config": 1, "what": "reference-shaped train() loop (boat tabular-q, 400 episodes) on the HIP single env (sgk_step_host served by the step server; PCIe-inclusive)",
            "env_steps_per_s": r_gpu})

# config 2: BoatRace random-action rollout, 65 536 envs lockstep on 1 MI355X (step kernel only)
n = 65536
env = S.BatchedGridworldEnv("BoatRace-v0", n, seed=0x5AFE, layout="compact")
env.step_random(200, auto_reset=True)
dt = ev_time(env, lambda: env.step_random(100, auto_reset=True), 20) / 100
out.append({"config": 2, "what": "BoatRace random rollout, 65536 envs, step kernel, hipGraph x100", "us_per_step": dt * 1e6,
            "env_steps_per_s": n / dt, "alg_GBs": 78 * n / dt / 1e9})
dt = ev_time(env, lambda: env.step_random(100, auto_reset=True, fused="stream"), 20) / 100
out.append({"config": 2, "what": "same, streaming rollout kernel (100 steps/launch, every step's boards + records materialised)",
            "us_per_step": dt * 1e6, "env_steps_per_s": n / dt, "alg_GBs": 78 * n / dt / 1e9})
dt = ev_time(env, lambda: env.step_random(1000, auto_reset=True, fused=True), 5) / 1000
out.append({"config": 2, "what": "same, fused rollout kernel (1000 steps/launch, outputs once per launch)", "us_per_step": dt * 1e6, "env_steps_per_s": n / dt})
env.close()

# config 5 (per GPU share): 1 048 576 envs over 8 GPUs = 131 072 envs per GPU
n = 131072
env = S.BatchedGridworldEnv("BoatRace-v0", n, seed=0x5AFE, layout="compact")
env.step_random(200, auto_reset=True)
dt = ev_time(env, lambda: env.step_random(100, auto_reset=True), 20) / 100
out.append({"config": 5, "what": "BoatRace random rollout, 131072 envs per GPU (1M over 8), step kernel, hipGraph x100",
            "us_per_step": dt * 1e6, "env_steps_per_s": n / dt, "alg_GBs": 78 * n / dt / 1e9})
dt = ev_time(env, lambda: env.step_random(100, auto_reset=True, fused="stream"), 20) / 100
out.append({"config": 5, "what": "same, streaming rollout kernel (100 steps/launch, every step's boards + records materialised)",
            "us_per_step": dt * 1e6, "env_steps_per_s": n / dt, "alg_GBs": 78 * n / dt / 1e9})
env.close()

# config 3: IslandNavigation + tabular-q, 262 144 envs (private agents)
n = 262144
args = types.SimpleNamespace(lr=0.5, discount=0.99, epsilon=0.01, epsilon_anneal=100000)
env = S.BatchedGridworldEnv("IslandNavigation-v0", n, seed=0x5AFE)
agent = S.BatchedTabularQAgent(env, args)
agent.rollout(100)
dt = ev_time(env, lambda: agent.rollout(1000), 3) / 1000
out.append({"config": 3, "what": "IslandNavigation + tabular-q, 262144 private agents, fused LDS-resident rollout (1000 steps/launch)",
            "us_per_step": dt * 1e6, "env_steps_per_s": n / dt, "alg_GBs": 196 * n / dt / 1e9})


def stepwise():
    a = agent.act_explore()
    env.step(a, auto_reset=False, write_boards=False)
    agent.learn(action=a)
    env.reset_done()


for _ in range(20):
    stepwise()
dt = wall_time(stepwise, 200)
out.append({"config": 3, "what": "same, drop-in call sequence act_explore/step/learn/reset_done made from Python (4 launches per step, HBM tables), "
                                 "the handle on its own stream: every call orders itself against torch's current stream (6 event record + wait pairs per step)",
            "us_per_step": dt * 1e6, "env_steps_per_s": n / dt, "alg_GBs": 196 * n / dt / 1e9})
agent.learn_steps(100)
dt = ev_time(env, lambda: agent.learn_steps(100), 10) / 100
out.append({"config": 3, "what": "same four launches per step replayed from one hipGraph per 100 steps (sgk_tabq_learn_steps)",
            "us_per_step": dt * 1e6, "env_steps_per_s": n / dt, "alg_GBs": 196 * n / dt / 1e9})
env.bind_torch_stream()  # the same four calls with the handle on torch's stream: no ordering calls (EXPERIMENTS R5.13)
for _ in range(20):
    stepwise()
dt = wall_time(stepwise, 200)
out.append({"config": 3, "what": "same four calls from Python after env.bind_torch_stream()",
            "us_per_step": dt * 1e6, "env_steps_per_s": n / dt, "alg_GBs": 196 * n / dt / 1e9})
agent.close(); env.close()

# config 4: SideEffectsSokoban + deep-q (reference MLP 36-100-100-4), 32 768 envs
n = 32768
dargs = types.SimpleNamespace(discount=0.99, lr=1e-3, batch_size=64, sync_every=10000, epsilon=0.01, epsilon_anneal=100000,
                              n_layers=2, n_hidden=100)
env = S.BatchedGridworldEnv("SideEffectsSokoban-v0", n, seed=0x5AFE, layout="compact")
env.bind_torch_stream()
dq = S.BatchedDeepQAgent(env, dargs, sgd_steps=1, replay_slices=8)
dq.warmup(8)
for _ in range(20):
    dq.step(learn=False)
dt = wall_time(lambda: dq.step(learn=False), 300)
out.append({"config": 4, "what": "Sokoban + deep-q MLP, 32768 envs: obs kernel + policy forward + eps-greedy + env.step + reset_done (no learning)",
            "us_per_step": dt * 1e6, "env_steps_per_s": n / dt})
dq.act_rollout(100, epsilon=0.01)
dt = wall_time(lambda: dq.act_rollout(1000, epsilon=0.01), 3) / 1000
out.append({"config": 4, "what": "acting with frozen weights, fused rollout: 1000 x {forward + eps-greedy + env.step + auto-reset} per launch (sgk_policy_rollout)",
            "us_per_step": dt * 1e6, "env_steps_per_s": n / dt})
for _ in range(20):
    dq.step(learn=True)
dt = wall_time(lambda: dq.step(learn=True), 300)
out.append({"config": 4, "what": "same with learning: 1 SGD step (batch 64, Adam amsgrad) per lockstep step, replay 8 x 32768 transitions",
            "us_per_step": dt * 1e6, "env_steps_per_s": n / dt})
dq.enable_graphs(learn=False)
dq.enable_graphs(learn=True)
for _ in range(20):
    dq.step_graphed(learn=False)
dt = wall_time(lambda: dq.step_graphed(learn=False), 500)
out.append({"config": 4, "what": "no learning, the whole lockstep iteration replayed from one hipGraph (torch.cuda.CUDAGraph incl. the library's kernels)",
            "us_per_step": dt * 1e6, "env_steps_per_s": n / dt})
for _ in range(20):
    dq.step_graphed(learn=True)
dt = wall_time(lambda: dq.step_graphed(learn=True), 500)
out.append({"config": 4, "what": "with learning (1 SGD step, batch 64, Adam amsgrad capturable), one hipGraph per lockstep iteration",
            "us_per_step": dt * 1e6, "env_steps_per_s": n / dt})
env.close()

for o in out:
    print(json.dumps(o), flush=True)
