"""Batched PPO timing on one GPU: rollout gather (env-steps/s, us per lockstep step) and the epochs, per env / body.
Run on the GPU box: python tools/bench_ppo.py (log kept under profiles/rNN/bench_ppo.log)."""
import sys, os, types, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "safe-grid-agents_amd"))
import torch
import safe_grid_agents_amd as S

def run(name, n, body, hidden=100):
    torch.manual_seed(0)
    env = S.BatchedGridworldEnv(name, n, seed=5)
    env.bind_torch_stream()
    a = types.SimpleNamespace(discount=0.99, lr=1e-3, batch_size=4096, rollouts=1, epochs=16, clipping=0.2, entropy_bonus=0.0,
                              critic_coeff=1e-4, n_layers=2, n_hidden=hidden, n_channels=5, device=0, log_gradients=False, cheat=False)
    agent = S.BatchedPPOAgent(env, a, body=body)
    learn_ms = {}
    for graphed in (False, True):
        agent.graph_epochs = graphed
        for it in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ro = agent.gather_rollout()
            torch.cuda.synchronize(); t1 = time.perf_counter()
            agent.learn(ro); agent.sync()
            torch.cuda.synchronize(); t2 = time.perf_counter()
        learn_ms[graphed] = 1e3 * (t2 - t1)
    steps = int(ro.lengths.sum().item())
    print(f"{name} n={n} body={body} fused={agent.fused_policy}: gather {1e3*(t1-t0):.1f} ms ({steps/(t1-t0):.3e} env-steps/s, "
          f"{1e6*(t1-t0)/ro.actions.shape[0]:.1f} us/lockstep), learn (16 epochs x 4096 rows) eager {learn_ms[False]:.1f} ms, "
          f"one hipGraph {learn_ms[True]:.2f} ms", flush=True)
    env.close()

def run_reference_batch(name, n, hidden=100, epochs=16):
    """The reference's own minibatch size (batch-size 64, its default): learn() as ONE kernel (sgk_ppo_epochs) against the
    same update as torch launches, eager and as one hipGraph."""
    torch.manual_seed(0)
    env = S.BatchedGridworldEnv(name, n, seed=5)
    env.bind_torch_stream()
    a = types.SimpleNamespace(discount=0.99, lr=1e-3, batch_size=64, rollouts=1, epochs=epochs, clipping=0.2, entropy_bonus=0.01,
                              critic_coeff=1.0, n_layers=2, n_hidden=hidden, n_channels=5, device=0, log_gradients=False, cheat=False)
    agent = S.BatchedPPOAgent(env, a)
    ms = {}
    for mode in ("eager", "graph", "kernel"):
        agent.fused_learn = mode == "kernel"
        agent.graph_epochs = mode != "eager"
        for it in range(5):
            ro = agent.gather_rollout()
            torch.cuda.synchronize(); t1 = time.perf_counter()
            agent.learn(ro)
            torch.cuda.synchronize(); t2 = time.perf_counter()
            agent.sync()
        ms[mode] = 1e3 * (t2 - t1)
    print(f"{name} n={n} H={hidden}: learn ({epochs} epochs x 64 rows) eager {ms['eager']:.2f} ms, one hipGraph {ms['graph']:.2f} ms, "
          f"sgk_ppo_epochs {ms['kernel']:.3f} ms ({1e3 * ms['kernel'] / epochs:.1f} us/epoch)", flush=True)
    env.close()

def run_unfused_gather(name, n, body, hidden=100, fused_conv=False, rollout=False):
    """Bodies without a fused ROLLOUT kernel: the T-step gather loop eager vs replayed from one hipGraph. fused_conv: ppo-cnn's trunk +
    actor forward + draw as one launch per step (sgk_convq_sample) instead of the torch module + sgk_categorical_sample."""
    ms = {}
    for graphed in (False, True):
        torch.manual_seed(0)
        env = S.BatchedGridworldEnv(name, n, seed=5)
        env.bind_torch_stream()
        a = types.SimpleNamespace(discount=0.99, lr=1e-3, batch_size=64, rollouts=1, epochs=4, clipping=0.2, entropy_bonus=0.01,
                                  critic_coeff=1.0, n_layers=2, n_hidden=hidden, n_channels=5, device=0, log_gradients=False, cheat=False)
        agent = S.BatchedPPOAgent(env, a, body=body, fused_conv=fused_conv)
        agent.fused_rollout = rollout  # (True: the whole gather is one sgk_convq_rollout launch; `graphed` then changes nothing)
        agent.graph_gather = graphed
        for it in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ro = agent.gather_rollout()
            torch.cuda.synchronize(); t1 = time.perf_counter()
        ms[graphed] = 1e3 * (t1 - t0)
        env.close()
    print(f"{name} n={n} body={body} H={hidden} {('sgk_convq_rollout' if rollout else 'sgk_convq_sample') if fused_conv else 'unfused'} gather: eager {ms[False]:.1f} ms "
          f"({1e3 * ms[False] / ro.actions.shape[0]:.1f} us/lockstep), one hipGraph {ms[True]:.1f} ms "
          f"({1e3 * ms[True] / ro.actions.shape[0]:.1f} us/lockstep)", flush=True)

for name in ("BoatRace-v0", "SideEffectsSokoban-v0", "DistributionalShift-v0"):  # ppo-cnn's gather: the fused conv kernels against the torch module
    run_unfused_gather(name, 32768, "cnn", fused_conv=True, rollout=True)
    run_unfused_gather(name, 32768, "cnn", fused_conv=True)
    run_unfused_gather(name, 32768, "cnn", fused_conv=False)
if len(sys.argv) > 1 and sys.argv[1] == "cnn":
    sys.exit(0)
for n in (1024, 32768):
    run_unfused_gather("BoatRace-v0", n, "cnn")
    run_unfused_gather("BoatRace-v0", n, "mlp", hidden=32)
for name in ("BoatRace-v0", "SideEffectsSokoban-v0", "IslandNavigation-v0", "DistributionalShift-v0"):
    run_reference_batch(name, 32768)
run_reference_batch("BoatRace-v0", 32768, hidden=64)
run_reference_batch("BoatRace-v0", 32768, epochs=1)
for n in (4096, 32768, 262144):
    run("BoatRace-v0", n, "mlp")
run("BoatRace-v0", 32768, "mlp", hidden=64)
run("BoatRace-v0", 32768, "cnn")
run("SideEffectsSokoban-v0", 32768, "mlp")
run("IslandNavigation-v0", 32768, "mlp")
