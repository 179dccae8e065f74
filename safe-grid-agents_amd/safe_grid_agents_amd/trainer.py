"""`train(args)` and the CLI surface of the reference, for the hot-path scope.

train():            reference train.py:21-81 (seeding, object wiring, warm-up, episode loop, eval cadence).
prepare_parser():   reference parsing/parse.py:91-124 + the three YAML files, restated as a table: same
                    positional grammar `<core flags> <env alias> <agent alias> <agent flags>`, same flag names,
                    aliases and defaults (core_parser_configs.yaml:1-47, agent_parser_configs.yaml:1-63).
"""
import argparse
import random

import numpy as np

from . import envs as _envs
from .agents import AGENT_MAP
from .loops import EVAL_MAP, LEARN_MAP, WARMUP_MAP
from .metering import NullWriter, make_meters

ENV_MAP = _envs.ENV_MAP

_CORE_FLAGS = [
    # (long, alias, kwargs)
    ("seed", "S", dict(type=int)),
    ("episodes", "E", dict(type=int, default=2000)),
    ("eval-timesteps", "V", dict(type=int, default=2000)),
    ("eval-every", "EE", dict(type=int, default=100)),
    ("eval-visualize-episodes", "EV", dict(type=int, default=4)),
    ("discount", "D", dict(type=float, default=0.99)),
    ("cheat", "C", dict(action="store_true")),
    ("log-dir", "L", dict(type=str)),
    ("disable-cuda", "dc", dict(action="store_true")),
    # extension: N > 0 runs N envs with N private agents in lockstep on the GPU (train_batched)
    ("n-envs", "N", dict(type=int, default=0)),
    # extension: shard the -N env batch over this many GPUs of the node, one process per GPU (contiguous env-id blocks, one
    # RCCL metrics all-reduce per reporting period). `python -m safe_grid_agents_amd --devices G ...` starts the G ranks itself;
    # under torch.distributed.run the launcher's WORLD_SIZE decides and this flag is only checked against it.
    ("devices", "G", dict(type=int, default=1)),
]
_LR = ("lr", "l", dict(type=float, required=True))
_EPS = ("epsilon", "e", dict(type=float, default=0.01))
_ANNEAL = ("epsilon-anneal", "dl", dict(type=int, default=100000))
_BATCH = ("batch-size", "b", dict(type=int, default=64))
_LAYERS = ("n-layers", "ls", dict(type=int, default=2))
_DEVICE = ("device", "dv", dict(type=int, default=0))
_GRADLOG = ("log-gradients", "lg", dict(action="store_true"))
_PPO_FLAGS = [_LR,
              ("rollouts", "r", dict(type=int, required=True)),
              ("epochs", "e", dict(type=int, required=True)),
              _BATCH,
              ("clipping", "c", dict(type=float, default=0.2)),
              ("critic-coeff", "cc", dict(type=float, default=1.0)),
              ("entropy-bonus", "eb", dict(type=float, default=0.01)),
              _LAYERS]
_AGENT_FLAGS = {
    "random": [],
    "single": [("action", "a", dict(type=int, default=0))],
    "tabular-q": [_LR, _EPS, _ANNEAL],
    "deep-q": [_LR, _EPS, _ANNEAL,
               ("replay-capacity", "r", dict(type=int, default=10000)),
               ("sync-every", "s", dict(type=int, default=10000)),
               ("n-layers", "ls", dict(type=int, default=2)),
               ("n-hidden", "hd", dict(type=int, default=100)),
               ("batch-size", "b", dict(type=int, default=64)),
               ("device", "dv", dict(type=int, default=0)),
               ("log-gradients", "lg", dict(action="store_true")),
               # extension, batched trainer (-N) only, NOT the reference's DeepQAgent (an MLP: value.py:148-158): a convolutional
               # Q-body built like policy_cnn.py:17-81, through PyTorch-ROCm; no parity claim (BASELINE config 4 says "conv policy")
               ("q-body", "qb", dict(type=str, default="mlp", choices=("mlp", "cnn"),
                                     help="mlp: the reference's network (default, pinned); cnn: non-parity conv body, -N only")),
               ("n-channels", "ch", dict(type=int, default=5))],
    "ppo-mlp": _PPO_FLAGS + [("n-hidden", "hd", dict(type=int, default=100)), _DEVICE, _GRADLOG],
    "ppo-cnn": [("n-channels", "ch", dict(type=int, default=5))] + _PPO_FLAGS + [_DEVICE, _GRADLOG],
}


def _add(parser, flags):
    for long, alias, kw in flags:
        parser.add_argument("-" + alias, "--" + long, **kw)


def prepare_parser():
    parser = argparse.ArgumentParser(description="Safety gridworld agents on MI355X (safe-grid-agents CLI surface)")
    _add(parser, _CORE_FLAGS)
    env_sub = parser.add_subparsers(dest="env_alias", help="gridworld environment")
    env_sub.required = True
    for env_alias in ENV_MAP:
        env_parser = env_sub.add_parser(env_alias)
        agent_sub = env_parser.add_subparsers(dest="agent_alias", help="agent")
        agent_sub.required = True
        for agent_alias, flags in _AGENT_FLAGS.items():
            _add(agent_sub.add_parser(agent_alias), flags)
    return parser


def _noop(*args, **kwargs):
    pass


def _default_writer(log_dir):
    """tensorboardX.SummaryWriter when the user has it (reference train.py:14,44); else this repo's own event-file writer
    (same calls, files TensorBoard reads); a null writer when there is no log directory."""
    try:
        from tensorboardX import SummaryWriter  # not in this image

        return SummaryWriter(log_dir)
    except ImportError:
        if not log_dir:
            return NullWriter(log_dir)
        from .eventfile import EventFileWriter

        return EventFileWriter(log_dir)


def train(args, config=None, reporter=_noop, env_factory=None, writer_factory=None):
    """One training run, exactly the reference's control flow. `env_factory(name)` defaults to the HIP-backed
    `make`; `writer_factory(log_dir)` defaults to tensorboardX when present, else a null writer."""
    import torch

    if config is not None:
        vars(args).update(config)
    random.seed(args.seed)
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)

    env_name = ENV_MAP[args.env_alias]
    agent_class = AGENT_MAP[args.agent_alias]
    warmup_fn = WARMUP_MAP[args.agent_alias]
    learn_fn = LEARN_MAP[args.agent_alias]
    eval_fn = EVAL_MAP[args.agent_alias]

    history, eval_history = make_meters({}), make_meters({})
    writer = (writer_factory or _default_writer)(getattr(args, "log_dir", None))
    for key, value in vars(args).items():
        writer.add_text("data/{}".format(key), str(value))
    history["writer"] = eval_history["writer"] = writer

    env = (env_factory or _envs.make)(env_name)
    env.seed(args.seed)
    agent = agent_class(env, args)
    agent, env, history, args = warmup_fn(agent, env, history, args)

    history["t"], history["t_learn"] = 0, 0
    history["episode"], eval_history["period"] = 0, 0
    for _ in range(args.episodes):
        env_state = (env.reset(), 0.0, False, {"hidden_reward": 0.0, "observed_reward": 0.0})
        history["episode"] += 1
        env_state, history, eval_next = learn_fn(agent, env, env_state, history, args)
        info = env_state[3]
        reporter(hidden_reward=info["hidden_reward"], obs_reward=info["observed_reward"])
        if eval_next:
            eval_history = eval_fn(agent, env, eval_history, args)
    eval_history = eval_fn(agent, env, eval_history, args)
    return agent, history, eval_history


def train_batched(args, writer_factory=None, reporter=_noop):
    """The same experiment for `args.n_envs` independent (env, agent) pairs in lockstep, on one GPU -- or, launched by
    torch.distributed.run with one process per GPU, sharded over the GPUs of a node: rank r owns the contiguous env-id block
    dist.shard_range gives it (the counter RNG is keyed by global env id, so the trajectories do not depend on the number of
    ranks), nothing but the 16-word metrics vector crosses xGMI (one RCCL all-reduce per reporting period), rank 0 writes.
    The shared-policy agents (ppo-*) would need a gradient all-reduce and stay single-GPU.

    Keeps train()'s cadence in units of lockstep steps: one "episode" = `max_iterations` steps (every env finishes at least
    one episode in that span), an evaluation (greedy, batched_default_eval) after every `eval_every` of them and once at
    the end; metrics are the aggregate meters (BatchMetrics) written under the reference's tensorboard tags. Supports the
    agents whose learning runs on the device: tabular-q (private tables), deep-q (one shared Q-network, one SGD step per
    lockstep step), ppo-mlp / ppo-cnn (one shared policy; an "episode" is one PPO iteration = one episode per env + the
    epochs) and random."""
    import os

    from . import dist as sdist
    from .agents import BatchedTabularQAgent
    from .loops import batched_default_eval, batched_ppo_learn
    from .ppo import BatchedPPOAgent

    rank, local_rank, world = sdist.env_from_torchrun()
    if getattr(args, "devices", 1) > 1 and world != args.devices:
        raise SystemExit("--devices %d needs %d ranks (WORLD_SIZE is %d): run `python -m safe_grid_agents_amd --devices %d ...`, "
                         "which starts them, or torch.distributed.run --nproc-per-node %d" % (
                             args.devices, args.devices, world, args.devices, args.devices))
    if world > 1:
        if args.agent_alias in ("ppo-mlp", "ppo-cnn"):
            raise KeyError("train_batched shards independent agents (tabular-q, random) over GPUs; %r shares one policy"
                           % (args.agent_alias,))
        sdist.init_process_group(os.environ.get("SGK_DIST_BACKEND"))  # nccl (= RCCL) unless the test knob says gloo
    if os.environ.get("SGK_BENCH_ONE_DEVICE") == "1":
        local_rank = 0  # test knob: every rank on the one GPU of the box
    begin, end = sdist.shard_range(args.n_envs, rank, world)
    env_name = ENV_MAP[args.env_alias]
    writer = (writer_factory or _default_writer)(getattr(args, "log_dir", None)) if rank == 0 else NullWriter(None)
    for key, value in vars(args).items():
        writer.add_text("data/{}".format(key), str(value))
    env = _envs.make(env_name, n_envs=end - begin, seed=args.seed or 0, device=local_rank, env_index_base=begin)
    horizon = int(env.info.max_iterations)
    sdist.library_comm(env)  # (several ranks under nccl) the metrics all-reduce's RCCL communicator, before the first flush
    if args.agent_alias == "tabular-q":
        agent = BatchedTabularQAgent(env, args)
    elif args.agent_alias in ("ppo-mlp", "ppo-cnn"):
        import torch

        torch.manual_seed(args.seed or 0)
        agent = BatchedPPOAgent(env, args, body=args.agent_alias[4:])
    elif args.agent_alias == "deep-q":
        import torch

        from .deepq_batched import BatchedDeepQAgent

        if world > 1:
            raise KeyError("train_batched shards independent agents over GPUs; deep-q shares one network")
        torch.manual_seed(args.seed or 0)
        # the replay holds whole lockstep slices: as many as cover the reference's capacity, at least 2
        slices = max(2, -(-int(args.replay_capacity) // env.n_envs))
        agent = BatchedDeepQAgent(env, args, sgd_steps=1, replay_slices=slices)
        agent.warmup(slices)  # dqn_warmup (warmup.py:8-23): random-action transitions fill the replay
        env.reset()
    elif args.agent_alias == "random":
        agent = None
    else:
        raise KeyError("train_batched supports tabular-q, deep-q, ppo-mlp, ppo-cnn and random, not %r" % (args.agent_alias,))
    ppo = isinstance(agent, BatchedPPOAgent)
    deepq = args.agent_alias == "deep-q"
    history = {"writer": writer, "t": 0, "t_learn": 0}
    period = 0
    for episode in range(1, args.episodes + 1):
        if ppo:
            bm = batched_ppo_learn(agent, env, history, cheat=args.cheat)
            history["t"] += horizon
        else:
            env.metrics_reset()
            if agent is None:
                env.step_random(horizon, auto_reset=True)
            elif deepq:
                for _ in range(horizon):  # dqn_learn for every env: act_explore, step, replay add, one SGD step, epsilon, sync
                    agent.step(learn=True, cheat=args.cheat)
            else:
                agent.rollout(horizon, cheat=args.cheat)
                if hasattr(agent, "check_hash_overflow"):
                    agent.check_hash_overflow()  # hashed tables (tomato watering): a full table stops the run at THIS period
            bm = sdist.global_metrics(env)  # this shard's metrics, all-reduced over the ranks when there are several
        bm.write(writer, episode, prefix="Train/")
        if agent is not None and not ppo:
            writer.add_scalar("Train/epsilon", agent.epsilon, agent.t)
        if deepq and agent.last_loss is not None:
            writer.add_scalar("Train/value_loss", float(agent.last_loss.reshape(-1)[0]), agent.t)
        reporter(hidden_reward=bm.meter("safeties")["avg"], obs_reward=bm.meter("returns")["avg"])
        # evaluation cadence exactly as the reference's loops decide it: whiler (tabular-q / deep-q, learn.py:21-22) evaluates
        # after the episodes with episode % eval_every == eval_every - 1; ppo_learn (learn.py:100) after those with
        # episode % eval_every == 0 (episode > 0)
        if ppo:
            eval_next = episode > 0 and episode % args.eval_every == 0
        else:
            eval_next = episode % args.eval_every == args.eval_every - 1
        if agent is not None and eval_next:
            period = _batched_eval(agent, env, args, writer, period, rank, sdist)
    if agent is not None:  # train.py:81: one more evaluation after the loop, unconditionally (also when the last episode had one)
        period = _batched_eval(agent, env, args, writer, period, rank, sdist)
    return agent, env


def _batched_eval(agent, env, args, writer, period, rank, sdist):
    from .loops import batched_default_eval

    if rank == 0:
        print("#### EVAL ####")
    if hasattr(agent, "check_hash_tables"):
        agent.check_hash_tables()  # levels without a perfect hash of their boards: a full table is an error, not a silent state
    batched_default_eval(agent, env, args.eval_timesteps)
    sdist.global_metrics(env).write(writer, period, prefix="Evaluation/")
    env.reset()
    return period + 1
