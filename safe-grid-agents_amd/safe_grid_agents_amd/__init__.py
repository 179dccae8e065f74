"""safe_grid_agents_amd -- MI355X-native gridworld step / agent-rollout path behind the safe-grid-gym Env API and
the safe-grid-agents agent / loop API. The compute path is libsgk.so (hand-written HIP for gfx950, C-ABI in
include/sgk.h); importing an env without it raises. Nothing here falls back to the CPU.
"""
from . import _lib
from .agents import (AGENT_MAP, BatchedTabularQAgent, DeepQAgent, Experience, ExperienceBatch, RandomAgent,
                     ReplayBuffer, Rollout, SingleActionAgent, TabularQAgent)
from .ppo import BatchedPPOAgent, PPOBaseAgent, PPOCNNAgent, PPOMLPAgent, discounted_returns_f32
from .deepq_batched import BatchedDeepQAgent, DeviceReplay
from .envs import ENV_IDS, ENV_MAP, BatchedGridworldEnv, GridworldEnv, make
from .loops import (EVAL_MAP, LEARN_MAP, WARMUP_MAP, BatchedRollout, batched_default_eval, batched_gather_rollout, batched_ppo_learn, batched_random_rollout, batched_tabq_learn,
                    default_eval,
                    dqn_learn, dqn_warmup, noop_warmup, ppo_learn, tabq_learn, whiler)
from .eventfile import EventFileWriter, read_events
from .metering import AverageMeter, BatchMetrics, NullWriter, RecordingWriter, make_meters, track_metrics
from .trainer import prepare_parser, train, train_batched

__all__ = [
    "AGENT_MAP", "ENV_MAP", "ENV_IDS", "LEARN_MAP", "EVAL_MAP", "WARMUP_MAP",
    "make", "GridworldEnv", "BatchedGridworldEnv",
    "RandomAgent", "SingleActionAgent", "TabularQAgent", "DeepQAgent", "BatchedTabularQAgent", "BatchedDeepQAgent", "DeviceReplay",
    "ReplayBuffer", "Experience", "ExperienceBatch", "Rollout",
    "BatchedPPOAgent", "batched_ppo_learn", "PPOBaseAgent", "PPOMLPAgent", "PPOCNNAgent", "discounted_returns_f32",
    "whiler", "tabq_learn", "dqn_learn", "ppo_learn", "default_eval", "dqn_warmup", "noop_warmup",
    "batched_random_rollout", "batched_tabq_learn", "batched_default_eval", "batched_gather_rollout", "BatchedRollout",
    "EventFileWriter", "read_events", "AverageMeter", "make_meters", "track_metrics", "BatchMetrics", "NullWriter", "RecordingWriter",
    "prepare_parser", "train", "train_batched",
]
