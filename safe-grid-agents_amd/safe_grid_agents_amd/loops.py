"""Interaction loops with the reference's signatures and dispatch maps:

    learn_fn(agent, env, env_state, history, args) -> (env_state, history, eval_next)   LEARN_MAP
    eval_fn(agent, env, eval_history, args)        -> eval_history                       EVAL_MAP
    warmup_fn(agent, env, history, args)           -> (agent, env, history, args)        WARMUP_MAP

(reference safe_grid_agents/common/learn.py:8-113, eval.py:8-59, warmup.py:8-33), plus the batched lockstep
forms that keep N envs and their agents on the GPU between steps.
"""
import collections
import functools
from collections import defaultdict

import numpy as np

from .agents import RandomAgent
from .metering import BatchMetrics, track_metrics


# ---- single-env loops (drop-in) --------------------------------------------------------------------
def whiler(step_fn):
    """Run `step_fn` until the env reports done, then book the episode (reference learn.py:8-26)."""

    @functools.wraps(step_fn)
    def run_episode(agent, env, env_state, history, args):
        while True:
            env_state, history = step_fn(agent, env, env_state, history, args)
            history["t"] += 1
            if env_state[2]:
                break
        history = track_metrics(history, env)
        eval_next = history["episode"] % args.eval_every == args.eval_every - 1
        return env_state, history, eval_next

    return run_episode


def _maybe_cheat(args, action, reward, info):
    """--cheat: learn from the hidden reward and, when the env reports it, the action actually taken
    (reference learn.py:41-47,72-78)."""
    if not args.cheat:
        return action, reward
    reward = info["hidden_reward"]
    try:
        action = info["extra_observations"]["actual_actions"]
    except KeyError:
        pass
    return action, reward


@whiler
def tabq_learn(agent, env, env_state, history, args):
    state = env_state[0]
    t = history["t"]
    action = agent.act_explore(state)
    successor, reward, done, info = env.step(action)
    learn_action, learn_reward = _maybe_cheat(args, action, reward, info)
    agent.learn(state, learn_action, learn_reward, successor)
    history["writer"].add_scalar("Train/epsilon", agent.update_epsilon(), t)
    # the reference returns the (possibly substituted) reward in env_state; keep that
    return (successor, learn_reward, done, info), history


@whiler
def dqn_learn(agent, env, env_state, history, args):
    state = env_state[0]
    t = history["t"]
    action = agent.act_explore(state)
    successor, reward, done, info = env.step(action)
    learn_action, learn_reward = _maybe_cheat(args, action, reward, info)
    history = agent.learn(state, learn_action, learn_reward, successor, done, history)
    history["writer"].add_scalar("Train/epsilon", agent.update_epsilon(), t)
    if t % args.sync_every == args.sync_every - 1:
        agent.sync_target_Q()
    return (successor, learn_reward, done, info), history


def ppo_learn(agent, env, env_state, history, args):
    """One PPO iteration: gather `rollouts` episodes under the old policy, `epochs` minibatch updates, then make the
    updated policy the old one (reference learn.py:88-104). Evaluation follows every iteration whose episode counter
    is a multiple of eval_every; the env_state handed in is returned untouched, as in the reference."""
    rollout = agent.gather_rollout(env, env_state, history, args)
    history = agent.learn(*rollout, history, args)
    agent.sync()
    eval_next = history["episode"] > 0 and history["episode"] % args.eval_every == 0
    return env_state, history, eval_next


LEARN_MAP = {"deep-q": dqn_learn, "tabular-q": tabq_learn, "ppo-mlp": ppo_learn, "ppo-cnn": ppo_learn}


def default_eval(agent, env, eval_history, args):
    """Greedy rollout for at least args.eval_timesteps steps, ending on an episode boundary (reference eval.py:8-56).

    Every episode that ends before the step budget is spent is booked without a tensorboard write and followed by
    a reset; the episode that ends after it is booked once, with the write, by the final track_metrics.
    """
    print("#### EVAL ####")
    frames_wanted = args.eval_visualize_episodes > 0
    state, done, t = env.reset(), False, 0
    clip = [np.copy(env.render(mode="rgb_array"))]
    clips = []
    budget_spent = False
    while True:
        if done:
            if budget_spent:
                break
            eval_history = track_metrics(eval_history, env, eval=True, write=False)
            state, done = env.reset(), False
            if frames_wanted:
                clips.append(np.swapaxes(np.stack(clip), 0, 1))  # colour axis before time (eval.py:26-28)
                clip = [np.copy(env.render(mode="rgb_array"))]
                frames_wanted = args.eval_visualize_episodes > len(clips)
        state, _, done, _ = env.step(agent.act(state))
        t += 1
        budget_spent = t >= args.eval_timesteps
        if frames_wanted:
            clip.append(np.copy(env.render(mode="rgb_array")))
    if clips:
        eval_history["writer"].add_video("Evaluation/grid_animation", np.stack(clips, axis=0), eval_history["period"])
    eval_history = track_metrics(eval_history, env, eval=True, write=True)
    eval_history["returns"].reset(reset_history=True)
    for name in ("safeties", "margins", "margins_support"):
        eval_history[name].reset()
    eval_history["period"] += 1
    return eval_history


EVAL_MAP = defaultdict(lambda: default_eval, {})


def dqn_warmup(agent, env, history, args):
    """Fill the replay with args.replay_capacity random-action transitions (reference warmup.py:8-23).

    As in the reference the loop starts in the `done` state, so the very first iteration books whatever
    env._env.episode_return holds before the first reset."""
    walker = RandomAgent(env, args)
    print("#### WARMUP ####\n")
    done, state = True, None
    for _ in range(args.replay_capacity):
        if done:
            history["returns"].update(env._env.episode_return)
            state, done = env.reset(), False
        action = walker.act(None)
        successor, reward, done, _ = env.step(action)
        agent.replay.add(state, action, reward, successor, done)
    return agent, env, history, args


def noop_warmup(agent, env, history, args):
    return agent, env, history, args


WARMUP_MAP = defaultdict(lambda: noop_warmup, {"deep-q": dqn_warmup})


# ---- batched lockstep loops (GPU-resident) -----------------------------------------------------------
def batched_random_rollout(env, n_steps, fused=False, chunk=None):
    """RandomAgent over every env of a BatchedGridworldEnv for n_steps lockstep steps with reset-on-done
    (the loop shape of dqn_warmup, warmup.py:14-21, without the replay). Returns BatchMetrics of the episodes
    finished during the call. Nothing leaves the GPU but the 16-word metrics vector."""
    chunk = chunk or n_steps
    done_steps = 0
    while done_steps < n_steps:
        k = min(chunk, n_steps - done_steps)
        env.step_random(k, auto_reset=True, fused=fused)
        done_steps += k
    return BatchMetrics(env.metrics(), env.reward_scale)


def batched_default_eval(agent, env, eval_timesteps):
    """default_eval (reference eval.py:8-56) for N envs in lockstep, greedy `agent.act`.

    Per env the reference plays episodes back to back, resetting after every `done` that arrives before step
    `eval_timesteps`, and leaves its loop at the first `done` at or after it (eval.py:19-39): whole episodes only.
    Lockstep form: `eval_timesteps - 1` iterations of {step, reset finished envs}, then steps WITHOUT reset until every
    env's current episode has ended (at most `max_iterations` more; finished envs idle, their steps are no-ops).
    No host synchronisation inside the loop. Returns BatchMetrics of the evaluation (metrics are reset first)."""
    boards = bool(getattr(agent, "reads_boards", False))  # table agents act on the state word; networks need the cells
    env.metrics_reset()
    env.reset()
    weights = agent.greedy_weights() if hasattr(agent, "greedy_weights") else None
    if weights is not None:  # fused policy: each phase is one sgk_policy_rollout / sgk_convq_rollout launch (auto-reset == step + reset_done)
        if "wh" in weights:  # a conv body
            def rollout(k, auto_reset):
                env.convq_rollout(weights, k, int(weights["b1"].numel()), mode="greedy", epsilon=0.0, auto_reset=auto_reset)
        else:
            def rollout(k, auto_reset):
                env.policy_rollout(weights, k, mode="greedy", epsilon=0.0, auto_reset=auto_reset)
        if int(eval_timesteps) > 1:
            rollout(int(eval_timesteps) - 1, True)
        rollout(int(env.info.max_iterations), False)
        return BatchMetrics(env.metrics(), env.reward_scale)
    for _ in range(max(int(eval_timesteps) - 1, 0)):
        env.step(agent.act(), auto_reset=False, write_boards=boards)
        env.reset_done()
    for _ in range(int(env.info.max_iterations)):
        env.step(agent.act(), auto_reset=False, write_boards=boards)
    return BatchMetrics(env.metrics(), env.reward_scale)


BatchedRollout = collections.namedtuple("BatchedRollout", ["states", "actions", "rewards", "returns", "lengths"])


def rollout_buffers(env, horizon=None):
    """Device tensors one batched rollout needs (reusable across rollouts: a captured learner reads fixed addresses)."""
    import torch

    T = int(horizon or env.info.max_iterations)
    n, dev = env.n_envs, "cuda:%d" % env.device
    return {"states": torch.empty((T, n, env.n_cells), dtype=torch.int8, device=dev),
            "actions": torch.empty((T, n), dtype=torch.uint8, device=dev),
            "recs": torch.empty((T, n, 4), dtype=torch.int8, device=dev),  # reward, hidden reward, done, actual action
            "rewards": torch.empty((n, T), dtype=torch.float32, device=dev),
            "returns": torch.empty((n, T), dtype=torch.float32, device=dev),
            "lengths": torch.empty(n, dtype=torch.int32, device=dev)}


def batched_gather_rollout(policy, env, discount, cheat=False, horizon=None, buffers=None):
    """PPOBaseAgent.gather_rollout (reference policy_base.py:133-177) with one rollout = one episode PER ENV, all envs in
    lockstep: `policy(boards) -> uint8 actions [N]` acts on the int8 board view (a policy with the attribute
    `writes_out = True` is called as policy(boards, out=row) instead; one with `fused_rollout() -> (weights, draw0)` hands
    the whole loop to sgk_policy_rollout, one launch; one with `run_steps()` runs the T steps itself into `buffers`, e.g. by
    replaying a recorded hipGraph), finished envs idle until the horizon, and
    the discounted returns come from the bit-exact batched kernel (policy_base.py:179-186). Everything stays in HBM.

    Returns BatchedRollout(states int8 [T, N, cells], actions uint8 [T, N], rewards float32 [N, T], returns float32 [N, T],
    lengths int32 [N]); entries past an env's length are zero. The episodes are booked in the env's metrics vector
    (track_metrics, policy_base.py:168) and every env is reset afterwards (policy_base.py:174). `buffers`
    (rollout_buffers) are overwritten and returned when given."""
    import torch

    buf = buffers or rollout_buffers(env, horizon)
    states, actions, recs = buf["states"], buf["actions"], buf["recs"]
    rewards, returns, lengths = buf["rewards"], buf["returns"], buf["lengths"]
    T, n = actions.shape
    dev = actions.device
    env.reset()
    fused = getattr(policy, "fused_rollout", None)  # -> (MLP weights, first draw index): the whole loop is one launch
    if fused is not None:
        weights, draw0 = fused()
        if "wh" in weights:  # a conv body (PPOCNNAgent): sgk_convq_rollout
            env.convq_rollout(weights, T, int(weights["b1"].numel()), mode="sample", draw_index0=draw0, auto_reset=False, states=states,
                              actions=actions, recs=recs, mask_finished=True)
        else:
            env.policy_rollout(weights, T, mode="sample", draw_index0=draw0, auto_reset=False, states=states, actions=actions,
                               recs=recs, mask_finished=True)  # entries past an episode's end are stored as zeros by the kernel
    elif getattr(policy, "run_steps", None) is not None:  # the policy runs the T steps itself (a recorded hipGraph)
        policy.run_steps()
    else:
        record = env._device_views()["rec"]
        direct = bool(getattr(policy, "writes_out", False))  # policy(boards, out=row) stores its actions itself
        for t in range(T):  # four launches per lockstep step: policy, board copy, env step, record copy
            boards = env.boards().reshape(n, -1)
            if direct:
                policy(boards, out=actions[t])
            else:
                actions[t].copy_(policy(boards))
            states[t].copy_(boards)
            env.step(actions[t], auto_reset=False)
            recs[t].copy_(record)
    # a finished env idles: its later records read (0, 0, done, .), so everything per-episode follows from the done flags
    finished_steps = (recs[:, :, 2] != 0).sum(0, dtype=torch.int32)  # done stays set from the last step of the episode on
    lengths.copy_(torch.clamp(T - finished_steps + 1, max=T))
    if env.reward_scale != 1.0:  # integer rewards in units of env.reward_scale (TomatoWatering: 0.02 per watered tomato): the
        # reference's float is count * REWARD_FACTOR in float64, rounded to float32 when the returns are made (policy_base.py:179-186)
        rewards.copy_(recs[:, :, 1 if cheat else 0].t().to(torch.float64) * env.reward_scale)
    else:
        rewards.copy_(recs[:, :, 1 if cheat else 0].t())
    if cheat or fused is None:
        live = torch.arange(T, device=dev).unsqueeze(1) < lengths.unsqueeze(0)  # [T, N]
        if cheat:
            actions.copy_(recs[:, :, 3].view(torch.uint8))
        actions.mul_(live)
        if fused is None:
            states.mul_(live.unsqueeze(2))
    returns.zero_()
    env.discounted_returns(rewards, discount, lengths=lengths, out=returns)
    env.reset()
    return BatchedRollout(states, actions, rewards, returns, lengths)


def batched_ppo_learn(agent, env, history=None, cheat=False):
    """ppo_learn (reference learn.py:88-104) for a BatchedPPOAgent: one rollout of N episodes (one per env) under the old
    policy, the epochs, then the sync. Returns the BatchMetrics of the gathered episodes."""
    env.metrics_reset()
    rollout = agent.gather_rollout(cheat=cheat)
    bm = BatchMetrics(env.metrics(), env.reward_scale)
    agent.learn(rollout, history)
    agent.sync()
    return bm


def batched_tabq_learn(agent, env, n_steps, cheat=False, fused=True, chunk=100):
    """tabq_learn for N private agents in lockstep: act_explore -> env.step -> learn -> update_epsilon, with the
    episode loop of train.py:62-70 (reset after done) folded in. fused=True runs all n_steps in one launch with the
    Q-tables resident in LDS (or their rows in HBM); fused=False keeps the drop-in call sequence -- four launches per lockstep
    step (act_explore, step, learn, reset_done) -- and replays it from a hipGraph, `chunk` steps per replay; fused="calls" makes
    the four calls from Python (what the graph records)."""
    if fused is True:
        agent.rollout(n_steps, cheat=cheat)
    elif fused == "calls":
        for _ in range(n_steps):
            actions = agent.act_explore()
            env.step(actions, auto_reset=False, write_boards=False)
            agent.learn(action=actions, cheat=cheat)
            env.reset_done()
    else:
        done = 0
        while done < n_steps:
            k = min(int(chunk), n_steps - done)
            agent.learn_steps(k, cheat=cheat)
            done += k
    return BatchMetrics(env.metrics(), env.reward_scale)
