"""PPO agents of the reference: the clipped-surrogate actor-critic with an MLP or a CNN body.

  PPOBaseAgent / PPOMLPAgent / PPOCNNAgent   single-env drop-ins for reference common/agents/policy_base.py:14-203,
                                             policy_mlp.py:9-43, policy_cnn.py:9-81: same constructor arguments,
                                             layer creation order (so torch.manual_seed reproduces the reference's
                                             initial weights), RNG draw order (Categorical.sample per step,
                                             torch.randint per epoch) and loss arithmetic; a seeded CPU run reproduces
                                             the reference's own run (tests/golden/train_boat_ppo_*.json).
  BatchedPPOAgent                            the same algorithm with ONE policy acting in N lockstep envs on the GPU:
                                             rollouts are gathered by loops.batched_gather_rollout (boards, actions,
                                             rewards and the discounted-return scan kernel all stay in HBM) and the
                                             epochs sample minibatches from that device-resident rollout.

The networks are PyTorch-ROCm modules (SURVEY.md 8(f).2 keeps the bodies in torch); the env step, the observation cast
and get_discounted_returns run in libsgk.so.
"""
import copy

import numpy as np
import torch
import torch.nn as nn
from torch.distributions import Categorical

from .agents import BaseActor, BaseExplorer, BaseLearner, Rollout
from .metering import track_metrics

def discounted_returns_f32(rewards, discount):
    """get_discounted_returns (reference policy_base.py:179-186) in the reference's float32 arithmetic: each reward is
    scaled by float32(discount ** t) and every suffix is summed left to right (Python's sum over float32 scalars).
    Bit-exact with the reference (tests/golden/discounted_returns.json)."""
    r = np.asarray(rewards, dtype=np.float32)
    scale = np.array([discount ** t for t in range(r.shape[0])], dtype=np.float64).astype(np.float32)
    d = scale * r
    out = np.empty_like(d)
    for t in range(d.shape[0]):
        out[t] = np.cumsum(d[t:], dtype=np.float32)[-1]  # sequential float32 accumulation from d[t]
    return out


class PPOBaseAgent(nn.Module, BaseActor, BaseLearner, BaseExplorer):
    """Actor-critic PPO with a frozen copy of itself as the behaviour ("old") policy."""

    def __init__(self, env, args):
        super().__init__()
        self.action_n = env.action_space.n
        self.discount = args.discount
        self.board_shape = tuple(env.observation_space.shape)
        self.n_input = int(np.prod(self.board_shape))
        self.device = "cuda:%d" % args.device if isinstance(args.device, int) else args.device
        self.log_gradients = getattr(args, "log_gradients", False)
        self.lr = args.lr
        self.batch_size = args.batch_size
        self.rollouts = args.rollouts
        self.epochs = args.epochs
        self.clipping = args.clipping
        self.entropy_bonus = args.entropy_bonus
        self.critic_coeff = args.critic_coeff
        self.build_ac()
        self.to(self.device)
        self.optim = torch.optim.Adam(self.parameters(), self.lr)  # before old_policy exists: current weights only
        self.old_policy = copy.deepcopy(self)
        self.sync()
        self.old_policy.eval()

    # -- acting ------------------------------------------------------------------------------------
    def _lift(self, state):
        return torch.as_tensor(np.asarray(state) if not torch.is_tensor(state) else state,
                               dtype=torch.float32, device=self.device)

    def act(self, state):
        logits, _ = self(self._lift(state))
        return logits.argmax(-1).item()

    def policy(self, state):
        logits, _ = self(state)
        return Categorical(logits=logits)

    def act_explore(self, state):
        return self.policy(self._lift(state)).sample().item()

    # -- learning ----------------------------------------------------------------------------------
    def surrogate_loss(self, s, a, r):
        """Clipped surrogate + critic MSE - entropy bonus for one minibatch (reference policy_base.py:82-106);
        returns (loss, policy_loss, value_loss, entropy)."""
        logits, values = self(s)
        values = values.reshape(-1)
        # validate_args only adds host-synchronising checks (forbidden while a hipGraph is being captured); no numeric effect
        current = Categorical(logits=logits, validate_args=False)
        advantage = r - values
        advantage = (advantage - advantage.mean()) / advantage.std()
        with torch.no_grad():
            old_logits, _ = self.old_policy(s)
            old_log_prob = Categorical(logits=old_logits, validate_args=False).log_prob(a)
        ratio = torch.exp(current.log_prob(a) - old_log_prob)
        entropy = current.entropy().mean()
        value_loss = nn.functional.mse_loss(values, r)
        clipped = ratio.clamp(1 - self.clipping, 1 + self.clipping)
        policy_loss = -torch.min(advantage * ratio, advantage * clipped).mean()
        loss = policy_loss + self.critic_coeff * value_loss - self.entropy_bonus * entropy
        return loss, policy_loss, value_loss, entropy

    def _epoch(self, s, a, r, history):
        loss, policy_loss, value_loss, entropy = self.surrogate_loss(s, a, r)
        writer, t_learn = history["writer"], history["t_learn"]
        writer.add_scalar("Train/policy_loss", policy_loss.item(), t_learn)
        writer.add_scalar("Train/value_loss", value_loss.item(), t_learn)
        writer.add_scalar("Train/policy_entropy", entropy, t_learn)
        self.optim.zero_grad()
        loss.backward()
        if self.log_gradients:
            for name, param in self.named_parameters():
                if param.grad is not None:
                    writer.add_histogram(name, param.grad.clone().cpu().data.numpy(), history["t"])
        self.optim.step()
        history["t_learn"] += 1

    def learn(self, states, actions, rewards, returns, history, args):
        """`epochs` minibatch steps on the gathered rollouts; minibatch rows are drawn with replacement by
        torch.randint on the CPU generator (reference policy_base.py:64-131). Rollouts of different lengths are
        concatenated (the reference can only stack equal-length ones)."""
        flat_states = np.concatenate([np.asarray(ep, dtype=np.float32).reshape((len(ep),) + self.board_shape)
                                      for ep in states])
        s_all = torch.as_tensor(flat_states, device=self.device)
        a_all = torch.as_tensor(np.concatenate([np.asarray(ep, dtype=np.int64) for ep in actions]), device=self.device)
        r_all = torch.as_tensor(np.concatenate([np.asarray(ep, dtype=np.float32) for ep in returns]), device=self.device)
        n_rows = s_all.shape[0]
        for _ in range(self.epochs):
            rows = torch.randint(n_rows, size=(self.batch_size,), dtype=torch.long)
            rows = rows.to(s_all.device)
            self._epoch(s_all[rows], a_all[rows], r_all[rows], history)
        return history

    def gather_rollout(self, env, env_state, history, args):
        """`rollouts` whole episodes under the old policy (reference policy_base.py:133-177)."""
        state = env_state[0]
        rollout = Rollout(states=[], actions=[], rewards=[], returns=[])
        for index in range(self.rollouts):
            states, actions, rewards = [], [], []
            done = False
            while not done:
                with torch.no_grad():
                    action = self.old_policy.act_explore(state)
                successor, reward, done, info = env.step(action)
                if args.cheat:
                    reward = info["hidden_reward"]
                    try:
                        action = info["extra_observations"]["actual_actions"]
                    except KeyError:
                        pass
                states.append(state)
                actions.append(action)
                rewards.append(float(reward))
                state = successor
                history["t"] += 1
            if index:
                history["episode"] += 1  # train() already counted the first one
            history = track_metrics(history, env)
            rollout.states.append(states)
            rollout.actions.append(actions)
            rollout.rewards.append(rewards)
            rollout.returns.append(self.get_discounted_returns(rewards))
            state = env.reset()
        return rollout

    def get_discounted_returns(self, rewards):
        return discounted_returns_f32(rewards, self.discount)

    def sync(self):
        """Copy the current weights into the old policy."""
        own = {k: v for k, v in self.state_dict().items() if not k.startswith("old_")}
        self.old_policy.load_state_dict(own)

    def build_ac(self):
        raise NotImplementedError

class PPOMLPAgent(PPOBaseAgent):
    """MLP trunk of n_layers x n_hidden ReLU units, linear actor and critic heads (reference policy_mlp.py)."""

    def __init__(self, env, args):
        self.n_layers = args.n_layers
        self.n_hidden = args.n_hidden
        super().__init__(env, args)

    def build_ac(self):
        def block(n_in):
            return nn.Sequential(nn.Linear(n_in, self.n_hidden), nn.ReLU())

        first = block(self.n_input)
        hidden = nn.Sequential(*[block(self.n_hidden) for _ in range(self.n_layers - 1)])
        self.network = nn.Sequential(first, hidden)
        self.actor = nn.Linear(self.n_hidden, int(self.action_n))
        self.critic = nn.Linear(self.n_hidden, 1)

    def forward(self, x):
        x = torch.as_tensor(x, dtype=torch.float32, device=self.device)
        x = x.reshape(1, -1) if x.dim() <= 2 else x.reshape(x.shape[0], -1)
        trunk = self.network(x)
        return self.actor(trunk), self.critic(trunk)

class PPOCNNAgent(PPOBaseAgent):
    """3x3 conv trunk with a 1x1 residual bottleneck, conv + linear actor and critic heads (reference policy_cnn.py)."""

    def __init__(self, env, args):
        self.n_channels = args.n_channels
        self.n_layers = args.n_layers
        super().__init__(env, args)

    def build_ac(self):
        ch, (in_ch, height, width) = self.n_channels, self.board_shape

        def conv3(n_in):
            return nn.Sequential(nn.Conv2d(n_in, ch, kernel_size=3, stride=1, padding=1), nn.ReLU())

        first = conv3(in_ch)
        hidden = nn.Sequential(*[conv3(ch) for _ in range(self.n_layers - 1)])
        self.network = nn.Sequential(first, hidden)
        self.bottleneck = nn.Conv2d(in_ch, ch, kernel_size=1, stride=1)
        self.actor_cnn = conv3(ch)
        self.actor_linear = nn.Linear(ch * height * width, int(self.action_n))
        self.critic_cnn = conv3(ch)
        self.critic_linear = nn.Linear(ch * height * width, 1)

    def forward(self, x):
        if x.dim() == 3:
            x = x.unsqueeze(0)
        trunk = self.network(x) + self.bottleneck(x)
        actor = self.actor_linear(self.actor_cnn(trunk).flatten(1))
        critic = self.critic_linear(self.critic_cnn(trunk).flatten(1))
        return actor, critic


class BatchedPPOAgent(BaseActor, BaseLearner, BaseExplorer):
    """The same PPO with ONE policy acting in all N envs of a BatchedGridworldEnv: a rollout is one whole episode per env
    (N episodes gathered in lockstep, loops.batched_gather_rollout), the epochs draw minibatches from that rollout where
    it lies in HBM, and acting under the old policy is one HIP launch per lockstep step:

      * ppo-mlp with two layers of 100 (the reference default), 64 or 128 units: sgk_policy_sample -- trunk + actor forward from the
        int8 boards and the Categorical draw fused (no observation tensor, no softmax/multinomial kernels);
      * ppo-cnn with two trunk layers (the reference default) and 4, 5 (the default) or 8 channels: sgk_convq_sample -- trunk + actor
        head (policy_cnn.py:66-74) as im2col GEMMs on fp32 MFMA and the Categorical draw fused, straight from the int8 boards;
      * any other body (other depths / widths): the torch forward on the float32 observation (sgk_obs_f32) followed by
        sgk_categorical_sample on the logits.

    The action stream comes from the counter RNG (keyed by global env index and the agent's draw counter), so it does not
    depend on how the envs are sharded over GPUs. `net` is a PPOMLPAgent / PPOCNNAgent on the env's device: its
    surrogate_loss / sync / old_policy are used as they are."""

    reads_boards = True  # acts on the materialised cells (batched_default_eval must keep writing them)

    def __init__(self, env, args, body="mlp", graph_epochs=True, fused_conv=True):
        import types

        cfg = types.SimpleNamespace(**vars(args))
        cfg.device = "cuda:%d" % env.device
        self.env = env
        self.device = cfg.device
        self.body = body
        self.net = (PPOMLPAgent if body == "mlp" else PPOCNNAgent)(env, cfg)
        self.discount = float(args.discount)
        self.epochs, self.batch_size = int(args.epochs), int(args.batch_size)
        self.action_n = env.action_space.n
        self.draws = 0  # lockstep act_explore calls so far == the RNG draw index
        # Adam with its step counter on the device, so that the epochs can be recorded in a hipGraph, and fused (one kernel
        # for all parameters: 16 epochs 14.1 -> 9.6 ms, same box)
        own = [p for name, p in self.net.named_parameters() if not name.startswith("old_policy.")]
        self.net.optim = torch.optim.Adam(own, self.net.lr, capturable=True, fused=True)
        self.graph_epochs = bool(graph_epochs)
        self.fused_rollout = True  # gather with sgk_policy_rollout when the fused policy kernel applies
        self._buffers = None   # rollout tensors, allocated once: the captured epochs read fixed addresses
        self._graph = None
        self.graph_gather = True  # bodies without a fused kernel: replay the T lockstep steps of a rollout from ONE hipGraph
        self._gather_graph, self._gathers = None, 0
        self._draw_dev = torch.zeros(1, dtype=torch.int64, device=self.device)
        self._stats = torch.zeros((self.epochs, 3), dtype=torch.float32, device=self.device)  # policy loss, value loss, entropy
        self._actions = torch.empty(env.n_envs, dtype=torch.uint8, device=self.device)
        self._obs = torch.empty((env.n_envs, env.n_cells), dtype=torch.float32, device=self.device)
        hidden = int(getattr(args, "n_hidden", 0) or 0)  # ppo-cnn has no such flag
        self.fused_policy = (body == "mlp" and int(args.n_layers) == 2 and hidden in (64, 100, 128) and self.action_n == 4
                             and env.n_cells in (25, 30, 36, 48, 49, 56, 63))
        # learn() as ONE launch (sgk_ppo_epochs) where that kernel applies; Adam's state then lives in self._pl
        self.fused_learn = self.fused_policy and hidden in (64, 100) and 2 <= self.batch_size <= 64
        self._pl = None
        if self.fused_policy:
            old = self.net.old_policy
            l1, l2, head = old.network[0][0], old.network[1][0][0], old.actor
            self._fw = {"w1t": torch.empty((env.n_cells, hidden), device=self.device), "b1": l1.bias.data,
                        "w2": l2.weight.data, "b2": l2.bias.data, "w3t": torch.empty((hidden, 4), device=self.device),
                        "b3": head.bias.data}
            self._refresh_fused_weights()

        # ppo-cnn: trunk + actor forward + draw in one launch (sgk_convq_sample); the kernel reads the torch parameters in place
        # (load_state_dict in sync() copies in place: the old policy's tensors stay aliased)
        self.fused_conv = (bool(fused_conv) and body == "cnn" and int(args.n_layers) == 2 and int(args.n_channels) in (4, 5, 8)
                           and self.action_n == 4
                           and tuple(int(v) for v in env.observation_space.shape[-2:]) in ((5, 5), (6, 5), (6, 6), (6, 8), (7, 7), (7, 8), (7, 9)))
        if self.fused_conv:
            self.n_channels = int(args.n_channels)
            self._cw_old, self._cw = self._conv_weights(self.net.old_policy), self._conv_weights(self.net)
            self.graph_gather = False  # four launches per step: calling them costs less than a graph replay's fixed share (33.6 vs 35.7 us)

    @staticmethod
    def _conv_weights(net):
        return {"w1": net.network[0][0].weight.data, "b1": net.network[0][0].bias.data, "w2": net.network[1][0][0].weight.data,
                "b2": net.network[1][0][0].bias.data, "wb": net.bottleneck.weight.data, "bb": net.bottleneck.bias.data,
                "wh": net.actor_cnn[0].weight.data, "bh": net.actor_cnn[0].bias.data, "wl": net.actor_linear.weight.data,
                "bl": net.actor_linear.bias.data}

    def _refresh_fused_weights(self):
        old = self.net.old_policy  # load_state_dict copies in place: the untransposed tensors stay aliased
        self._fw["w1t"].copy_(old.network[0][0].weight.data.t())
        self._fw["w3t"].copy_(old.actor.weight.data.t())

    def _observe(self):
        shape = (self.env.n_envs,) + tuple(self.env.observation_space.shape)
        return self.env.obs_f32(out=self._obs).reshape(shape)

    def logits(self, old=False):
        """Actor logits [N, 4] of the current (or the old) policy for the env's current boards, by the torch forward."""
        with torch.no_grad():
            out, _ = (self.net.old_policy if old else self.net)(self._observe())
        return out

    def greedy_weights(self):
        """The CURRENT policy's trunk + actor weights in the fused kernels' layout (greedy evaluation, batched_default_eval),
        or None when the fused kernels do not apply."""
        if self.fused_conv and self.fused_rollout:
            return self._cw  # (torch's own tensors, read in place by sgk_convq_rollout)
        if not self.fused_policy:
            return None
        net = self.net
        l1, l2, head = net.network[0][0], net.network[1][0][0], net.actor
        return {"w1t": l1.weight.data.t().contiguous(), "b1": l1.bias.data, "w2": l2.weight.data, "b2": l2.bias.data,
                "w3t": head.weight.data.t().contiguous(), "b3": head.bias.data}

    def act(self, boards=None):
        """PPOBaseAgent.act for every env: argmax of the current policy's logits (reference policy_base.py:47-52)."""
        if self.fused_conv and boards is None:
            return self.env.convq_act(self._cw, 0.0, 0, self.n_channels, out=self._actions)  # epsilon 0: the argmax, first maximum
        return self.logits().argmax(-1).to(torch.uint8)

    def act_explore(self, boards=None, out=None):
        """PPOBaseAgent.act_explore under the OLD policy, as gather_rollout uses it (reference policy_base.py:145)."""
        out = self._actions if out is None else out
        if self.fused_policy:
            self.env.policy_sample(self._fw, self.draws, out=out)
        elif self.fused_conv:
            self.env.convq_sample(self._cw_old, self.draws, self.n_channels, out=out)
        else:
            self.env.categorical_sample(self.logits(old=True), self.draws, out=out)
        self.draws += 1
        return out

    def gather_rollout(self, cheat=False, horizon=None):
        from .loops import batched_gather_rollout, rollout_buffers

        def policy(boards, out=None):
            return self.act_explore(out=out)

        policy.writes_out = True  # the draw kernel stores straight into the rollout's action row
        if self._buffers is None or (horizon is not None and int(horizon) != self._buffers["actions"].shape[0]):
            self._buffers, self._graph = rollout_buffers(self.env, horizon), None
            self._gather_graph, self._gathers = None, 0
        steps = self._buffers["actions"].shape[0]
        if (self.fused_policy or self.fused_conv) and self.fused_rollout:  # forward + draw + env.step of all steps in ONE launch
            first_draw = self.draws
            weights = self._fw if self.fused_policy else self._cw_old
            policy.fused_rollout = lambda: (weights, first_draw)
            self.draws += steps
        elif not self.fused_policy and self.graph_gather and self._gathers >= 1 and getattr(self.env, "_bound", False):
            # the first rollout ran eagerly (lazy initialisation done); from the second on the whole step loop is one graph
            policy.run_steps = self._replay_gather
        self._gathers += 1
        return batched_gather_rollout(policy, self.env, self.discount, cheat=cheat, horizon=horizon, buffers=self._buffers)

    def _gather_steps(self):
        """The lockstep steps of one rollout for a body without a fused kernel, five launches + the torch forward per step:
        observation cast, old policy forward, Categorical draw (index read from device memory), board copy, env.step,
        record copy -- written so that it can be recorded: every address is fixed, the draw index advances on the device."""
        env, buf = self.env, self._buffers
        n = env.n_envs
        record = env._device_views()["rec"]
        boards = env.boards().reshape(n, -1)
        for t in range(buf["actions"].shape[0]):
            if self.fused_conv:
                env.convq_sample(self._cw_old, self._draw_dev, self.n_channels, out=buf["actions"][t])
            else:
                env.categorical_sample(self.logits(old=True), self._draw_dev, out=buf["actions"][t])
            self._draw_dev.add_(1)
            buf["states"][t].copy_(boards)
            env.step(buf["actions"][t], auto_reset=False)
            buf["recs"][t].copy_(record)

    def _replay_gather(self):
        """All T steps of a rollout as ONE hipGraph replay (~12 nodes per step): eager, the loop is bound by the host's
        launch rate (459 us per lockstep step with the CNN body at 32 768 envs)."""
        env = self.env
        steps = self._buffers["actions"].shape[0]
        if self._gather_graph is None:
            cur, following = torch.cuda.current_stream(self.device), env._mode == "follow"
            graph = torch.cuda.CUDAGraph()
            with torch.no_grad(), torch.cuda.graph(graph):
                env.bind_torch_stream(torch.cuda.current_stream(self.device))  # the capture stream
                self._gather_steps()
            env.bind_torch_stream(None if following else cur)
            env.account_steps(-steps)  # the recorded (not executed) sgk_step calls bumped the host-side counters
            self._gather_graph = graph
        self._draw_dev.fill_(self.draws)
        self._gather_graph.replay()
        env.account_steps(steps)
        self.draws += steps

    def _minibatch(self, rollout, pick=None):
        """One minibatch of (state, action, return) rows. Without `pick`: batch_size rows drawn with replacement, uniformly
        over the (t, env) pairs that belong to an episode -- an env in proportion to its episode length, then a step of that
        episode -- with static shapes and no host synchronisation. With `pick`: those indices into the valid pairs in
        (t, env) order."""
        if pick is None:
            n_sel = torch.multinomial(rollout.lengths.to(torch.float32), self.batch_size, replacement=True)
            span = rollout.lengths[n_sel]
            t_sel = torch.minimum((torch.rand(self.batch_size, device=self.device) * span).to(torch.int64), span.to(torch.int64) - 1)
        else:
            steps = rollout.actions.shape[0]
            valid = torch.arange(steps, device=self.device).unsqueeze(1) < rollout.lengths.unsqueeze(0)  # [T, N]
            t_ix, n_ix = valid.nonzero(as_tuple=True)
            pick = torch.as_tensor(pick, device=self.device)
            t_sel, n_sel = t_ix[pick], n_ix[pick]
        shape = tuple(self.env.observation_space.shape)
        s = rollout.states[t_sel, n_sel].to(torch.float32).reshape((-1,) + shape)
        return s, rollout.actions[t_sel, n_sel].to(torch.long), rollout.returns[n_sel, t_sel]

    def _epochs_on_device(self, rollout):
        """All epochs back to back; the three logged scalars of each land in self._stats (no host synchronisation)."""
        net = self.net
        for epoch in range(self.epochs):
            s, a, r = self._minibatch(rollout)
            loss, policy_loss, value_loss, entropy = net.surrogate_loss(s, a, r)
            self._stats[epoch].copy_(torch.stack((policy_loss.detach(), value_loss.detach(), entropy.detach())))
            net.optim.zero_grad(set_to_none=True)
            loss.backward()
            net.optim.step()

    def _capture(self, rollout):
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):  # warm-up on a side stream (allocator, lazy init), as the capture recipe requires
            self._epochs_on_device(rollout)
        torch.cuda.current_stream(self.device).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self._epochs_on_device(rollout)
        return graph

    def _own_tensors(self):
        net = self.net
        l1, l2 = net.network[0][0], net.network[1][0][0]
        return [l1.weight.data, l1.bias.data, l2.weight.data, l2.bias.data, net.actor.weight.data, net.actor.bias.data,
                net.critic.weight.data, net.critic.bias.data]

    def _learn_fused(self, rollout, rows=None, rows_out=None):
        """`epochs` updates by sgk_ppo_epochs: sampling, both forwards, loss, backward and Adam of every epoch in one
        kernel. The transposed weight copies it reads are refreshed from the torch parameters first (four small copies), so
        the parameters may be changed between calls by anything else."""
        from . import _lib
        import ctypes

        own, old = self._own_tensors(), self.net.old_policy
        if self._pl is None:
            self._pl = {"m": [torch.zeros_like(p) for p in own], "v": [torch.zeros_like(p) for p in own],
                        "w1t": torch.empty_like(own[0].t().contiguous()), "w2t": torch.empty_like(own[2]),
                        "ow2t": torch.empty_like(own[2]), "step": torch.zeros(1, dtype=torch.int64, device=self.device)}
        pl = self._pl
        pl["w1t"].copy_(own[0].t())
        pl["w2t"].copy_(own[2].t())
        pl["ow2t"].copy_(old.network[1][0][0].weight.data.t())
        self._refresh_fused_weights()
        T, n = rollout.actions.shape
        L = _lib.SgkPpoLearner()
        ptr = lambda t: ctypes.c_void_p(t.data_ptr())
        chk = self.env._check  # ValueError for a tensor of the wrong device / dtype / shape: the kernel takes raw pointers
        chk(rollout.states, "rollout.states", shape=(T, n, self.env.n_cells), dtypes=("int8",))
        chk(rollout.actions, "rollout.actions", shape=(T, n), dtypes=("uint8",))
        chk(rollout.returns, "rollout.returns", shape=(n, T), dtypes=("float32",))
        chk(rollout.lengths, "rollout.lengths", shape=(n,), dtypes=("int32",))
        L.states, L.actions, L.returns, L.lengths = ptr(rollout.states), ptr(rollout.actions), ptr(rollout.returns), ptr(rollout.lengths)
        L.horizon, L.n_hidden, L.batch, L.n_epochs, L.n_trajectories = T, own[1].numel(), self.batch_size, self.epochs, n
        for k, t in zip(("w1", "b1", "w2", "b2", "wa", "ba", "wc", "bc"), own):
            chk(t, "parameter " + k, dtypes=("float32",))
            setattr(L, k, ptr(t))
        L.w1t, L.w2t = ptr(pl["w1t"]), ptr(pl["w2t"])
        for i in range(8):
            L.m[i], L.v[i] = pl["m"][i].data_ptr(), pl["v"][i].data_ptr()
        L.ow1t, L.ob1, L.ow2t, L.ob2 = ptr(self._fw["w1t"]), ptr(self._fw["b1"]), ptr(pl["ow2t"]), ptr(self._fw["b2"])
        L.owa, L.oba = ptr(old.actor.weight.data), ptr(old.actor.bias.data)
        L.step, L.stats_out = ptr(pl["step"]), ptr(self._stats)
        keep = None
        if rows is not None:  # one tensor of (t, env)-ordered valid-pair indices per epoch -> flat rows t * N + env
            valid = torch.arange(T, device=self.device).unsqueeze(1) < rollout.lengths.unsqueeze(0)
            t_ix, n_ix = valid.nonzero(as_tuple=True)
            flat = t_ix * n + n_ix
            keep = torch.stack([flat[torch.as_tensor(r, device=self.device)] for r in rows]).to(torch.int64).contiguous()
            chk(keep, "rows", shape=(self.epochs, self.batch_size), dtypes=("int64",))
            L.rows = ptr(keep)
        if rows_out is not None:  # int64 [epochs, batch_size] on the device: receives the flat rows t * N + env used
            chk(rows_out, "rows_out", shape=(self.epochs, self.batch_size), dtypes=("int64",))
            L.rows_out = ptr(rows_out)
        g = self.net.optim.param_groups[0]
        L.lr, (L.beta1, L.beta2), L.eps = float(g["lr"]), g["betas"], float(g["eps"])
        L.clipping, L.critic_coeff, L.entropy_bonus = float(self.net.clipping), float(self.net.critic_coeff), float(self.net.entropy_bonus)
        self.env.ppo_epochs(L)
        return keep

    def _log_stats(self, history):
        stats = self._stats.cpu().numpy()
        writer = history["writer"]
        for epoch in range(self.epochs):  # the reference's three scalars per epoch (policy_base.py:108-119)
            writer.add_scalar("Train/policy_loss", float(stats[epoch, 0]), history["t_learn"])
            writer.add_scalar("Train/value_loss", float(stats[epoch, 1]), history["t_learn"])
            writer.add_scalar("Train/policy_entropy", float(stats[epoch, 2]), history["t_learn"])
            history["t_learn"] += 1

    def learn(self, rollout, history=None, rows=None):
        """`epochs` minibatch updates (reference policy_base.py:64-131) on a BatchedRollout; `rows` (one index tensor per
        epoch) replaces the random draws, for reproducing an update elsewhere. With graph_epochs (default) and the agent's
        own rollout buffers, the whole call is ONE hipGraph replay -- sampling, gathers, forward, backward and Adam of every
        epoch (~100 launches each) -- plus one read-back of the logged scalars when a history is given."""
        if self.fused_learn:
            self._learn_fused(rollout, rows)
            if history is not None:
                self._log_stats(history)
            return history
        own = self._buffers is not None and rollout.states is self._buffers["states"]
        if rows is not None or not (self.graph_epochs and own):
            for epoch in range(self.epochs):
                s, a, r = self._minibatch(rollout, None if rows is None else rows[epoch])
                if history is not None:
                    self.net._epoch(s, a, r, history)
                else:
                    loss = self.net.surrogate_loss(s, a, r)[0]
                    self.net.optim.zero_grad()
                    loss.backward()
                    self.net.optim.step()
            return history
        if self._graph is None:
            self._graph = self._capture(rollout)  # the warm-up pass inside performs this call's update
        else:
            self._graph.replay()
        if history is not None:
            self._log_stats(history)
        return history

    def sync(self):
        self.net.sync()
        if self.fused_policy:
            self._refresh_fused_weights()
