"""DeepQAgent over a BatchedGridworldEnv: one shared Q-network (PyTorch-ROCm MLP, the reference's architecture and
defaults: value.py:148-158, agent_parser_configs.yaml:40-54) acting for N envs per lockstep step, everything resident in
HBM. BASELINE.json config 4 (SideEffectsSokoban + deep-q, 32 768 envs).

What is kept from the reference (value.py:61-187, learn.py:29-58, warmup.py:8-23): network, target network and
`sync_every`, Adam(amsgrad), MSE on `discount * max target_Q(s') * (1 - terminal) + r`, grad-clip 10, the epsilon
schedule, epsilon-greedy = Categorical(eps/n + (1 - eps) on the argmax), uniform replay sampling with replacement.

What necessarily differs, and is stated next to any number: the reference does ONE SGD step on 64
samples per SINGLE env-step of its single env; with N envs in lockstep one call produces N transitions, so the ratio is
a parameter here: `sgd_steps` SGD steps of `batch_size` samples per LOCKSTEP step, replay of `replay_slices` x N
transitions (a ring of whole lockstep slices). With N = 1, sgd_steps = 1 and replay_slices = replay_capacity the schedule IS
the reference's, and tests/golden/batched_dqn_*.npz (the reference's own DeepQAgent + dqn_warmup + dqn_learn, its random
calls answered from the counter RNG) is reproduced: actions exactly, losses and weights to fp32 tolerance.

Kept AS WRITTEN, because it is what the reference computes (`reference_loss_broadcast=True`, the default): the [B,1]-vs-[B]
mse_loss broadcast of value.py:119-123 (loss = mean over (i, j) of (Q_i - e_j)^2: every sample regresses towards the
minibatch's mean target); the target network's own random initialisation (value.py:82-84: no sync before the first
`sync_every`); dqn_warmup's `state` that is never advanced inside an episode (warmup.py:17-21: every warm-up row keeps its
episode's first board as `state`). Each has a switch to the textbook form, labelled non-reference.

`q_body="cnn"` (NOT the reference's DeepQAgent, which is an MLP: value.py:148-158 -- offered because BASELINE.json's config 4 is
worded "conv policy"): the Q-network is the convolutional body of the reference's PPO agent (policy_cnn.py:17-81: n_layers 3x3
convolutions of n_channels with a 1x1 residual bottleneck, then a 3x3 head convolution and a linear layer, here onto the four
Q-values). It runs through PyTorch-ROCm (MIOpen) on the float32 observation (sgk_obs_f32), acts through sgk_epsilon_greedy and
learns with torch autograd + Adam: no fused kernel, no parity claim.
"""


class DeviceReplay:
    """Ring of lockstep slices in HBM: int8 boards (state, successor), uint8 action, int8 reward, bool terminal."""

    def __init__(self, n_envs, n_cells, slices, device):
        import torch

        self.n, self.slices, self.device = n_envs, slices, device
        self.states = torch.empty((slices, n_envs, n_cells), dtype=torch.int8, device=device)
        self.successors = torch.empty_like(self.states)
        self.actions = torch.empty((slices, n_envs), dtype=torch.uint8, device=device)
        self.rewards = torch.empty((slices, n_envs), dtype=torch.int8, device=device)
        self.terminals = torch.empty((slices, n_envs), dtype=torch.bool, device=device)
        self.head = 0      # next slice to write
        self.filled = 0    # slices holding data
        self.head_dev = torch.zeros(1, dtype=torch.long, device=device)  # the same head, for graph-captured adds
        self.head_dev_stale = False
        # env._version at which states[head] was filled with the boards the next action is chosen on (reset_store below; None: not)
        self.ready_version = None

    def __len__(self):
        return self.filled * self.n

    def add_slice(self, states, actions, rewards, successors, terminals):
        k = self.head
        self.states[k].copy_(states)
        self.actions[k].copy_(actions)
        self.rewards[k].copy_(rewards)
        self.successors[k].copy_(successors)
        self.terminals[k].copy_(terminals)
        self._advance()
        self.head_dev_stale = True

    def _advance(self):
        self.head = (self.head + 1) % self.slices
        self.filled = min(self.filled + 1, self.slices)

    def store(self, env, phase, actions=None, cheat=False, captured=False):
        """ReplayBuffer.add for every env as two kernel launches (sgk_replay_store): phase 0 before env.step (boards ->
        states[head]), phase 1 after it (boards -> successors[head]; action, reward, terminal from the step records). With
        `captured` the slice index is read from device memory (head_dev) so that the launch can be recorded in a graph; the
        caller advances the heads (phase 1 of the eager form advances the host head itself)."""
        import ctypes

        from . import _lib

        ptr = lambda x: None if x is None else ctypes.c_void_p(x.data_ptr())  # noqa: E731
        env._sync_torch_to_lib()
        _lib.check(env.lib.sgk_replay_store(env._h.ptr, int(phase), ptr(actions), int(bool(cheat)), int(self.head),
                                            ptr(self.head_dev) if captured else None, ptr(self.states), ptr(self.successors),
                                            ptr(self.actions), ptr(self.rewards), ptr(self.terminals)))
        env._sync_lib_to_torch()
        if phase == 1 and not captured:
            self._advance()
            self.head_dev_stale = True

    def note_replayed_add(self):
        self._advance()

    # ---- the two halves of the add fused into the launches around them (sgk_step_store / sgk_reset_done_store) ----
    def states_ready(self, env):
        """states[head] already holds the boards the next action will be chosen on (the previous step's reset_store put them there and
        nothing has changed the envs since)."""
        return self.ready_version is not None and self.ready_version == env._version

    def step_store(self, env, actions, cheat=False, captured=False):
        """env.step(actions) + the add's second half in one launch; advances the host head unless captured."""
        rings = (self.successors, self.actions, self.rewards, self.terminals)
        if captured:
            env.step_store(actions, 0, rings, cheat=cheat, slice_dev=self.head_dev)
        else:
            env.step_store(actions, self.head, rings, cheat=cheat)
            self._advance()
            self.head_dev_stale = True

    def reset_store(self, env, captured=False):
        """reset_done() + the add's first half for the NEXT step (states[head] = the boards after the reset) in one launch."""
        if captured:
            env.reset_done_store(self.states, 0, slice_dev=self.head_dev)
        else:
            env.reset_done_store(self.states, self.head)
        self.ready_version = env._version

    def reset_store_args(self, captured=False):
        """(states ring, slice, slice_dev) of the reset_store that follows a learning step: for sgk_dqn_sgd_step_reset_store, which does
        it inside the SGD step's second launch; the caller then calls note_reset_store."""
        return (self.states, 0, self.head_dev) if captured else (self.states, self.head, None)

    def note_reset_store(self, env):
        self.ready_version = env._version

    def sample(self, batch):
        """Uniform with replacement over everything stored (contain.py:19-22), indices drawn on the device."""
        import torch

        total = self.filled * self.n
        ix = torch.randint(0, total, (batch,), device=self.device)
        flat = lambda t: t.reshape(self.slices * self.n, *t.shape[2:])  # noqa: E731
        return (flat(self.states)[ix], flat(self.actions)[ix], flat(self.rewards)[ix], flat(self.successors)[ix],
                flat(self.terminals)[ix])


def _ConvQ(nn, height, width, channels, n_layers, n_actions):
    """The conv body of policy_cnn.py:17-81 with a Q head: rows of n_cells cell values in, n_actions scores out."""

    class ConvQ(nn.Module):
        def __init__(self):
            super().__init__()

            def conv3(n_in):
                return nn.Sequential(nn.Conv2d(n_in, channels, kernel_size=3, stride=1, padding=1), nn.ReLU())

            self.network = nn.Sequential(conv3(1), *[conv3(channels) for _ in range(n_layers - 1)])
            self.bottleneck = nn.Conv2d(1, channels, kernel_size=1, stride=1)
            self.head_cnn = conv3(channels)
            self.head_linear = nn.Linear(channels * height * width, n_actions)

        def forward(self, x):
            x = x.reshape(-1, 1, height, width)
            trunk = self.network(x) + self.bottleneck(x)
            return self.head_linear(self.head_cnn(trunk).flatten(1))

    return ConvQ()


class BatchedDeepQAgent:
    reads_boards = True  # acts on the materialised cells (batched_default_eval must keep writing them)

    def __init__(self, env, args, sgd_steps=1, replay_slices=8, fused_learn=True, q_body=None, reference_loss_broadcast=True,
                 sync_target_at_start=False, fused_conv=True):
        """reference_loss_broadcast: F.mse_loss on Qs [B,1] vs expected_Qs [B] as value.py:119-123 writes it (False: squeezed, the
        textbook per-sample loss -- NOT the reference). sync_target_at_start: copy Q into the target network in the constructor
        (NOT the reference: value.py:82-84 initialises the two networks independently)."""
        import torch

        self.torch = torch
        self.env = env
        self.q_body = q_body or getattr(args, "q_body", None) or "mlp"
        if self.q_body not in ("mlp", "cnn"):
            raise ValueError("q_body must be 'mlp' (the reference's DeepQAgent) or 'cnn' (non-parity option)")
        self.n_channels = int(getattr(args, "n_channels", None) or 5)  # policy_cnn.py's default (agent_parser_configs.yaml:107-111)
        self.device = "cuda:%d" % env.device
        self.action_n = env.action_space.n
        self.n_input = env.n_cells
        self.discount, self.lr = float(args.discount), float(args.lr)
        self.batch_size, self.sync_every = int(args.batch_size), int(args.sync_every)
        self.sgd_steps = int(sgd_steps)
        self.eps0, self.anneal = float(args.epsilon), int(args.epsilon_anneal)
        self.t = 0  # lockstep steps taken == update_epsilon() calls
        n_layers, n_hidden = int(args.n_layers), int(args.n_hidden)
        self.fused_learn = False  # set below; sync_target_Q looks at it
        self.fuse_reset = True  # with the fused learner: reset_done + next-states store inside the SGD step's Adam launch
        self.Q = self.build_Q(self.n_input, n_layers, n_hidden).to(self.device).eval()
        self.target_Q = self.build_Q(self.n_input, n_layers, n_hidden).to(self.device).eval()
        self.reference_loss_broadcast = bool(reference_loss_broadcast)
        if sync_target_at_start:
            self.sync_target_Q()
        # capturable: Adam's step counters live on the device, so optim.step() can be recorded in a hipGraph; fused: one
        # kernel for all parameters instead of a dozen foreach kernels (learning iteration 483 -> 266 us, same box)
        self.optim = torch.optim.Adam(self.Q.parameters(), lr=self.lr, amsgrad=True, capturable=True, fused=True)
        self.replay = DeviceReplay(env.n_envs, env.n_cells, replay_slices, self.device)
        self._obs = torch.empty((env.n_envs, env.n_cells), dtype=torch.float32, device=self.device)
        self.last_loss = None
        self._actions = torch.empty(env.n_envs, dtype=torch.uint8, device=self.device)
        # fused forward + act_explore kernel (sgk_policy_act): two layers of 100 (the reference default), 64 or 128 units
        self.fused_policy = (self.q_body == "mlp" and n_layers == 2 and n_hidden in (64, 100, 128) and self.action_n == 4
                             and env.n_cells in (25, 30, 36, 48, 49, 56, 63))
        if self.fused_policy:
            l1, l2, l3 = self.Q[0][0], self.Q[1][0][0], self.Q[2]
            self._fw = {"w1t": torch.empty((env.n_cells, n_hidden), device=self.device), "b1": l1.bias.data,
                        "w2": l2.weight.data, "b2": l2.bias.data, "w3t": torch.empty((n_hidden, 4), device=self.device),
                        "b3": l3.bias.data}
            self._fw_stale = True
        # the conv body's forward + act_explore as ONE kernel (sgk_convq_act; the kernel reads torch's parameters in place)
        self.fused_conv = (fused_conv and self.q_body == "cnn" and n_layers == 2 and self.n_channels in (4, 5, 8) and self.action_n == 4
                           and (int(env.H), int(env.W)) in ((5, 5), (6, 5), (6, 6), (6, 8), (7, 7), (7, 8), (7, 9)))
        if self.fused_conv:
            q = self.Q
            self._cw = {"w1": q.network[0][0].weight.data, "b1": q.network[0][0].bias.data, "w2": q.network[1][0].weight.data,
                        "b2": q.network[1][0].bias.data, "wb": q.bottleneck.weight.data, "bb": q.bottleneck.bias.data,
                        "wh": q.head_cnn[0].weight.data, "bh": q.head_cnn[0].bias.data, "wl": q.head_linear.weight.data,
                        "bl": q.head_linear.bias.data}
        # DeepQAgent.learn as ONE kernel (sgk_dqn_sgd_step: sampling, both forwards, TD target, backward, grad clip, Adam
        # amsgrad) for the two-layer topology with up to 128 units and minibatches up to 64; else torch autograd + Adam
        lds_need = 4 * (4 * 64 * n_hidden + n_hidden * n_hidden + 12 * n_hidden + 872) + 128 * ((env.n_cells + 3) & ~3) + 64
        lanes_need = (n_hidden // 4) ** 2 + n_hidden  # one lane per 4 x 4 tile of W2 plus the bias lanes, of 1 024
        if (fused_learn and self.fused_policy and n_hidden in (64, 100) and lanes_need <= 1024 and self.batch_size <= 64
                and lds_need <= 160 * 1024):
            params = [p.data for p in self.Q.parameters()]  # w1, b1, w2, b2, w3, b3 (torch registration order)
            zeros = lambda: [torch.zeros_like(p) for p in params]  # noqa: E731
            self._fl = {"w2t": torch.empty((n_hidden, n_hidden), device=self.device), "m": zeros(), "v": zeros(), "vmax": zeros(),
                        "tw1t": torch.empty((env.n_cells, n_hidden), device=self.device),
                        "tw2t": torch.empty((n_hidden, n_hidden), device=self.device),
                        "step": torch.zeros(1, dtype=torch.int64, device=self.device),
                        "loss": torch.zeros(1, dtype=torch.float32, device=self.device)}
            self._refresh_fused_weights()
            self._fl["w2t"].copy_(self.Q[1][0][0].weight.data.t())
            self.fused_learn = True
            self._refresh_target_transposes()
        self._eps_dev = torch.ones(1, dtype=torch.float64, device=self.device)
        self._draw_dev = torch.zeros(1, dtype=torch.int64, device=self.device)
        self._graphs = {}

    def build_Q(self, n_input, n_layers, n_hidden):
        nn = self.torch.nn
        if self.q_body == "cnn":
            return _ConvQ(nn, int(self.env.H), int(self.env.W), self.n_channels, n_layers, int(self.action_n))
        first = nn.Sequential(nn.Linear(n_input, n_hidden), nn.ReLU())
        hidden = nn.Sequential(*[nn.Sequential(nn.Linear(n_hidden, n_hidden), nn.ReLU()) for _ in range(n_layers - 1)])
        return nn.Sequential(first, hidden, nn.Linear(n_hidden, int(self.action_n)))

    # epsilon schedule of DeepQAgent (value.py:70-76,142-146): NO overwrite to 0.0, so step 0 acts with 1.0
    @property
    def epsilon(self):
        t = min(self.t, self.anneal - 1)
        return 1.0 - (1 - self.eps0) * t / self.anneal

    def update_epsilon(self):
        self.t += 1
        return self.epsilon

    def sync_target_Q(self):
        self.target_Q.load_state_dict(self.Q.state_dict())
        if self.fused_learn:
            self._refresh_target_transposes()

    def _refresh_target_transposes(self):
        self._fl["tw1t"].copy_(self.target_Q[0][0].weight.data.t())
        self._fl["tw2t"].copy_(self.target_Q[1][0][0].weight.data.t())

    def _learn_batch_fused(self, rows=None, rows_out=None, reset_store=None):
        """One call of sgk_dqn_sgd_step; the kernel also keeps W1^T / W2^T / W3^T current, so the fused policy kernel needs no
        refresh afterwards. rows: int64 device tensor [batch] of transition indices (slice * n_envs + env) to train on instead of
        the kernel's own draw; rows_out: int64 device tensor [batch] that receives the indices used. reset_store = (states ring, slice,
        slice_dev): sgk_dqn_sgd_step_reset_store -- the lockstep step's reset_done + next-states store ride in the Adam launch."""
        import ctypes

        from . import _lib

        rp, fl, fw = self.replay, self._fl, self._fw
        for t, what in ((rows, "rows"), (rows_out, "rows_out")):
            if t is not None:
                self.env._check(t, what, shape=(self.batch_size,), dtypes=("int64",))
        q = [p.data for p in self.Q.parameters()]
        t1, t2, t3 = self.target_Q[0][0], self.target_Q[1][0][0], self.target_Q[2]
        ptr = lambda x: ctypes.c_void_p(x.data_ptr())  # noqa: E731
        arr = lambda xs: (ctypes.c_void_p * 6)(*[x.data_ptr() for x in xs])  # noqa: E731
        L = _lib.SgkDqnLearner(
            states=ptr(rp.states), successors=ptr(rp.successors), actions=ptr(rp.actions), rewards=ptr(rp.rewards),
            terminals=ptr(rp.terminals), slices_filled=int(rp.filled), n_hidden=int(q[1].numel()), batch=int(self.batch_size),
            loss_mode=_lib.DQN_LOSS_REFERENCE if self.reference_loss_broadcast else _lib.DQN_LOSS_PER_SAMPLE, w1=ptr(q[0]), b1=ptr(q[1]), w2=ptr(q[2]), b2=ptr(q[3]), w3=ptr(q[4]), b3=ptr(q[5]), w1t=ptr(fw["w1t"]),
            w2t=ptr(fl["w2t"]), w3t=ptr(fw["w3t"]), m=arr(fl["m"]), v=arr(fl["v"]), vmax=arr(fl["vmax"]), tw1t=ptr(fl["tw1t"]),
            tb1=ptr(t1.bias.data), tw2t=ptr(fl["tw2t"]), tb2=ptr(t2.bias.data), tw3=ptr(t3.weight.data), tb3=ptr(t3.bias.data),
            step=ptr(fl["step"]), loss_out=ptr(fl["loss"]), lr=self.lr, beta1=0.9, beta2=0.999, eps=1e-8,
            discount=self.discount, max_grad_norm=10.0, rows=None if rows is None else ptr(rows),
            rows_out=None if rows_out is None else ptr(rows_out))
        env = self.env
        if reset_store is not None:
            ring, sl, sl_dev = reset_store
            S_ = int(ring.shape[0])
            env._check(ring, "states ring", shape=(S_, env.n_envs, env.n_cells), dtypes=("int8",))
            sd = None if sl_dev is None else ctypes.c_void_p(env._check(sl_dev, "slice_dev", numel=1, dtypes=("int64",)).data_ptr())
            env._version += 1
            env._sync_torch_to_lib()
            _lib.check(env.lib.sgk_dqn_sgd_step_reset_store(env._h.ptr, ctypes.byref(L), 0, int(sl), sd, S_, ctypes.c_void_p(ring.data_ptr())))
        else:
            env._sync_torch_to_lib()
            _lib.check(env.lib.sgk_dqn_sgd_step(env._h.ptr, ctypes.byref(L)))
        env._sync_lib_to_torch()
        self._fw_stale = False
        self.last_loss = fl["loss"]
        return self.last_loss

    def greedy_weights(self):
        """Q-network weights in the fused kernels' layout (greedy evaluation, batched_default_eval), or None."""
        if self.fused_conv:
            return self._cw  # (torch's own tensors: sgk_convq_rollout reads them in place)
        if not self.fused_policy:
            return None
        if self._fw_stale:
            self._refresh_fused_weights()
        return self._fw

    def act_rollout(self, n_steps, epsilon=0.0, auto_reset=True):
        """n_steps of acting with the current (frozen) Q-network in one launch: forward, epsilon-greedy draw with a FIXED
        epsilon, env.step (evaluation and data collection; learning schedules epsilon per step and uses step())."""
        weights = self.greedy_weights()
        if weights is None:
            raise ValueError("act_rollout needs a fused policy kernel (two layers of 64 / 100 / 128 units, or the conv body)")
        if self.fused_conv:
            self.env.convq_rollout(weights, n_steps, self.n_channels, mode="greedy", epsilon=epsilon, draw_index0=self.t, auto_reset=auto_reset)
        else:
            self.env.policy_rollout(weights, n_steps, mode="greedy", epsilon=epsilon, draw_index0=self.t, auto_reset=auto_reset)
        self.t += int(n_steps)

    def _refresh_fused_weights(self):
        """The kernel wants W1 and W3 transposed (rows of W1^T / W3^T are what one input cell / one hidden unit multiplies);
        the other four tensors are read in place from the torch parameters (optimiser steps update them in place)."""
        self._fw["w1t"].copy_(self.Q[0][0].weight.data.t())
        self._fw["w3t"].copy_(self.Q[2].weight.data.t())
        self._fw_stale = False

    def scores(self, obs=None):
        obs = self.env.obs_f32(self._obs) if obs is None else obs
        with self.torch.no_grad():
            return self.Q(obs)

    def act(self, obs=None):
        if self.fused_policy and obs is None:
            if self._fw_stale:
                self._refresh_fused_weights()
            return self.env.policy_act(self._fw, 0.0, self.t, out=self._actions)
        if self.fused_conv and obs is None:
            return self._conv_act(0.0, self.t)
        return self.scores(obs).argmax(1).to(self.torch.uint8)

    def act_explore(self, obs=None):
        """Categorical(eps/n everywhere + (1 - eps) on the argmax), sampled for every env (value.py:94-111): one HIP kernel
        (argmax + counter-RNG draw) when the action space is the usual 4, a torch composition otherwise."""
        torch = self.torch
        if self.fused_policy and obs is None:
            if self._fw_stale:
                self._refresh_fused_weights()
            return self.env.policy_act(self._fw, self.epsilon, self.t, out=self._actions)
        if self.fused_conv and obs is None:
            return self._conv_act(self.epsilon, self.t)
        scores = self.scores(obs)
        if self.action_n == 4:
            return self.env.epsilon_greedy(scores, self.epsilon, self.t, out=self._actions)
        greedy = scores.argmax(1)
        n = greedy.shape[0]
        explore = torch.rand(n, device=self.device) < self.epsilon
        rand_a = torch.randint(0, self.action_n, (n,), device=self.device)
        return torch.where(explore, rand_a, greedy).to(torch.uint8)

    def _conv_act(self, epsilon, draw_index, scores_out=None):
        return self.env.convq_act(self._cw, epsilon, draw_index, self.n_channels, out=self._actions, scores_out=scores_out)

    def learn_batch(self):
        torch = self.torch
        if self.fused_learn:
            return self._learn_batch_fused()
        states, actions, rewards, successors, terminals = self.replay.sample(self.batch_size)
        self.Q.train()
        q_sa = self.Q(states.float()).gather(1, actions.long().unsqueeze(1)).squeeze(1)
        with torch.no_grad():
            next_q = self.target_Q(successors.float()).max(1)[0]
            next_q = torch.where(terminals, torch.zeros_like(next_q), next_q)
            expected = self.discount * next_q + (rewards.double() * float(self.env.reward_scale)).float()  # f64 product, then f32
        if self.reference_loss_broadcast:  # value.py:119-123: mse_loss([B,1], [B]) = the mean over the [B,B] broadcast
            loss = (q_sa.unsqueeze(1) - expected.unsqueeze(0)).square().mean()
        else:
            loss = torch.nn.functional.mse_loss(q_sa, expected)
        self.optim.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(self.Q.parameters(), 10.0)
        self.optim.step()
        self.Q.eval()
        if self.fused_policy:
            self._fw_stale = True
        self.last_loss = loss.detach()
        return self.last_loss

    def step(self, learn=True, cheat=False, explore=True):
        """One lockstep iteration of dqn_learn for every env: act_explore -> env.step -> replay.add -> learn ->
        update_epsilon -> (sync target) -> reset finished envs (the episode loop of train.py:62-70)."""
        env, rp = self.env, self.replay
        if learn and not rp.states_ready(env):
            rp.store(env, 0)  # the boards the agents act on are the transitions' states (already there when the previous step left them)
        if self.fused_policy:
            if self._fw_stale:
                self._refresh_fused_weights()
            actions = env.policy_act(self._fw, self.epsilon if explore else 0.0, self.t, out=self._actions)
        elif self.fused_conv:
            actions = self._conv_act(self.epsilon if explore else 0.0, self.t)
        else:
            env.obs_f32(self._obs)
            actions = self.act_explore(self._obs) if explore else self.act(self._obs)
        fused_reset = learn and self.fused_learn and self.fuse_reset  # the reset rides in the last SGD step's Adam launch
        if learn:
            # env.step + the rest of the add in ONE launch: successor boards, action, reward (hidden when cheating), terminal
            rp.step_store(env, actions, cheat)
            for i in range(self.sgd_steps):
                if fused_reset and i == self.sgd_steps - 1:
                    self._learn_batch_fused(reset_store=rp.reset_store_args())
                    rp.note_reset_store(env)
                else:
                    self.learn_batch()
        else:
            env.step(actions, auto_reset=True)  # (step + reset of the finished envs in one launch: nobody needs the terminal boards)
        t = self.t
        self.update_epsilon()
        if learn and t % self.sync_every == self.sync_every - 1:
            self.sync_target_Q()
        if learn and not fused_reset:
            rp.reset_store(env)  # reset_done + the NEXT transition's states in one launch
        return actions

    # ---- the same lockstep iteration replayed from ONE hipGraph (torch.cuda.CUDAGraph) -------------------------------
    # Eager PyTorch costs ~10 us of host time per op; one iteration is ~15 ops without learning and ~100 with, so at
    # 32 768 envs the GPU idles most of the time. Capturing the iteration (the library's obs / step / reset kernels are
    # plain launches on the capture stream) removes that: everything that varies between replays lives in device memory
    # (epsilon scalar, replay ring head, Adam's capturable step counter).
    def _captured_iteration(self, learn, cheat=False):
        """act -> step (+ the add's second half) -> [SGD] -> reset (+ the NEXT transition's states): states[head] must hold the current
        boards when it starts (enable_graphs / step_graphed see to it: DeviceReplay.states_ready)."""
        torch = self.torch
        env = self.env
        if self.fused_policy:  # epsilon and the draw index are read from device memory: they advance between replays
            if (learn and not self.fused_learn) or self._fw_stale:
                self._refresh_fused_weights()  # recorded in the learn graph: torch's update leaves the transposes behind
            actions = env.policy_act(self._fw, self._eps_dev, self._draw_dev, out=self._actions)
        elif self.fused_conv:
            actions = self._conv_act(self._eps_dev, self._draw_dev)
        elif self.action_n == 4:
            env.obs_f32(self._obs)
            with torch.no_grad():
                scores = self.Q(self._obs)
            actions = env.epsilon_greedy(scores, self._eps_dev, self._draw_dev, out=self._actions)
        else:
            env.obs_f32(self._obs)
            with torch.no_grad():
                scores = self.Q(self._obs)
            greedy = scores.argmax(1)
            n = greedy.shape[0]
            explore = torch.rand(n, device=self.device) < self._eps_dev
            rand_a = torch.randint(0, self.action_n, (n,), device=self.device)
            actions = torch.where(explore, rand_a, greedy).to(torch.uint8)
        if learn:
            self.replay.step_store(env, actions, cheat, captured=True)  # --cheat: hidden reward + executed action (learn.py:41-47)
            self.replay.head_dev.add_(1).remainder_(self.replay.slices)
            fused_reset = self.fused_learn and self.fuse_reset
            for i in range(self.sgd_steps):
                if fused_reset and i == self.sgd_steps - 1:
                    self._learn_batch_fused(reset_store=self.replay.reset_store_args(captured=True))
                    self.replay.note_reset_store(env)
                else:
                    self.learn_batch()
            if not fused_reset:
                self.replay.reset_store(env, captured=True)  # (head_dev already names the next slice)
        else:
            env.step(actions, auto_reset=True)

    def enable_graphs(self, learn=True, cheat=False):
        """Capture one lockstep iteration. Needs a full replay ring (run warmup(replay_slices) first) so that the sampling
        range is a constant, and Adam(capturable=True). `cheat` (learn.py:41-47: learn from the hidden reward and the action
        the env executed) is part of what is recorded: graphs are kept per (learn, cheat)."""
        torch = self.torch
        cheat = bool(cheat) and bool(learn)
        if (learn, cheat) in self._graphs:
            return
        if learn:
            if self.replay.filled != self.replay.slices:
                raise ValueError("fill the replay ring (warmup) before capturing the learn graph")
        env = self.env
        if learn:
            self.replay.head_dev.fill_(self.replay.head)
            self.replay.head_dev_stale = False
        with torch.no_grad() if not learn else torch.enable_grad():
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):
                env.bind_torch_stream(side)
                if learn and not self.replay.states_ready(env):
                    self.replay.store(env, 0, captured=True)  # states[head_dev] = the boards the first recorded action is chosen on
                for _ in range(3):  # warm-up on the side stream (allocator, lazy init), as the capture recipe requires
                    self._captured_iteration(learn, cheat)
                    if learn:
                        self.replay.note_replayed_add()
            torch.cuda.current_stream(self.device).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                env.bind_torch_stream(torch.cuda.current_stream(self.device))  # the capture stream
                self._captured_iteration(learn, cheat)
            env.bind_torch_stream()  # follow torch's current stream again: replays are launched on it
        env.account_steps(-1)  # the recorded (not executed) sgk_step bumped the host-side counters once
        if learn:
            self.replay.ready_version = env._version  # the last warm-up iteration's reset_store filled states[head]
        self._graphs[(learn, cheat)] = graph

    def step_graphed(self, learn=True, cheat=False):
        """One lockstep iteration = one graph replay (+ two scalar updates); enable_graphs(learn, cheat) first."""
        cheat = bool(cheat) and bool(learn)
        self._eps_dev.fill_(self.epsilon)
        self._draw_dev.fill_(self.t)
        if learn and self.replay.head_dev_stale:  # eager adds moved the host-side head since the last replay
            self.replay.head_dev.fill_(self.replay.head)
            self.replay.head_dev_stale = False
        if self.fused_policy and self._fw_stale and not learn:
            self._refresh_fused_weights()  # the no-learning graph does not record the transposes
        if learn and not self.replay.states_ready(self.env):
            self.replay.store(self.env, 0, captured=True)  # (something else moved the envs since the last learning step)
        self._graphs[(learn, cheat)].replay()
        self.env.account_steps(1)
        if learn:
            self.replay.note_replayed_add()
            self.replay.ready_version = self.env._version  # the replayed reset_store filled states[head]
        else:
            self.replay.ready_version = None
        t = self.t
        self.update_epsilon()
        if learn and t % self.sync_every == self.sync_every - 1:
            self.sync_target_Q()

    def warmup(self, n_steps, reference_state=True):
        """dqn_warmup (warmup.py:8-23) for every env in ONE launch of the streamed random rollout (sgk_rollout_random_stream, the
        headline kernel): RandomAgent's actions from counter-RNG stream 0, every step's successor board straight into the replay's
        `successors` ring, the step records into a scratch ring that three strided copies split into actions / rewards / terminals.
        reference_state=True keeps warmup.py:17-21 as written: `state` is assigned at env.reset() only, so every row of an episode
        stores the episode's FIRST board as its state; False stores the board each action was taken on.
        (A finished episode's row holds the next episode's first board as successor -- the ring convention; DeepQAgent.learn zeroes
        next_Q there, value.py:121, and never reads it. The stored action is the executed one: WhiskyGold's replaced actions
        appear as executed.)"""
        torch = self.torch
        env, rp = self.env, self.replay
        n_steps = int(n_steps)
        if n_steps <= 0:
            return
        if n_steps > rp.slices:
            raise ValueError("warmup(%d) exceeds the replay ring of %d slices" % (n_steps, rp.slices))
        n, cells = env.n_envs, env.n_cells
        initial = env.boards().reshape(n, cells).clone()  # (a copy also when the rows are pitched: the view's logical shape is [N,1,H,W])
        recs = torch.empty((rp.slices, n, 4), dtype=torch.int8, device=self.device)
        first = rp.head
        env.rollout_random_stream(n_steps, boards=rp.successors, recs=recs, first_slice=first)
        idx = (first + torch.arange(n_steps, device=self.device)) % rp.slices
        rec = recs[idx]
        rp.actions[idx] = rec[:, :, 3].contiguous().view(torch.uint8)
        rp.rewards[idx] = rec[:, :, 0]
        term = rec[:, :, 2] != 0
        rp.terminals[idx] = term
        before = torch.cat([initial[None], rp.successors[idx][:-1]])  # the board every step acted on
        if reference_state:
            starts = torch.cat([torch.ones((1, n), dtype=torch.bool, device=self.device), term[:-1]])  # step k opens an episode
            k = torch.arange(n_steps, device=self.device)[:, None] * starts
            k = torch.cummax(k, dim=0).values
            before = before.gather(0, k[:, :, None].expand(-1, -1, cells))
        rp.states[idx] = before
        rp.head = (first + n_steps) % rp.slices
        rp.filled = min(rp.filled + n_steps, rp.slices)
        rp.head_dev_stale = True
