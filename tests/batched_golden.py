"""Loader for tests/golden/batched_*.npz: the reference's own TabularQAgent / tabq_learn / RandomAgent / dqn_warmup run once per
env index with np.random answered from the batched path's counter RNG (tests/golden/make_golden.py, golden_batched_*). Shared by
the CPU test (the oracle reproduces them) and the GPU tests (the HIP kernels reproduce them)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

TABQ_FIXTURES = ["batched_tabq_boat.npz", "batched_tabq_island.npz", "batched_tabq_sokoban_cheat.npz",
                 "batched_tabq_whisky_cheat.npz", "batched_tabq_lava.npz", "batched_tabq_super.npz",
                 "batched_tabq_interrupt_cheat.npz", "batched_tabq_belt.npz", "batched_tabq_bandit.npz", "batched_tabq_tomato.npz"]
WARMUP_FIXTURES = ["batched_warmup_boat.npz", "batched_warmup_island.npz", "batched_warmup_sokoban.npz"]


def _num(v):
    return float.fromhex(v) if isinstance(v, str) else v


class TabqFixture:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name))
        m = json.loads(str(z["meta"]))
        self.meta = m
        self.env, self.cheat, self.seed = m["env"], m["cheat"], m["seed"]
        self.n, self.steps, self.eval_timesteps = m["n_agents"], m["steps"], m["eval_timesteps"]
        self.lr, self.discount, self.eps0, self.anneal = m["lr"], m["discount"], m["epsilon"], m["epsilon_anneal"]
        self.epsilon_used = m["epsilon_used"]  # hex f64, one per agent step (the same for every agent)
        self.agents = m["agents"]
        # [steps, n] like the oracle's / the product's action matrices
        self.actions = np.array([[int(c) for c in a["actions"]] for a in self.agents], dtype=np.uint8).T.copy()
        self.q_agent, self.q_boards, self.q_rows = z["q_agent"], z["q_boards"], z["q_rows"]
        self.final_boards = np.array([a["final_board"] for a in self.agents], dtype=np.int8)
        # what one unit of the integer rewards is worth: the reference sees floats on TomatoWatering (count * REWARD_FACTOR, summed
        # step by step), the batch counts tomatoes
        self.scale = 0.02 if self.env == "TomatoWatering-v0" else 1.0

    def units(self, v):
        """A reference reward / return (hex float or int) in the batch's integer units."""
        x = _num(v) / self.scale
        r = int(round(x))
        if not abs(x - r) < 1e-6 * max(1.0, abs(x)):
            raise AssertionError((v, x))
        return r

    def args(self):
        import types

        return types.SimpleNamespace(lr=self.lr, discount=self.discount, epsilon=self.eps0, epsilon_anneal=self.anneal)

    def rows_of(self, i):
        sel = np.nonzero(self.q_agent == i)[0]
        return [(self.q_boards[k], self.q_rows[k]) for k in sel]

    def metrics(self):
        """The 12 used words of the batch's metrics vector from the reference's per-episode TensorBoard scalars
        (meters.py:86-96: Train/returns, /safeties, /margins, /margins_support) of all agents. Integer-reward levels only."""
        sums = {k: 0 for k in ("returns", "safeties", "margins", "margins_support")}
        maxs = {k: None for k in sums}
        counts = {k: 0 for k in sums}
        for a in self.agents:
            for k in sums:
                for v in a["episodes"][k]:
                    v = self.units(v)
                    sums[k] += v
                    counts[k] += 1
                    maxs[k] = v if maxs[k] is None else max(maxs[k], v)
        return sums, counts, maxs


    def eval_metrics(self):
        """What the batch's metrics vector must hold after batched_default_eval: the reference's default_eval (eval.py:8-56) of every
        agent -- each episode's (episode_return, get_last_performance()) as its track_metrics calls saw them."""
        m = {"sum_return": 0, "sum_safety": 0, "sum_margin": 0, "sum_margin_pos": 0, "episodes": 0, "margin_pos_count": 0,
             "max_return": None, "max_safety": None, "max_margin": None, "max_margin_pos": None}

        def up(k, v):
            m[k] = v if m[k] is None else max(m[k], v)

        for a in self.agents:
            assert len(a["eval_episodes"]) >= 1
            for ret, perf in a["eval_episodes"]:
                ret, perf = self.units(ret), self.units(perf)
                margin = ret - perf
                m["sum_return"] += ret; m["sum_safety"] += perf; m["sum_margin"] += margin; m["episodes"] += 1
                up("max_return", ret); up("max_safety", perf); up("max_margin", margin)
                if margin > 0:
                    m["sum_margin_pos"] += margin; m["margin_pos_count"] += 1
                    up("max_margin_pos", margin)
        return m


def assert_eval_metrics(vec, fx, oracle_module):
    """vec: the 16-word metrics vector after the evaluation (oracle or device)."""
    O = oracle_module
    want = fx.eval_metrics()
    got = {"sum_return": vec[O.M_SUM_RETURN], "sum_safety": vec[O.M_SUM_SAFETY], "sum_margin": vec[O.M_SUM_MARGIN],
           "sum_margin_pos": vec[O.M_SUM_MARGIN_POS], "episodes": vec[O.M_EPISODES], "margin_pos_count": vec[O.M_MARGIN_POS_COUNT],
           "max_return": vec[O.M_MAX_RETURN], "max_safety": vec[O.M_MAX_SAFETY], "max_margin": vec[O.M_MAX_MARGIN],
           "max_margin_pos": vec[O.M_MAX_MARGIN_POS]}
    for k, v in want.items():
        if v is not None:
            if int(got[k]) != v:  # (raised, not asserted: this helper module is not one pytest rewrites -- under python -O a bare assert here checks nothing)
                raise AssertionError((k, int(got[k]), v))


class WarmupFixture:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name))
        m = json.loads(str(z["meta"]))
        self.meta = m
        self.env, self.seed, self.n, self.steps = m["env"], m["seed"], m["n_agents"], m["steps"]
        self.meters = m["returns_meters"]
        # [n, steps, ...] as recorded -> [steps, n, ...] like a trajectory ring
        self.states = z["states"].transpose(1, 0, 2).copy()
        self.successors = z["successors"].transpose(1, 0, 2).copy()
        self.actions = z["actions"].T.copy()
        self.rewards = z["rewards"].T.copy()
        self.terminals = z["terminals"].T.copy()


def hexes(row):
    return [float(x).hex() for x in row]


PPO_FIXTURES = ["batched_ppo_boat.npz", "batched_ppo_boat_cheat.npz", "batched_ppo_tomato.npz", "batched_ppo_island_gather.npz",
                "batched_ppo_whisky_cheat_gather.npz"]
# the reference's PPOCNNAgent (policy_cnn.py; 5 / 8 / 4 channels): what sgk_convq_sample fuses
PPO_CNN_FIXTURES = ["batched_ppo_cnn_boat.npz", "batched_ppo_cnn_sokoban_gather.npz", "batched_ppo_cnn_island_cheat_gather.npz"]


class PpoFixture:
    """tests/golden/batched_ppo_*.npz (make_golden.py:golden_batched_ppo): the reference's train() with PPOMLPAgent, rollout r of
    every gather_rollout = the episode of env index base + r, Categorical.sample() and torch.randint answered from the batch's
    counter RNG. Iteration k holds what gather_rollout returned (states / actions / rewards / returns [n, horizon], zero past an
    episode's end), the old policy's logits at every step, the minibatch rows of its epochs, and the weights after its learn."""

    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name))
        m = json.loads(str(z["meta"]))
        self.z, self.meta = z, m
        self.env, self.cheat, self.seed, self.base = m["env"], m["cheat"], m["seed"], m["base"]
        self.n, self.horizon, self.iterations, self.learn = m["n"], m["horizon"], m["iterations"], m["learn"]
        self.eval_timesteps, self.agents = m["eval_timesteps"], m["agents"]
        self.scale = 0.02 if self.env == "TomatoWatering-v0" else 1.0

    units = TabqFixture.units
    eval_metrics = TabqFixture.eval_metrics

    def args(self, device):
        import types

        m = self.meta
        return types.SimpleNamespace(discount=m["discount"], lr=m["lr"], batch_size=m["batch_size"], rollouts=self.n, epochs=m["epochs"],
                                     clipping=m["clipping"], entropy_bonus=m["entropy_bonus"], critic_coeff=m["critic_coeff"],
                                     n_layers=m["n_layers"], n_hidden=m["n_hidden"], n_channels=m.get("n_channels"), device=device,
                                     log_gradients=False, cheat=self.cheat)

    def it(self, k, what):
        return self.z["it%d_%s" % (k, what)]

    def weights(self, k):
        """The agent's own parameters before iteration k's learn (k = iterations: at the end), keyed like its state_dict."""
        return {key: self.z["w%d_%s" % (k, key)] for key in self.meta["weight_keys"]}

    def losses(self, k):
        """[epochs, 3] float64: policy loss, value loss, entropy of iteration k's epochs as the reference logged them."""
        e = self.meta["epochs"]
        rows = [c for c in self.meta["losses"] if k * e <= c[2] < (k + 1) * e]
        out = np.zeros((e, 3))
        for tag, v, step in rows:
            out[step - k * e, ("Train/policy_loss", "Train/value_loss", "Train/policy_entropy").index(tag)] = _num(v)
        return out

    def gather_metrics(self, k):
        """The metrics vector's sums / counts / maxima of iteration k's gathered episodes (track_metrics, policy_base.py:168)."""
        m = {"sum_return": 0, "sum_safety": 0, "sum_margin": 0, "sum_margin_pos": 0, "episodes": 0, "margin_pos_count": 0}
        for it, _, ret, perf in self.meta["episodes"]:
            if it != k:
                continue
            ret, perf = self.units(ret), self.units(perf)
            margin = ret - perf
            m["sum_return"] += ret; m["sum_safety"] += perf; m["sum_margin"] += margin; m["episodes"] += 1
            if margin > 0:
                m["sum_margin_pos"] += margin; m["margin_pos_count"] += 1
        return m


DQN_FIXTURES = ["batched_dqn_sokoban.npz", "batched_dqn_boat_cheat.npz"]


class DqnFixture:
    """tests/golden/batched_dqn_*.npz (make_golden.py:golden_batched_dqn): the reference's train() with its DeepQAgent, dqn_warmup and
    dqn_learn on ONE env index, np.random.randint / Categorical.sample / np.random.choice answered from the batch's counter RNG (streams
    0 / 2 / 4). Holds both networks' initial weights, the replay after the warm-up (row k = warm-up step k), every agent step's action,
    epsilon, loss, Q-values and minibatch ring slots, the weights at every sync and at the end, and the final greedy evaluation."""

    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name))
        m = json.loads(str(z["meta"]))
        self.z, self.meta = z, m
        self.env, self.cheat, self.seed, self.index, self.steps = m["env"], m["cheat"], m["seed"], m["index"], m["steps"]
        self.capacity, self.batch, self.sync_every = m["replay_capacity"], m["batch_size"], m["sync_every"]
        self.eval_timesteps = m["eval_timesteps"]
        self.agents = [{"eval_episodes": m["eval_episodes"], "eval_actions": m["eval_actions"]}]
        self.epsilon_used = [float.fromhex(e) for e in m["epsilon_used"]]
        self.actions, self.losses, self.rows, self.scores, self.gaps = z["actions"], z["losses"], z["rows"], z["scores"], z["gaps"]
        self.explored = np.array(m["explored"], dtype=bool)
        self.scale = 1.0

    units = TabqFixture.units
    eval_metrics = TabqFixture.eval_metrics

    def args(self):
        import types

        m = self.meta
        return types.SimpleNamespace(discount=m["discount"], lr=m["lr"], batch_size=m["batch_size"], sync_every=m["sync_every"],
                                     epsilon=m["epsilon"], epsilon_anneal=m["epsilon_anneal"], n_layers=m["n_layers"],
                                     n_hidden=m["n_hidden"], replay_capacity=m["replay_capacity"], cheat=self.cheat)

    def weights(self, tag):
        """tag: init_Q, init_T, final_Q, final_T, sync<i>_Q -> arrays keyed like the network's state_dict."""
        return {key: self.z["%s_%s" % (tag, key.replace(".", "_"))] for key in self.meta["weight_keys"]}

    def warm(self, what):
        return self.z["warm_" + what]

    def episode_metrics(self):
        """Sums / counts of the training episodes the reference's track_metrics booked (Train/returns, ... per episode)."""
        e = self.meta["episodes"]
        rets = [self.units(v) for v in e["returns"]]
        perfs = [self.units(v) for v in e["safeties"]]
        return {"episodes": len(rets), "sum_return": sum(rets), "sum_safety": sum(perfs)}
