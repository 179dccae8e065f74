// sgk_kernels.h -- internal interface between the C-ABI layer (sgk_api.hip) and the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sgk.h"
#include "sgk_mailbox.h"
#include "sgk_rules.h"

#define SGK_METRIC_SLOTS 2048

namespace sgk {

// The device buffer behind Shard::rules_dev: the rule table padded to SGK_RULES_IMAGE_BYTES, then a blank 64-env COMPACT board tile
// (the level's backdrop 64 times over, n_cells bytes each) -- what the per-step kernel's waves load at entry (sgk_device.h).
constexpr int SGK_RULES_IMAGE_BYTES = 2048;
constexpr int SGK_RULES_DEV_BYTES = SGK_RULES_IMAGE_BYTES + 64 * SGK_CELLS;

// Device-resident arrays of one shard (N envs on one GPU). Structure-of-arrays over envs.
struct Shard {
  int env_id = 0, layout = 0, device = 0;
  int n_cells = 0, pitch = 0, n_states = 0;
  int n_cus = 256, max_grid = 2048, stream_grid = 2048, rollout_grid = 2048;
  int ring_nt_mode = -1;  // SGK_RING_NT: 0 / 1 = ring stores never / always non-temporal; -1 = by ring size and slices (sgk_step.hip)
  int64_t n = 0;
  uint64_t seed = 0, env_base = 0, lockstep_t = 0;
  SgkRules rules_host;
  SgkRules *rules_dev = nullptr;
  uint64_t *state = nullptr;      // [n] packed: agent cell | box cell | frame | flags | int16 return | int16 hidden
  uint32_t *rec = nullptr;        // [n] sgk_step_rec
  int8_t *boards = nullptr;       // [n_padded][pitch] int8 cells
  int32_t *last_return = nullptr; // [n] episode_return of the last finished episode
  int32_t *last_perf = nullptr;   // [n] get_last_performance()
  int32_t *n_episodes = nullptr;  // [n]
  int32_t *n_resets = nullptr;    // [n] how often the env has been reset (create included); maintained for the envs with
                                  // draws of their own (whisky, absent supervisor, safe interruptibility), whose key it is
  double *aux = nullptr;          // [n][6] float64 side state that outlives episodes: friend or foe's bandit estimates (null elsewhere)
  int64_t *metrics = nullptr;     // [SGK_METRICS_LEN] reduced vector (valid after launch_metrics_reduce)
  int64_t *metric_slab = nullptr; // [SGK_METRIC_SLOTS][SGK_METRICS_LEN] per-workgroup partials
  int32_t *wg_count = nullptr;    // compaction scratch
  int64_t *wg_offset = nullptr;
  int64_t *finished_total = nullptr;
};

struct TabqShard {
  double *table = nullptr;   // [n_states][n][4] float64, STATE-major: row (s, e) at ((s * n) + e) * 4 (sgk_tabq.hip: row_of)
  uint64_t *tags = nullptr;     // [n] low word: state index the last action was chosen from (0xffffffff = env was over);
                                //     high word: state index of the row in row_cache (0xffffffff = none)
  double *row_cache = nullptr;  // [n][4] the Q row of the state named by the tag: what the per-step kernels hand each other
  // Levels whose boards have no perfect hash (tomato watering: 63 cells x 2^13 watered sets): every agent's table is an
  // open-addressing hash table of `hash_cap` slots (a power of two, 64 .. 2^24), keys[n][hash_cap] (0xffffffff = empty) beside the
  // rows table[hash_cap][n][4]; a slot is claimed the first time a board is looked up -- the defaultdict of value.py:31-36.
  uint32_t *keys = nullptr;
  int32_t hash_cap = 0;          // 0: perfect-hash level
  int32_t *hash_overflow = nullptr;  // [2]: [0] set when some agent's table was full and a board found no row (it then reads zeros
                                     // and learns nothing: re-create larger); [1] scratch of sgk_tabq_hash_info's slot count
  int32_t n_states = 0;          // rows per agent: the level's n_states, or hash_cap
  double lr = 0, discount = 0, eps0 = 0;
  int64_t anneal = 0;
  int64_t t_agent = 0;
  const long long *t_ptr = nullptr;  // when set (hipGraph replays): the global agent step is *t_ptr + t_agent
};

hipError_t launch_step(const Shard &sh, const uint8_t *actions, uint32_t flags, hipStream_t st);
// the single-env step server (its mailbox: sgk_mailbox.h)
hipError_t launch_env_server(const Shard &sh, const uint8_t *actions, SgkMailbox *mb, uint32_t last, hipStream_t st);
hipError_t launch_rollout_random(const Shard &sh, int32_t n_steps, uint32_t flags, hipStream_t st);
// the same loop with every step's board tile and step record materialised: into the env's own buffers (rings == nullptr) or
// into trajectory rings boards [ring][n][n_cells] / recs [ring][n], step k -> slice (slice0 + k) % ring
hipError_t launch_rollout_stream(const Shard &sh, int32_t n_steps, uint32_t flags, int8_t *boards_ring, uint32_t *recs_ring,
                                 int32_t ring, int32_t slice0, hipStream_t st);
// mode 0: reset all (mask == nullptr) or masked envs; 1: reset envs whose episode is over; 2: re-materialise boards only;
// | 4: touch no boards (state words only)
hipError_t launch_reset(const Shard &sh, const uint8_t *mask, int mode, hipStream_t st);
hipError_t launch_ring_probe(const Shard &sh, int8_t *boards_ring, uint32_t *recs_ring, int32_t ring, uint32_t flags, hipStream_t st);
hipError_t launch_metrics_init(const Shard &sh, hipStream_t st);
hipError_t launch_aux_init(const Shard &sh, hipStream_t st);
// out_host: optional second destination in pinned device-mapped host memory (the synchronising reader then needs no copy)
hipError_t launch_metrics_reduce(const Shard &sh, hipStream_t st, long long *out_host = nullptr);
hipError_t launch_obs_f32(const Shard &sh, float *dst, hipStream_t st);
struct PolicyWeights {
  const float *w1t, *b1, *w2, *b2, *w3t, *b3;
  int n_hidden;
};
// mode 0: epsilon-greedy over the scores (DeepQAgent.act_explore); mode 1: Categorical(logits = scores).sample() (PPO)
hipError_t launch_policy_act(const Shard &sh, int mode, const PolicyWeights &w, uint8_t *actions, float *scores, double eps,
                             uint64_t draw, const double *eps_dev, const uint64_t *draw_dev, hipStream_t st);
// n_steps of {forward, draw, env.step} in one launch; trajectory outputs optional ([n_steps][n]...); SGK_F_AUTO_RESET in flags
hipError_t launch_policy_rollout(const Shard &sh, int mode, const PolicyWeights &w, double eps, uint64_t draw0, int32_t n_steps,
                                 uint32_t flags, int8_t *states_out, uint8_t *actions_out, uint32_t *recs_out, hipStream_t st);
hipError_t launch_eps_greedy(const Shard &sh, int mode, const float *scores, uint8_t *actions, double eps, uint64_t draw,
                             const double *eps_dev, const uint64_t *draw_dev, hipStream_t st);
// the conv Q-body's forward + act_explore in one launch (sgk_convq.hip; a labelled NON-parity option: the reference's DeepQAgent is an MLP)
struct ConvQWeights {
  const float *w1, *b1;  // conv3x3 1 -> C   [C][1][3][3], [C]
  const float *w2, *b2;  // conv3x3 C -> C   [C][C][3][3], [C]
  const float *wb, *bb;  // conv1x1 1 -> C   [C][1][1][1], [C]   (the residual bottleneck)
  const float *wh, *bh;  // conv3x3 C -> C   (head)
  const float *wl, *bl;  // linear C*H*W -> 4   [4][C*H*W], [4]
};
// mode 0: epsilon-greedy on the four scores (DeepQAgent.act_explore); 1: Categorical(logits = scores).sample() (PPOBaseAgent.act_explore)
hipError_t launch_convq_act(const Shard &sh, const ConvQWeights &w, int n_channels, int mode, uint8_t *actions, float *scores, double eps, uint64_t draw,
                            const double *eps_dev, const uint64_t *draw_dev, hipStream_t st);
// n_steps of {conv forward, draw, env.step} in one launch (sgk_convq_rollout.hip); outputs as for launch_policy_rollout
hipError_t launch_convq_rollout(const Shard &sh, const ConvQWeights &w, int n_channels, int mode, double eps, uint64_t draw0, int32_t n_steps,
                                uint32_t flags, int8_t *states_out, uint8_t *actions_out, uint32_t *recs_out, hipStream_t st);
// DeepQAgent.learn as one kernel (sgk_learn.hip); all pointers are device pointers
struct DqnLearner {
  const int8_t *states, *successors;
  const uint8_t *actions;
  const int8_t *rewards;
  const uint8_t *terminals;
  int slices_filled;
  float *w1, *b1, *w2, *b2, *w3, *b3, *w1t, *w2t, *w3t;
  float *m[6], *v[6], *vmax[6];
  const float *tw1t, *tb1, *tw2t, *tb2, *tw3, *tb3;
  long long *step;
  float *loss_out;
  int n_hidden, batch;
  int loss_mode;  // SGK_DQN_LOSS_*
  const long long *rows;  // caller's minibatch or null
  long long *rows_out;    // minibatch used, or null
  void *scratch;          // dqn_sgd_scratch_bytes() of device memory (zeroed once): Adam runs as a second, chip-wide launch; null: inside the one kernel
  int multi_wg;           // (-DSGK_DQN_MULTI_WG experiment build only) run the four-workgroup kernel on `scratch`
  double lr, beta1, beta2, eps, discount, max_grad_norm;
  // sgk_dqn_sgd_step_reset_store: the Adam launch also resets the finished envs and stores the next transitions' states (what
  // launch_reset_done_store does), as extra workgroups of the same grid
  int reset_store = 0;
  uint32_t rs_flags = 0;
  int64_t rs_slice = 0;
  const long long *rs_slice_dev = nullptr;
  int32_t rs_ring = 0;
  int8_t *rs_states_ring = nullptr;
};  // (the replay's int8 rewards are in units of the level's reward_scale: launch_dqn_sgd takes it from the shard's rules)
// env.step + the second half of ReplayBuffer.add / reset_done + the first half for the next step, one launch each (sgk_step.hip)
hipError_t launch_step_store(const Shard &sh, const uint8_t *actions, uint32_t flags, int cheat, int64_t slice, const long long *slice_dev,
                             int32_t ring, int8_t *successors, uint8_t *r_actions, int8_t *r_rewards, uint8_t *r_terminals, hipStream_t st);
hipError_t launch_reset_done_store(const Shard &sh, uint32_t flags, int64_t slice, const long long *slice_dev, int32_t ring, int8_t *states,
                                   hipStream_t st);
hipError_t launch_replay_store(const Shard &sh, int phase, const uint8_t *actions, int cheat, int64_t head, const long long *head_dev,
                               int8_t *states, int8_t *successors, uint8_t *r_actions, int8_t *r_rewards, uint8_t *r_terminals,
                               hipStream_t st);
size_t dqn_sgd_lds_bytes(int n_cells, int n_hidden);
size_t dqn_sgd_scratch_bytes(int n_cells, int n_hidden);
hipError_t launch_dqn_sgd(const Shard &sh, const DqnLearner &L, hipStream_t st);
struct PpoLearner {
  const int8_t *states;
  const uint8_t *actions;
  const float *returns;
  const int32_t *lengths;
  int horizon, n_hidden, batch, n_epochs;
  int64_t n_trajectories;
  float *w1, *b1, *w2, *b2, *wa, *ba, *wc, *bc, *w1t, *w2t;
  float *m[8], *v[8];
  const float *ow1t, *ob1, *ow2t, *ob2, *owa, *oba;
  long long *step;
  float *stats_out;
  const long long *rows;
  long long *rows_out;
  double lr, beta1, beta2, eps, clipping, critic_coeff, entropy_bonus;
};
size_t ppo_epochs_lds_bytes(int n_cells, int n_hidden);
hipError_t launch_ppo_epochs(const Shard &sh, const PpoLearner &P, hipStream_t st);
hipError_t launch_discounted_returns(const Shard &sh, const float *rewards, const int32_t *lengths, const float *gamma_pow,
                                     float *returns, int64_t n, int t_max, hipStream_t st);
hipError_t launch_render_rgb(const Shard &sh, uint8_t *dst, hipStream_t st);
hipError_t launch_dense_boards(const Shard &sh, int8_t *dst, hipStream_t st);
hipError_t launch_finished(const Shard &sh, int32_t *ids, int32_t *ret, int32_t *perf, hipStream_t st);
hipError_t launch_tabq_act(const Shard &sh, const TabqShard &tq, int explore, uint8_t *actions_out, hipStream_t st);
hipError_t launch_tabq_forget_rows(const Shard &sh, const TabqShard &tq, hipStream_t st);
hipError_t launch_tabq_hash_used(const Shard &sh, const TabqShard &tq, int32_t *max_used_dev, hipStream_t st);
hipError_t launch_tabq_learn(const Shard &sh, const TabqShard &tq, const uint8_t *actions, int cheat, hipStream_t st);
// one lockstep step of tabq_learn in one launch (act_explore, env.step, learn, reset of the finished envs); actions_out may be null
hipError_t launch_tabq_step(const Shard &sh, const TabqShard &tq, int cheat, uint32_t flags, uint8_t *actions_out, hipStream_t st);
hipError_t launch_tabq_rollout(const Shard &sh, const TabqShard &tq, int64_t n_steps, int cheat, hipStream_t st);
hipError_t launch_tabq_rollout_hbm(const Shard &sh, const TabqShard &tq, int64_t n_steps, int cheat, hipStream_t st);
size_t tabq_rollout_lds_bytes(const Shard &sh);

int host_random_action(uint64_t seed, uint64_t env, uint64_t t);  // sgk_host_debug.cpp
int host_debug_transition(const SgkRules &R, int agent_cell, int box_cell, int action, int out[5]);
int host_debug_step(const SgkRules &R, uint64_t word, int n_resets, int action, uint64_t seed, uint64_t env, uint64_t *word_out,
                    int out[4], double *aux);
uint64_t host_reset_word(const SgkRules &R, uint64_t seed, uint64_t env, int n_resets, const double *aux);
double host_epsilon_at(double eps0, int64_t anneal, int64_t t);

}  // namespace sgk
