// sgk_tabq.hip -- TabularQAgent.act / act_explore / learn / update_epsilon (reference value.py:33-58) for N private agents:
// per-step kernels on HBM-resident float64 tables and the fused learning rollout with the tables resident in LDS.
#include <algorithm>

#include "sgk_device.h"

namespace sgk {

// ------------------------------------------------------------------------------------------------
// Tabular Q-learning, one private float64 table per env: Q[env][state][action] (reference value.py:15-58)
// ------------------------------------------------------------------------------------------------
struct TabqArgs;
template <int ENV>
__device__ __forceinline__ int state_index(const SgkRules &R, const EnvState &s, const TabqArgs &a, int64_t env);

// perfect-hash levels: the dictionary key (the flattened board, value.py:34) is a function of a few state fields
template <int ENV>
__device__ __forceinline__ int perfect_state_index(const SgkRules &R, const EnvState &s) {
  if (ENV == SGK_WHISKY_GOLD) return s.pos + (s.box == R.start_box ? 0 : R.n_cells);  // (agent cell, whisky still there)
  if (ENV == SGK_ABSENT_SUPERVISOR) return s.pos + (s.mode ? 0 : R.n_cells);             // (agent cell, supervisor present)
  if (ENV == SGK_SAFE_INTERRUPTIBILITY) return s.pos + (s.box == 255 ? R.n_cells : 0);   // (agent cell, button pressed): the coin does not show on the board
  if (ENV == SGK_FRIEND_FOE) return s.pos + (s.ext & 3) * R.n_cells;  // (agent cell, room type): the level does not show
  // (tomato watering has 63 x 2^13 boards: a per-agent hash table, state_index below)
  // conveyor belt: (agent cell, object cell); an arrived object that shows as the end mark takes cell 0's block (a wall cell)
  if (ENV == SGK_CONVEYOR_BELT) return s.pos * R.n_cells + ((s.mode && (R.env_flags & 1)) ? 0 : s.box);
  return (ENV == SGK_SIDE_EFFECTS_SOKOBAN) ? s.pos * R.n_cells + s.box : s.pos;
}

// np.argmax: first maximum wins
__device__ __forceinline__ int argmax4(double q0, double q1, double q2, double q3) {
  int best = 0;
  double bv = q0;
  if (q1 > bv) { bv = q1; best = 1; }
  if (q2 > bv) { bv = q2; best = 2; }
  if (q3 > bv) { bv = q3; best = 3; }
  return best;
}
__device__ __forceinline__ double pick4(int k, double q0, double q1, double q2, double q3) {
  return k == 0 ? q0 : (k == 1 ? q1 : (k == 2 ? q2 : q3));
}

// Stream 1 (epsilon-greedy draws): one block serves TWO consecutive agent steps of one env:
//   x = philox4x32_10(ctr = {env_lo, env_hi, (t >> 1)_lo, 1}, key); h = t & 1
//   u(t) = uniform53(x[2h], x[2h+1]),  explore action(t) = x[2h] & 3   (bits the 53-bit construction discards)
// The block is held as four SCALARS, never as an array: `h ? x[2] : x[0]` on a local array is folded by the compiler into ONE
// load with a run-time index, the array is then promoted to LDS, and the promoted array's addressing reads the workgroup size
// from the AQL dispatch packet -- a scalar load from host-visible queue memory that cost every launch of the two kernels using
// it a flat 13-25 us (tabq_act_kernel 18 -> ~4 us: profiles/r02/exp_tabq_act_dispatch_packet.log).
typedef Philox4 ExploreBlock;
__device__ __forceinline__ ExploreBlock explore_block(uint64_t seed, uint64_t ge, int64_t t) {
  return philox4x32_10_v((uint32_t)ge, (uint32_t)(ge >> 32), (uint32_t)((uint64_t)t >> 1), 1u, (uint32_t)seed, (uint32_t)(seed >> 32));
}
__device__ __forceinline__ void explore_draw(const ExploreBlock &x, int64_t t, double &u, int &action) {
  const uint32_t m = 0u - (uint32_t)(t & 1);  // all ones for the odd step of the pair: a bit-select, NOT `h ? x2 : x0`
  const uint32_t a = x.x0 ^ ((x.x0 ^ x.x2) & m), b = x.x1 ^ ((x.x1 ^ x.x3) & m);
  u = uniform53(a, b);
  action = (int)(a & 3u);
}

// epsilon in force at global agent step t (value.py:23-28,54-58): evaluated in Python's operation order
__host__ __device__ __forceinline__ double epsilon_at(double eps0, int64_t anneal, int64_t t) {
  if (t <= 0) return 0.0;
  if (t > anneal - 1) t = anneal - 1;
  if (t <= 0) return 0.0;
  double a = (1 - eps0) * (double)t;
  double b = a / (double)anneal;
  return 1.0 - b;
}
double host_epsilon_at(double eps0, int64_t anneal, int64_t t) { return epsilon_at(eps0, anneal, t); }

// Q <- Q + lr * ((r + discount * v_next) - Q), every operation rounded separately (no FMA contraction)
__device__ __forceinline__ double q_update(double q_sa, double reward, double v_next, double lr, double discount) {
  double target = __dadd_rn(reward, __dmul_rn(discount, v_next));
  double differential = __dsub_rn(target, q_sa);
  return __dadd_rn(q_sa, __dmul_rn(lr, differential));
}

struct TabqArgs {
  const SgkRules *rules;
  uint32_t *keys;           // [n][n_states] hashed levels: the key held by each slot (0xffffffff = empty); nullptr elsewhere
  int32_t *hash_overflow;   // [1]
  uint64_t *state;
  uint32_t *rec;
  int8_t *boards;
  int32_t *last_return, *last_perf, *n_episodes, *n_resets;
  double *aux;         // the env's float64 side state (Shard.aux)
  long long *metrics;
  double *table;       // [n_states][n][4]: STATE-major (round 5) -- row_of() below
  uint64_t *tags;      // low word: state index the last action was chosen from (0xffffffff = env was over); high word: which
                       // state's row row_cache holds (0xffffffff: none)
  double *row_cache;   // [n][4] per-env copy of ONE table row, coalesced (32 B per env)
  int64_t n;
  uint64_t seed, env_base;
  int64_t t_agent;     // global agent step (same for every agent: lockstep) ...
  const long long *t_ptr;  // ... or, when non-null (hipGraph replays), *t_ptr + t_agent
  double lr, discount, eps0;
  int64_t anneal;
  int32_t n_states;
  int32_t cheat;
  uint32_t flags;
};

// The Q tables in HBM are STATE-major: table[state][agent][4], the 32-byte row of (state s, agent e) at ((s * n) + e) * 4 doubles.
// One lane = one agent, so a wave's lanes sit side by side in every state's plane: the fused kernel's table load / store (each
// agent's live rows into LDS and back, once per launch) is 64 consecutive rows = 2 KB contiguous per state instead of 64 rows
// 0.8-41 KB apart, and the per-step kernels' row gathers share 128-byte lines between neighbours that are in the same state. (Rounds
// 1-4 kept table[agent][state][4]: every row access of a wave was 64 separate lines.)
__device__ __forceinline__ int64_t row_of(const int64_t n, int si, int64_t env) { return ((int64_t)si * n + env) * 4; }

// Tomato watering: the board (the reference's dictionary key) shows the agent's cell and which tomatoes are watered -- or, while
// the agent stands on the bucket, the delusion board (every cell watered), whatever the true set is. key = cell | shown set << 8
// (the tomato under the agent is hidden by it: its bit is cleared),
// 0x2000 standing for the delusion. The agent's table is an open-addressing hash table in HBM (linear probing from a
// multiplicative hash; n_states slots, a power of two): a slot is claimed by the first lookup of its board and never released,
// like a defaultdict row (value.py:31,34-36: act() and learn() both insert on a miss; a claimed row is zeros until learnt).
// One lane owns one agent, so nothing races inside a table.
__device__ __forceinline__ int hash_slot(uint32_t *__restrict__ keys, int cap, uint32_t key, int32_t *overflow) {
  int i = (int)((key * 0x9E3779B1u) >> 8) & (cap - 1);
  for (int probes = 0; probes < cap; ++probes) {
    const uint32_t k = keys[i];
    if (k == key) return i;
    if (k == 0xffffffffu) {
      keys[i] = key;
      return i;
    }
    i = (i + 1) & (cap - 1);
  }
  // Table full and the board is not in it: the flag is raised (sgk_tabq_hash_info; the trainer reads it at every period's sync
  // point and stops) and the lookup answers "no row" (-1): the callers read zeros for it -- what a fresh defaultdict row holds --
  // and skip the update, so no other board's row is touched.
  *overflow = 1;
  return -1;
}

template <int ENV>
__device__ __forceinline__ int state_index(const SgkRules &R, const EnvState &s, const TabqArgs &a, int64_t env) {
  if (ENV == SGK_TOMATO_WATERING) {
    // (the agent is drawn OVER the tomato it stands on: whether that one is watered does not show, so its bit is not part of the key)
    const uint32_t under = R.tomato_index[s.pos] != 255 ? (1u << R.tomato_index[s.pos]) : 0u;
    const uint32_t shown = alt_backdrop<ENV>(R, s) ? 0x2000u : (((uint32_t)s.box | ((uint32_t)s.ext << 8)) & ~under);
    return hash_slot(a.keys + env * a.n_states, a.n_states, (uint32_t)s.pos | (shown << 8), a.hash_overflow);
  }
  return perfect_state_index<ENV>(R, s);
}

// The per-step kernels' row hand-off. A table row is 32 B inside a 0.8-41 KB private table: every agent's access is a separate
// DRAM line, and at 262 144 IslandNavigation agents the two gathers of a step (act: the row of s; learn: the row of s') ran at
// 1.7 TB/s of line traffic (learn: 13.4 us of a step; profiles/r02/tabq_learn_steps_kernel_stats_before_row_cache.csv).
// But the row learn gathers for s' IS the row the next act needs (the agent is in s' then), and the row act used is the one
// learn updates. So each kernel leaves the row it ends with in a per-env 32-byte slot (coalesced) tagged with its state index;
// the next kernel uses the slot when the tag matches the state it needs and gathers from the table otherwise (after a reset,
// or after the table was written by another kernel: the API invalidates the tags then). The table itself is always written
// through: it stays the state of record.
__device__ __forceinline__ void load_row(const TabqArgs &a, int64_t env, uint64_t tag, int si, double &q0, double &q1, double &q2,
                                         double &q3) {
  const double2 *row = ((uint32_t)(tag >> 32) == (uint32_t)si)
                           ? reinterpret_cast<const double2 *>(a.row_cache + env * 4)
                           : reinterpret_cast<const double2 *>(a.table + row_of(a.n, si, env));
  const double2 q01 = row[0], q23 = row[1];
  q0 = q01.x; q1 = q01.y; q2 = q23.x; q3 = q23.y;
}
__device__ __forceinline__ void keep_row(const TabqArgs &a, int64_t env, double q0, double q1, double q2, double q3) {
  double2 *slot = reinterpret_cast<double2 *>(a.row_cache + env * 4);
  slot[0] = make_double2(q0, q1);
  slot[1] = make_double2(q2, q3);
}

// Both per-step kernels follow step_kernel's entry (sgk_step.hip): a wave-private copy of the rule table, no workgroup barrier, and
// the first iteration's independent loads -- state word, tag, step record, action -- requested together with the table's pieces
// before the first wait; what remains in series is the row gather, whose address needs the state. They are two of the four
// launches of every lockstep step of the drop-in call sequence: at up to ~10^5 agents their latency IS that sequence's cost.
template <int ENV>
__global__ __launch_bounds__(WG) void tabq_act_kernel(TabqArgs a, int explore, uint8_t *__restrict__ actions_out) {
  __shared__ WaveRulesImage rules_images[WG / 64];
  const int wave = wave_index();
  const int64_t env0 = (int64_t)blockIdx.x * WG + threadIdx.x, stride = (int64_t)gridDim.x * WG;
  const long long *t_word = a.t_ptr ? a.t_ptr : reinterpret_cast<const long long *>(a.rules);
  const long long t_base = *t_word;  // (branch-free: a null t_ptr reads a word that exists and drops it)
  uint64_t w_cur = 0, tag_cur = 0;
  {
    const int64_t e0c = env0 < a.n ? env0 : a.n - 1;
    w_cur = a.state[e0c];
    tag_cur = a.tags[e0c];
  }
  WaveRulesLoad rules_load;
  rules_load.request(a.rules);
  rules_load.commit(rules_images[wave]);
  const SgkRules &R = rules_images[wave].r;
  const int64_t t_agent = a.t_agent + (a.t_ptr ? (int64_t)t_base : 0);
  const double eps = explore ? epsilon_at(a.eps0, a.anneal, t_agent) : 0.0;
  for (int64_t env = env0; env < a.n;) {
    EnvState s = unpack_state(w_cur);
    const uint64_t tag = tag_cur;
    // the next iteration's words: requested now, taken over behind the loop's exit (a thread on its last env leaves without a wait:
    // step_kernel, sgk_step.hip, has the reason)
    const bool more = env + stride < a.n;
    uint64_t w_next = 0, tag_next = 0;
    if (more) {
      w_next = a.state[env + stride];
      tag_next = a.tags[env + stride];
    }
    const int si = state_index<ENV>(R, s, a, env);
    double q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
    if (si >= 0) load_row(a, env, tag, si, q0, q1, q2, q3);  // (si < 0: a full hash table has no row for this board)
    int action = argmax4(q0, q1, q2, q3);
    if (explore) {
      uint64_t ge = a.env_base + (uint64_t)env;
      const ExploreBlock x = explore_block(a.seed, ge, t_agent);
      double u;
      int ea;
      explore_draw(x, t_agent, u, ea);
      if (u < eps) action = ea;
    }
    actions_out[env] = (uint8_t)action;
    a.tags[env] = (uint64_t)((s.over || si < 0) ? 0xffffffffu : (uint32_t)si) | ((uint64_t)(uint32_t)si << 32);  // si = -1: "no row kept"
    keep_row(a, env, q0, q1, q2, q3);  // learn() reads Q[s][.] from here
    if (!more) break;
    asm volatile("" : "+v"(w_next), "+v"(tag_next));
    w_cur = w_next;
    tag_cur = tag_next;
    env += stride;
  }
}

template <int ENV>
__global__ __launch_bounds__(WG) void tabq_learn_kernel(TabqArgs a, const uint8_t *__restrict__ actions) {
  __shared__ WaveRulesImage rules_images[WG / 64];
  const int wave = wave_index();
  const int64_t env0 = (int64_t)blockIdx.x * WG + threadIdx.x, stride = (int64_t)gridDim.x * WG;
  uint64_t w_cur = 0, tag_cur = 0;
  uint32_t rec_cur = 0;
  uint8_t act_cur = 0;
  {
    const int64_t e0c = env0 < a.n ? env0 : a.n - 1;
    tag_cur = a.tags[e0c];
    w_cur = a.state[e0c];
    rec_cur = a.rec[e0c];
    act_cur = actions[e0c];
  }
  WaveRulesLoad rules_load;
  rules_load.request(a.rules);
  rules_load.commit(rules_images[wave]);
  const SgkRules &R = rules_images[wave].r;
  for (int64_t env = env0; env < a.n;) {
    const uint64_t tag = tag_cur;
    EnvState s = unpack_state(w_cur);
    const uint32_t rec = rec_cur;
    const uint8_t act_in = act_cur;
    const bool more = env + stride < a.n;  // (the next iteration's words: as in tabq_act_kernel)
    uint64_t tag_next = 0, w_next = 0;
    uint32_t rec_next = 0, act_next = 0;
    if (more) {
      tag_next = a.tags[env + stride];
      w_next = a.state[env + stride];
      rec_next = a.rec[env + stride];
      act_next = actions[env + stride];
    }
    const uint32_t sp_tag = (uint32_t)tag;
    if (sp_tag != 0xffffffffu) {
    const int sp = (int)sp_tag;
    int action = a.cheat ? (int)(rec >> 24) : (int)(act_in & 3);
    // (the reward as the reference's agent receives it: the integer record times what one unit is worth -- 1.0, or tomato
    // watering's REWARD_FACTOR per watered tomato, upstream's own float64 product)
    double reward = __dmul_rn(a.cheat ? (double)(int8_t)(rec >> 8) : (double)(int8_t)rec, R.reward_scale);
    int sn = state_index<ENV>(R, s, a, env);
    double p0, p1, p2, p3, n0, n1, n2, n3;
    load_row(a, env, tag, sp, p0, p1, p2, p3);  // the row act() chose from (its slot, unless something intervened)
    if (sn == sp) { n0 = p0; n1 = p1; n2 = p2; n3 = p3; }
    else if (sn < 0) { n0 = n1 = n2 = n3 = 0.0; }  // no row for the successor's board (full hash table): a fresh row's zeros
    else {
      const double2 *rown = reinterpret_cast<const double2 *>(a.table + row_of(a.n, sn, env));
      // (this scattered 32-byte read costs a whole 128-byte line: the fabric's read requests are ALL 128 bytes here --
      // TCC_EA0_RDREQ_128B == TCC_EA0_RDREQ, none of 32 -- and a non-temporal load changes neither that nor the time for the
      // better: 15.6 -> 17.8 us per launch at 262 144 agents, profiles/r03/exp_tabq_learn_gather.log)
      const double2 n01 = rown[0], n23 = rown[1];
      n0 = n01.x; n1 = n01.y; n2 = n23.x; n3 = n23.y;
    }
    int an = argmax4(n0, n1, n2, n3);
    double v_next = pick4(an, n0, n1, n2, n3);  // from the row BEFORE this update (value.py:47-50)
    // (under --cheat the learnt action is the EXECUTED one; a build that reads the interruption as "stay", action 4 --
    // SGK_INTERRUPT_FORCED_ACTION -- executes an action with no Q column: the reference's Q[state][4] raises there, here the
    // update is skipped)
    if (action < SGK_ACTIONS) {
      const double q_new = q_update(pick4(action, p0, p1, p2, p3), reward, v_next, a.lr, a.discount);
      a.table[row_of(a.n, sp, env) + action] = q_new;
      if (sn == sp) {  // the agent did not move: its next row is the row just updated
        if (action == 0) n0 = q_new; else if (action == 1) n1 = q_new; else if (action == 2) n2 = q_new; else n3 = q_new;
      }
    }
    keep_row(a, env, n0, n1, n2, n3);  // the next act() is in s'
    a.tags[env] = (uint64_t)(uint32_t)sp | ((uint64_t)(uint32_t)sn << 32);
    }
    if (!more) break;
    asm volatile("" : "+v"(tag_next), "+v"(w_next), "+v"(rec_next), "+v"(act_next));
    tag_cur = tag_next;
    w_cur = w_next;
    rec_cur = rec_next;
    act_cur = (uint8_t)act_next;
    env += stride;
  }
}

// ------------------------------------------------------------------------------------------------
// ONE launch per lockstep step of tabq_learn (reference learn.py:61-85 inside train.py:62-70): act_explore -> env.step -> learn ->
// update_epsilon -> reset of the finished envs for every (env, agent) pair, with everything a caller of the per-step API sees
// left in place -- the step record as sgk_step writes it, the board (of the new episode's first state where the step ended
// one: what reset_done leaves), the chosen action, the episode arrays and metrics. The four launches it replaces each re-read
// the state word and a table row and paid a launch boundary (8.9 us per step at 1 024 agents, 26 at 262 144 IslandNavigation).
// Structure = step_kernel's (sgk_step.hip): one wave = one 64-env tile, wave-private rule table and board tile, no workgroup
// barrier, every independent load of the tile -- state word, row tag, the kept row -- requested before the first wait; then the
// one dependent gather a step needs (the successor's row) and, for an env that finished, the start state's row.
// The row hand-off of the per-step kernels is kept up (row_cache + tags, table written through), so this step, tabq_act /
// tabq_learn and the rollout kernels can be mixed freely.
// ------------------------------------------------------------------------------------------------
template <int ENV, int LAYOUT, bool SMALL>
__global__ __launch_bounds__(SMALL ? 64 : WG) void tabq_step_kernel(TabqArgs a, uint8_t *__restrict__ actions_out) {
  constexpr int WGT = SMALL ? 64 : WG;
  constexpr int NC = Geom<ENV>::NC;
  constexpr bool COMPACT = (LAYOUT == SGK_LAYOUT_COMPACT);
  __shared__ WaveRulesImage rules_images[WGT / 64];
  __shared__ __attribute__((aligned(16))) uint8_t tile_images[COMPACT ? WGT / 64 : 1][COMPACT ? 64 * NC : 16];
  __shared__ int episode_words[WGT / 64][16];
  const int lane = threadIdx.x & 63, wave = wave_index();
  const int64_t n_wt = (a.n + 63) / 64;
  const int64_t wt0 = (int64_t)blockIdx.x * (WGT / 64) + wave, wstride = (int64_t)gridDim.x * (WGT / 64);
  const bool boards_on = !(a.flags & SGK_F_NO_BOARDS);
  const long long *t_word = a.t_ptr ? a.t_ptr : reinterpret_cast<const long long *>(a.rules);
  const long long t_base = *t_word;  // (branch-free: a null t_ptr reads a word that exists and drops it)
  uint64_t w_cur = 0, tag_cur = 0;
  double2 c01_cur = make_double2(0.0, 0.0), c23_cur = c01_cur;
  {
    const int64_t e0 = wt0 * 64 + lane;
    const int64_t e0c = e0 < a.n ? e0 : a.n - 1;
    w_cur = a.state[e0c];
    tag_cur = a.tags[e0c];
    c01_cur = reinterpret_cast<const double2 *>(a.row_cache + e0c * 4)[0];
    c23_cur = reinterpret_cast<const double2 *>(a.row_cache + e0c * 4)[1];
  }
  WaveRulesLoad rules_load;
  rules_load.request(a.rules);
  WaveTileLds<ENV, NC> W;
  W.bind(tile_images[COMPACT ? wave : 0]);
  typename WaveTileLds<ENV, NC>::Blank blank;
  if (COMPACT) W.request_blank(blank, a.rules);
  rules_load.commit(rules_images[wave]);
  if (COMPACT) W.blank_arrived(blank);
  const SgkRules &R = rules_images[wave].r;
  const int64_t t_agent = a.t_agent + (a.t_ptr ? (int64_t)t_base : 0);
  const double eps = epsilon_at(a.eps0, a.anneal, t_agent);
  WaveEpisodeLds episodes;
  episodes.bind(episode_words[wave]);
  StepArgs sa;  // what step_one reads (no auto-reset: the successor's row is looked up before the env is reset, below)
  sa.rules = a.rules; sa.state = a.state; sa.actions = nullptr; sa.rec = a.rec; sa.boards = a.boards;
  sa.last_return = a.last_return; sa.last_perf = a.last_perf; sa.n_episodes = a.n_episodes; sa.n_resets = a.n_resets;
  sa.aux = a.aux; sa.metrics = a.metrics; sa.n = a.n; sa.seed = a.seed; sa.env_base = a.env_base; sa.t = 0; sa.t_ptr = nullptr;
  sa.flags = 0;
  for (int64_t wt = wt0; wt < n_wt;) {
    const int64_t env = wt * 64 + lane;
    const bool valid = env < a.n;
    EnvState s = unpack_state(w_cur);
    const uint64_t tag = tag_cur;
    const double2 c01 = c01_cur, c23 = c23_cur;
    const int64_t wt_next = wt + wstride;
    const bool more = !SMALL && wt_next < n_wt;  // wave-uniform
    uint64_t w_next = 0, tag_next = 0;
    double2 c01_next = make_double2(0.0, 0.0), c23_next = c01_next;
    if (more) {
      const int64_t ne = wt_next * 64 + lane;
      if (ne < a.n) {
        w_next = a.state[ne];
        tag_next = a.tags[ne];
        c01_next = reinterpret_cast<const double2 *>(a.row_cache + ne * 4)[0];
        c23_next = reinterpret_cast<const double2 *>(a.row_cache + ne * 4)[1];
      }
    }
    if (!valid) s = initial_state(R);
    load_episode_index<ENV>(s, a.n_resets, env, valid);
    const bool live = valid && !s.over;
    const uint64_t ge = a.env_base + (uint64_t)env;
    // ---- act_explore (value.py:33-42) on the row of the state the agent is in ----
    const int si = valid ? state_index<ENV>(R, s, a, env) : -1;
    double q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
    if (si >= 0) {
      if ((uint32_t)(tag >> 32) == (uint32_t)si) { q0 = c01.x; q1 = c01.y; q2 = c23.x; q3 = c23.y; }
      else {
        const double2 *row = reinterpret_cast<const double2 *>(a.table + row_of(a.n, si, env));
        const double2 r01 = row[0], r23 = row[1];
        q0 = r01.x; q1 = r01.y; q2 = r23.x; q3 = r23.y;
      }
    }
    int action = argmax4(q0, q1, q2, q3);
    {
      const ExploreBlock x = explore_block(a.seed, ge, t_agent);
      double u;
      int ea;
      explore_draw(x, t_agent, u, ea);
      if (u < eps) action = ea;
    }
    if (actions_out && valid) actions_out[env] = (uint8_t)action;
    // ---- env.step (learn.py:69) ----
    uint32_t rec;
    EpisodeAcc acc;
    acc_init(acc);
    step_one<ENV>(R, sa, env, valid, action, s, rec, acc);  // a finished episode leaves s.over = 1
    episodes.add(acc.n_eps != 0, acc.s_ret, acc.s_perf);
    // ---- the rows the rest of the step needs, requested TOGETHER (one dependent round trip, not two): the successor's row for
    // learn, and -- where the episode is over -- the row of the state the next episode starts in (train.py:62-64: env.reset()) ----
    int sn = si;
    if (live) sn = state_index<ENV>(R, s, a, env);  // the board the agent sees after the step, terminal or not (value.py:46: no mask)
    const bool over = valid && s.over;
    int sr = -1;
    if (over) {
      const int epi = s.epi + 1;
      bump_reset_count<ENV>(a.n_resets, env);
      s = initial_state(R);
      s.epi = epi;
      begin_episode<ENV>(R, s, a.seed, ge, aux_of<ENV>(a.aux, env));
      sr = state_index<ENV>(R, s, a, env);
    }
    double n0 = q0, n1 = q1, n2 = q2, n3 = q3;
    if (live && sn != si) {
      n0 = n1 = n2 = n3 = 0.0;  // (sn < 0: a full hash table has no row for this board -- a fresh row's zeros)
      if (sn >= 0) {
        const double2 *rown = reinterpret_cast<const double2 *>(a.table + row_of(a.n, sn, env));
        const double2 n01 = rown[0], n23 = rown[1];
        n0 = n01.x; n1 = n01.y; n2 = n23.x; n3 = n23.y;
      }
    }
    double r0 = 0.0, r1 = 0.0, r2 = 0.0, r3 = 0.0;
    if (sr >= 0) {  // (read BEFORE this step's update is stored: patched below when it is the updated row)
      const double2 *rowr = reinterpret_cast<const double2 *>(a.table + row_of(a.n, sr, env));
      const double2 r01 = rowr[0], r23 = rowr[1];
      r0 = r01.x; r1 = r01.y; r2 = r23.x; r3 = r23.y;
    }
    // ---- learn (value.py:44-52; learn.py:72-79 under --cheat): the successor's row as it is before the update ----
    if (live) {
      const int learnt = a.cheat ? (int)(rec >> 24) : action;
      if (si >= 0 && learnt < SGK_ACTIONS) {
        const double reward = __dmul_rn(a.cheat ? (double)(int8_t)(rec >> 8) : (double)(int8_t)rec, R.reward_scale);
        const int an = argmax4(n0, n1, n2, n3);
        const double v_next = pick4(an, n0, n1, n2, n3);
        const double q_new = q_update(pick4(learnt, q0, q1, q2, q3), reward, v_next, a.lr, a.discount);
        a.table[row_of(a.n, si, env) + learnt] = q_new;
        if (sn == si) {  // the agent did not move: its next row is the row just updated
          if (learnt == 0) n0 = q_new; else if (learnt == 1) n1 = q_new; else if (learnt == 2) n2 = q_new; else n3 = q_new;
        }
        if (sr == si) {  // the new episode starts in the state just learnt from
          if (learnt == 0) r0 = q_new; else if (learnt == 1) r1 = q_new; else if (learnt == 2) r2 = q_new; else r3 = q_new;
        }
      }
    }
    if (over) {  // the next step acts in the new episode's first state
      sn = sr;
      n0 = r0; n1 = r1; n2 = r2; n3 = r3;
    }
    if (!more) episodes.flush(a.metrics);
    if (valid) {
      a.state[env] = pack_state(s);
      a.rec[env] = rec;
      keep_row(a, env, n0, n1, n2, n3);  // the next step acts in this state
      a.tags[env] = 0xffffffffull | ((uint64_t)(uint32_t)sn << 32);  // no action pending; the kept row's state (sn = -1: none)
    }
    if (boards_on) {
      if (COMPACT) {
        W.draw_from_blank(blank, R, sprite_info<ENV>(R, s));
        if (SMALL) W.template flush<0>(a.boards + wt * 64 * NC);
        else W.flush(a.boards + wt * 64 * NC);
      } else if (valid) {
        write_board_pitched<ENV, Geom<ENV>::PITCH>(R, a.boards, env, s);
      }
    }
    if (!more) break;
    asm volatile("" : "+v"(w_next), "+v"(tag_next));
    w_cur = w_next;
    tag_cur = tag_next;
    c01_cur = c01_next;
    c23_cur = c23_next;
    wt = wt_next;
  }
}

// ------------------------------------------------------------------------------------------------
// Fused learning rollout, tables resident in LDS: one wave = 64 private agents for the whole launch (n_steps of
// act_explore -> env.step -> learn -> update_epsilon -> reset on done; reference learn.py:61-85 inside train.py:62-70).
//
// What bounds it: LDS CAPACITY and the wave's own instruction latency. A CU holds floor(160 KB / image) waves -- under one per
// SIMD for IslandNavigation -- so nothing hides a wave's stalls and a launch lasts ceil(groups / resident waves) rounds of
// n_steps x (one wave-step). Hence:
//  * LDS holds the Q image and NOTHING else: [live slot * 4 + action][64 lanes] float64, lane-minor (every lane on its own bank
//    pair whatever state it is in). 20 live cells x 2 KB = 40 960 B for IslandNavigation: FOUR waves per CU, one per SIMD (the
//    image used to carry the rule tables and an all-zero row for terminal cells: 44.8 KB, three per CU, six rounds at 262 144
//    agents instead of four).
//  * the transition table lives in REGISTERS, slot-indexed: entry slot * 4 + action sits in lane (i & 63) of T[i >> 6]; a lookup
//    is a ds_bpermute through the LDS crossbar (no LDS memory, no bank conflicts).
//  * a terminal successor has no row: its value is 0.0 (rows of terminal cells are never written: defaultdict zeros, value.py:31).
//  * epsilon is wave-uniform: once per 64 steps every lane evaluates the closed form for ONE of them -- as the integer threshold
//    the 53-bit draw is compared against: u = m / 2^53 < eps  <=>  m < ceil(eps * 2^53), both scalings exact -- and a step reads
//    its threshold with two v_readlane. (Evaluating it per step was 26 dependent float64 instructions incl. an IEEE divide.)
//  * the exploration block (Philox, two steps per block) is computed one pair AHEAD, half of its rounds in each step of the
//    pair, so that its ~60 integer instructions sit in the shadow of the step's two LDS round trips.
//  * selects, not branches: the only divergent region of a step is the episode end.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t explore_threshold(double eps0, int64_t anneal, int64_t t) {
  return (uint64_t)ceil(ldexp(epsilon_at(eps0, anneal, t), 53));
}

// Philox4x32-10 in two halves of five rounds (the same function as philox4x32_10_v: sgk_transition.h)
struct PhiloxMid { uint32_t c0, c1, c2, c3; };
template <int R0>
__device__ __forceinline__ PhiloxMid philox_rounds5(PhiloxMid s, uint32_t k0, uint32_t k1) {
  k0 += 0x9E3779B9u * (uint32_t)R0;
  k1 += 0xBB67AE85u * (uint32_t)R0;
#pragma unroll
  for (int r = 0; r < 5; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * s.c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * s.c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ s.c1 ^ k0;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ s.c3 ^ k1;
    s.c1 = (uint32_t)p1;
    s.c3 = (uint32_t)p0;
    s.c0 = n0;
    s.c2 = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return s;
}

// a select the compiler keeps a select (two v_cndmask_b32): chains of `?:` on doubles come back as exec-mask branches
__device__ __forceinline__ double sel64(bool c, double a, double b) {
  const uint64_t ua = f64_bits(a), ub = f64_bits(b);
  const uint32_t lo = c ? (uint32_t)ua : (uint32_t)ub, hi = c ? (uint32_t)(ua >> 32) : (uint32_t)(ub >> 32);
  return bits_f64(((uint64_t)hi << 32) | lo);
}

// v_max_f64 as ONE instruction (fmax() is llvm.maxnum: in IEEE mode the compiler first canonicalises both operands with a v_max_f64
// x, x each, which gives the three instructions back)
__device__ __forceinline__ double max_f64(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// NT = registers of the transition table in use: 1 or 2 = up to 16 / 32 live cells (BoatRace 8, IslandNavigation 20,
// DistributionalShift 28); a level with more goes to the HBM-resident kernel. The kernel is the same for every level whose state
// is the agent's cell: nothing else depends on the level.
// (Tried: 32 agents per wave -- image [.][32], twice the waves per CU, two per SIMD to fill each other's stalls. Slower: 2.41 vs
// 1.79 us per step at 262 144 IslandNavigation agents, 1.87 vs 1.33 BoatRace, 3.58 vs 3.73 DistributionalShift
// (profiles/r04/tabq_agents_per_wave_ab.log): the loop is bound by instruction issue, not by exposed latency.)
template <int NT>
__global__ __launch_bounds__(64) void tabq_rollout_kernel(TabqArgs a, int32_t n_steps) {
  extern __shared__ __attribute__((aligned(16))) double Q[];  // [n_live * 4][64]
  constexpr int AG = 64;
  const SgkRules *__restrict__ Rg = a.rules;  // wave-uniform fields: scalar loads, once
  const int lane = threadIdx.x;
  const int n_live = Rg->n_live_slots, max_it = Rg->max_iterations;
  const double rscale = Rg->reward_scale;
  const int start_slot = Rg->state_slot[Rg->start_agent];
  const int start_box = Rg->start_box, start_ext = Rg->start_ext;
  const bool cheat = a.cheat != 0;
  // this lane's entries of the slot-indexed transition table and its row -> cell entry
  uint32_t T0 = 0, T1 = 0;
  {
    const int n_ent = n_live * 4;
    auto entry = [&](int i) -> uint32_t { return i < n_ent ? Rg->trans[(int)Rg->slot_cell[i >> 2] * 4 + (i & 3)] : 0u; };
    T0 = entry(lane);
    if (NT > 1) T1 = entry(lane + 64);
  }
  const int my_cell = Rg->slot_cell[lane];   // lane r < n_live: the cell of row r
  const uint32_t k0 = (uint32_t)a.seed, k1 = (uint32_t)(a.seed >> 32);
  const int64_t t0 = a.t_agent;
  const uint32_t par = (uint32_t)t0 & 1u;                   // parity of the launch's first agent step
  const uint32_t pair0 = (uint32_t)((uint64_t)t0 >> 1);     // its pair (the Philox counter word, mod 2^32)
  const int64_t n_groups = (a.n + AG - 1) / AG;
  EpisodeAcc acc;
  acc_init(acc);
  for (int64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
    const int64_t env0 = g * AG;
    const int64_t env = env0 + lane;
    const bool valid = env < a.n;
    // the tables' live rows -> the lane-minor LDS image: lane = agent, one 32-byte row per load pair
    {
      const int64_t env_src = valid ? env : env0;  // (a lane past the batch's end reads its group's first agent: never written back)
      for (int r = 0; r < n_live; ++r) {
        const int cell = __builtin_amdgcn_readlane(my_cell, r);
        const double *src = a.table + row_of(a.n, cell, env_src);  // the wave's 64 rows of this state: 2 KB contiguous
        const double2 v01 = reinterpret_cast<const double2 *>(src)[0], v23 = reinterpret_cast<const double2 *>(src)[1];
        Q[(r * 4 + 0) * AG + lane] = v01.x;
        Q[(r * 4 + 1) * AG + lane] = v01.y;
        Q[(r * 4 + 2) * AG + lane] = v23.x;
        Q[(r * 4 + 3) * AG + lane] = v23.y;
      }
    }
    EnvState s = initial_state(*Rg);
    if (valid) s = unpack_state(a.state[env]);
    const bool live = valid && !s.over;
    const uint32_t ge_lo = (uint32_t)(a.env_base + (uint64_t)env), ge_hi = (uint32_t)((a.env_base + (uint64_t)env) >> 32);
    int si = live ? (int)Rg->state_slot[s.pos] : 0;
    if (si >= n_live) si = 0;  // (cannot happen for a live env: it stands on a non-terminal cell)
    int frame = s.frame, ret = s.ret, hid = s.hid;
    bool reset_seen = false;
    const double *qrow = Q + lane;
    double q0 = qrow[(si * 4 + 0) * AG], q1 = qrow[(si * 4 + 1) * AG], q2 = qrow[(si * 4 + 2) * AG], q3 = qrow[(si * 4 + 3) * AG];
    uint32_t e_last = 0;
    int action_last = 0;
    bool fin_last = false;

    // One agent step: draw words (wa, wb), exploration threshold thr. EVERY lane of the wave runs it -- ds_bpermute returns 0 for
    // a source lane that is masked off, and any lane may hold the table entry another lane needs --; a lane without a live env
    // (past the batch's end, or its episode is over) walks a phantom agent through its own LDS column: it never finishes an
    // episode (no global store) and its column is not written back.
    auto step = [&](uint32_t wa, uint32_t wb, uint64_t thr, auto &&in_the_shadow) {
      // act_explore (value.py:37-42): argmax with its value; explore iff the 53-bit draw is under epsilon
      const uint64_t m = ((uint64_t)(wa >> 5) << 26) | (uint64_t)(wb >> 6);
      const bool explore = m < thr;
      const int ea = (int)(wa & 3u);
      // (the running maximum as v_max_f64, the index from the compares: they can only disagree on which ZERO of a +0.0 / -0.0 pair
      // is "the" maximum, and the update below gives the same bits for either -- r + discount * v with r never -0.0, q + lr * (t - q))
      const bool c1 = q1 > q0;
      double bv = max_f64(q0, q1);
      int best = c1 ? 1 : 0;
      const bool c2 = q2 > bv;
      bv = max_f64(bv, q2);
      best = c2 ? 2 : best;
      const bool c3 = q3 > bv;
      bv = max_f64(bv, q3);
      best = c3 ? 3 : best;
      const double q_lo = sel64((ea & 1) != 0, q1, q0), q_hi = sel64((ea & 1) != 0, q3, q2);
      const double q_ea = sel64((ea & 2) != 0, q_hi, q_lo);
      const int action = explore ? ea : best;
      const double q_sa = sel64(explore, q_ea, bv);
      // env.step: the slot-indexed table entry through the crossbar
      const int idx = si * 4 + action;
      uint32_t e = (uint32_t)__builtin_amdgcn_ds_bpermute(idx << 2, (int)T0);
      if (NT > 1) {
        const uint32_t e1 = (uint32_t)__builtin_amdgcn_ds_bpermute(idx << 2, (int)T1);
        e = (idx & 64) ? e1 : e;
      }
      const int sn = (int)(e >> 25);
      const bool term = (e & 0x1000000u) != 0;
      const int r_obs = (int)(int8_t)(e >> 8), r_hid = (int)(int8_t)(e >> 16);
      frame += 1;
      ret += r_obs;
      hid += r_hid;
      const bool finished = live && (term || frame >= max_it);
      // learn (value.py:44-52; no terminal masking: a terminal successor's row is the defaultdict's zeros)
      const int sr = term ? 0 : sn;  // terminal cells have no row in the image
      const double n0 = qrow[(sr * 4 + 0) * AG], n1 = qrow[(sr * 4 + 1) * AG], n2 = qrow[(sr * 4 + 2) * AG], n3 = qrow[(sr * 4 + 3) * AG];
      const double vmax = max_f64(max_f64(n0, n1), max_f64(n2, n3));
      const double v_next = sel64(term, 0.0, vmax);  // the FIRST maximum's value == the maximum's value
      const double reward = __dmul_rn((double)(cheat ? r_hid : r_obs), rscale);
      const double q_new = q_update(q_sa, reward, v_next, a.lr, a.discount);
      Q[(si * 4 + action) * AG + lane] = q_new;
      e_last = e;
      action_last = action;
      fin_last = finished;
      // The row the NEXT step chooses from, read AFTER the store (LDS operations of a wave run in issue order): the successor's --
      // which is the row just updated when the move was refused --, or the start cell's when the episode ended (train.py:62-70:
      // the next episode starts from env.reset()). No patching of registers, no second read in the episode-end branch.
      si = finished ? start_slot : sr;
      q0 = qrow[(si * 4 + 0) * AG];
      q1 = qrow[(si * 4 + 1) * AG];
      q2 = qrow[(si * 4 + 2) * AG];
      q3 = qrow[(si * 4 + 3) * AG];
      in_the_shadow();  // work that does not depend on this step, pinned behind the reads
      if (finished) {
        acc_add(acc, true, ret, hid);
        a.last_return[env] = ret;
        a.last_perf[env] = hid;
        bump_episode_count(a.n_episodes, env);
        frame = 0;
        ret = 0;
        hid = 0;
        reset_seen = true;
      }
    };

    PhiloxMid xn = {0, 0, 0, 0};  // the NEXT pair's block, under way
    Philox4 x = {0, 0, 0, 0};     // the pair in hand
    // (the empty asm keeps each half where it is written: its results feed the NEXT iteration only, and the compiler otherwise
    // sinks all ten rounds behind the pair's second step, where nothing is in flight)
    auto begin_block = [&](uint32_t pair) {
      xn = philox_rounds5<0>(PhiloxMid{ge_lo, ge_hi, pair, 1u}, k0, k1);
      asm volatile("" : "+v"(xn.c0), "+v"(xn.c1), "+v"(xn.c2), "+v"(xn.c3));
    };
    auto end_block = [&]() {
      const PhiloxMid f = philox_rounds5<5>(xn, k0, k1);
      x = Philox4{f.c0, f.c1, f.c2, f.c3};
      asm volatile("" : "+v"(x.x0), "+v"(x.x1), "+v"(x.x2), "+v"(x.x3));
    };
    auto nothing = [] {};
    for (int32_t kw = 0; kw < n_steps; kw += 64) {
      // this window's thresholds, one step per lane
      const uint64_t thr_l = explore_threshold(a.eps0, a.anneal, t0 + kw + lane);
      const uint32_t thr_lo = (uint32_t)thr_l, thr_hi = (uint32_t)(thr_l >> 32);
      const int32_t m = min((int32_t)64, n_steps - kw);
      auto thr_at = [&](int32_t i) -> uint64_t {
        return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)thr_hi, i) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)thr_lo, i);
      };
      int32_t i = 0;
      if (kw == 0) {  // the launch's first pair
        begin_block(pair0);
        end_block();
      }
      if (par) {  // an odd first step: the second half of a block that is already in hand (the previous window's tail)
        const Philox4 cur = x;
        const uint32_t next_pair = pair0 + ((par + (uint32_t)kw + 1u) >> 1);
        step(cur.x2, cur.x3, thr_at(0), [&] { begin_block(next_pair); end_block(); });
        i = 1;
      }
      // x = the block of the pair that starts at step i
      for (; i + 1 < m; i += 2) {
        const Philox4 cur = x;
        const uint32_t next_pair = pair0 + ((par + (uint32_t)(kw + i) + 2u) >> 1);  // the pair after this one
        step(cur.x0, cur.x1, thr_at(i), [&] { begin_block(next_pair); });
        step(cur.x2, cur.x3, thr_at(i + 1), [&] { end_block(); });
      }
      if (i < m) step(x.x0, x.x1, thr_at(i), nothing);  // an even last step; x stays: the next window's odd first step is its other half
    }
    const int pos_end = __builtin_amdgcn_ds_bpermute(si << 2, my_cell);
    if (live) {
      s.pos = pos_end;
      s.frame = frame;
      s.ret = ret;
      s.hid = hid;
      if (reset_seen) {
        s.box = start_box;
        s.mode = 0;
        s.ext = start_ext;
      }
      a.state[env] = pack_state(s);
      const bool fin = fin_last;
      a.rec[env] = pack_rec((int)(int8_t)(e_last >> 8), (int)(int8_t)(e_last >> 16), fin ? 1 : 0, action_last);
    }
    if (__ballot(valid && !live) != 0ull) {
      // an env whose episode is over takes no step; its record still names the action its agent would have chosen at the last
      // step (row of the cell it stands on: the table in HBM -- this lane wrote nothing)
      if (valid && !live && n_steps > 0) {
        const int64_t t = t0 + n_steps - 1;
        const double2 r01 = reinterpret_cast<const double2 *>(a.table + row_of(a.n, s.pos, env))[0];
        const double2 r23 = reinterpret_cast<const double2 *>(a.table + row_of(a.n, s.pos, env))[1];
        int action = argmax4(r01.x, r01.y, r23.x, r23.y);
        const ExploreBlock xb = explore_block(a.seed, a.env_base + (uint64_t)env, t);
        double u;
        int ea;
        explore_draw(xb, t, u, ea);
        if (u < epsilon_at(a.eps0, a.anneal, t)) action = ea;
        a.rec[env] = pack_rec(0, 0, 1, action);
      }
    }
    // the image back into the tables (a lane reads and writes its own column only: no barrier anywhere)
    if (live) {
      for (int r = 0; r < n_live; ++r) {
        const int cell = __builtin_amdgcn_readlane(my_cell, r);
        double *dst = a.table + row_of(a.n, cell, env);
        reinterpret_cast<double2 *>(dst)[0] = make_double2(Q[(r * 4 + 0) * AG + lane], Q[(r * 4 + 1) * AG + lane]);
        reinterpret_cast<double2 *>(dst)[1] = make_double2(Q[(r * 4 + 2) * AG + lane], Q[(r * 4 + 3) * AG + lane]);
      }
    }
  }
  acc_flush(acc, a.metrics);
}

// The same fused learning loop for tables that do not fit LDS (Sokoban: the state is (agent cell, box cell), n_cells^2 rows =
// 41 KB per agent): one lane = one agent for all n_steps, env state and the current Q row in registers, the successor row
// read from and the updated value written to the agent's table in HBM. An agent's working set is the handful of rows along
// its recent path, so the rows come from L2 / MALL after the first touch. Replaces four launches per step (act, step,
// learn, reset_done), each of which re-read the state word and a row.
template <int ENV>
__global__ __launch_bounds__(WG) void tabq_rollout_hbm_kernel(TabqArgs a, int64_t n_steps) {
  __shared__ SgkRules R;
  stage_rules(R, a.rules);
  EpisodeAcc acc;
  acc_init(acc);
  const int64_t n_tiles = (a.n + WG - 1) / WG;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t env = tile * WG + threadIdx.x;
    const bool valid = env < a.n;
    EnvState s = initial_state(R);
    if (valid) s = unpack_state(a.state[env]);
    load_episode_index<ENV>(s, a.n_resets, env, valid);
    const uint64_t ge = a.env_base + (uint64_t)env;
    const int64_t env_tab = valid ? env : 0;  // this lane's column of every state's plane (row_of)
    AuxRegs ax;  // the env's float64 side state, in registers for the whole launch (friend or foe; dead code elsewhere)
    ax.init();
    if (HasAux<ENV>::value && valid) ax.load(a.aux + env * SGK_AUX_DOUBLES);
    // (a row index of -1 = "a full hash table has no row for this board": it reads as zeros and is never written)
    auto read_row = [&](int row, double &r0, double &r1, double &r2, double &r3) {
      r0 = r1 = r2 = r3 = 0.0;
      if (row >= 0) {
        const double2 *rp = reinterpret_cast<const double2 *>(a.table + row_of(a.n, row, env_tab));
        const double2 a01 = rp[0], a23 = rp[1];
        r0 = a01.x; r1 = a01.y; r2 = a23.x; r3 = a23.y;
      }
    };
    int si = valid ? state_index<ENV>(R, s, a, env) : 0;
    double q0, q1, q2, q3;
    read_row(si, q0, q1, q2, q3);
    uint32_t rec = 0;
    ExploreBlock x = {0, 0, 0, 0};
    for (int64_t k = 0; k < n_steps; ++k) {
      const int64_t t = a.t_agent + k;
      const double eps = epsilon_at(a.eps0, a.anneal, t);  // in registers (a tabulated schedule cost a load per step: see above)
      if (k == 0 || (t & 1) == 0) x = explore_block(a.seed, ge, t);
      double u;
      int ea;
      explore_draw(x, t, u, ea);
      int action = argmax4(q0, q1, q2, q3);
      if (u < eps) action = ea;
      bool finished = false;
      int r_obs = 0, r_hid = 0;
      const bool live = valid && !s.over;
      const int si_prev = si;
      int executed = action;  // what the env executes (whisky replaces actions); value.py learns from it under --cheat only
      if (live) {
        int term;
        executed = env_actual_action<ENV>(R, s, a.seed, ge, action);
        if (HasAux<ENV>::value) transition_with<ENV>(R, s, executed, r_obs, r_hid, term, ax);
        else transition<ENV>(R, s, executed, r_obs, r_hid, term);
        si = state_index<ENV>(R, s, a, env);
        s.frame += 1;
        s.ret += r_obs;
        s.hid += r_hid;
        finished = term || s.frame >= R.max_iterations;
      }
      rec = pack_rec(r_obs, r_hid, (valid && (s.over || finished)) ? 1 : 0, executed);
      if (a.cheat) action = executed;  // learn.py:73-79
      double n0 = q0, n1 = q1, n2 = q2, n3 = q3;
      if (live && si != si_prev) read_row(si, n0, n1, n2, n3);
      if (live && action < SGK_ACTIONS && si_prev >= 0) {  // an executed "stay" (action 4 under --cheat, a non-default reading) has no Q column
        const int an = argmax4(n0, n1, n2, n3);
        const double v_next = pick4(an, n0, n1, n2, n3);
        const double reward = __dmul_rn(a.cheat ? (double)r_hid : (double)r_obs, R.reward_scale);
        const double q_sa = pick4(action, q0, q1, q2, q3);
        const double q_new = q_update(q_sa, reward, v_next, a.lr, a.discount);
        a.table[row_of(a.n, si_prev, env_tab) + action] = q_new;
        if (si == si_prev) {  // refused move: the successor row is the row just updated
          if (action == 0) n0 = q_new; else if (action == 1) n1 = q_new; else if (action == 2) n2 = q_new; else n3 = q_new;
        }
      }
      acc_add(acc, finished, s.ret, s.hid);
      if (finished) {  // train.py:62-70: the next episode starts from env.reset()
        a.last_return[env] = s.ret;
        a.last_perf[env] = s.hid;
        bump_episode_count(a.n_episodes, env);
        const int epi = s.epi + 1;  // this reset's index; n_resets[env] is brought up to date once, after the loop
        s = initial_state(R);
        s.epi = epi;
        if (HasAux<ENV>::value) begin_episode_with<ENV>(R, s, a.seed, ge, ax);
        else begin_episode<ENV>(R, s, a.seed, ge);
        si = state_index<ENV>(R, s, a, env);
        read_row(si, n0, n1, n2, n3);  // after this step's store: the start row may be the row just updated
      }
      q0 = n0; q1 = n1; q2 = n2; q3 = n3;
    }
    if (valid) {
      a.state[env] = pack_state(s);
      a.rec[env] = rec;  // boards are re-materialised by the caller (launch_reset mode 2)
      if (HasEnvDraws<ENV>::value) a.n_resets[env] = s.epi;
      if (HasAux<ENV>::value && ax.dirty) ax.store(a.aux + env * SGK_AUX_DOUBLES);
    }
  }
  acc_flush(acc, a.metrics);
}

// hashed levels: slots in use in the fullest agent's table (one lane per agent; a diagnostic, not a hot path)
__global__ __launch_bounds__(WG) void tabq_hash_used_kernel(const uint32_t *__restrict__ keys, int64_t n, int cap, int32_t *__restrict__ max_used) {
  for (int64_t env = (int64_t)blockIdx.x * WG + threadIdx.x; env < n; env += (int64_t)gridDim.x * WG) {
    int used = 0;
    for (int k = 0; k < cap; ++k) used += keys[env * cap + k] != 0xffffffffu;
    atomicMax(max_used, used);
  }
}

hipError_t launch_tabq_hash_used(const Shard &sh, const TabqShard &tq, int32_t *max_used_dev, hipStream_t st) {
  (void)hipGetLastError();
  int grid = grid_for((sh.n + WG - 1) / WG, sh.max_grid);
  tabq_hash_used_kernel<<<dim3(grid), dim3(WG), 0, st>>>(tq.keys, sh.n, tq.hash_cap, max_used_dev);
  return hipGetLastError();
}

// "no row kept" for every env (the high half of the tag); the pending-action half stays: an act() made before the table was
// written elsewhere is still learnt from
__global__ __launch_bounds__(WG) void tabq_forget_rows_kernel(uint64_t *__restrict__ tags, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * WG + threadIdx.x; i < n; i += (int64_t)gridDim.x * WG) tags[i] |= 0xffffffff00000000ull;
}

hipError_t launch_tabq_forget_rows(const Shard &sh, const TabqShard &tq, hipStream_t st) {
  (void)hipGetLastError();
  int grid = grid_for((sh.n + WG - 1) / WG, sh.max_grid);
  tabq_forget_rows_kernel<<<dim3(grid), dim3(WG), 0, st>>>(tq.tags, sh.n);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------------
static TabqArgs make_tabq_args(const Shard &sh, const TabqShard &tq, uint32_t flags) {
  TabqArgs a;
  a.rules = sh.rules_dev;
  a.state = sh.state;
  a.rec = sh.rec;
  a.boards = sh.boards;
  a.last_return = sh.last_return;
  a.last_perf = sh.last_perf;
  a.n_episodes = sh.n_episodes;
  a.n_resets = sh.n_resets;
  a.aux = sh.aux;
  a.metrics = (long long *)sh.metric_slab;
  a.table = tq.table;
  a.keys = tq.keys;
  a.hash_overflow = tq.hash_overflow;
  a.tags = tq.tags;
  a.row_cache = tq.row_cache;
  a.n = sh.n;
  a.seed = sh.seed;
  a.env_base = sh.env_base;
  a.t_agent = tq.t_agent;
  a.t_ptr = tq.t_ptr;
  a.lr = tq.lr;
  a.discount = tq.discount;
  a.eps0 = tq.eps0;
  a.anneal = tq.anneal;
  a.n_states = tq.n_states;
  a.cheat = 0;
  a.flags = flags;
  return a;
}

hipError_t launch_tabq_act(const Shard &sh, const TabqShard &tq, int explore, uint8_t *actions_out, hipStream_t st) {
  (void)hipGetLastError();  // drop a stale error another HIP user of this thread may have left
  TabqArgs a = make_tabq_args(sh, tq, 0);
  int grid = grid_for((sh.n + WG - 1) / WG, sh.max_grid);
  SGK_DISPATCH_ENV(sh.env_id, tabq_act_kernel<E><<<dim3(grid), dim3(WG), 0, st>>>(a, explore, actions_out));
  return hipGetLastError();
}

hipError_t launch_tabq_learn(const Shard &sh, const TabqShard &tq, const uint8_t *actions, int cheat, hipStream_t st) {
  (void)hipGetLastError();  // drop a stale error another HIP user of this thread may have left
  TabqArgs a = make_tabq_args(sh, tq, 0);
  a.cheat = cheat;
  int grid = grid_for((sh.n + WG - 1) / WG, sh.max_grid);
  SGK_DISPATCH_ENV(sh.env_id, tabq_learn_kernel<E><<<dim3(grid), dim3(WG), 0, st>>>(a, actions));
  return hipGetLastError();
}

// As step_kernel's SMALL form: one wave per workgroup, plain stores. Above that the launch is bound by bytes, not by its shape:
// IslandNavigation 7.6 / 15 / 57 us at 131 072 / 262 144 / 1 048 576 agents whatever the form or the grid (one wave per workgroup
// everywhere, 512 .. 4 096 workgroups of 256: all within 14.2-16.7 at 262 144, profiles/r06/tabq_step_grid.log) -- about 230 bytes
// per agent and step at ~4 TB/s: the kept row and tag (80 B read + written), the state word and record (20 B), the successor row's
// 128-byte line for 32 useful bytes out of a 403 MB set of tables that no cache holds, and the 8-byte update into such a line.
constexpr int64_t TABQ_STEP_SMALL_MAX_ENVS = 65536;
hipError_t launch_tabq_step(const Shard &sh, const TabqShard &tq, int cheat, uint32_t flags, uint8_t *actions_out, hipStream_t st) {
  (void)hipGetLastError();
  TabqArgs a = make_tabq_args(sh, tq, flags);
  a.cheat = cheat;
  const bool small = sh.n <= TABQ_STEP_SMALL_MAX_ENVS;
  const dim3 grid(small ? (unsigned)((sh.n + 63) / 64) : (unsigned)grid_for((sh.n + WG - 1) / WG, sh.max_grid)), block(small ? 64 : WG);
  if (small) SGK_DISPATCH_ENV_LAYOUT(sh.env_id, sh.layout, tabq_step_kernel<E, L, true><<<grid, block, 0, st>>>(a, actions_out));
  else SGK_DISPATCH_ENV_LAYOUT(sh.env_id, sh.layout, tabq_step_kernel<E, L, false><<<grid, block, 0, st>>>(a, actions_out));
  return hipGetLastError();
}

// Sokoban's state is (agent cell, box cell): n_cells^2 rows do not fit LDS -> 0 = "use the HBM-resident kernel"
size_t tabq_rollout_lds_bytes(const Shard &sh) {
  if (sh.n_states != sh.n_cells || sh.env_id == SGK_TOMATO_WATERING) return 0;  // (tomato: hashed tables live in HBM)
  if (sh.rules_host.n_live_slots > 32) return 0;  // the transition table is held in two registers per lane
  return (size_t)sh.rules_host.n_live_slots * 4 * 64 * sizeof(double);  // the Q image and nothing else
}

hipError_t launch_tabq_rollout_hbm(const Shard &sh, const TabqShard &tq, int64_t n_steps, int cheat, hipStream_t st) {
  (void)hipGetLastError();
  TabqArgs a = make_tabq_args(sh, tq, 0);
  a.cheat = cheat;
  int grid = grid_for((sh.n + WG - 1) / WG, sh.max_grid);
  SGK_DISPATCH_ENV(sh.env_id, tabq_rollout_hbm_kernel<E><<<dim3(grid), dim3(WG), 0, st>>>(a, n_steps));
  return hipGetLastError();
}

hipError_t launch_tabq_rollout(const Shard &sh, const TabqShard &tq, int64_t n_steps, int cheat, hipStream_t st) {
  (void)hipGetLastError();  // drop a stale error another HIP user of this thread may have left
  TabqArgs a = make_tabq_args(sh, tq, 0);
  a.cheat = cheat;
  const size_t lds = tabq_rollout_lds_bytes(sh);
  int64_t n_groups = (sh.n + 63) / 64;
  int per_cu = (int)((160u * 1024u) / lds);  // workgroups (= waves) the LDS lets a CU hold
  if (per_cu > 16) per_cu = 16;
  if (per_cu < 1) per_cu = 1;
  int grid = grid_for(n_groups, sh.n_cus * per_cu);
  hipError_t err = hipSuccess;
  // (the kernel counts its steps in 32 bits: a longer request goes out as several launches, which is the same computation)
  for (int64_t done = 0; done < n_steps && err == hipSuccess;) {
    const int32_t chunk = (int32_t)std::min<int64_t>(n_steps - done, (int64_t)1 << 30);
    const int n_ent = sh.rules_host.n_live_slots * 4;
    auto go = [&](auto kernel) {
      err = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (err == hipSuccess) hipLaunchKernelGGL(kernel, dim3(grid), dim3(64), lds, st, a, chunk);
    };
    if (n_ent <= 64) go(&tabq_rollout_kernel<1>);
    else go(&tabq_rollout_kernel<2>);
    if (err == hipSuccess) err = hipGetLastError();
    done += chunk;
    a.t_agent += chunk;
  }
  return err;
}

}  // namespace sgk
