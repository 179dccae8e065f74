// sgk_convq_rollout.hip -- n_steps of {conv body forward, action draw, env.step} in ONE launch (sgk_convq_rollout): the inner loop of
// PPOBaseAgent.gather_rollout (reference policy_base.py:142-163: old_policy.act_explore -> env.step -> store state / action / reward)
// for a PPOCNNAgent (policy_cnn.py:17-81), or acting with a frozen conv Q-network (the batched DeepQ agent's non-parity option). What
// sgk_policy_rollout (sgk_policy.hip) is for the MLP bodies, on the kernel pieces of sgk_convq.h.
//
// A workgroup owns the ENVS envs of a pass for ALL n_steps: their state words live in the registers of ENVS lanes of wave 0, their
// boards as int8 rows in LDS. Per step: all lanes turn the rows into plane 0 (and store them as the trajectory's `states`), the three
// convolutions run as in sgk_convq_act (four waves, three barriers), wave 0 sums the outputs and draws, and its owner lanes step their
// envs (the kernels' one step_one), store action and record, and re-draw their rows. No board, action or record crosses HBM between
// steps except as trajectory output: per lockstep step the caller's four launches (forward + draw, board copy, sgk_step, record copy)
// become one pass of this loop.
#include "sgk_convq.h"

namespace sgk {

template <int ENV>
struct ConvDims;  // the level's board (sgk_levels.h); the conv kernels are specialised on (H, W)
template <> struct ConvDims<SGK_BOAT_RACE> { static constexpr int H = 5, W = 5; };
template <> struct ConvDims<SGK_ISLAND_NAVIGATION> { static constexpr int H = 6, W = 8; };
template <> struct ConvDims<SGK_SIDE_EFFECTS_SOKOBAN> { static constexpr int H = 6, W = 6; };
template <> struct ConvDims<SGK_DISTRIBUTIONAL_SHIFT> { static constexpr int H = 7, W = 9; };
template <> struct ConvDims<SGK_WHISKY_GOLD> { static constexpr int H = 6, W = 8; };
template <> struct ConvDims<SGK_ABSENT_SUPERVISOR> { static constexpr int H = 6, W = 8; };
template <> struct ConvDims<SGK_SAFE_INTERRUPTIBILITY> { static constexpr int H = 7, W = 8; };
template <> struct ConvDims<SGK_CONVEYOR_BELT> { static constexpr int H = 7, W = 7; };
template <> struct ConvDims<SGK_TOMATO_WATERING> { static constexpr int H = 7, W = 9; };
template <> struct ConvDims<SGK_FRIEND_FOE> { static constexpr int H = 6, W = 5; };

struct ConvRolloutArgs {
  StepArgs env;          // state / rec / episode arrays / metrics / rules / n / seed / env_base / flags
  double eps;            // mode 0
  uint64_t draw0;        // draw index of the first step; step k uses draw0 + k
  int32_t n_steps, mode; // mode 0: epsilon-greedy (Philox stream 2), 1: Categorical sample (stream 3)
  int8_t *states_out;    // [n_steps][n][NC] boards the policy acted on, or null
  uint8_t *actions_out;  // [n_steps][n] or null
  uint32_t *recs_out;    // [n_steps][n] step records or null
};

template <int ENV, int C>
struct ConvRolloutLds {
  typedef ConvQGeom<ConvDims<ENV>::H, ConvDims<ENV>::W, C> G;
  static constexpr size_t o_tile = (G::lds_bytes + 15) & ~(size_t)15;                     // int8 [ENVS][NC]
  static constexpr size_t o_flags = (o_tile + (size_t)G::ENVS * G::NC + 15) & ~(size_t)15;  // int32 action[ENVS], over[ENVS]
  static constexpr size_t o_rules = (o_flags + 2 * sizeof(int) * G::ENVS + 15) & ~(size_t)15;
  static constexpr size_t bytes = o_rules + sizeof(SgkRules) + 16;
};

// The register budget of sgk_convq_act, although the env state, the episode accumulators and the step's temporaries then spill 16-256
// bytes per lane to scratch (outside the convolutions): one wave per SIMD fewer has no spills and is SLOWER -- 28.3 against 23.7 us per
// lockstep step at 32 768 Sokoban envs and five channels, 55 against 45 on DistributionalShift (tools/gpu_convq_ab.sh).
#ifndef CQ_ROLLOUT_WAVES_FOR
#define CQ_ROLLOUT_WAVES_FOR(C) CQ_WAVES_FOR(C)
#endif
template <int ENV, int C>
__global__ __launch_bounds__(CQ_WG, CQ_ROLLOUT_WAVES_FOR(C)) void convq_rollout_kernel(
    ConvRolloutArgs a, const float *__restrict__ w1r, const float *__restrict__ b1r, const float *__restrict__ w2r,
    const float *__restrict__ b2r, const float *__restrict__ wbr, const float *__restrict__ bbr, const float *__restrict__ whr,
    const float *__restrict__ bhr, const float *__restrict__ wlr, const float *__restrict__ blr) {
  constexpr int HH = ConvDims<ENV>::H, WW = ConvDims<ENV>::W;
  typedef ConvQGeom<HH, WW, C> G;
  typedef ConvRolloutLds<ENV, C> LD;
  static_assert(G::NC == Geom<ENV>::NC, "level geometry");
  extern __shared__ __attribute__((aligned(16))) unsigned char convq_smem[];
  float *Lf = reinterpret_cast<float *>(convq_smem);
  float *WL = Lf + G::O_WL, *act = Lf + G::O_ACT;
  int8_t *tile = reinterpret_cast<int8_t *>(convq_smem + LD::o_tile);
  int *act_sh = reinterpret_cast<int *>(convq_smem + LD::o_flags), *over_sh = act_sh + G::ENVS;
  SgkRules &R = *reinterpret_cast<SgkRules *>(convq_smem + LD::o_rules);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  CqLane<G> L;
  cq_setup<G, WW, C>(L, WL, act, w1r, w2r, whr, wlr);
  // this lane's board bytes of a pass: element i = (env e, cell pos) of the row tile -> its place in plane 0
  int b_lds[G::NB];
#pragma unroll
  for (int u = 0; u < G::NB; ++u) {
    const int i = t + CQ_WG * u;
    const bool live = i < G::ENVS * G::NC;
    const int e = live ? i / G::NC : 0, pos = live ? i - e * G::NC : 0;
    const int y = pos / WW, x = pos - y * WW;
    b_lds[u] = live ? e * G::ENV_F + G::CENTRE + y * G::PW + x : -1;
  }
  stage_rules(R, a.env.rules);  // ends with a workgroup barrier: weights, planes and rules are in place
  const int64_t n = a.env.n;
  const int64_t n_pass = (n + G::ENVS - 1) / G::ENVS;
  const bool mask_finished = (a.env.flags & SGK_F_MASK_FINISHED) != 0;
  const bool owner = t < G::ENVS;  // lanes of wave 0: one env each
  EpisodeAcc acc;
  acc_init(acc);
  for (int64_t pass = blockIdx.x; pass < n_pass; pass += gridDim.x) {
    const int64_t env0 = pass * G::ENVS;
    const int hz = (int)(pass >> 44);  // (always 0: see cq_network)
    const int64_t env = env0 + t;
    const bool valid = owner && env < n;
    EnvState s = initial_state(R);
    if (valid) s = unpack_state(a.env.state[env]);
    load_episode_index<ENV>(s, a.env.n_resets, env, valid);
    int8_t *row = tile + (owner ? t : 0) * G::NC;
    if (owner) {
      write_row_bytes<ENV, G::NC>(R, row, s);  // draw this env's board from its state word
      over_sh[t] = (mask_finished && s.over) ? 1 : 0;
    }
    uint32_t rec = 0;
    const int64_t left = n - env0;
    const int lim = left < G::ENVS ? (int)left : G::ENVS;  // envs of this pass that exist
    for (int k = 0; k < a.n_steps; ++k) {
      __syncthreads();  // the rows (and the finished flags) of this step are complete; the previous step's sums have been read
      // ---- rows -> plane 0 (float) and -> the trajectory's states (zeros for envs whose episode is over, SGK_F_MASK_FINISHED) ----
      int8_t *dst = a.states_out ? a.states_out + ((int64_t)k * n + env0) * G::NC : nullptr;
#pragma unroll
      for (int u = 0; u < G::NB; ++u)
        if (b_lds[u] >= 0) {
          const int i = t + CQ_WG * u;
          const int8_t v = tile[i];
          act[b_lds[u]] = (float)v;
          if (dst) {
            const int e = i / G::NC;
            if (e < lim) dst[i] = over_sh[e] ? (int8_t)0 : v;
          }
        }
      __syncthreads();
      cq_network<G, C>(L, act, WL, hz, w1r, b1r, w2r, b2r, wbr, bbr, whr, bhr);
      if (wave == 0) {
        // ---- the outputs and the draw per env (lane = (env, action) quads), handed to the env's owner lane through LDS ----
        constexpr int NIT = (G::ENVS * 4 + 63) / 64;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const int idx = it * 64 + lane;
          float q0, q1, q2, q3;
          cq_outputs<G>(act, blr, idx, q0, q1, q2, q3);
          if (idx < G::ENVS * 4 && (idx & 3) == 0) {
            const uint64_t ge = a.env.env_base + (uint64_t)(env0 + (idx >> 2));
            act_sh[idx >> 2] = a.mode == 0 ? pick_action<0>(q0, q1, q2, q3, ge, a.draw0 + (uint64_t)k, a.env.seed, a.eps)
                                           : pick_action<1>(q0, q1, q2, q3, ge, a.draw0 + (uint64_t)k, a.env.seed, a.eps);
          }
        }
        __builtin_amdgcn_wave_barrier();  // (one wave: its LDS operations execute in order)
        // ---- env.step by the owner lanes; action and record into the trajectory; the row re-drawn ----
        if (owner) {
          const int action = act_sh[t];
          const bool was_over = mask_finished && s.over;
          const int old_pos = s.pos, old_box = s.box;
          const int old_alt = HasAltBackdrop<ENV>::value ? alt_backdrop<ENV>(R, s) : 0;
          step_one<ENV>(R, a.env, env, valid, action, s, rec, acc);
          if (valid) {
            if (a.actions_out) a.actions_out[(int64_t)k * n + env] = was_over ? (uint8_t)0 : (uint8_t)action;
            if (a.recs_out) a.recs_out[(int64_t)k * n + env] = rec;
          }
          const int new_alt = HasAltBackdrop<ENV>::value ? alt_backdrop<ENV>(R, s) : 0;
          if (HasMask<ENV>::value || (HasAltBackdrop<ENV>::value && new_alt != old_alt)) {
            // the other backdrop (an auto-reset flipped the supervisor's coin; the button was pressed; the agent stepped on or off the
            // bucket) or a level whose cells change by themselves (tomatoes dry): the whole row
            write_row_bytes<ENV, G::NC>(R, row, s);
          } else if (s.pos != old_pos || s.box != old_box) {  // re-draw the cells this step changed (a reset included)
            const uint8_t *backdrop = backdrop_of(R, new_alt);
            row[old_pos] = (int8_t)backdrop[old_pos];
            if (HasSprite2<ENV>::value) {
              if (old_box < G::NC) row[old_box] = (int8_t)backdrop[old_box];
              if (s.box < G::NC) row[s.box] = (int8_t)sprite2_value<ENV>(R, s);
            }
            row[s.pos] = (int8_t)R.agent_value[s.pos];
          }
          over_sh[t] = (mask_finished && s.over) ? 1 : 0;
        }
      }
    }
    if (valid) {
      a.env.state[env] = pack_state(s);
      a.env.rec[env] = rec;  // the env's own boards are re-materialised by the caller (launch_reset mode 2)
    }
    __syncthreads();  // nobody still reads this pass's rows when the next pass's owners draw theirs
  }
  acc_flush(acc, a.env.metrics);
}

hipError_t launch_convq_rollout(const Shard &sh, const ConvQWeights &w, int n_channels, int mode, double eps, uint64_t draw0, int32_t n_steps,
                                uint32_t flags, int8_t *states_out, uint8_t *actions_out, uint32_t *recs_out, hipStream_t st) {
  (void)hipGetLastError();
  ConvRolloutArgs a;
  a.env = make_step_args(sh, nullptr, flags);
  a.eps = eps;
  a.draw0 = draw0;
  a.n_steps = n_steps;
  a.mode = mode;
  a.states_out = states_out;
  a.actions_out = actions_out;
  a.recs_out = recs_out;
#define SGK_CONVQ_ROLLOUT(E, CV)                                                                                           \
  do {                                                                                                                     \
    typedef ConvRolloutLds<E, CV> LD;                                                                                      \
    static_assert(LD::bytes <= 160u * 1024u, "convq rollout LDS plan");                                                   \
    if (ConvDims<E>::H != sh.rules_host.height || ConvDims<E>::W != sh.rules_host.width) return hipErrorInvalidValue;      \
    constexpr size_t lds = LD::bytes;                                                                                      \
    const int64_t n_pass = (sh.n + LD::G::ENVS - 1) / LD::G::ENVS;                                                         \
    static std::atomic<unsigned long long> opted_in{0};                                                                    \
    if (!((opted_in.load() >> (sh.device & 63)) & 1ull)) {                                                                 \
      hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void *>(&convq_rollout_kernel<E, CV>),                    \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                           \
      if (ae != hipSuccess) return ae;                                                                                     \
      opted_in.fetch_or(1ull << (sh.device & 63));                                                                         \
    }                                                                                                                      \
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(CQ_ROLLOUT_WAVES_FOR(CV), (160u * 1024u) / lds));         \
    const int grid = grid_for(n_pass, sh.n_cus * per_cu);                                                                  \
    convq_rollout_kernel<E, CV><<<dim3(grid), dim3(CQ_WG), lds, st>>>(a, w.w1, w.b1, w.w2, w.b2, w.wb, w.bb, w.wh, w.bh, w.wl, w.bl); \
  } while (0)
  SGK_DISPATCH_ENV(sh.env_id, {
    if (n_channels == 5) SGK_CONVQ_ROLLOUT(E, 5);
    else if (n_channels == 4) SGK_CONVQ_ROLLOUT(E, 4);
    else if (n_channels == 8) SGK_CONVQ_ROLLOUT(E, 8);
    else return hipErrorInvalidValue;
  });
#undef SGK_CONVQ_ROLLOUT
  return hipGetLastError();
}

}  // namespace sgk
