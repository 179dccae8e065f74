// sgk_convq.hip -- the reference's convolutional body, forward + draw for every env in ONE launch (sgk_convq_sample, sgk_convq_act).
//
// The network is PPOCNNAgent's (policy_cnn.py:17-81) trunk and ONE four-way head:
//     trunk = relu(conv3x3(relu(conv3x3(x, 1 -> C)), C -> C)) + conv1x1(x, 1 -> C)          (n_layers = 2, the reference's default)
//     out   = linear(flatten(relu(conv3x3(trunk, C -> C))), C * H * W -> 4)
//   * sgk_convq_sample: head = actor_cnn / actor_linear, action = Categorical(logits = out).sample() -- PPOBaseAgent.act_explore
//     (policy_base.py:54-64) as ppo-cnn's gather_rollout calls it per step (policy_base.py:145). Pinned to the reference's own ppo-cnn
//     runs (tests/golden/batched_ppo_cnn_*.npz).
//   * sgk_convq_act: `out` read as four Q-values + DeepQAgent.act_explore's epsilon-greedy draw -- the batched DeepQ agent's labelled
//     NON-PARITY option q_body="cnn" (the reference's DeepQAgent is an MLP: value.py:148-158; BASELINE.json words config 4 as "conv
//     policy").
// Through PyTorch / MIOpen that forward costs 390 us per lockstep step at 32 768 envs (six tiny-spatial convolutions, an observation
// cast, a dozen launches); it is 18.7 k multiply-adds per 6 x 6 board.
//
// The shape of the problem: every convolution is an im2col GEMM with FIVE output channels (policy_cnn.py's default) -- on the
// 16 x 16 x 4 MFMA five of sixteen rows would be real (the first form of this kernel: 101 us per launch, EXPERIMENTS.md R6.6). Hence
// v_mfma_f32_4x4x1_16b_f32: sixteen independent 4 x 4 outer products per instruction, exact fp32, the same 64 flop / clk / SIMD.
//   * lane = one output SLOT (block = lane / 4, column = lane % 4); its B operand is that slot's window tap, read from LDS at a
//     compile-time offset; the A operand -- four output channels' weights for one tap -- is broadcast from one block of a register
//     that holds sixteen taps (cbsz = 4, abid = tap % 16): all weights of the three convolutions live in 14 VGPRs per lane.
//     The accumulator leaves the lane with all channels of its own slot: bias, ReLU, the 1 x 1 residual and the linear head's
//     per-slot products are register epilogues.
//   * activations live in LDS as planes with ONE zero column shared by the end of a row and the start of the next (row pitch W + 1)
//     and a zero row above and below; the slots of a pass enumerate those addresses in order (border slots included: W / (W + 1) of
//     the lanes do useful work), and the env-to-env pitch is congruent to the slots of one env modulo 32 -- so the 32 lanes of an LDS
//     access group always touch 32 consecutive banks: every window read and every store is conflict-free.
//   * a workgroup of 4 waves takes ENVS = 512 / slots-per-env envs per pass, two 64-slot groups per wave, four barriers per pass;
//     the next pass's boards are requested before the first convolution; one wave sums the scores and draws. 31-35 KB of LDS: three
//     (five channels) or four workgroups per CU.
//   * five channels = one full row tile on the MFMA + a lone fifth channel: its weights are uniform, so they sit in scalar registers
//     and multiply the same window values on the VALU (per slot and window row one v_pk_fma_f32 on the ds_read2 pair + one v_fmac):
//     8 cycles instead of three MFMAs with one live row in four (24).
// Measured (32 768 Sokoban boards, 5 channels; EXPERIMENTS.md R6.6): 28.3 us per launch = 43 TFLOP/s of useful fp32 (0.28 of the
// matrix peak; 57 TFLOP/s at 1 M boards); the lockstep step around it 33 us against 390 through torch. What is left is instruction
// issue, not the MFMA: per wave and pass 198 MFMAs (1 584 cycles) beside ~450 other vector instructions (epilogues, LDS address
// adds for ds_read2's 8-bit offsets, the Philox draw).
// The epsilon-greedy draw is sgk_epsilon_greedy's (Philox stream 2, keyed by global env index). fp32 with another summation order than
// MIOpen / rocBLAS: scores agree with the torch module to fp32 tolerance (tests/test_gpu_convq.py: rtol 1e-4).
#include <algorithm>
#include <atomic>
#include <type_traits>

#include "sgk_device.h"
#include "sgk_draws.h"
#include "sgk_kernels.h"

namespace sgk {

constexpr int CQ_WG = 256;  // 4 waves
#ifndef CQ_GPW
#define CQ_GPW 2  // 64-slot groups per wave and pass (A/B: tools/gpu_convq_ab.sh)
#endif
// waves per SIMD the register allocation aims at = workgroups per CU: four, but three with five channels (the fifth channel's 45
// scalar weights next to 21 pointer arguments spill scalar registers into vector lanes; at four waves those spill on to scratch:
// 31.8 us against 28.3 at 32 768 Sokoban boards; with four or eight channels four waves win by 2-8 %)
#ifdef CQ_MIN_WAVES
#define CQ_WAVES_FOR(C) CQ_MIN_WAVES
#else
#define CQ_WAVES_FOR(C) ((C) % 4 == 1 ? 3 : 4)
#endif

template <int N, int I = 0, class F>
__device__ __forceinline__ void cq_static_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    cq_static_for<N, I + 1>(f);
  }
}

template <int HH, int WW, int C>
struct ConvQGeom {
  static constexpr int NC = HH * WW;                     // cells
  static constexpr int PW = WW + 1;                      // row pitch: W cells + the zero column shared with the next row
  static constexpr int SE = HH * PW;                     // slots of one env (interior rows, border column included)
  static constexpr int PL = (HH + 2) * PW + 1;           // floats per plane: a zero row above and below, one leading zero
  static constexpr int PLANES = 1 + 2 * C;               // x | h1 (later: the linear head's per-slot products) | trunk
  static constexpr int GPW = CQ_GPW, GROUPS = 4 * GPW, SLOTS = 64 * GROUPS;  // 64-slot groups per pass: GPW per wave
  static constexpr int ENVS = SLOTS / SE;                // envs per pass
  static constexpr int ENV_F0 = PLANES * PL;
  static constexpr int ENV_F = ENV_F0 + (((SE - ENV_F0) % 32) + 32) % 32;  // env pitch == SE (mod 32): consecutive slots, consecutive banks
  static constexpr int CR = (C % 4 == 1) ? 1 : 0;        // a lone fifth channel: scalar-weight FMAs instead of a 1/4-full MFMA tile
  static constexpr int MT = (C + 3) / 4 - CR;            // 4-channel row tiles on the MFMA
  static constexpr int CM = CR ? 4 * MT : C;             // channels on the MFMA
  static constexpr int K1 = 9, K2 = 9 * C, KC2 = (K2 + 15) / 16;
  static constexpr int NF = C * NC;                      // linear inputs
  static constexpr int WLR = 4 * C;                      // the linear weights of one slot: [action][channel]
  static constexpr int O_WL = 0, O_ACT = (SE * WLR + 3) & ~3;
  static constexpr int CENTRE = PW + 1;                  // slot r's own cell relative to its window's top-left corner
  static constexpr size_t lds_bytes = sizeof(float) * (size_t)(O_ACT + ENVS * ENV_F);
  static constexpr int NB = (ENVS * NC + CQ_WG - 1) / CQ_WG;  // board bytes per lane and pass
  static_assert(ENVS >= 1 && 4 * PL <= C * PL && SE <= PL, "convq geometry");
};

typedef float cq_f4 __attribute__((ext_vector_type(4)));
typedef float cq_f2 __attribute__((ext_vector_type(2)));

// the taps of one convolution for this lane's two slots: acc[j][m] += W[4 m .. 4 m + 3][k] (x) window_k(slot j), k = 0 .. K - 1.
// Tap k = (ci, dy, dx) reads plane IN_PLANE + ci at window offset dy * PW + dx; a[m][k / 16] holds W[4 m + lane % 4][16 q + lane / 4].
// With five channels the fifth's weights (wrem[k], uniform: scalar registers) multiply the same window values on the VALU: per slot
// and window row one v_pk_fma_f32 + one v_fmac (8 cycles) instead of three MFMAs with one live row in four (24 cycles).
template <class G, int K, int KC, int IN_PLANE>
__device__ __forceinline__ void cq_taps(const float *act, const int (&base)[G::GPW], const float (&a)[G::MT][KC], const float *__restrict__ wrem,
                                        cq_f4 (&acc)[G::GPW][G::MT], float (&racc)[G::GPW]) {
  cq_f2 racc2[G::GPW] = {};
  float racc1[G::GPW] = {};
  // one input channel (nine taps, eighteen window values) at a time, the next channel's values requested before this one's arithmetic:
  // the compiler left alone requests all 2 K values first (152 VGPRs at five channels: three waves per SIMD instead of four)
  constexpr int CIN = K / 9;
  // a window row = the pair (dx 0, dx 1), one ds_read2_b32 into an aligned register pair, and the single dx 2
  cq_f2 pr[2][G::GPW][3];  // [parity of the input channel][slot][dy]
  float sg[2][G::GPW][3];
  auto request = [&](auto cic) {
    constexpr int ci = decltype(cic)::value;
    cq_static_for<3>([&](auto dc) {
      constexpr int dy = decltype(dc)::value;
      constexpr int off = (IN_PLANE + ci) * G::PL + dy * G::PW;
#pragma unroll
      for (int j = 0; j < G::GPW; ++j) {
        pr[ci & 1][j][dy] = cq_f2{act[base[j] + off], act[base[j] + off + 1]};
        sg[ci & 1][j][dy] = act[base[j] + off + 2];
      }
    });
  };
  request(std::integral_constant<int, 0>{});
  cq_static_for<CIN>([&](auto cic) {
    constexpr int ci = decltype(cic)::value;
    if constexpr (ci + 1 < CIN) request(std::integral_constant<int, ci + 1>{});
    __builtin_amdgcn_sched_barrier(0);
    cq_static_for<3>([&](auto dc) {
      constexpr int dy = decltype(dc)::value;
      constexpr int k = 9 * ci + 3 * dy;
      cq_static_for<3>([&](auto xc) {
        constexpr int dx = decltype(xc)::value;
        cq_static_for<G::MT>([&](auto mc) {
          constexpr int m = decltype(mc)::value;
#pragma unroll
          for (int j = 0; j < G::GPW; ++j) {
            const float b = dx == 2 ? sg[ci & 1][j][dy] : pr[ci & 1][j][dy][dx];
            acc[j][m] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[m][(k + dx) / 16], b, acc[j][m], 4, (k + dx) % 16, 0);
          }
        });
      });
      if constexpr (G::CR == 1) {  // two taps of one slot per v_pk_fma_f32: scalar register pair x the ds_read2 pair
        const cq_f2 wp = {wrem[k], wrem[k + 1]};
        const float ws = wrem[k + 2];
#pragma unroll
        for (int j = 0; j < G::GPW; ++j) {
          racc2[j] = __builtin_elementwise_fma(wp, pr[ci & 1][j][dy], racc2[j]);
          racc1[j] = fmaf(ws, sg[ci & 1][j][dy], racc1[j]);
        }
      }
    });
    __builtin_amdgcn_sched_barrier(0);
  });
  if constexpr (G::CR == 1) {
#pragma unroll
    for (int j = 0; j < G::GPW; ++j) racc[j] = (racc2[j][0] + racc2[j][1]) + racc1[j];
  }
}

// channel c of a slot's accumulators
template <class G>
__device__ __forceinline__ float cq_channel(const cq_f4 (&acc)[G::MT], float racc, int c) {
  return c < G::CM ? acc[c / 4][c % 4] : racc;
}

// (the ten weight tensors as restrict-qualified kernel arguments, not as a struct of pointers: only then may the per-channel constants
// and the fifth channel's weights be read through the scalar cache inside the pass loop, next to the kernel's own stores)
template <int HH, int WW, int C>
__global__ __launch_bounds__(CQ_WG, CQ_WAVES_FOR(C)) void convq_act_kernel(
    const int8_t *__restrict__ boards, int pitch, const float *__restrict__ w1r, const float *__restrict__ b1r, const float *__restrict__ w2r,
    const float *__restrict__ b2r, const float *__restrict__ wbr, const float *__restrict__ bbr, const float *__restrict__ whr,
    const float *__restrict__ bhr, const float *__restrict__ wlr, const float *__restrict__ blr, uint8_t *__restrict__ actions,
    float *__restrict__ scores_out, int64_t n, int mode, double eps, uint64_t seed, uint64_t env_base, uint64_t draw,
    const double *__restrict__ eps_ptr, const uint64_t *__restrict__ draw_ptr) {
  typedef ConvQGeom<HH, WW, C> G;
  extern __shared__ __attribute__((aligned(16))) unsigned char convq_smem[];
  float *L = reinterpret_cast<float *>(convq_smem);
  float *WL = L + G::O_WL, *act = L + G::O_ACT;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (eps_ptr) eps = *eps_ptr;  // device-resident scalars: the launch can be replayed from a graph
  if (draw_ptr) draw = *draw_ptr;
  // ---- the convolutions' weights as A operands, in registers for the whole launch ----
  float a1[G::MT][1], a2[G::MT][G::KC2], ah[G::MT][G::KC2];
  {
    const int i = lane & 3, kk = lane >> 2;
#pragma unroll
    for (int m = 0; m < G::MT; ++m) {
      const int c = 4 * m + i;
      a1[m][0] = (c < G::CM && kk < G::K1) ? w1r[c * 9 + kk] : 0.0f;
#pragma unroll
      for (int q = 0; q < G::KC2; ++q) {
        const int k = 16 * q + kk;
        const bool live = c < G::CM && k < G::K2;
        a2[m][q] = live ? w2r[c * G::K2 + k] : 0.0f;  // [c][ci][dy][dx] flattened = c * 9 C + k
        ah[m][q] = live ? whr[c * G::K2 + k] : 0.0f;
      }
    }
  }
  // ---- the linear head's weights per slot: WL[r][action][channel], zero rows for the border slots; the planes zeroed (borders stay
  // zero: the epilogues store zeros there) ----
  for (int i = t; i < G::SE * G::WLR; i += CQ_WG) {
    const int r = i / G::WLR, ac = i - r * G::WLR, a = ac / C, c = ac - a * C;
    const int y = r / G::PW, x = r - y * G::PW;
    WL[i] = x < WW ? wlr[a * G::NF + c * G::NC + y * WW + x] : 0.0f;
  }
  for (int i = t; i < G::ENVS * G::ENV_F; i += CQ_WG) act[i] = 0.0f;
  // ---- this lane's two slots (the same in every pass): group g = wave + 4 j, slot s = 64 g + lane = (env e, r) ----
  int base[G::GPW], wlrow[G::GPW];
  bool valid[G::GPW], interior[G::GPW];
#pragma unroll
  for (int j = 0; j < G::GPW; ++j) {
    const int s = 64 * (wave + 4 * j) + lane;
    valid[j] = s < G::ENVS * G::SE;
    const int e = valid[j] ? s / G::SE : 0, r = valid[j] ? s - e * G::SE : 0;
    base[j] = e * G::ENV_F + r;  // the window's top-left corner in plane 0
    wlrow[j] = r * G::WLR;
    interior[j] = valid[j] && (r % G::PW) < WW;
  }
  // ---- this lane's board bytes of a pass ----
  int b_lds[G::NB], b_env[G::NB], b_goff[G::NB];
  int8_t cur[G::NB];
#pragma unroll
  for (int u = 0; u < G::NB; ++u) {
    const int i = t + CQ_WG * u;
    const bool live = i < G::ENVS * G::NC;
    const int e = live ? i / G::NC : 0, pos = live ? i - e * G::NC : 0;
    const int y = pos / WW, x = pos - y * WW;
    b_env[u] = live ? e : 0x7fffffff;
    b_goff[u] = e * pitch + pos;
    b_lds[u] = e * G::ENV_F + G::CENTRE + y * G::PW + x;
    cur[u] = 0;
  }
  const int64_t n_pass = (n + G::ENVS - 1) / G::ENVS;
  auto request_boards = [&](int64_t pass) {
    const int8_t *src = boards + pass * G::ENVS * (int64_t)pitch;  // (uniform)
    const int64_t left = n - pass * G::ENVS;
    const int lim = left < G::ENVS ? (int)left : G::ENVS;
#pragma unroll
    for (int u = 0; u < G::NB; ++u) cur[u] = b_env[u] < lim ? src[b_goff[u]] : (int8_t)0;
  };
  if ((int64_t)blockIdx.x < n_pass) request_boards(blockIdx.x);
  __syncthreads();
  for (int64_t pass = blockIdx.x; pass < n_pass; pass += gridDim.x) {
    const int64_t env0 = pass * G::ENVS;
    // (always 0, but not to the compiler: the fifth channel's scalar weights are then loaded per pass into scalar registers instead of
    // being hoisted into ~100 vector registers for the whole launch, which halves the occupancy)
    const int hz = (int)(pass >> 44);
    // ---- the boards -> plane 0 (float); the next pass's bytes requested ----
#pragma unroll
    for (int u = 0; u < G::NB; ++u)
      if (b_env[u] < G::ENVS) act[b_lds[u]] = (float)cur[u];
    if (pass + gridDim.x < n_pass) request_boards(pass + gridDim.x);
    __syncthreads();
    // ---- conv3x3 1 -> C, ReLU: planes 1 .. C (border slots are never written: they stay zero) ----
    {
      cq_f4 acc[G::GPW][G::MT] = {};
      float racc[G::GPW] = {};
      cq_taps<G, G::K1, 1, 0>(act, base, a1, w1r + G::CM * G::K1 + hz, acc, racc);
#pragma unroll
      for (int j = 0; j < G::GPW; ++j)
        if (interior[j]) {
#pragma unroll
          for (int c = 0; c < C; ++c) act[base[j] + (1 + c) * G::PL + G::CENTRE] = fmaxf(cq_channel<G>(acc[j], racc[j], c) + b1r[c + hz], 0.0f);
        }
    }
    __syncthreads();
    // ---- conv3x3 C -> C, ReLU, + the 1 x 1 bottleneck of the board: planes C + 1 .. 2 C (the trunk) ----
    {
      cq_f4 acc[G::GPW][G::MT] = {};
      float racc[G::GPW] = {};
      cq_taps<G, G::K2, G::KC2, 1>(act, base, a2, w2r + G::CM * G::K2 + hz, acc, racc);
#pragma unroll
      for (int j = 0; j < G::GPW; ++j)
        if (interior[j]) {
          const float xin = act[base[j] + G::CENTRE];
#pragma unroll
          for (int c = 0; c < C; ++c)
            act[base[j] + (1 + C + c) * G::PL + G::CENTRE] = fmaxf(cq_channel<G>(acc[j], racc[j], c) + b2r[c + hz], 0.0f) + fmaf(wbr[c + hz], xin, bbr[c + hz]);
        }
    }
    __syncthreads();
    // ---- head conv3x3 C -> C, ReLU, times this slot's rows of the linear layer: four per-slot products into planes 1 .. 4 ----
    {
      cq_f4 acc[G::GPW][G::MT] = {};
      float racc[G::GPW] = {};
      cq_taps<G, G::K2, G::KC2, 1 + C>(act, base, ah, whr + G::CM * G::K2 + hz, acc, racc);
#pragma unroll
      for (int j = 0; j < G::GPW; ++j)
        if (interior[j]) {
          float wr[G::WLR];
          const cq_f4 *wp = reinterpret_cast<const cq_f4 *>(WL + wlrow[j]);
#pragma unroll
          for (int q = 0; q < C; ++q) {
            const cq_f4 v4 = wp[q];
            wr[4 * q] = v4[0];
            wr[4 * q + 1] = v4[1];
            wr[4 * q + 2] = v4[2];
            wr[4 * q + 3] = v4[3];
          }
          float hv[C];
#pragma unroll
          for (int c = 0; c < C; ++c) hv[c] = fmaxf(cq_channel<G>(acc[j], racc[j], c) + bhr[c + hz], 0.0f);
#pragma unroll
          for (int a = 0; a < 4; ++a) {
            float sacc = 0.0f;
#pragma unroll
            for (int c = 0; c < C; ++c) sacc = fmaf(hv[c], wr[a * C + c], sacc);
            act[base[j] + (1 + a) * G::PL + G::CENTRE] = sacc;
          }
        }
    }
    __syncthreads();
    // ---- the four scores of an env and its draw, by ONE wave (the other three go on to the next pass's boards): lane = (env, action)
    // sums that env's per-slot products in slot order (border slots hold zeros) ----
    if (wave == 0) {
      constexpr int NIT = (G::ENVS * 4 + 63) / 64;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int idx = it * 64 + lane;
        const bool live = idx < G::ENVS * 4;
        const int e = live ? idx >> 2 : 0, a = idx & 3;
        const float *p = act + e * G::ENV_F + (1 + a) * G::PL + G::CENTRE;
        float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
        constexpr int SE3 = G::SE / 3;
#pragma unroll 2
        for (int r = 0; r < SE3; ++r) {
          s0 += p[r];
          s1 += p[SE3 + r];
          s2 += p[2 * SE3 + r];
        }
#pragma unroll
        for (int r = 3 * SE3; r < G::SE; ++r) s0 += p[r];
        const float tot = blr[a] + ((s0 + s1) + s2);
        const int l0 = lane & ~3;
        const float q0 = __shfl(tot, l0), q1 = __shfl(tot, l0 + 1), q2 = __shfl(tot, l0 + 2), q3 = __shfl(tot, l0 + 3);
        const int64_t env = env0 + e;
        if (live && a == 0 && env < n) {
          // mode 0: DeepQAgent.act_explore's epsilon-greedy draw on four scores; mode 1: Categorical(logits).sample() (PPOCNNAgent)
          actions[env] = (uint8_t)(mode == 0 ? pick_action<0>(q0, q1, q2, q3, env_base + (uint64_t)env, draw, seed, eps)
                                             : pick_action<1>(q0, q1, q2, q3, env_base + (uint64_t)env, draw, seed, eps));
          if (scores_out) *reinterpret_cast<float4 *>(scores_out + 4 * env) = make_float4(q0, q1, q2, q3);
        }
      }
    }
    // (the next pass writes plane 0 first and passes a barrier before planes 1 .. C are written again)
  }
}

hipError_t launch_convq_act(const Shard &sh, const ConvQWeights &w, int n_channels, int mode, uint8_t *actions, float *scores, double eps, uint64_t draw,
                            const double *eps_dev, const uint64_t *draw_dev, hipStream_t st) {
  (void)hipGetLastError();
  const int H = sh.rules_host.height, W = sh.rules_host.width;
#define SGK_CONVQ_LAUNCH(HV, WV, CV)                                                                                       \
  do {                                                                                                                     \
    constexpr size_t lds = ConvQGeom<HV, WV, CV>::lds_bytes;                                                               \
    const int64_t n_pass = (sh.n + ConvQGeom<HV, WV, CV>::ENVS - 1) / ConvQGeom<HV, WV, CV>::ENVS;                                                               \
    static_assert(lds <= 160u * 1024u, "convq LDS plan");                                                                 \
    static std::atomic<unsigned long long> opted_in{0};                                                                    \
    if (!((opted_in.load() >> (sh.device & 63)) & 1ull)) {                                                                 \
      hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void *>(&convq_act_kernel<HV, WV, CV>),                   \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                           \
      if (ae != hipSuccess) return ae;                                                                                     \
      opted_in.fetch_or(1ull << (sh.device & 63));                                                                         \
    }                                                                                                                      \
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(CQ_WAVES_FOR(CV), (160u * 1024u) / lds));                                \
    const int grid = grid_for(n_pass, sh.n_cus * per_cu);                                                                  \
    convq_act_kernel<HV, WV, CV><<<dim3(grid), dim3(CQ_WG), lds, st>>>(sh.boards, sh.pitch, w.w1, w.b1, w.w2, w.b2, w.wb, w.bb, w.wh, w.bh, \
                                                                       w.wl, w.bl, actions, scores, sh.n, mode, eps, sh.seed, \
                                                                       sh.env_base, draw, eps_dev, draw_dev);              \
  } while (0)
#define SGK_CONVQ_LAUNCH_C(HV, WV)                                                                                         \
  do {                                                                                                                     \
    if (n_channels == 5) SGK_CONVQ_LAUNCH(HV, WV, 5);                                                                      \
    else if (n_channels == 4) SGK_CONVQ_LAUNCH(HV, WV, 4);                                                                 \
    else if (n_channels == 8) SGK_CONVQ_LAUNCH(HV, WV, 8);                                                                 \
    else return hipErrorInvalidValue;                                                                                      \
  } while (0)
  if (H == 5 && W == 5) SGK_CONVQ_LAUNCH_C(5, 5);
  else if (H == 6 && W == 5) SGK_CONVQ_LAUNCH_C(6, 5);
  else if (H == 6 && W == 6) SGK_CONVQ_LAUNCH_C(6, 6);
  else if (H == 6 && W == 8) SGK_CONVQ_LAUNCH_C(6, 8);
  else if (H == 7 && W == 7) SGK_CONVQ_LAUNCH_C(7, 7);
  else if (H == 7 && W == 8) SGK_CONVQ_LAUNCH_C(7, 8);
  else if (H == 7 && W == 9) SGK_CONVQ_LAUNCH_C(7, 9);
  else return hipErrorInvalidValue;
#undef SGK_CONVQ_LAUNCH_C
#undef SGK_CONVQ_LAUNCH
  return hipGetLastError();
}

}  // namespace sgk
