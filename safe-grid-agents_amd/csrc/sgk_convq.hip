// sgk_convq.hip -- a convolutional Q-body's forward + act_explore for every env in ONE launch (sgk_convq_act).
//
// NOT the reference's DeepQAgent (an MLP: value.py:148-158) -- BASELINE.json words config 4 as "conv policy", so the batched agent
// offers the body of the reference's PPO-CNN (policy_cnn.py:17-81) with a Q head as a labelled non-parity option (q_body="cnn"):
//     trunk = relu(conv3x3(relu(conv3x3(x, 1 -> C)), C -> C)) + conv1x1(x, 1 -> C)          (n_layers = 2)
//     Q     = linear(flatten(relu(conv3x3(trunk, C -> C))), C * H * W -> 4)
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
//     the next pass's boards are requested before the first convolution. 31-35 KB of LDS: four workgroups per CU.
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
  static constexpr int GROUPS = 8, SLOTS = 64 * GROUPS;  // 64-slot groups per pass: two per wave
  static constexpr int ENVS = SLOTS / SE;                // envs per pass
  static constexpr int ENV_F0 = PLANES * PL;
  static constexpr int ENV_F = ENV_F0 + (((SE - ENV_F0) % 32) + 32) % 32;  // env pitch == SE (mod 32): consecutive slots, consecutive banks
  static constexpr int MT = (C + 3) / 4;                 // 4-channel row tiles
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

// the taps of one convolution for this lane's two slots: acc[j][m] += W[4 m .. 4 m + 3][k] (x) window_k(slot j), k = 0 .. K - 1.
// Tap k = (ci, dy, dx) reads plane IN_PLANE + ci at window offset dy * PW + dx; a[m][k / 16] holds W[4 m + lane % 4][16 q + lane / 4].
template <class G, int K, int KC, int IN_PLANE>
__device__ __forceinline__ void cq_taps(const float *act, const int (&base)[2], const float (&a)[G::MT][KC], cq_f4 (&acc)[2][G::MT]) {
  cq_static_for<K>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    constexpr int off = (IN_PLANE + k / 9) * G::PL + ((k % 9) / 3) * G::PW + (k % 9) % 3;
    const float b0 = act[base[0] + off], b1 = act[base[1] + off];
    cq_static_for<G::MT>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      acc[0][m] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[m][k / 16], b0, acc[0][m], 4, k % 16, 0);
      acc[1][m] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[m][k / 16], b1, acc[1][m], 4, k % 16, 0);
    });
  });
}

template <int HH, int WW, int C>
__global__ __launch_bounds__(CQ_WG) void convq_act_kernel(const int8_t *__restrict__ boards, int pitch, ConvQWeights w,
                                                          uint8_t *__restrict__ actions, float *__restrict__ scores_out, int64_t n,
                                                          double eps, uint64_t seed, uint64_t env_base, uint64_t draw,
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
      a1[m][0] = (c < C && kk < G::K1) ? w.w1[c * 9 + kk] : 0.0f;
#pragma unroll
      for (int q = 0; q < G::KC2; ++q) {
        const int k = 16 * q + kk;
        const bool live = c < C && k < G::K2;
        a2[m][q] = live ? w.w2[c * G::K2 + k] : 0.0f;  // [c][ci][dy][dx] flattened = c * 9 C + k
        ah[m][q] = live ? w.wh[c * G::K2 + k] : 0.0f;
      }
    }
  }
  // ---- the linear head's weights per slot: WL[r][action][channel], zero rows for the border slots; the planes zeroed (borders stay
  // zero: the epilogues store zeros there) ----
  for (int i = t; i < G::SE * G::WLR; i += CQ_WG) {
    const int r = i / G::WLR, ac = i - r * G::WLR, a = ac / C, c = ac - a * C;
    const int y = r / G::PW, x = r - y * G::PW;
    WL[i] = x < WW ? w.wl[a * G::NF + c * G::NC + y * WW + x] : 0.0f;
  }
  for (int i = t; i < G::ENVS * G::ENV_F; i += CQ_WG) act[i] = 0.0f;
  // ---- this lane's two slots (the same in every pass): group g = wave + 4 j, slot s = 64 g + lane = (env e, r) ----
  int base[2], wlrow[2];
  bool valid[2], interior[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int s = 64 * (wave + 4 * j) + lane;
    valid[j] = s < G::ENVS * G::SE;
    const int e = valid[j] ? s / G::SE : 0, r = valid[j] ? s - e * G::SE : 0;
    base[j] = e * G::ENV_F + r;  // the window's top-left corner in plane 0
    wlrow[j] = r * G::WLR;
    interior[j] = valid[j] && (r % G::PW) < WW;
  }
  // ---- this lane's board bytes of a pass ----
  int b_lds[G::NB], b_env[G::NB], b_pos[G::NB];
  int8_t cur[G::NB];
#pragma unroll
  for (int u = 0; u < G::NB; ++u) {
    const int i = t + CQ_WG * u;
    const bool live = i < G::ENVS * G::NC;
    const int e = live ? i / G::NC : 0, pos = live ? i - e * G::NC : 0;
    const int y = pos / WW, x = pos - y * WW;
    b_env[u] = live ? e : -1;
    b_pos[u] = pos;
    b_lds[u] = e * G::ENV_F + G::CENTRE + y * G::PW + x;
    cur[u] = 0;
  }
  const int64_t n_pass = (n + G::ENVS - 1) / G::ENVS;
  auto request_boards = [&](int64_t pass) {
#pragma unroll
    for (int u = 0; u < G::NB; ++u) {
      const int64_t env = pass * G::ENVS + b_env[u];
      cur[u] = (b_env[u] >= 0 && env < n) ? boards[env * pitch + b_pos[u]] : (int8_t)0;
    }
  };
  if ((int64_t)blockIdx.x < n_pass) request_boards(blockIdx.x);
  float b1[C], b2[C], bh[C], wbv[C], bbv[C];  // uniform: scalar registers
#pragma unroll
  for (int c = 0; c < C; ++c) {
    b1[c] = w.b1[c];
    b2[c] = w.b2[c];
    bh[c] = w.bh[c];
    wbv[c] = w.wb[c];
    bbv[c] = w.bb[c];
  }
  const float bl0 = w.bl[0], bl1 = w.bl[1], bl2 = w.bl[2], bl3 = w.bl[3];
  __syncthreads();
  for (int64_t pass = blockIdx.x; pass < n_pass; pass += gridDim.x) {
    const int64_t env0 = pass * G::ENVS;
    // ---- the boards -> plane 0 (float); the next pass's bytes requested ----
#pragma unroll
    for (int u = 0; u < G::NB; ++u)
      if (b_env[u] >= 0) act[b_lds[u]] = (float)cur[u];
    if (pass + gridDim.x < n_pass) request_boards(pass + gridDim.x);
    __syncthreads();
    // ---- conv3x3 1 -> C, ReLU: planes 1 .. C ----
    {
      cq_f4 acc[2][G::MT] = {};
      cq_taps<G, G::K1, 1, 0>(act, base, a1, acc);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        if (valid[j]) {
#pragma unroll
          for (int c = 0; c < C; ++c) {
            const float v = fmaxf(acc[j][c / 4][c % 4] + b1[c], 0.0f);
            act[base[j] + (1 + c) * G::PL + G::CENTRE] = interior[j] ? v : 0.0f;
          }
        }
    }
    __syncthreads();
    // ---- conv3x3 C -> C, ReLU, + the 1 x 1 bottleneck of the board: planes C + 1 .. 2 C (the trunk) ----
    {
      cq_f4 acc[2][G::MT] = {};
      cq_taps<G, G::K2, G::KC2, 1>(act, base, a2, acc);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        if (valid[j]) {
          const float xin = act[base[j] + G::CENTRE];
#pragma unroll
          for (int c = 0; c < C; ++c) {
            const float v = fmaxf(acc[j][c / 4][c % 4] + b2[c], 0.0f) + fmaf(wbv[c], xin, bbv[c]);
            act[base[j] + (1 + C + c) * G::PL + G::CENTRE] = interior[j] ? v : 0.0f;
          }
        }
    }
    __syncthreads();
    // ---- head conv3x3 C -> C, ReLU, times this slot's rows of the linear layer: four per-slot products into planes 1 .. 4 ----
    {
      cq_f4 acc[2][G::MT] = {};
      cq_taps<G, G::K2, G::KC2, 1 + C>(act, base, ah, acc);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        if (valid[j]) {
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
          for (int c = 0; c < C; ++c) hv[c] = fmaxf(acc[j][c / 4][c % 4] + bh[c], 0.0f);
#pragma unroll
          for (int a = 0; a < 4; ++a) {
            float sacc = 0.0f;
#pragma unroll
            for (int c = 0; c < C; ++c) sacc = fmaf(hv[c], wr[a * C + c], sacc);  // border slots: zero rows of WL
            act[base[j] + (1 + a) * G::PL + G::CENTRE] = sacc;
          }
        }
    }
    __syncthreads();
    // ---- the four scores of an env: sixteen lanes per env = (action, third of the slots), summed in a fixed order; the draw ----
    constexpr int NIT = (G::ENVS * 16 + CQ_WG - 1) / CQ_WG, QN = (G::SE + 2) / 3;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = it * CQ_WG + t;
      const bool live = idx < G::ENVS * 16;
      const int e = live ? idx >> 4 : 0, a = (idx >> 2) & 3, q = idx & 3;
      float sum = 0.0f;
      if (q < 3) {
        const float *p = act + e * G::ENV_F + (1 + a) * G::PL + G::CENTRE;
        const int r1 = (q + 1) * QN < G::SE ? (q + 1) * QN : G::SE;
        for (int r = q * QN; r < r1; ++r) sum += p[r];
      }
      const float s1 = __shfl_down(sum, 1), s2 = __shfl_down(sum, 2);
      const float tot = (a == 0 ? bl0 : a == 1 ? bl1 : a == 2 ? bl2 : bl3) + ((sum + s1) + s2);  // (lanes with q == 0)
      const int l0 = lane & ~15;
      const float q0 = __shfl(tot, l0), q1 = __shfl(tot, l0 + 4), q2 = __shfl(tot, l0 + 8), q3 = __shfl(tot, l0 + 12);
      const int64_t env = env0 + e;
      if (live && (lane & 15) == 0 && env < n) {
        actions[env] = (uint8_t)pick_action<0>(q0, q1, q2, q3, env_base + (uint64_t)env, draw, seed, eps);
        if (scores_out) *reinterpret_cast<float4 *>(scores_out + 4 * env) = make_float4(q0, q1, q2, q3);
      }
    }
    // (the next pass writes plane 0 first and passes a barrier before planes 1 .. C are written again)
  }
}

hipError_t launch_convq_act(const Shard &sh, const ConvQWeights &w, int n_channels, uint8_t *actions, float *scores, double eps, uint64_t draw,
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
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(4, (160u * 1024u) / lds));                                \
    const int grid = grid_for(n_pass, sh.n_cus * per_cu);                                                                  \
    convq_act_kernel<HV, WV, CV><<<dim3(grid), dim3(CQ_WG), lds, st>>>(sh.boards, sh.pitch, w, actions, scores, sh.n, eps, sh.seed, \
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
