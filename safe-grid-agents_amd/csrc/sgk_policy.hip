// sgk_policy.hip -- the network-facing kernels: action draws on scores / logits (DeepQAgent.act_explore, reference
// value.py:94-111; PPOBaseAgent.act_explore, policy_base.py:54-64), the fused MLP forward + draw on the matrix cores
// (value.py:89-111,148-158; policy_mlp.py:17-43) and PPOBaseAgent.get_discounted_returns (policy_base.py:179-186).
#include <atomic>

#include "sgk_device.h"
#include "sgk_draws.h"

namespace sgk {

// One lane per env: a 16-byte load of the scores, a Philox block, a byte store. Replaces six PyTorch kernels (argmax,
// rand, lt, randint, where, cast -- or softmax, multinomial, cast) per lockstep step.
template <int MODE>
__global__ __launch_bounds__(WG) void eps_greedy_kernel(const float4 *__restrict__ scores, uint8_t *__restrict__ actions,
                                                        int64_t n, double eps, uint64_t seed, uint64_t env_base,
                                                        uint64_t draw, const double *__restrict__ eps_ptr,
                                                        const uint64_t *__restrict__ draw_ptr) {
  if (eps_ptr) eps = *eps_ptr;     // device-resident scalars: the launch can be replayed from a graph
  if (draw_ptr) draw = *draw_ptr;
  for (int64_t env = (int64_t)blockIdx.x * WG + threadIdx.x; env < n; env += (int64_t)gridDim.x * WG) {
    const float4 q = scores[env];
    actions[env] = (uint8_t)pick_action<MODE>(q.x, q.y, q.z, q.w, env_base + (uint64_t)env, draw, seed, eps);
  }
}

// DeepQAgent's Q-network forward + act_explore (MODE 0), or PPOMLPAgent's trunk + actor forward + Categorical draw (MODE 1),
// for every env in ONE launch (reference value.py:89-111,148-158 / policy_mlp.py:17-43 with the default topology
// n_layers = 2: Linear(K0,H)+ReLU, Linear(H,H)+ReLU, Linear(H,4)). PyTorch needs five kernels for the three small GEMMs
// (M = n_envs, K,N <= 100: 48 us at 32 768 envs, profiles/r01) plus the observation cast and six more for the epsilon-greedy
// mix. Weights are read in place from the torch parameters: w1t = W1^T [K0][H], w2 = W2 [H][H] (torch layout),
// w3t = W3^T [H][4]. fp32 throughout; the summation order differs from rocBLAS, so parity with the torch forward is to fp32
// tolerance (tests: rtol 1e-4), not bit-exact.
// Round-1 history (profiles/r01/policy_kernel.md): a one-lane-per-env VALU kernel (weights broadcast from LDS, v_pk_fma_f32)
// took 38 us at 32 768 envs -- a lone wave per SIMD issues a packed FMA only every ~8 cycles; this MFMA formulation 14.5 us.
// ------------------------------------------------------------------------------------------------
// The forward runs on the matrix cores: v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate: a k-ordered fmaf chain, exact
// f32 -- no reduced precision), transposed formulation out^T[n][env] = W[n][k] * in^T[k][env]:
//   A operand (16 x 4)  = 16 output neurons x 4 input features of the weight matrix   (lane l: A[l & 15][l >> 4])
//   B operand (4 x 16)  = 4 input features x 16 envs                                  (lane l: B[l >> 4][l & 15])
//   C/D       (16 x 16) = 16 output neurons x 16 envs, 4 VGPRs: lane l, reg r = C[4 * (l >> 4) + r][l & 15]
// The layers chain in registers: the C registers of layer i ARE the B operands of layer i + 1, because the order of the
// K summation is free -- k-step (mt, r) of the next layer takes feature 16 mt + 4 (l >> 4) + r from lane l, which is
// exactly C register r of neuron tile mt, and the weights are staged in LDS in that k order (one ds_read_b128 = the A
// operands of four k-steps). Hidden width H is padded to MT = ceil(H / 16) tiles with zero weights and biases (ReLU(0) = 0
// contributes nothing downstream); the 4 action scores occupy rows 0..3 of one more tile.
// A wave owns NT = 2 env tiles (32 envs) per pass so that every A operand read from LDS feeds two MFMAs and 14 independent
// accumulators cover the 40-cycle dependent latency; a workgroup of 4 waves = 128 envs, i.e. at 32 768 envs one wave per
// SIMD on all 1 024 SIMDs. Work per 32 envs: (MT * ceil(K0 / 4) + MT * 4 MT + 4 MT) * 2 MFMAs of 32 cycles
// = 574 MFMAs = 18.4 k cycles at H = 100, K0 = 36 (8.4 us at the ~2.2 GHz the kernel runs at; 526 since round 4: the partial
// tile's k-steps, PolicyMfmaGeom below). Measured in-kernel (round 1, 574 MFMAs) at 32 768
// envs (clock64): 5.1 k cycles until the first tile and W1 are staged (one cold memory round trip), 5.7 k layer 1, 14.8 k
// layer 2 (incl. the W2 commit), 4.0 k layer 3 + draw + stores = 14.5 us per launch; 10.0 us per 128-env pass in steady
// state at 1 M envs (84 % of the MFMA issue bound; the useful-FLOP rate is 82 TFLOP/s f32 because of the 100 -> 112 padding).
constexpr int PMFMA_WG = 256;           // 4 waves
constexpr int PMFMA_NT = 2;             // env tiles (of 16) per wave and pass
constexpr int PMFMA_ENVS = (PMFMA_WG / 64) * PMFMA_NT * 16;  // 128 envs per workgroup and pass

template <int K0, int H>
struct PolicyMfmaGeom {
  static constexpr int MT = (H + 15) / 16;   // neuron tiles of the hidden layers
  static constexpr int KS1 = (K0 + 3) / 4;   // k-steps of the first layer
  // Where the hidden neurons sit. A full tile t < FT holds neurons 16 t .. 16 t + 15 in row order. The last, partial tile (H = 100:
  // four neurons) spreads its neurons over the rows 4 g + j with j < RK -- one per lane group g and k-step j -- because a k-step
  // (tile, r) of the NEXT layer consumes rows {4 g + r : g = 0..3} of that tile: packed like this its R neurons cost R / 4 k-steps
  // instead of four (H = 100: 25 k-steps per hidden K dimension instead of 28; 526 MFMAs per 32 envs instead of 574). The order of
  // neurons inside a layer is free as long as the next layer's K order follows it: `neuron()` is the one place that decides it.
  static_assert(H % 4 == 0, "hidden widths are multiples of four");
  static constexpr int FT = H / 16, RK = (H % 16) / 4;
  static __host__ __device__ constexpr int neuron(int t, int row) {
    return t < FT ? 16 * t + row : ((t == FT && (row & 3) < RK) ? 16 * FT + (row >> 2) * RK + (row & 3) : -1);
  }
  static __host__ __device__ constexpr int ksteps(int t) { return t < FT ? 4 : RK; }  // k-steps tile t feeds the next layer with
  static constexpr int W1 = MT * KS1 * 64;   // floats: [mt][s][lane]
  static constexpr int W2 = MT * MT * 64 * 4;  // floats: [mt_out][mt_k][lane][r]
  static constexpr int W3 = MT * 64 * 4;     // floats: [mt_k][lane][r]
  static constexpr int B = MT * 16;          // padded bias vectors
  static constexpr int TILE_DW = PMFMA_ENVS * K0 / 4;                           // dwords of one board tile
  static constexpr int TILE_LD = (TILE_DW + PMFMA_WG - 1) / PMFMA_WG;            // dword loads per thread and tile
  static constexpr size_t lds_bytes = sizeof(float) * (W1 + W2 + W3 + 2 * B + 16) + 2 * (size_t)PMFMA_ENVS * K0 + 16;
};

// The three layers for the 32 envs of one wave, from the board bytes in `tile` to the four scores of env (lane & 31) in
// lanes 0..31. `between_layers()` runs after layer 1 (the single-launch kernel commits W2 / W3 to LDS there on its first pass).
template <int K0, int H, class Between>
__device__ __forceinline__ __attribute__((ext_vector_type(4))) float policy_forward(
    const float *lw1, const float *lw2, const float *lw3, const float *lb1, const float *lb2, const float *lb3,
    const int8_t *tile, int wave_env, int lane, Between between_layers) {
  typedef PolicyMfmaGeom<K0, H> G;
  constexpr int MT = G::MT, KS1 = G::KS1, NT = PMFMA_NT;
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int col = lane & 15, grp = lane >> 4;
  // ---- layer 1: h1^T = relu(W1 x^T + b1) --------------------------------------------------------------
  f4 h1[NT][MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const f4 bias = *reinterpret_cast<const f4 *>(lb1 + 16 * mt + 4 * grp);
#pragma unroll
    for (int e = 0; e < NT; ++e) h1[e][mt] = bias;
  }
#pragma unroll
  for (int s = 0; s < KS1; ++s) {
    float x[NT];
    const int k = 4 * s + grp;
#pragma unroll
    for (int e = 0; e < NT; ++e) x[e] = (k < K0) ? (float)tile[(wave_env + 16 * e + col) * K0 + k] : 0.0f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const float a = lw1[(mt * KS1 + s) * 64 + lane];
#pragma unroll
      for (int e = 0; e < NT; ++e) h1[e][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, x[e], h1[e][mt], 0, 0, 0);
    }
  }
#pragma unroll
  for (int e = 0; e < NT; ++e)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) h1[e][mt] = __builtin_elementwise_max(h1[e][mt], (f4){0.0f, 0.0f, 0.0f, 0.0f});
  between_layers();
  // ---- layer 2: h2^T = relu(W2 h1^T + b2); k-step (mk, r) reads register r of h1 tile mk --------------------
  f4 h2[NT][MT];
#pragma unroll
  for (int mo = 0; mo < MT; ++mo) {
    const f4 bias = *reinterpret_cast<const f4 *>(lb2 + 16 * mo + 4 * grp);
#pragma unroll
    for (int e = 0; e < NT; ++e) h2[e][mo] = bias;
  }
#pragma unroll
  for (int mk = 0; mk < MT; ++mk) {
    f4 a[MT];
#pragma unroll
    for (int mo = 0; mo < MT; ++mo) a[mo] = *reinterpret_cast<const f4 *>(lw2 + ((mo * MT + mk) * 64 + lane) * 4);
#pragma unroll
    for (int r = 0; r < G::ksteps(mk); ++r)
#pragma unroll
      for (int mo = 0; mo < MT; ++mo)
#pragma unroll
        for (int e = 0; e < NT; ++e)
          h2[e][mo] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mo][r], h1[e][mk][r], h2[e][mo], 0, 0, 0);
  }
#pragma unroll
  for (int e = 0; e < NT; ++e)
#pragma unroll
    for (int mo = 0; mo < MT; ++mo) h2[e][mo] = __builtin_elementwise_max(h2[e][mo], (f4){0.0f, 0.0f, 0.0f, 0.0f});
  // ---- layer 3: scores^T = W3 h2^T + b3; rows 0..3 of the tile = lanes 0..15, registers 0..3 ---------------
  f4 sc[NT];
  {
    const f4 bias = *reinterpret_cast<const f4 *>(lb3 + 4 * grp);
#pragma unroll
    for (int e = 0; e < NT; ++e) sc[e] = bias;
  }
#pragma unroll
  for (int mk = 0; mk < MT; ++mk) {
    const f4 a = *reinterpret_cast<const f4 *>(lw3 + (mk * 64 + lane) * 4);
#pragma unroll
    for (int r = 0; r < G::ksteps(mk); ++r)
#pragma unroll
      for (int e = 0; e < NT; ++e) sc[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], h2[e][mk][r], sc[e], 0, 0, 0);
  }
  // lanes 0..15 hold the four scores of env tile e; bring tile 1 to lanes 16..31 so that 32 lanes pick 32 actions at once
  static_assert(NT == 2, "the epilogue pairs two env tiles");
  f4 mine;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float other = __shfl(sc[1][r], lane - 16, 64);
    mine[r] = grp == 1 ? other : sc[0][r];
  }
  return mine;
}

template <int K0, int H, int MODE>
__global__ __launch_bounds__(PMFMA_WG) void policy_mfma_kernel(const int8_t *__restrict__ boards, int pitch,
                                                               const float *__restrict__ w1t, const float *__restrict__ b1,
                                                               const float *__restrict__ w2, const float *__restrict__ b2,
                                                               const float *__restrict__ w3t, const float *__restrict__ b3,
                                                               uint8_t *__restrict__ actions, float *__restrict__ scores_out,
                                                               int64_t n, double eps, uint64_t seed, uint64_t env_base,
                                                               uint64_t draw, const double *__restrict__ eps_ptr,
                                                               const uint64_t *__restrict__ draw_ptr) {
  typedef PolicyMfmaGeom<K0, H> G;
  constexpr int MT = G::MT, KS1 = G::KS1, NT = PMFMA_NT;
  typedef float f4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) unsigned char policy_smem[];
  float *lw1 = reinterpret_cast<float *>(policy_smem);
  float *lw2 = lw1 + G::W1;
  float *lw3 = lw2 + G::W2;
  float *lb1 = lw3 + G::W3;
  float *lb2 = lb1 + G::B;
  float *lb3 = lb2 + G::B;  // [16]: rows 0..3 = the action biases
  int8_t *tiles = reinterpret_cast<int8_t *>(lb3 + 16);  // 2 x [PMFMA_ENVS][K0]: the board tile in use and the next one
  // Board tiles travel global -> registers -> LDS: the loads of the first tile are issued before the weight staging (one
  // exposed memory round trip instead of two), those of tile i + 1 before the MFMAs of tile i (hidden entirely).
  const int64_t n_tiles = (n + PMFMA_ENVS - 1) / PMFMA_ENVS;
  const bool dense = pitch == K0;  // rows back to back (COMPACT layout): a tile is PMFMA_ENVS * K0 contiguous bytes
  uint32_t pre[G::TILE_LD];
  auto tile_fetch = [&](int64_t t) {
    // env0 * K0 is a multiple of 128 (dword aligned for every K0) and the boards allocation is padded to a multiple of 256
    // envs, so a whole tile can always be read; rows of envs >= n hold stale cells whose results are never stored
    const uint32_t *src = reinterpret_cast<const uint32_t *>(boards + t * PMFMA_ENVS * K0);
#pragma unroll
    for (int j = 0; j < G::TILE_LD; ++j) pre[j] = src[min(j * PMFMA_WG + (int)threadIdx.x, G::TILE_DW - 1)];
  };
  auto tile_commit = [&](int8_t *dst, int64_t t) {
    if (dense) {
#pragma unroll
      for (int j = 0; j < G::TILE_LD; ++j) {
        const int i = j * PMFMA_WG + threadIdx.x;
        if (i < G::TILE_DW) reinterpret_cast<uint32_t *>(dst)[i] = pre[j];
      }
    } else {  // padded rows (PITCHED layout): gathered byte by byte, not prefetched
      const int64_t env0 = t * PMFMA_ENVS;
      const int lim = (int)min((int64_t)PMFMA_ENVS, n - env0) * K0;
      for (int i = threadIdx.x; i < PMFMA_ENVS * K0; i += PMFMA_WG)
        dst[i] = i < lim ? boards[(env0 + i / K0) * pitch + i % K0] : (int8_t)0;
    }
  };
  if (dense && (int64_t)blockIdx.x < n_tiles) tile_fetch(blockIdx.x);
  // stage the weights in operand order (w1t = W1^T [K0][H], w2 = W2 [H][H], w3t = W3^T [H][4]; zero padding). Two phases,
  // every loop fully unrolled and branch-free (clamped address + select): first ALL global loads of the thread go out, then
  // the LDS stores. Written as load/store pairs the compiler waits out one L2 round trip per element (23 us per workgroup
  // with rolled loops, 4 us unrolled but paired -- measured); W2 moves as 16-byte rows of four consecutive k.
  static_assert(H % 4 == 0, "W2 rows are staged as float4");
  constexpr int N2 = (G::W2 / 4 + PMFMA_WG - 1) / PMFMA_WG, N1 = (G::W1 + PMFMA_WG - 1) / PMFMA_WG,
                N3 = (G::W3 + PMFMA_WG - 1) / PMFMA_WG;
  f4 r2[N2];
  float r1[N1], r3[N3];
  // what layer 1 needs goes out first (loads return in order): W1 and the biases; W2 / W3 follow and are only written to
  // LDS after the first tile's layer 1, so their round trip hides behind its MFMAs
#pragma unroll
  for (int it = 0; it < N1; ++it) {
    const int i = it * PMFMA_WG + threadIdx.x;  // (mt, s, lane)
    const int l = i & 63, s = (i >> 6) % KS1, mt = (i >> 6) / KS1;
    const int nrn = G::neuron(mt, l & 15), k = 4 * s + (l >> 4);
    const float v = w1t[min(k, K0 - 1) * H + max(nrn, 0)];
    r1[it] = (nrn >= 0 && mt < MT && k < K0) ? v : 0.0f;
  }
  const int bias_nrn = G::neuron((int)threadIdx.x >> 4, (int)threadIdx.x & 15);  // the neuron at padded position threadIdx.x
  const float rb1 = b1[max(bias_nrn, 0)], rb2 = b2[max(bias_nrn, 0)];
  const float rb3 = b3[threadIdx.x & 3];
#pragma unroll
  for (int it = 0; it < N2; ++it) {
    const int i = it * PMFMA_WG + threadIdx.x;  // (mo, mk, lane)
    const int l = i & 63, mk = (i >> 6) % MT, mo = (i >> 6) / MT;
    const int nrn = G::neuron(mo, l & 15), g = l >> 4;
    const float *wrow = w2 + max(nrn, 0) * H;
    // the four features (mk, 4 g + r): consecutive in a full tile; in the partial one the r < RK first are neurons, the rest padding
    f4 v = *reinterpret_cast<const f4 *>(wrow + min(16 * mk + 4 * g, H - 4));
    if (G::RK > 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float part = wrow[min(16 * G::FT + g * G::RK + min(r, G::RK - 1), H - 1)];
        if (mk >= G::FT) v[r] = r < G::RK ? part : 0.0f;
      }
    }
    r2[it] = (nrn >= 0 && mo < MT && mk < MT) ? v : (f4){0.0f, 0.0f, 0.0f, 0.0f};
  }
#pragma unroll
  for (int it = 0; it < N3; ++it) {
    const int i = it * PMFMA_WG + threadIdx.x;  // (mk, lane, r)
    const int r = i & 3, l = (i >> 2) & 63, mk = i >> 8;
    const int a = l & 15, k = G::neuron(mk, 4 * (l >> 4) + r);
    const float v = w3t[max(k, 0) * 4 + (a & 3)];
    r3[it] = (a < 4 && k >= 0 && mk < MT) ? v : 0.0f;
  }
#pragma unroll
  for (int it = 0; it < N1; ++it) {
    const int i = it * PMFMA_WG + threadIdx.x;
    if (i < G::W1) lw1[i] = r1[it];
  }
  static_assert(G::B <= PMFMA_WG, "one thread per padded bias entry");
  if (threadIdx.x < G::B) {
    lb1[threadIdx.x] = bias_nrn >= 0 ? rb1 : 0.0f;
    lb2[threadIdx.x] = bias_nrn >= 0 ? rb2 : 0.0f;
  }
  if (threadIdx.x < 16) lb3[threadIdx.x] = threadIdx.x < 4 ? rb3 : 0.0f;
  if (eps_ptr) eps = *eps_ptr;
  if (draw_ptr) draw = *draw_ptr;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if ((int64_t)blockIdx.x < n_tiles) tile_commit(tiles, blockIdx.x);
  __syncthreads();
  int buf = 0;
  for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const int64_t env0 = t * PMFMA_ENVS;
    const int8_t *tile = tiles + buf * (PMFMA_ENVS * K0);
    const int64_t t_next = t + gridDim.x;
    if (dense && t_next < n_tiles) tile_fetch(t_next);
    const int wave_env = wave * NT * 16;  // first env of this wave inside the workgroup tile
    const int64_t env = env0 + wave_env + (lane & 31);  // the env whose action this lane (of lanes 0..31) picks at the end
    double u;
    uint32_t x2;
    draw_block<MODE>(env_base + (uint64_t)env, draw, seed, u, x2);  // VALU work in the shadow of the MFMAs below
    const f4 mine = policy_forward<K0, H>(lw1, lw2, lw3, lb1, lb2, lb3, tile, wave_env, lane, [&]() {
      if (t == (int64_t)blockIdx.x) {  // first pass: W2 / W3 have arrived by now
#pragma unroll
        for (int it = 0; it < N2; ++it) {
          const int i = it * PMFMA_WG + threadIdx.x;
          if (i < G::W2 / 4) reinterpret_cast<f4 *>(lw2)[i] = r2[it];
        }
#pragma unroll
        for (int it = 0; it < N3; ++it) {
          const int i = it * PMFMA_WG + threadIdx.x;
          if (i < G::W3) lw3[i] = r3[it];
        }
        __syncthreads();
      }
    });
    if (lane < 32 && env < n) {
      actions[env] = (uint8_t)select_action<MODE>(mine[0], mine[1], mine[2], mine[3], u, x2, eps);
      if (scores_out) reinterpret_cast<float4 *>(scores_out)[env] = make_float4(mine[0], mine[1], mine[2], mine[3]);
    }
    if (t_next < n_tiles) tile_commit(tiles + (buf ^ 1) * (PMFMA_ENVS * K0), t_next);
    __syncthreads();  // the next tile is complete, and nobody still reads the one just used
    buf ^= 1;
  }
}

// ------------------------------------------------------------------------------------------------
// n_steps of {policy forward, action draw, env.step} in ONE launch: PPOBaseAgent.gather_rollout's inner loop (reference
// policy_base.py:142-163: old_policy.act_explore -> env.step -> store state / action / reward) or DeepQAgent acting with
// frozen weights (eval.py:33-36), for every env, with the network weights staged in LDS once and the env state in registers
// for the whole rollout. A wave owns 32 envs (lanes 0..31 hold their state words); their boards live as rows of the
// workgroup's LDS tile -- the MFMA forward reads them there -- and are kept current by re-drawing only the cells a step
// changed. Everything a wave touches in the loop is its own (its 32 tile rows, its lanes' registers), so the step loop has
// no workgroup barrier. Per step and wave: 526 MFMAs (H = 100), one Philox block, one table lookup, <= 4 LDS byte writes,
// and the optional trajectory stores (board rows as dwords, one action byte, one 4-byte record per env).
// ------------------------------------------------------------------------------------------------
struct RolloutArgs {
  StepArgs env;              // state / rec / episode arrays / metrics / rules / n / seed / env_base / flags
  PolicyWeights w;
  double eps;                // MODE 0
  uint64_t draw0;            // draw index of the first step; step k uses draw0 + k
  int32_t n_steps;
  int8_t *states_out;        // [n_steps][n][K0] boards the policy acted on, or null
  uint8_t *actions_out;      // [n_steps][n] or null
  uint32_t *recs_out;        // [n_steps][n] step records or null
};

template <int ENV, int H, int MODE>
__global__ __launch_bounds__(PMFMA_WG) void policy_rollout_kernel(RolloutArgs a) {
  constexpr int K0 = Geom<ENV>::NC;
  typedef PolicyMfmaGeom<K0, H> G;
  constexpr int MT = G::MT, KS1 = G::KS1;
  typedef float f4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) unsigned char policy_smem[];
  float *lw1 = reinterpret_cast<float *>(policy_smem);
  float *lw2 = lw1 + G::W1;
  float *lw3 = lw2 + G::W2;
  float *lb1 = lw3 + G::W3;
  float *lb2 = lb1 + G::B;
  float *lb3 = lb2 + G::B;
  int8_t *tile = reinterpret_cast<int8_t *>(lb3 + 16);                      // [PMFMA_ENVS][K0]
  SgkRules &R = *reinterpret_cast<SgkRules *>(tile + 2 * PMFMA_ENVS * K0);  // behind the (here unused) second tile buffer
  const PolicyWeights &w = a.w;
  // weights in operand order (same arrangement as policy_mfma_kernel; staged once per launch, so plainly)
  for (int i = threadIdx.x; i < G::W1; i += PMFMA_WG) {
    const int l = i & 63, s = (i >> 6) % KS1, mt = (i >> 6) / KS1;
    const int nrn = G::neuron(mt, l & 15), k = 4 * s + (l >> 4);
    lw1[i] = (nrn >= 0 && k < K0) ? w.w1t[k * H + nrn] : 0.0f;
  }
  for (int i = threadIdx.x; i < G::W2 / 4; i += PMFMA_WG) {
    const int l = i & 63, mk = (i >> 6) % MT, mo = (i >> 6) / MT;
    const int nrn = G::neuron(mo, l & 15);
    f4 v = {0.0f, 0.0f, 0.0f, 0.0f};
    if (nrn >= 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = G::neuron(mk, 4 * (l >> 4) + r);
        if (k >= 0) v[r] = w.w2[nrn * H + k];
      }
    }
    reinterpret_cast<f4 *>(lw2)[i] = v;
  }
  for (int i = threadIdx.x; i < G::W3; i += PMFMA_WG) {
    const int r = i & 3, l = (i >> 2) & 63, mk = i >> 8;
    const int ac = l & 15, k = G::neuron(mk, 4 * (l >> 4) + r);
    lw3[i] = (ac < 4 && k >= 0) ? w.w3t[k * 4 + ac] : 0.0f;
  }
  for (int i = threadIdx.x; i < G::B; i += PMFMA_WG) {
    const int nrn = G::neuron(i >> 4, i & 15);
    lb1[i] = nrn >= 0 ? w.b1[nrn] : 0.0f;
    lb2[i] = nrn >= 0 ? w.b2[nrn] : 0.0f;
  }
  if (threadIdx.x < 16) lb3[threadIdx.x] = threadIdx.x < 4 ? w.b3[threadIdx.x] : 0.0f;
  stage_rules(R, a.env.rules);  // ends with a workgroup barrier: weights and rules are in place

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wave_env = wave * PMFMA_NT * 16;
  const bool owner = lane < 32;  // lanes 32..63 only take part in the MFMAs and the row stores
  const int64_t n = a.env.n;
  const int64_t n_tiles = (n + PMFMA_ENVS - 1) / PMFMA_ENVS;
  EpisodeAcc acc;
  acc_init(acc);
  for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const int64_t env = t * PMFMA_ENVS + wave_env + (lane & 31);
    const bool valid = owner && env < n;
    EnvState s = initial_state(R);
    if (valid) s = unpack_state(a.env.state[env]);
    load_episode_index<ENV>(s, a.env.n_resets, env, valid);
    int8_t *row = tile + (wave_env + (lane & 31)) * K0;
    if (owner) write_row_bytes<ENV, K0>(R, row, s);  // draw this env's board from its state word
    uint32_t rec = 0;
    for (int k = 0; k < a.n_steps; ++k) {
      __builtin_amdgcn_wave_barrier();  // the rows written by lanes 0..31 are read by all 64 lanes below
      if (a.states_out) {  // the boards the policy is about to act on: this wave's 32 rows are contiguous in the output
        const int64_t first = t * PMFMA_ENVS + wave_env;
        const int rows = (int)max((int64_t)0, min((int64_t)32, n - first));
        int8_t *dst = a.states_out + ((int64_t)k * n + first) * K0;
        const int8_t *src = tile + wave_env * K0;
        // SGK_F_MASK_FINISHED: rows of envs whose episode is over (bit r = row r of this wave) are stored as zeros
        const uint32_t over = (a.env.flags & SGK_F_MASK_FINISHED) ? (uint32_t)__ballot(owner && s.over) : 0u;
        if ((reinterpret_cast<uintptr_t>(dst) & 3) == 0 && over == 0u) {
          for (int i = lane; i < rows * K0 / 4; i += 64) reinterpret_cast<uint32_t *>(dst)[i] = reinterpret_cast<const uint32_t *>(src)[i];
          for (int i = (rows * K0 / 4) * 4 + lane; i < rows * K0; i += 64) dst[i] = src[i];
        } else if ((reinterpret_cast<uintptr_t>(dst) & 3) == 0) {
          for (int i = lane; i < rows * K0 / 4; i += 64) {
            uint32_t v = reinterpret_cast<const uint32_t *>(src)[i];
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if ((over >> ((4 * i + j) / K0)) & 1u) v &= ~(0xffu << (8 * j));
            reinterpret_cast<uint32_t *>(dst)[i] = v;
          }
          for (int i = (rows * K0 / 4) * 4 + lane; i < rows * K0; i += 64) dst[i] = ((over >> (i / K0)) & 1u) ? (int8_t)0 : src[i];
        } else {
          for (int i = lane; i < rows * K0; i += 64) dst[i] = ((over >> (i / K0)) & 1u) ? (int8_t)0 : src[i];
        }
      }
      const bool was_over = (a.env.flags & SGK_F_MASK_FINISHED) && s.over;
      double u;
      uint32_t x2;
      draw_block<MODE>(a.env.env_base + (uint64_t)env, a.draw0 + (uint64_t)k, a.env.seed, u, x2);
      const f4 sc = policy_forward<K0, H>(lw1, lw2, lw3, lb1, lb2, lb3, tile, wave_env, lane, []() {});
      const int action = select_action<MODE>(sc[0], sc[1], sc[2], sc[3], u, x2, a.eps);
      const int old_pos = s.pos, old_box = s.box;
      const int old_alt = HasAltBackdrop<ENV>::value ? alt_backdrop<ENV>(R, s) : 0;
      step_one<ENV>(R, a.env, env, valid, action, s, rec, acc);
      if (valid) {
        if (a.actions_out) a.actions_out[(int64_t)k * n + env] = was_over ? (uint8_t)0 : (uint8_t)action;
        if (a.recs_out) a.recs_out[(int64_t)k * n + env] = rec;
      }
      const int new_alt = HasAltBackdrop<ENV>::value ? alt_backdrop<ENV>(R, s) : 0;
      if (owner && (HasMask<ENV>::value || (HasAltBackdrop<ENV>::value && new_alt != old_alt))) {
        // the other backdrop (an auto-reset flipped the supervisor's coin; the button was pressed; the agent stepped on or off the
        // bucket) or a level whose cells change by themselves (tomatoes dry): the whole row
        write_row_bytes<ENV, K0>(R, row, s);
      } else if (owner && (s.pos != old_pos || s.box != old_box)) {  // re-draw the cells this step changed (a reset included)
        const uint8_t *backdrop = backdrop_of(R, new_alt);
        row[old_pos] = (int8_t)backdrop[old_pos];
        if (HasSprite2<ENV>::value) {
          if (old_box < K0) row[old_box] = (int8_t)backdrop[old_box];
          if (s.box < K0) row[s.box] = (int8_t)sprite2_value<ENV>(R, s);
        }
        row[s.pos] = (int8_t)R.agent_value[s.pos];
      }
    }
    if (valid) {
      a.env.state[env] = pack_state(s);
      a.env.rec[env] = rec;  // the env's own boards are re-materialised by the caller (launch_reset mode 2)
    }
    __builtin_amdgcn_wave_barrier();
  }
  acc_flush(acc, a.env.metrics);
}

template <int ENV, int H>
constexpr size_t policy_rollout_lds_bytes() {
  return PolicyMfmaGeom<Geom<ENV>::NC, H>::lds_bytes + sizeof(SgkRules) + 16;
}

// PPOBaseAgent.get_discounted_returns (reference policy_base.py:179-186) for a batch of trajectories.
// The reference is an O(T^2) Python double loop per trajectory; its float32 rounding order is kept exactly:
//   d[t] = float32(discount ** t) * r[t];   returns[t] = ((d[t] + d[t+1]) + d[t+2]) + ...   (Python sum(): left to right)
// One wave per trajectory: the wave stages d[] in LDS, then lane t accumulates its own suffix serially -- consecutive
// lanes read consecutive LDS words at every iteration (conflict-free). gamma_pow[t] is computed on the host in double.
constexpr int RET_TMAX = 1024;
__global__ __launch_bounds__(WG) void discounted_returns_kernel(const float *__restrict__ rewards,
                                                                const int32_t *__restrict__ lengths,
                                                                const float *__restrict__ gamma_pow,
                                                                float *__restrict__ returns, int64_t n, int t_max) {
  __shared__ float d[WG / 64][RET_TMAX];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int64_t traj = (int64_t)blockIdx.x * (WG / 64) + wave; traj < n; traj += (int64_t)gridDim.x * (WG / 64)) {
    const int len = lengths ? min(lengths[traj], t_max) : t_max;
    const float *r = rewards + traj * t_max;
    for (int t = lane; t < len; t += 64) d[wave][t] = __fmul_rn(gamma_pow[t], r[t]);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's LDS writes have landed
    for (int t0 = 0; t0 < len; t0 += 64) {
      const int t = t0 + lane;
      float acc = 0.0f;
      if (t < len) {
        acc = d[wave][t];
        for (int k = t + 1; k < len; ++k) acc = __fadd_rn(acc, d[wave][k]);
        returns[traj * t_max + t] = acc;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ------------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------------
hipError_t launch_eps_greedy(const Shard &sh, int mode, const float *scores, uint8_t *actions, double eps, uint64_t draw,
                             const double *eps_dev, const uint64_t *draw_dev, hipStream_t st) {
  (void)hipGetLastError();
  int grid = grid_for((sh.n + WG - 1) / WG, sh.max_grid);
  const float4 *sc = reinterpret_cast<const float4 *>(scores);
  if (mode == 0)
    eps_greedy_kernel<0><<<dim3(grid), dim3(WG), 0, st>>>(sc, actions, sh.n, eps, sh.seed, sh.env_base, draw, eps_dev, draw_dev);
  else
    eps_greedy_kernel<1><<<dim3(grid), dim3(WG), 0, st>>>(sc, actions, sh.n, eps, sh.seed, sh.env_base, draw, eps_dev, draw_dev);
  return hipGetLastError();
}

hipError_t launch_policy_act(const Shard &sh, int mode, const PolicyWeights &w, uint8_t *actions, float *scores, double eps,
                             uint64_t draw, const double *eps_dev, const uint64_t *draw_dev, hipStream_t st) {
  (void)hipGetLastError();
  int grid = grid_for((sh.n + PMFMA_ENVS - 1) / PMFMA_ENVS, sh.n_cus);
#define SGK_POLICY_LAUNCH_M(K0, HID, MODE)                                                                                 \
  do {                                                                                                                     \
    constexpr size_t lds = PolicyMfmaGeom<K0, HID>::lds_bytes;                                                             \
    static std::atomic<unsigned long long> opted_in{0}; /* > 64 KB of dynamic LDS needs the opt-in, once per kernel and device */      \
    if (!((opted_in.load() >> (sh.device & 63)) & 1ull)) {                                                                        \
      hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void *>(&policy_mfma_kernel<K0, HID, MODE>),              \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                           \
      if (ae != hipSuccess) return ae;                                                                                     \
      opted_in.fetch_or(1ull << (sh.device & 63));                                                                         \
    }                                                                                                                      \
    policy_mfma_kernel<K0, HID, MODE><<<dim3(grid), dim3(PMFMA_WG), lds, st>>>(sh.boards, sh.pitch, w.w1t, w.b1, w.w2, w.b2, \
                                                                              w.w3t, w.b3, actions, scores, sh.n, eps,     \
                                                                              sh.seed, sh.env_base, draw, eps_dev,         \
                                                                              draw_dev);                                   \
  } while (0)
#define SGK_POLICY_LAUNCH_H(K0, HID)                                                                                       \
  do {                                                                                                                     \
    if (mode == 0) SGK_POLICY_LAUNCH_M(K0, HID, 0);                                                                        \
    else SGK_POLICY_LAUNCH_M(K0, HID, 1);                                                                                  \
  } while (0)
  // hidden widths with an instantiation: the reference default (100) and the two common powers of two
#define SGK_POLICY_LAUNCH(K0)                                                                                              \
  do {                                                                                                                     \
    switch (w.n_hidden) {                                                                                                  \
    case 64: SGK_POLICY_LAUNCH_H(K0, 64); break;                                                                           \
    case 100: SGK_POLICY_LAUNCH_H(K0, 100); break;                                                                         \
    case 128: SGK_POLICY_LAUNCH_H(K0, 128); break;                                                                         \
    default: return hipErrorInvalidValue;                                                                                  \
    }                                                                                                                      \
  } while (0)
  switch (sh.n_cells) {
  case 25: SGK_POLICY_LAUNCH(25); break;
  case 30: SGK_POLICY_LAUNCH(30); break;
  case 36: SGK_POLICY_LAUNCH(36); break;
  case 48: SGK_POLICY_LAUNCH(48); break;
  case 49: SGK_POLICY_LAUNCH(49); break;
  case 56: SGK_POLICY_LAUNCH(56); break;
  case 63: SGK_POLICY_LAUNCH(63); break;
  default: return hipErrorInvalidValue;
  }
#undef SGK_POLICY_LAUNCH
#undef SGK_POLICY_LAUNCH_H
#undef SGK_POLICY_LAUNCH_M
  return hipGetLastError();
}

hipError_t launch_policy_rollout(const Shard &sh, int mode, const PolicyWeights &w, double eps, uint64_t draw0, int32_t n_steps,
                                 uint32_t flags, int8_t *states_out, uint8_t *actions_out, uint32_t *recs_out, hipStream_t st) {
  (void)hipGetLastError();
  RolloutArgs a;
  a.env = make_step_args(sh, nullptr, flags);
  a.w = w;
  a.eps = eps;
  a.draw0 = draw0;
  a.n_steps = n_steps;
  a.states_out = states_out;
  a.actions_out = actions_out;
  a.recs_out = recs_out;
  int grid = grid_for((sh.n + PMFMA_ENVS - 1) / PMFMA_ENVS, sh.n_cus);
#define SGK_ROLLOUT_LAUNCH_M(E, HID, MODE)                                                                                 \
  do {                                                                                                                     \
    constexpr size_t lds = policy_rollout_lds_bytes<E, HID>();                                                             \
    static std::atomic<unsigned long long> opted_in{0};                                                                                \
    if (!((opted_in.load() >> (sh.device & 63)) & 1ull)) {                                                                        \
      hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void *>(&policy_rollout_kernel<E, HID, MODE>),            \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                           \
      if (ae != hipSuccess) return ae;                                                                                     \
      opted_in.fetch_or(1ull << (sh.device & 63));                                                                         \
    }                                                                                                                      \
    policy_rollout_kernel<E, HID, MODE><<<dim3(grid), dim3(PMFMA_WG), lds, st>>>(a);                                       \
  } while (0)
#define SGK_ROLLOUT_LAUNCH_H(E, HID)                                                                                       \
  do {                                                                                                                     \
    if (mode == 0) SGK_ROLLOUT_LAUNCH_M(E, HID, 0);                                                                        \
    else SGK_ROLLOUT_LAUNCH_M(E, HID, 1);                                                                                  \
  } while (0)
  SGK_DISPATCH_ENV(sh.env_id, {
    switch (w.n_hidden) {
    case 64: SGK_ROLLOUT_LAUNCH_H(E, 64); break;
    case 100: SGK_ROLLOUT_LAUNCH_H(E, 100); break;
    case 128: SGK_ROLLOUT_LAUNCH_H(E, 128); break;
    default: return hipErrorInvalidValue;
    }
  });
#undef SGK_ROLLOUT_LAUNCH_H
#undef SGK_ROLLOUT_LAUNCH_M
  return hipGetLastError();
}

hipError_t launch_discounted_returns(const Shard &sh, const float *rewards, const int32_t *lengths, const float *gamma_pow,
                                     float *returns, int64_t n, int t_max, hipStream_t st) {
  (void)hipGetLastError();  // drop a stale error another HIP user of this thread may have left
  int grid = grid_for((n + (WG / 64) - 1) / (WG / 64), sh.max_grid * 2);
  hipLaunchKernelGGL(discounted_returns_kernel, dim3(grid), dim3(WG), 0, st, rewards, lengths, gamma_pow, returns, n, t_max);
  return hipGetLastError();
}

}  // namespace sgk
