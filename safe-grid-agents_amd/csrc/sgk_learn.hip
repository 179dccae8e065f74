// sgk_learn.hip -- DeepQAgent.learn (reference value.py:113-136) for the reference's default topology as ONE kernel:
// sample a minibatch from the device replay ring, forward the Q-network and the target network, the TD target and the MSE
// loss, the backward pass, clip_grad_norm_(10) and the Adam(amsgrad) update -- what PyTorch runs as ~80 small kernels
// (a 64-sample minibatch through a 36-100-100-4 MLP: every kernel is launch latency). One workgroup of 1 024 lanes; the
// minibatch, both networks' activations and the two back-propagated activations live in LDS, every lane keeps the
// gradients of the parameters it owns in registers, and the weights are read from L2 in the orientation that makes the
// lanes' addresses consecutive (a transposed copy of W1 / W2 / W3 is kept current by the update itself; the fused policy
// kernels read W1^T and W3^T from it).
//
// fp32 throughout, same formulas as torch (mse_loss over the [B,1]-vs-[B] broadcast of value.py:119-123 by default -- loss_mode,
// see the kernel; clip coefficient max_norm / (norm + 1e-6) clamped to 1, Adam with bias corrections and amsgrad); the summation
// order inside the dot products differs from rocBLAS / torch's CPU kernels, so parity is to fp32 tolerance, not bit for bit: the
// reference's own DeepQAgent runs (tests/golden/batched_dqn_*.npz, 320-330 steps across three target syncs) are reproduced with
// every action equal, losses and weights within rtol 2e-4 (tests/test_gpu_batched_golden.py).
#include <algorithm>
#include <atomic>

#include "sgk_device.h"
#include "sgk_step_store.h"

extern "C" __device__ float __ockl_wfred_add_f32(float);

namespace sgk {

// -DSGK_LEARN_TIMELINE (tools/gpu_dqn_timeline.sh builds it into a library of its own, never the product's): lane 0 stamps
// wall_clock64() (100 MHz) behind every barrier of dqn_sgd_kernel; sgk_debug_learn_stamps() reads them back.
#ifdef SGK_LEARN_TIMELINE
__device__ unsigned long long learn_stamps[32];
// (stamps 30 / 31: the shader clock counter, s_memtime, at the first and at the latest stamp: in-kernel clock = their difference over
// the wall-clock difference x 100 MHz)
#define SGK_STAMP(i) do { if (threadIdx.x == 0) { learn_stamps[i] = wall_clock64(); learn_stamps[(i) == 0 ? 30 : 31] = __builtin_amdgcn_s_memtime(); } } while (0)
#define SGK_MSTAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) { learn_stamps[i] = wall_clock64(); learn_stamps[(i) == 0 ? 30 : 31] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define SGK_STAMP(i) do { } while (0)
#define SGK_MSTAMP(i) do { } while (0)
#endif

constexpr int LWG = 1024;    // lanes of the one workgroup
constexpr int LB = 64;       // samples per minibatch handled (batch <= 64)

struct LearnArgs {
  // replay ring [slices][n_envs][...]
  const int8_t *states, *successors;
  const uint8_t *actions;
  const int8_t *rewards;
  const uint8_t *terminals;
  int64_t n_envs, total;  // sampling range: total = filled slices * n_envs
  int32_t n_cells;
  // Q-network (torch layouts [out][in]) updated in place; transposed copies kept current; Adam state per tensor
  float *w1, *b1, *w2, *b2, *w3, *b3;
  float *w1t, *w2t, *w3t;
  float *m[6], *v[6], *vmax[6];  // order: w1, b1, w2, b2, w3, b3
  // target network: transposed hidden weights, output weights as they are [4][H]
  const float *tw1t, *tb1, *tw2t, *tb2, *tw3, *tb3;
  long long *step;   // Adam step counter on the device
  float *loss_out;   // may be null
  const long long *rows;  // the caller's minibatch (transition indices into the ring), or null: drawn here
  long long *rows_out;    // the minibatch used, or null
  float *adam_scratch;    // null: Adam inside dqn_sgd_kernel; else AdamHeader + the flat gradient go here and dqn_adam_kernel follows
  int32_t n_hidden, batch;
  int32_t loss_mode;  // 0 = the reference's [B,1]-vs-[B] broadcast mse_loss (value.py:119-123), 1 = per-sample (squeezed) mse_loss
  uint64_t seed;  // minibatch indices: Philox stream 4, ctr = {sample, 0, Adam step before this update, 4}
  float lr, beta1, beta2, eps, discount, max_norm;
  double reward_scale;  // what one unit of the int8 rewards is worth (SgkRules.reward_scale; 1 except tomato watering): the
                        // reference's reward is the float64 product, rounded to float32 when the batch tensor is made (value.py:170-171)
};

typedef float f4 __attribute__((ext_vector_type(4)));

// Cooperative global -> LDS copy of n floats (n % 4 == 0, both 16-byte aligned): every lane's loads are in flight together,
// ONE memory round trip for the whole block. The kernel is a chain of dependent phases run by a single workgroup, so what
// it costs is (number of exposed round trips) x (latency): reading the weights straight from L2 inside the dot-product
// loops was 80 round trips = 85-180 us; staged like this the kernel has about ten.
__device__ __forceinline__ void stage(float *dst, const float *__restrict__ src, int n) {
  for (int i = threadIdx.x * 4; i < n; i += LWG * 4) *reinterpret_cast<f4 *>(dst + i) = *reinterpret_cast<const f4 *>(src + i);
}

// Row stride of a staged weight matrix [rows][H]: H padded so that stride % 32 == 8. The MFMA A operand of dense_layer is
// wt[(4 u + grp) * stride + n0 + col] over the lanes (col = 0..15, grp = 0..3): with stride 100 the four groups start 4 banks
// apart and overlap (up to 4 lanes per bank: four LDS passes), with stride % 32 == 8 they tile the 32 banks exactly (two
// lanes per bank, the minimum for 64 lanes).
constexpr int weight_stride(int h) { return h + (40 - h % 32) % 32; }

// stage() for a [rows][H] matrix into rows of weight_stride(H) floats
template <int H>
__device__ __forceinline__ void stage_rows(float *dst, const float *__restrict__ src, int rows) {
  constexpr int WS = weight_stride(H);
  for (int i = threadIdx.x * 4; i < rows * H; i += LWG * 4) {
    const int r = i / H, c = i - r * H;
    *reinterpret_cast<f4 *>(dst + r * WS + c) = *reinterpret_cast<const f4 *>(src + i);
  }
}

// stage_rows() in two halves: the loads into registers BEFORE a compute phase, the LDS stores behind it -- the matrix's trip from
// L2 runs in the shadow of the phase instead of in front of the next one (one staging buffer, re-filled per layer: the
// activations of 64 samples leave LDS no room for a second)
template <int H, int ROWS>
struct RowsInFlight {
  static constexpr int N = (ROWS * H / 4 + LWG - 1) / LWG;
  f4 v[N];
};
template <int H, int ROWS>
__device__ __forceinline__ void rows_request(RowsInFlight<H, ROWS> &r, const float *__restrict__ src) {
#pragma unroll
  for (int n = 0; n < RowsInFlight<H, ROWS>::N; ++n) {
    const int i = ((int)threadIdx.x + n * LWG) * 4;
    r.v[n] = (f4){0.0f, 0.0f, 0.0f, 0.0f};
    if (i < ROWS * H) r.v[n] = *reinterpret_cast<const f4 *>(src + i);
  }
}
template <int H, int ROWS>
__device__ __forceinline__ void rows_commit(float *dst, const RowsInFlight<H, ROWS> &r) {
  constexpr int WS = weight_stride(H);
#pragma unroll
  for (int n = 0; n < RowsInFlight<H, ROWS>::N; ++n) {
    const int i = ((int)threadIdx.x + n * LWG) * 4;
    if (i < ROWS * H) {
      const int rr = i / H, c = i - rr * H;
      *reinterpret_cast<f4 *>(dst + rr * WS + c) = r.v[n];
    }
  }
}

__device__ __forceinline__ f4 load_x4(const float *row, int k) { return *reinterpret_cast<const f4 *>(row + k); }
__device__ __forceinline__ f4 load_x4(const int8_t *row, int k) {  // four board cells -> four floats
  const uint32_t w = *reinterpret_cast<const uint32_t *>(row + k);
  return (f4){(float)(int8_t)(w & 0xff), (float)(int8_t)((w >> 8) & 0xff), (float)(int8_t)((w >> 16) & 0xff), (float)(int8_t)(w >> 24)};
}

__device__ __forceinline__ float load_x1(const float *row, int k) { return row[k]; }
__device__ __forceinline__ float load_x1(const int8_t *row, int k) { return (float)row[k]; }

// out[b][n] = epilogue(bias[n] + sum_k in[b][k] * wt[k][n]) for the LB samples and n < H, everything in LDS, on the matrix
// cores (v_mfma_f32_16x16x4_f32: exact fp32): out^T = W in^T in 16 x 16 tiles, A operand = 16 neurons x 4 k of wt (lanes
// along n: consecutive words), B operand = 4 k x 16 samples of `in`, C = 16 neurons x 16 samples. Two 4-byte LDS reads feed
// 1 024 FMAs; the VALU form before it (4 neurons x 2 samples per lane) moved 3 bytes of LDS per FMA and spent 8.6 us per
// 100 x 100 layer on LDS bandwidth alone. The ceil(H / 16) x 4 output tiles are dealt round-robin to the 16 waves.
// KP = row stride of `in` in elements; `mask` zeroes out[b][n] where mask[b][n] <= 0 (ReLU'); neuron rows >= H of the last
// tile are computed from whatever lies behind the row (still inside LDS) and dropped.
template <int K, int H, int WST = H, class T>
__device__ __forceinline__ void dense_layer(const T *in, int KP, const float *wt, const float *bias, float *out, bool relu,
                                            const float *mask) {  // WST = row stride of wt
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
  const int col = lane & 15, grp = lane >> 4;
  constexpr int MT = (H + 15) / 16, KS = (K + 3) / 4;  // neuron tiles, k-steps
  constexpr int CH = KS < 8 ? KS : 8;                    // k-steps whose operands are fetched together
  // (two tiles per wave in flight -- two independent accumulator chains behind one batch of 32 LDS reads -- was slower:
  // 41.5 -> 44.5 us for the DeepQ step, 653 -> 727 us for 16 PPO epochs)
  for (int tile = wave; tile < MT * (LB / 16); tile += n_waves) {
    const int n0 = 16 * (tile % MT), b0 = 16 * (tile / MT);
    f4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    if (bias && n0 + 4 * grp < H) acc = *reinterpret_cast<const f4 *>(bias + n0 + 4 * grp);
    const T *xrow = in + (b0 + col) * KP;
    const float *wcol = wt + n0 + col;
    // compile-time shapes: the loop nest unrolls completely and the operand reads of a chunk are issued before its MFMAs
    // (with a runtime K the compiler waited out every LDS read in front of the MFMA using it: 15 k cycles per 100 x 100
    // layer against an issue bound of 6.4 k)
#pragma unroll
    for (int c0 = 0; c0 < KS; c0 += CH) {
      float av[CH], bv[CH];
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        const int k = 4 * (c0 + u) + grp;
        if (4 * (c0 + u) + 3 < K) {  // whole k-step inside K: no guard
          av[u] = wcol[k * WST];
          bv[u] = load_x1(xrow, k);
        } else if (4 * (c0 + u) < K) {  // the ragged last k-step
          const int kc = k < K ? k : K - 1;
          av[u] = k < K ? wcol[kc * WST] : 0.0f;
          bv[u] = k < K ? load_x1(xrow, kc) : 0.0f;
        } else {
          av[u] = 0.0f;
          bv[u] = 0.0f;
        }
      }
#pragma unroll
      for (int u = 0; u < CH; ++u)
        if (4 * (c0 + u) < K) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
    }
    const int n = n0 + 4 * grp;  // this lane's four neurons, sample b0 + col
    if (n < H) {
      if (relu) acc = __builtin_elementwise_max(acc, (f4){0.0f, 0.0f, 0.0f, 0.0f});
      if (mask) {
        const f4 m = *reinterpret_cast<const f4 *>(mask + (b0 + col) * H + n);
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = m[c] > 0.0f ? acc[c] : 0.0f;
      }
      *reinterpret_cast<f4 *>(out + (b0 + col) * H + n) = acc;
    }
  }
}

// q[b][a] = b3[a] + sum_k h[b][k] * w3[a][k] (all LDS): one lane per (b, a)
__device__ __forceinline__ void head_forward(const float *h, int H, const float *w3, const float *b3, float *q) {
  if (threadIdx.x < LB * 4) {
    const int b = threadIdx.x >> 2, a = threadIdx.x & 3;
    float acc = b3[a];
    for (int k0 = 0; k0 < H; k0 += 4) {  // H % 4 == 0 (checked by the launcher)
      const f4 w = *reinterpret_cast<const f4 *>(w3 + a * H + k0);
      const f4 x = *reinterpret_cast<const f4 *>(h + b * H + k0);
      acc = fmaf(x[0], w[0], acc);
      acc = fmaf(x[1], w[1], acc);
      acc = fmaf(x[2], w[2], acc);
      acc = fmaf(x[3], w[3], acc);
    }
    q[b * 4 + a] = acc;
  }
}

// grad[r][c] = sum_b d[b][row0 + r] * x[b][col0 + c] for a 4 x 4 tile (the weight gradient of a dense layer): two LDS reads
// per sample for 16 FMAs
template <class T>
__device__ __forceinline__ void weight_grad_tile(const float *d, int DS, int row0, const T *x, int XS, int col0, f4 g[4]) {
#pragma unroll
  for (int r = 0; r < 4; ++r) g[r] = (f4){0.0f, 0.0f, 0.0f, 0.0f};
  for (int b = 0; b < LB; ++b) {
    const f4 dv = *reinterpret_cast<const f4 *>(d + b * DS + row0);
    const f4 xv = load_x4(x + b * XS, col0);
#pragma unroll
    for (int r = 0; r < 4; ++r) g[r] = __builtin_elementwise_fma((f4){dv[r], dv[r], dv[r], dv[r]}, xv, g[r]);
  }
}

// One 16 x 16 tile of a weight gradient on the matrix cores: G[r][c] = sum_b d[b][row0 + r] * x[b][col0 + c] over the LB samples
// (K = the batch). A operand = d[b][row0 + (lane & 15)], B operand = x[b][col0 + (lane & 15)], b = 4 s + (lane >> 4): both
// are consecutive words across the lanes. Result in the C layout: lane (col, grp), register r = G[4 grp + r][col].
template <class T>
__device__ __forceinline__ f4 weight_grad_mfma(const float *d, int DS, int row0, const T *x, int XS, int col0, int lane) {
  const int col = lane & 15, grp = lane >> 4;
  f4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
  float av[LB / 4], bv[LB / 4];
#pragma unroll
  for (int s = 0; s < LB / 4; ++s) {
    av[s] = d[(4 * s + grp) * DS + row0 + col];
    bv[s] = load_x1(x + (4 * s + grp) * XS, col0 + col);
  }
#pragma unroll
  for (int s = 0; s < LB / 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s], acc, 0, 0, 0);
  return acc;
}

// The same tile with the operands exchanged: the result comes out TRANSPOSED in the C layout -- lane (col, grp), register r =
// G[row0 + col][col0 + 4 grp + r]: four CONSECUTIVE columns of one row, i.e. 16 contiguous bytes of a row-major [rows][cols] tensor.
// The owner's Adam traffic on w, m, v and vmax is then four 16-byte loads and four 16-byte stores per quad instead of sixteen and
// sixteen dword ones (one CU's memory pipeline was what bounded that phase: 13 us for 13.9 k parameters).
template <class T>
__device__ __forceinline__ f4 weight_grad_mfma_t(const float *d, int DS, int row0, const T *x, int XS, int col0, int lane) {
  const int col = lane & 15, grp = lane >> 4;
  f4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
  float av[LB / 4], bv[LB / 4];
#pragma unroll
  for (int s = 0; s < LB / 4; ++s) {
    av[s] = load_x1(x + (4 * s + grp) * XS, col0 + col);
    bv[s] = d[(4 * s + grp) * DS + row0 + col];
  }
#pragma unroll
  for (int s = 0; s < LB / 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s], acc, 0, 0, 0);
  return acc;
}

// Column sums on the matrix cores (a bias gradient): S[r] = sum_b d[b][row0 + r] over the LB samples -- the same tile product
// with a B operand of ones. Lane (col, grp), register r = S[4 grp + r], the same in every column. A 64-iteration serial loop
// per bias lane (64 dependent LDS reads, two waves busy, fourteen idle) cost 2-3 us per layer.
__device__ __forceinline__ f4 column_sum_mfma(const float *d, int DS, int row0, int lane) {
  const int col = lane & 15, grp = lane >> 4;
  f4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
  float av[LB / 4];
#pragma unroll
  for (int s = 0; s < LB / 4; ++s) av[s] = d[(4 * s + grp) * DS + row0 + col];
#pragma unroll
  for (int s = 0; s < LB / 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], 1.0f, acc, 0, 0, 0);
  return acc;
}

__device__ __forceinline__ float block_sum(float x, float *scratch) {  // scratch: LWG / 64 floats
  x = __ockl_wfred_add_f32(x);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = x;
  __syncthreads();
  float total = 0.0f;
  for (int w = 0; w < LWG / 64; ++w) total += scratch[w];
  return total;
}

struct AdamCoef {
  float lr_bc1, inv_bc2_sqrt, beta1, beta2, eps;  // lr / bias_correction1, 1 / sqrt(bias_correction2)
};

// One element of Adam(amsgrad), torch's formulas: exp_avg.lerp_(grad, 1 - beta1); exp_avg_sq = beta2 * exp_avg_sq + (1 - beta2) grad^2;
// denom = sqrt(max_exp_avg_sq) / sqrt(bias_correction2) + eps; param -= lr / bias_correction1 * exp_avg / denom.
// The square root and the two quotients use the hardware's one-ulp forms (v_sqrt_f32, v_rcp_f32; 1 / sqrt(bias_correction2) is
// formed once per launch): with IEEE sqrtf() and `/` the phase was VALU-bound -- ~60 instructions per element, 28 element slots per
// lane, four waves per SIMD at four cycles per wave64 instruction = 11 of the 13 us the timeline showed -- and a relative 1e-7 in
// an update that is itself ~1e-3 of the weight is far inside what the float32 summation order already differs by.
__device__ __forceinline__ float adam_scalar(float p, float &m, float &v, float &vmax, float g, const AdamCoef &c) {
  m = m + (1.0f - c.beta1) * (g - m);
  v = c.beta2 * v + (1.0f - c.beta2) * g * g;
  vmax = fmaxf(vmax, v);
  const float denom = __builtin_amdgcn_sqrtf(vmax) * c.inv_bc2_sqrt + c.eps;
  return p - c.lr_bc1 * (m * __builtin_amdgcn_rcpf(denom));
}

// Adam on four consecutive parameters (16-byte aligned); returns the updated values
__device__ __forceinline__ f4 adam_row(float *p, float *m, float *v, float *vmax, f4 g, const AdamCoef &c) {
  f4 pv = *reinterpret_cast<f4 *>(p), mv = *reinterpret_cast<f4 *>(m), vv = *reinterpret_cast<f4 *>(v), xv = *reinterpret_cast<f4 *>(vmax);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float mi = mv[i], vi = vv[i], xi = xv[i];
    pv[i] = adam_scalar(pv[i], mi, vi, xi, g[i], c);
    mv[i] = mi; vv[i] = vi; xv[i] = xi;
  }
  *reinterpret_cast<f4 *>(p) = pv;
  *reinterpret_cast<f4 *>(m) = mv;
  *reinterpret_cast<f4 *>(v) = vv;
  *reinterpret_cast<f4 *>(vmax) = xv;
  return pv;
}

// LDS plan (floats unless noted): A, Bq, C, D [LB][H] -- A = Q hidden 1, Bq = Q hidden 2, C = target hidden 1 then dL/dh2,
// D = target hidden 2 then dL/dh1; ST [H][H] the weight matrix of the running phase; SM the small tensors (W3, target W3, the
// biases); S, S2 int8 [LB][KP] the boards; the per-sample scalars.
struct LearnLds {
  float *A, *Bq, *C, *D, *ST, *w3, *tw3, *b1, *b2, *b3, *tb1, *tb2, *tb3, *q, *tq, *y, *scratch, *rew;
  int8_t *S, *S2;
  int *idx, *act, *term;
};

__device__ __forceinline__ LearnLds carve(unsigned char *base, int KP, int H) {
  LearnLds L;
  float *f = reinterpret_cast<float *>(base);
  L.A = f; f += LB * H;
  L.Bq = f; f += LB * H;
  L.C = f; f += LB * H;
  L.D = f; f += LB * H;
  L.ST = f; f += H * weight_stride(H);
  L.w3 = f; f += 4 * H;
  L.tw3 = f; f += 4 * H;
  L.b1 = f; f += H;
  L.b2 = f; f += H;
  L.tb1 = f; f += H;
  L.tb2 = f; f += H;
  L.b3 = f; f += 4;
  L.tb3 = f; f += 4;
  L.q = f; f += LB * 4;
  L.tq = f; f += LB * 4;
  L.y = f; f += LB;
  L.scratch = f; f += 32;
  L.rew = f; f += LB;
  L.idx = reinterpret_cast<int *>(f); f += LB;
  L.act = reinterpret_cast<int *>(f); f += LB;
  L.term = reinterpret_cast<int *>(f); f += LB;
  L.S = reinterpret_cast<int8_t *>(f);
  L.S2 = L.S + LB * KP;
  return L;
}

// flat parameter vector in torch's registration order: w1 [H][K0], b1 [H], w2 [H][H], b2 [H], w3 [4][H], b3 [4]
template <int K0, int H>
struct ParamMap {
  static constexpr int o_w1 = 0, o_b1 = H * K0, o_w2 = o_b1 + H, o_b2 = o_w2 + H * H, o_w3 = o_b2 + H, o_b3 = o_w3 + 4 * H, P = o_b3 + 4;
};

// Two launches instead of one (round 6): the SGD step's last phase -- Adam on 13.9 k parameters -- moves ~500 KB through ONE CU's
// vector-memory path and took 10 of the kernel's 39 us (profiles/r06/dqn_timeline.log). With a.adam_scratch set, dqn_sgd_kernel stops
// behind the norm: it stores the gradient (flat, torch's parameter order, 16 bytes per lane where the MFMA layout gives four
// consecutive elements) and {clip coefficient, lr / bias_correction1, 1 / sqrt(bias_correction2)} to the handle's scratch block, and
// dqn_adam_kernel -- one lane per parameter, every CU, coalesced -- applies the update behind the kernel boundary (a boundary costs
// ~1.5 us; a grid barrier inside one launch 6-8: the experiment further down).
struct AdamHeader {
  float coef, lr_bc1, inv_bc2_sqrt, pad;
};

template <int K0, int H>
__global__ __launch_bounds__(LWG) void dqn_sgd_kernel(LearnArgs a) {
  SGK_STAMP(0);
  extern __shared__ __attribute__((aligned(16))) unsigned char learn_smem[];
  const int B = a.batch;
  constexpr int KP = (K0 + 3) & ~3;  // row stride of the board matrices (padding columns hold zeros)
  const LearnLds L = carve(learn_smem, KP, H);
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6, col = lane & 15, grp = lane >> 4;
  constexpr int MT = (H + 15) / 16, KT1 = (K0 + 15) / 16;  // 16-wide tiles along neurons / along board cells

  // ---- entry: ONE dependent chain of two memory round trips (the Adam step counter, then everything keyed by it) ----
  // minibatch: uniform with replacement over the stored transitions (contain.py:19-22), counter RNG. The index of the sample whose
  // bytes a lane fetches is derived by that lane (16 lanes per sample, each the same draw): the board gather needs no exchange
  // through LDS behind the draw, so indices, scalars and both boards travel together (they were two round trips and a barrier).
  const long long step0 = *a.step;
  // the first weight matrix and the small tensors are independent of it: requested first, so that they are in flight meanwhile
  RowsInFlight<H, K0> first;
  rows_request<H, K0>(first, a.tw1t);
  stage(L.w3, a.w3, 4 * H);
  stage(L.tw3, a.tw3, 4 * H);
  stage(L.b1, a.b1, H);
  stage(L.b2, a.b2, H);
  stage(L.tb1, a.tb1, H);
  stage(L.tb2, a.tb2, H);
  if (t < 4) { L.b3[t] = a.b3[t]; L.tb3[t] = a.tb3[t]; }
  {
    const int b = t >> 4, kk = t & 15;  // sample, byte lane (64 samples x 16 lanes = the workgroup)
    int id = 0;
    if (b < B) {
      if (a.rows) {
        const long long r = a.rows[b];
        id = (r >= 0 && r < a.total) ? (int)r : 0;  // (an index outside the stored transitions reads transition 0, not wild memory)
      } else {
        uint32_t x[4];
        philox4x32_10((uint32_t)b, 0u, (uint32_t)step0, 4u, (uint32_t)a.seed, (uint32_t)(a.seed >> 32), x);
        const unsigned long long r = ((unsigned long long)x[0] << 32) | x[1];
        id = (int)__umul64hi(r, (unsigned long long)a.total);
      }
    }
    if (kk == 0) {
      if (b < B && a.rows_out) a.rows_out[b] = id;
      L.act[b] = a.actions[id] & 3;
      L.rew[b] = (float)((double)a.rewards[id] * a.reward_scale);
      L.term[b] = a.terminals[id] ? 1 : 0;
    }
    int8_t sv[(KP + 15) / 16], s2v[(KP + 15) / 16];
#pragma unroll
    for (int j = 0; j < (KP + 15) / 16; ++j) {  // all of a lane's bytes requested before the first is used
      const int k = kk + 16 * j;
      const bool live = b < B && k < K0;
      sv[j] = live ? a.states[(int64_t)id * K0 + k] : (int8_t)0;
      s2v[j] = live ? a.successors[(int64_t)id * K0 + k] : (int8_t)0;
    }
#pragma unroll
    for (int j = 0; j < (KP + 15) / 16; ++j) {
      const int k = kk + 16 * j;
      if (k < KP) {
        L.S[b * KP + k] = sv[j];
        L.S2[b * KP + k] = s2v[j];
      }
    }
  }
  rows_commit<H, K0>(L.ST, first);
  if (t == LWG - 1) {  // Adam's bias corrections (two double-precision pow(), ~1.3 us on one lane) in the shadow of the gather
    const long long step = step0 + 1;
    L.scratch[16] = a.lr / (float)(1.0 - pow((double)a.beta1, (double)step));
    L.scratch[17] = 1.0f / sqrtf((float)(1.0 - pow((double)a.beta2, (double)step)));
  }
  __syncthreads();
  SGK_STAMP(1);
  // ---- target network on the successors; every next matrix is requested before the layer that precedes its use ----
  RowsInFlight<H, H> big;
  RowsInFlight<H, K0> small;
  rows_request<H, H>(big, a.tw2t);
  dense_layer<K0, H, weight_stride(H)>(L.S2, KP, L.ST, L.tb1, L.C, true, nullptr);
  __syncthreads();
  SGK_STAMP(2);
  rows_commit<H, H>(L.ST, big);
  __syncthreads();
  SGK_STAMP(3);
  rows_request<H, K0>(small, a.w1t);
  dense_layer<H, H, weight_stride(H)>(L.C, H, L.ST, L.tb2, L.D, true, nullptr);
  __syncthreads();
  SGK_STAMP(4);
  head_forward(L.D, H, L.tw3, L.tb3, L.tq);
  // ---- Q-network on the states ----
  rows_commit<H, K0>(L.ST, small);
  __syncthreads();
  SGK_STAMP(5);
  rows_request<H, H>(big, a.w2t);
  dense_layer<K0, H, weight_stride(H)>(L.S, KP, L.ST, L.b1, L.A, true, nullptr);
  __syncthreads();
  SGK_STAMP(6);
  rows_commit<H, H>(L.ST, big);
  __syncthreads();
  SGK_STAMP(7);
  rows_request<H, H>(big, a.w2);  // W2 as it is ([j][k]): the back-propagation through layer 2 wants it in this orientation
  dense_layer<H, H, weight_stride(H)>(L.A, H, L.ST, L.b2, L.Bq, true, nullptr);
  __syncthreads();
  SGK_STAMP(8);
  head_forward(L.Bq, H, L.w3, L.b3, L.q);
  rows_commit<H, H>(L.ST, big);
  __syncthreads();
  SGK_STAMP(9);
  // ---- the loss on y_j = r_j + discount * max_a' Q_target(s'_j, a') * (1 - terminal_j) and q_i = Q(s_i, a_i); dL/dq on the taken action.
  // loss_mode 0 (the reference, value.py:119-123): Qs is [B,1] and expected_Qs [B], so F.mse_loss broadcasts them to [B,B]:
  //   loss = mean_{i,j} (q_i - y_j)^2 = mean_i (q_i - ybar)^2 + mean_j (y_j - ybar)^2,   dL/dq_i = (2 / B) (q_i - ybar),  ybar = mean_j y_j
  //   (every sample is pulled towards the minibatch's MEAN target). loss_mode 1 (not the reference): the per-sample form,
  //   loss = mean_i (q_i - y_i)^2, dL/dq_i = (2 / B) (q_i - y_i).
  float sq = 0.0f;
  if (t < LB) {  // (exactly wave 0: the wave reduction below runs with all 64 lanes)
    float g = 0.0f, target = 0.0f, qsa = 0.0f;
    if (t < B) {
      const float nq = fmaxf(fmaxf(L.tq[t * 4], L.tq[t * 4 + 1]), fmaxf(L.tq[t * 4 + 2], L.tq[t * 4 + 3]));
      target = a.discount * (L.term[t] ? 0.0f : nq) + L.rew[t];
      qsa = L.q[t * 4 + L.act[t]];
    }
    if (a.loss_mode == 0) {
      const float ybar = __ockl_wfred_add_f32(target) / (float)B;  // (lanes >= B hold 0)
      if (t < B) {
        const float d = qsa - ybar, e = target - ybar;
        sq = fmaf(d, d, e * e);
        g = 2.0f * d / (float)B;
      }
    } else if (t < B) {
      const float d = qsa - target;
      sq = d * d;
      g = 2.0f * d / (float)B;
    }
    L.y[t] = g;
  }
  const float loss = block_sum(sq, L.scratch) / (float)B;  // (has barriers: y[] is visible afterwards)
  if (t < LB * 4) L.q[t] = ((t & 3) == L.act[t >> 2]) ? L.y[t >> 2] : 0.0f;  // q now holds dL/dq
  __syncthreads();
  SGK_STAMP(10);

  // ---- backward; every lane keeps the gradients of the parameters it owns in registers ----------------------------
  // dL/dh2 = relu'(h2) * (dq W3) -> C   (dense_layer with K = 4, "wt" = W3 [4][H])
  dense_layer<4, H>(L.q, 4, L.w3, nullptr, L.C, false, L.Bq);
  // W3 [4][H] on the matrix cores: wave w < MT takes the 16 columns k = 16 w .. of dq^T h2 (A = dq [b][4]: the tile's rows
  // 4 .. 15 are neighbouring samples' values and are dropped); the lanes of group 0 hold the four actions of column k.
  // b3 = column sums of dq: wave MT, lane 0 holds all four.
  f4 gw3 = {0.0f, 0.0f, 0.0f, 0.0f}, gb3 = {0.0f, 0.0f, 0.0f, 0.0f};
  if (wave < MT) gw3 = weight_grad_mfma(L.q, 4, 0, L.Bq, H, 16 * wave, lane);
  else if (wave == MT) gb3 = column_sum_mfma(L.q, 4, 0, lane);
  const bool own_w3 = wave < MT && grp == 0 && 16 * wave + col < H;
  const bool own_b3 = wave == MT && lane == 0;
  if (!own_w3) gw3 = (f4){0.0f, 0.0f, 0.0f, 0.0f};
  if (!own_b3) gb3 = (f4){0.0f, 0.0f, 0.0f, 0.0f};
  __syncthreads();
  SGK_STAMP(11);
  // dL/dh1 = relu'(h1) * (dh2 W2) -> D   ("wt" = W2 as it is)
  dense_layer<H, H, weight_stride(H)>(L.C, H, L.ST, nullptr, L.D, false, L.A);
  // W2 [H][H] in 16 x 16 MFMA tiles dealt round-robin to the 16 waves (tile = (neuron tile, input tile)); the gradients stay in
  // registers, TRANSPOSED in the C layout (weight_grad_mfma_t): lane (col, grp), register r <-> dW2[j0 + col][k0 + 4 grp + r], four
  // consecutive elements of a row of W2. MT more tiles are b2's column sums of dL/dh2 (rows j0 + 4 grp + r, the same in every
  // column: the lanes of column 0 own them).
  constexpr int T2 = MT * MT + MT, T1 = MT * KT1 + MT;
  constexpr int N2 = (T2 + LWG / 64 - 1) / (LWG / 64), N1 = (T1 + LWG / 64 - 1) / (LWG / 64);  // tiles per wave
  f4 gw2[N2];
#pragma unroll
  for (int i = 0; i < N2; ++i) {
    const int tile = wave + i * (LWG / 64);
    gw2[i] = (f4){0.0f, 0.0f, 0.0f, 0.0f};
    if (tile < MT * MT) gw2[i] = weight_grad_mfma_t(L.C, H, 16 * (tile / MT), L.A, H, 16 * (tile % MT), lane);
    else if (tile < T2) gw2[i] = column_sum_mfma(L.C, H, 16 * (tile - MT * MT), lane);
  }
  __syncthreads();
  SGK_STAMP(12);
  // W1 [H][K0] the same way (columns = board cells; cells >= K0 of the last tile are computed from what lies behind the
  // row in LDS and dropped) -- transposed like W2 where a row's quads are 16-byte aligned (K0 % 4 == 0), else in the plain C layout
  // (register r <-> dW1[j0 + 4 grp + r][k0 + col]) --, then b1's column sums of dL/dh1
  constexpr bool W1T = (K0 % 4) == 0;
  f4 gw1[N1];
#pragma unroll
  for (int i = 0; i < N1; ++i) {
    const int tile = wave + i * (LWG / 64);
    gw1[i] = (f4){0.0f, 0.0f, 0.0f, 0.0f};
    if (tile < MT * KT1)
      gw1[i] = W1T ? weight_grad_mfma_t(L.D, H, 16 * (tile / KT1), L.S, KP, 16 * (tile % KT1), lane)
                   : weight_grad_mfma(L.D, H, 16 * (tile / KT1), L.S, KP, 16 * (tile % KT1), lane);
    else if (tile < T1) gw1[i] = column_sum_mfma(L.D, H, 16 * (tile - MT * KT1), lane);
  }
  // ---- clip_grad_norm_(max_norm): coefficient from the global 2-norm of all gradients -----------------------------
  float ss = 0.0f;
#pragma unroll
  for (int c = 0; c < 4; ++c) ss = fmaf(gw3[c], gw3[c], fmaf(gb3[c], gb3[c], ss));
#pragma unroll
  for (int i = 0; i < N2; ++i) {
    const int tile = wave + i * (LWG / 64);
    const bool bias = tile >= MT * MT;
    // a weight tile's lane holds (row j0 + col, columns k0 + 4 grp + r); a bias tile's (rows j0 + 4 grp + r), counted once, in column 0
    const int j = 16 * (bias ? tile - MT * MT : tile / MT) + (bias ? 4 * grp : col), k = bias ? 0 : 16 * (tile % MT) + 4 * grp;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      // rows / columns past H belong to no parameter
      const bool live = tile < T2 && (bias ? (j + r < H && col == 0) : (j < H && k + r < H));
      gw2[i][r] = live ? gw2[i][r] : 0.0f;
      ss = fmaf(gw2[i][r], gw2[i][r], ss);
    }
  }
#pragma unroll
  for (int i = 0; i < N1; ++i) {
    const int tile = wave + i * (LWG / 64);
    const bool bias = tile >= MT * KT1;
    const bool tr = W1T && !bias;  // transposed ownership
    const int j = 16 * (bias ? tile - MT * KT1 : tile / KT1) + (tr ? col : 4 * grp), k = bias ? 0 : 16 * (tile % KT1) + (tr ? 4 * grp : col);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool live = tile < T1 && (bias ? (j + r < H && col == 0) : tr ? (j < H && k + r < K0) : (j + r < H && k < K0));
      gw1[i][r] = live ? gw1[i][r] : 0.0f;
      ss = fmaf(gw1[i][r], gw1[i][r], ss);
    }
  }
  // ---- Adam (amsgrad) on the owned parameters; the transposed copies follow ---------------------------------------
  // A lane owns NQ "quads": four parameters of one tensor at a fixed stride -- a column piece of a weight tile (rows j .. j + 3 at
  // column k: the MFMA C layout), four consecutive bias elements, W3's four rows of a column, or b3. Their old values (w, m, v,
  // vmax) are requested QD - 1 quads AHEAD: the first ones before the norm's reduction, quad i + QD - 1 before quad i is updated
  // and stored. (Per quad in turn -- load, update, store, next -- the seven quads were seven dependent round trips.)
  constexpr int NQ = 1 + N2 + N1;
  struct QuadRef {
    float *w, *m, *v, *x, *wt;  // tensor, Adam state, and where the quad goes in the transposed copy (null: none)
    int e0, stride, wt_stride;  // stride 1: four consecutive, 16-byte aligned elements
    bool on;
    int flat;                   // the tensor's offset in the flat parameter vector (ParamMap)
  };
  using PM = ParamMap<K0, H>;
  auto quad_ref = [&](int i) -> QuadRef {
    QuadRef r{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 1, 1, false, 0};
    if (i == 0) {
      if (own_w3) { const int k = 16 * wave + col; r = QuadRef{a.w3, a.m[4], a.v[4], a.vmax[4], a.w3t + 4 * k, k, H, 1, true, PM::o_w3}; }
      else if (own_b3) r = QuadRef{a.b3, a.m[5], a.v[5], a.vmax[5], nullptr, 0, 1, 1, true, PM::o_b3};
    } else if (i <= N2) {
      const int tile = wave + (i - 1) * (LWG / 64);
      if (tile < MT * MT) {
        const int j = 16 * (tile / MT) + col, k = 16 * (tile % MT) + 4 * grp;  // H % 4 == 0: the four columns k .. k + 3 are all inside
        if (j < H && k < H) r = QuadRef{a.w2, a.m[2], a.v[2], a.vmax[2], a.w2t + (size_t)k * H + j, j * H + k, 1, H, true, PM::o_w2};
      } else if (tile < T2) {
        const int jb = 16 * (tile - MT * MT) + 4 * grp;
        if (col == 0 && jb < H) r = QuadRef{a.b2, a.m[3], a.v[3], a.vmax[3], nullptr, jb, 1, 1, true, PM::o_b2};
      }
    } else {
      const int tile = wave + (i - 1 - N2) * (LWG / 64);
      if (tile < MT * KT1) {
        if (W1T) {
          const int j = 16 * (tile / KT1) + col, k = 16 * (tile % KT1) + 4 * grp;
          if (j < H && k < K0) r = QuadRef{a.w1, a.m[0], a.v[0], a.vmax[0], a.w1t + (size_t)k * H + j, j * K0 + k, 1, H, true, PM::o_w1};
        } else {
          const int j = 16 * (tile / KT1) + 4 * grp, k = 16 * (tile % KT1) + col;
          if (j < H && k < K0) r = QuadRef{a.w1, a.m[0], a.v[0], a.vmax[0], a.w1t + (size_t)k * H + j, j * K0 + k, K0, 1, true, PM::o_w1};
        }
      } else if (tile < T1) {
        const int jb = 16 * (tile - MT * KT1) + 4 * grp;
        if (col == 0 && jb < H) r = QuadRef{a.b1, a.m[1], a.v[1], a.vmax[1], nullptr, jb, 1, 1, true, PM::o_b1};
      }
    }
    return r;
  };
  struct QuadVals { f4 w, m, v, x; };
  auto quad_request = [&](QuadVals &q, const QuadRef &r) {
    q.w = q.m = q.v = q.x = (f4){0.0f, 0.0f, 0.0f, 0.0f};
    if (r.on) {
      if (r.stride == 1) {
        q.w = *reinterpret_cast<const f4 *>(r.w + r.e0); q.m = *reinterpret_cast<const f4 *>(r.m + r.e0);
        q.v = *reinterpret_cast<const f4 *>(r.v + r.e0); q.x = *reinterpret_cast<const f4 *>(r.x + r.e0);
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int e = r.e0 + c * r.stride;
          q.w[c] = r.w[e]; q.m[c] = r.m[e]; q.v[c] = r.v[e]; q.x[c] = r.x[e];
        }
      }
    }
  };
  // (how far ahead: measured with one and with two quads ahead -- 10.0 and 9.0 us for the phase, the second at the price of register
  // spills -- so the trips are not what bounds it. With -DSGK_LEARN_TIMELINE_ADAM wave 0 takes 0.5-1.3 us per quad whatever the
  // distance and then waits 2.9 us at the barrier for the other waves: the phase moves ~500 KB (13.9 k parameters x 4 tensors in, 5
  // out, padded tiles and 4-lane bias quads at full instruction cost) through ONE CU's 64 B/clk vector-memory path at ~40 % of
  // that rate. What would shorten it is more CUs, and a second workgroup costs a grid barrier: see the experiment below.)
  constexpr int QD = 2;
  if (a.adam_scratch) {  // (wave-uniform) two launches: the gradient leaves here, dqn_adam_kernel applies it
    float *gout = a.adam_scratch + sizeof(AdamHeader) / sizeof(float);
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const QuadRef r = quad_ref(i);
      const f4 g = i == 0 ? (own_w3 ? gw3 : gb3) : (i <= N2 ? gw2[i <= N2 ? i - 1 : 0] : gw1[i > N2 ? i - 1 - N2 : 0]);
      if (r.on) {
        if (r.stride == 1) *reinterpret_cast<f4 *>(gout + r.flat + r.e0) = g;
        else {
#pragma unroll
          for (int c = 0; c < 4; ++c) gout[r.flat + r.e0 + c * r.stride] = g[c];
        }
      }
    }
    const float norm2 = sqrtf(block_sum(ss, L.scratch));
    if (t == 0) {
      AdamHeader hd;
      hd.coef = fminf(a.max_norm / (norm2 + 1e-6f), 1.0f);
      hd.lr_bc1 = L.scratch[16];
      hd.inv_bc2_sqrt = L.scratch[17];
      hd.pad = 0.0f;
      *reinterpret_cast<AdamHeader *>(a.adam_scratch) = hd;
      *a.step = step0 + 1;
      if (a.loss_out) *a.loss_out = loss;
    }
    SGK_STAMP(13);
    return;
  }
  QuadVals qv[QD];
#pragma unroll
  for (int i = 0; i < QD - 1 && i < NQ; ++i) quad_request(qv[i], quad_ref(i));
  const float norm = sqrtf(block_sum(ss, L.scratch));
  SGK_STAMP(13);
  const float coef = fminf(a.max_norm / (norm + 1e-6f), 1.0f);
  AdamCoef ac;
  ac.lr_bc1 = L.scratch[16];  // (the bias corrections: computed at entry)
  ac.inv_bc2_sqrt = L.scratch[17];
  ac.beta1 = a.beta1; ac.beta2 = a.beta2; ac.eps = a.eps;
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    if (i + QD - 1 < NQ) quad_request(qv[(i + QD - 1) % QD], quad_ref(i + QD - 1));
#ifdef SGK_LEARN_TIMELINE_ADAM
    SGK_STAMP(14 + i);
#endif
    const QuadRef r = quad_ref(i);
    QuadVals &q = qv[i % QD];
    const f4 g = i == 0 ? (own_w3 ? gw3 : gb3) : (i <= N2 ? gw2[i <= N2 ? i - 1 : 0] : gw1[i > N2 ? i - 1 - N2 : 0]);
    if (r.on) {
      f4 nw, nm, nv, nx;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float m = q.m[c], v = q.v[c], x = q.x[c];
        nw[c] = adam_scalar(q.w[c], m, v, x, g[c] * coef, ac);
        nm[c] = m; nv[c] = v; nx[c] = x;
      }
      if (r.stride == 1) {
        *reinterpret_cast<f4 *>(r.w + r.e0) = nw; *reinterpret_cast<f4 *>(r.m + r.e0) = nm;
        *reinterpret_cast<f4 *>(r.v + r.e0) = nv; *reinterpret_cast<f4 *>(r.x + r.e0) = nx;
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int e = r.e0 + c * r.stride;
          r.w[e] = nw[c]; r.m[e] = nm[c]; r.v[e] = nv[c]; r.x[e] = nx[c];
        }
      }
      if (r.wt) {
        if (r.wt_stride == 1) *reinterpret_cast<f4 *>(r.wt) = nw;
        else {
#pragma unroll
          for (int c = 0; c < 4; ++c) r.wt[(size_t)c * r.wt_stride] = nw[c];
        }
      }
    }
  }
  if (t == 0) {
    *a.step = step0 + 1;
    if (a.loss_out) *a.loss_out = loss;
  }
#ifdef SGK_LEARN_TIMELINE
#ifdef SGK_LEARN_TIMELINE_ADAM
  SGK_STAMP(14 + NQ);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  SGK_STAMP(15 + NQ);
  __syncthreads();
  SGK_STAMP(16 + NQ);
#else
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's updates are out
  __syncthreads();
  SGK_STAMP(14);
#endif
#endif
}

// Adam(amsgrad) + the transposed copies, one lane per parameter (the second launch of sgk_dqn_sgd_step: see AdamHeader above)
template <int K0, int H>
__device__ __forceinline__ void adam_body(const LearnArgs &a, int vblock) {
  using PM = ParamMap<K0, H>;
  const AdamHeader hd = *reinterpret_cast<const AdamHeader *>(a.adam_scratch);
  const float *g = a.adam_scratch + sizeof(AdamHeader) / sizeof(float);
  AdamCoef ac;
  ac.lr_bc1 = hd.lr_bc1;
  ac.inv_bc2_sqrt = hd.inv_bc2_sqrt;
  ac.beta1 = a.beta1; ac.beta2 = a.beta2; ac.eps = a.eps;
  const int e = (int)(vblock * 256 + threadIdx.x);
  if (e >= PM::P) return;
  const int ten = e < PM::o_b1 ? 0 : e < PM::o_w2 ? 1 : e < PM::o_b2 ? 2 : e < PM::o_w3 ? 3 : e < PM::o_b3 ? 4 : 5;
  const int off = ten == 0 ? PM::o_w1 : ten == 1 ? PM::o_b1 : ten == 2 ? PM::o_w2 : ten == 3 ? PM::o_b2 : ten == 4 ? PM::o_w3 : PM::o_b3;
  float *wp = ten == 0 ? a.w1 : ten == 1 ? a.b1 : ten == 2 ? a.w2 : ten == 3 ? a.b2 : ten == 4 ? a.w3 : a.b3;
  const int le = e - off;
  float m = a.m[ten][le], v = a.v[ten][le], x = a.vmax[ten][le];
  const float nw = adam_scalar(wp[le], m, v, x, g[e] * hd.coef, ac);
  wp[le] = nw;
  a.m[ten][le] = m; a.v[ten][le] = v; a.vmax[ten][le] = x;
  // the transposed copies the policy kernels (W1^T, W3^T) and the next step's forward (W2^T) read
  if (ten == 0) { const int j = le / K0, k = le - j * K0; a.w1t[(size_t)k * H + j] = nw; }
  else if (ten == 2) { const int j = le / H, k = le - j * H; a.w2t[(size_t)k * H + j] = nw; }
  else if (ten == 4) { const int ai = le / H, k = le - ai * H; a.w3t[4 * k + ai] = nw; }
}

template <int K0, int H>
__global__ __launch_bounds__(256) void dqn_adam_kernel(LearnArgs a) {
  adam_body<K0, H>(a, (int)blockIdx.x);
}

// Adam AND the lockstep step's last launch -- reset_done + the next transitions' states into the replay ring (sgk_reset_done_store) --
// in one grid (sgk_dqn_sgd_step_reset_store): the two are independent (the reset neither reads the weights nor the minibatch, and the
// SGD kernel before this launch has finished sampling the ring), each alone is a 5 us launch that leaves most of the chip idle. The
// first adam_blocks workgroups run Adam, the others the reset kernel's body over their own virtual grid.
struct ResetStoreArgs {
  const SgkRules *rules;
  uint64_t *state;
  int8_t *boards;
  int32_t *n_resets;
  const double *aux;
  int64_t n;
  uint64_t seed, env_base;
  int32_t mode_flags;
  StepStore st;
};
template <int H, int ENV, int LAYOUT>
__global__ __launch_bounds__(256) void dqn_adam_reset_kernel(LearnArgs a, int adam_blocks, ResetStoreArgs r) {
  static_assert(WG == 256, "the reset body's workgroup");
  if ((int)blockIdx.x < adam_blocks) {  // (workgroup-uniform)
    adam_body<Geom<ENV>::NC, H>(a, (int)blockIdx.x);
  } else {
    reset_body<ENV, LAYOUT, true>(r.rules, r.state, r.boards, nullptr, r.mode_flags, r.n, r.seed, r.env_base, r.n_resets, r.aux, r.st,
                                  (int)blockIdx.x - adam_blocks, (int)gridDim.x - adam_blocks);
  }
}

// ------------------------------------------------------------------------------------------------
// EXPERIMENT (round 6; compiled only with -DSGK_DQN_MULTI_WG, never into the product library): the same SGD step on FOUR
// workgroups. The idea: dqn_sgd_kernel above is one workgroup on one CU -- 21 us of its 40 in fp32 MFMA phases at 70 % of one CU's
// matrix rate, 11 us in Adam through one CU's memory pipeline (profiles/r06/dqn_timeline_before.log) -- so give each of GW = 4
// workgroups 16 of the 64 samples (one MFMA sample tile; with 16 samples' activations all four weight matrices fit LDS at once:
// one batch of loads at entry, W2 held once with an odd row stride and read along either axis) and a quarter of Adam. What the
// split costs is three grid-wide barriers: B1 the targets of all 64 samples (the reference's loss needs their MEAN in every
// sample's gradient, value.py:119-123), B2 the four partial gradients (summed in the fixed order 0..3), B3 the partial norms.
// MEASURED (profiles/r06/dqn_timeline_multi.log, MI355X, 2.39 GHz in-kernel): 58 us against the one-workgroup kernel's 41. The
// barriers cost 6.8 / 8.3 / 6.1 us (agent-scope release + acquire: the XCDs' L2s are not coherent, a release writes an L2's dirty
// lines back and everything read behind an acquire comes from HBM: the phases behind the barriers take 4.9 and 5.9 us for a few KB),
// the entry 6.6 us (every workgroup's weights were last written by another XCD), and a layer of ONE tile per wave is
// latency-bound (3.0 us: 25 dependent MFMAs behind four batches of LDS reads, nothing to overlap them with). Correct -- it passes
// every DeepQ test, the reference-run fixtures included -- and slower; kept as the record of why the learner stays on one CU.
// ------------------------------------------------------------------------------------------------
#ifdef SGK_DQN_MULTI_WG
constexpr int GW = 4;         // workgroups
constexpr int NBW = LB / GW;  // samples per workgroup: one 16-wide MFMA tile

struct LearnShared {  // in the scratch block the library keeps per env handle (zeroed when it is made)
  unsigned int count, gen;
  unsigned int pad[14];
  float y[LB];         // every sample's TD target (B1)
  float ss[GW];        // partial squared gradient norms (B3)
  float loss[GW];      // partial loss sums (B2)
  float pad2[8];
};
static_assert(sizeof(LearnShared) % 16 == 0, "the gradient partials behind it are read as dwords but keep it aligned");

// every lane calls it; on return every global write made by any workgroup before its call is visible to every lane here
__device__ __forceinline__ void grid_barrier(LearnShared *sh) {
  __threadfence();  // release: this wave's stores are out of the CU and the XCD's L2 is written back (agent scope)
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int g = __hip_atomic_load(&sh->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__hip_atomic_fetch_add(&sh->count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(GW - 1)) {
      __hip_atomic_store(&sh->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&sh->gen, g + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (__hip_atomic_load(&sh->gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == g) __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  __threadfence();  // acquire: what this CU / XCD cached of the other workgroups' lines is dropped
}

// dense layer for ONE 16-sample tile: out[b][n] = epilogue(bias[n] + sum_k in[b][k] * wt(k, n)), wt(k, n) = w[k * RS + n * CS] (so a
// matrix held once serves both orientations), on v_mfma_f32_16x16x4_f32 as dense_layer above; the MT neuron tiles go to the waves
// wave_first .. wave_first + wave_count - 1 (two independent layers run side by side on disjoint wave sets).
template <int K, int H, class T>
__device__ __forceinline__ void dense16(const T *in, int KP, const float *w, int RS, int CS, const float *bias, float *out, bool relu,
                                        const float *mask, int wave_first, int wave_count) {
  const int lane = threadIdx.x & 63, wave = (int)(threadIdx.x >> 6) - wave_first;
  if (wave < 0 || wave >= wave_count) return;
  const int col = lane & 15, grp = lane >> 4;
  constexpr int MT = (H + 15) / 16, KS = (K + 3) / 4;
  constexpr int CH = KS < 8 ? KS : 8;
  for (int tile = wave; tile < MT; tile += wave_count) {
    const int n0 = 16 * tile;
    f4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    if (bias && n0 + 4 * grp < H) acc = *reinterpret_cast<const f4 *>(bias + n0 + 4 * grp);
    const T *xrow = in + col * KP;
    const float *wcol = w + (n0 + col) * CS;  // (neuron rows >= H of the last tile read what lies behind the matrix, inside LDS, and are dropped)
#pragma unroll
    for (int c0 = 0; c0 < KS; c0 += CH) {
      float av[CH], bv[CH];
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        const int k = 4 * (c0 + u) + grp;
        if (4 * (c0 + u) + 3 < K) {
          av[u] = wcol[k * RS];
          bv[u] = load_x1(xrow, k);
        } else if (4 * (c0 + u) < K) {
          const int kc = k < K ? k : K - 1;
          av[u] = k < K ? wcol[kc * RS] : 0.0f;
          bv[u] = k < K ? load_x1(xrow, kc) : 0.0f;
        } else {
          av[u] = 0.0f;
          bv[u] = 0.0f;
        }
      }
#pragma unroll
      for (int u = 0; u < CH; ++u)
        if (4 * (c0 + u) < K) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
    }
    const int n = n0 + 4 * grp;
    if (n < H) {
      if (relu) acc = __builtin_elementwise_max(acc, (f4){0.0f, 0.0f, 0.0f, 0.0f});
      if (mask) {
        const f4 m = *reinterpret_cast<const f4 *>(mask + col * H + n);
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = m[c] > 0.0f ? acc[c] : 0.0f;
      }
      *reinterpret_cast<f4 *>(out + col * H + n) = acc;
    }
  }
}

// G[r][c] = sum over the 16 samples of d[b][row0 + r] * x[b][col0 + c] (one 16 x 16 tile of a weight gradient; C layout: lane
// (col, grp), register r = G[4 grp + r][col]); ones = true: x = 1 (column sums: a bias gradient, the same in every column)
template <class T>
__device__ __forceinline__ f4 wgrad16(const float *d, int DS, int row0, const T *x, int XS, int col0, int lane, bool ones) {
  const int col = lane & 15, grp = lane >> 4;
  f4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
  float av[NBW / 4], bv[NBW / 4];
#pragma unroll
  for (int s = 0; s < NBW / 4; ++s) {
    av[s] = d[(4 * s + grp) * DS + row0 + col];
    bv[s] = ones ? 1.0f : load_x1(x + (4 * s + grp) * XS, col0 + col);
  }
#pragma unroll
  for (int s = 0; s < NBW / 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s], acc, 0, 0, 0);
  return acc;
}

constexpr int w2_stride(int h) { return h | 1; }  // odd: a column walk (n along the lanes, stride w2_stride) and a row walk both spread over the banks

struct MultiLds {
  float *w2, *tw2t, *w1t, *tw1t, *w3, *tw3, *b1, *b2, *b3, *tb1, *tb2, *tb3, *A, *Bq, *C, *D, *q, *tq, *y, *scratch, *rew;
  int8_t *S, *S2;
  int *act, *term;
};
template <int K0, int H>
__device__ __forceinline__ MultiLds carve_multi(unsigned char *base) {
  constexpr int KP = (K0 + 3) & ~3, WS = weight_stride(H);
  MultiLds L;
  float *f = reinterpret_cast<float *>(base);
  L.w2 = f; f += H * w2_stride(H) + 16;  // (+16: the partial last tile of a column walk reads up to 11 rows past the matrix; see dense16)
  f = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(f) + 15) & ~(uintptr_t)15);
  L.tw2t = f; f += H * WS;
  L.w1t = f; f += K0 * WS;
  L.tw1t = f; f += K0 * WS;
  L.w3 = f; f += 4 * H;
  L.tw3 = f; f += 4 * H;
  L.b1 = f; f += H;
  L.b2 = f; f += H;
  L.tb1 = f; f += H;
  L.tb2 = f; f += H;
  L.b3 = f; f += 4;
  L.tb3 = f; f += 4;
  L.A = f; f += NBW * H;
  L.Bq = f; f += NBW * H;
  L.C = f; f += NBW * H;
  L.D = f; f += NBW * H;
  L.q = f; f += LB * 4;   // (LB, not NBW: wgrad16 on dq reads rows past the 16 samples' 4 values; they are dropped)
  L.tq = f; f += NBW * 4;
  L.y = f; f += LB;
  L.scratch = f; f += 32;
  L.rew = f; f += NBW;
  L.act = reinterpret_cast<int *>(f); f += NBW;
  L.term = reinterpret_cast<int *>(f); f += NBW;
  L.S = reinterpret_cast<int8_t *>(f);
  L.S2 = L.S + NBW * KP;
  return L;
}
template <int K0, int H>
constexpr size_t multi_lds_bytes() {
  constexpr int KP = (K0 + 3) & ~3, WS = weight_stride(H);
  return sizeof(float) * (size_t)(H * w2_stride(H) + 16 + 4 + H * WS + 2 * K0 * WS + 8 * H + 4 * H + 8 + 4 * NBW * H + LB * 4 + NBW * 4 + LB + 32 + 3 * NBW) +
         2 * NBW * KP + 64;
}

template <int K0, int H>
__global__ __launch_bounds__(LWG) void dqn_sgd_multi_kernel(LearnArgs a, LearnShared *sh, float *partials) {
  extern __shared__ __attribute__((aligned(16))) unsigned char learn_smem[];
  using PM = ParamMap<K0, H>;
  constexpr int KP = (K0 + 3) & ~3, WS = weight_stride(H), W2S = w2_stride(H);
  constexpr int MT = (H + 15) / 16, KT1 = (K0 + 15) / 16;
  const MultiLds L = carve_multi<K0, H>(learn_smem);
  const int B = a.batch, t = threadIdx.x, wg = blockIdx.x;
  const int lane = t & 63, wave = t >> 6, col = lane & 15, grp = lane >> 4;
  SGK_MSTAMP(0);
  const long long step0 = *a.step;  // (workgroup 0 bumps it after the last barrier: every workgroup has read it by then)

  // ---- entry: ONE batch of loads. The minibatch index of the sample whose bytes a lane fetches is derived by that lane (16 lanes
  // per sample): no exchange between the draw and the gather. Everything else -- four weight matrices, the small tensors -- is
  // independent of it and travels at the same time.
  {
    const int bl = t >> 4, kk = t & 15;  // local sample (t < 256), byte lane
    if (bl < NBW) {
      const int b = wg * NBW + bl;       // sample of the minibatch
      int id = 0;
      if (b < B) {
        if (a.rows) {
          const long long r = a.rows[b];
          id = (r >= 0 && r < a.total) ? (int)r : 0;
        } else {
          uint32_t x[4];
          philox4x32_10((uint32_t)b, 0u, (uint32_t)step0, 4u, (uint32_t)a.seed, (uint32_t)(a.seed >> 32), x);
          const unsigned long long r = ((unsigned long long)x[0] << 32) | x[1];
          id = (int)__umul64hi(r, (unsigned long long)a.total);
        }
      }
      if (kk == 0) {
        if (b < B && a.rows_out) a.rows_out[b] = id;
        L.act[bl] = a.actions[id] & 3;
        L.rew[bl] = (float)((double)a.rewards[id] * a.reward_scale);
        L.term[bl] = a.terminals[id] ? 1 : 0;
      }
#pragma unroll
      for (int j = 0; j < (KP + 15) / 16; ++j) {
        const int k = kk + 16 * j;
        if (k < KP) {
          const bool live = b < B && k < K0;
          L.S[bl * KP + k] = live ? a.states[(int64_t)id * K0 + k] : (int8_t)0;
          L.S2[bl * KP + k] = live ? a.successors[(int64_t)id * K0 + k] : (int8_t)0;
        }
      }
    }
  }
  stage_rows<H>(L.tw1t, a.tw1t, K0);
  stage_rows<H>(L.w1t, a.w1t, K0);
  stage_rows<H>(L.tw2t, a.tw2t, H);
  for (int i = t * 4; i < H * H; i += LWG * 4) {  // W2 as torch holds it, rows of W2S floats (odd: 4-byte pieces)
    const f4 v = *reinterpret_cast<const f4 *>(a.w2 + i);
    const int r = i / H, c = i - r * H;
    float *d = L.w2 + r * W2S + c;
    d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
  }
  stage(L.w3, a.w3, 4 * H);
  stage(L.tw3, a.tw3, 4 * H);
  stage(L.b1, a.b1, H);
  stage(L.b2, a.b2, H);
  stage(L.tb1, a.tb1, H);
  stage(L.tb2, a.tb2, H);
  if (t < 4) { L.b3[t] = a.b3[t]; L.tb3[t] = a.tb3[t]; }
  if (t == LWG - 1) {  // Adam's bias corrections (two double-precision pow()) in the shadow of the loads
    const long long step = step0 + 1;
    L.scratch[16] = a.lr / (float)(1.0 - pow((double)a.beta1, (double)step));
    L.scratch[17] = 1.0f / sqrtf((float)(1.0 - pow((double)a.beta2, (double)step)));
  }
  __syncthreads();
  SGK_MSTAMP(1);
  // ---- both networks forward, side by side: target on the successors (waves 0..7), Q on the states (waves 8..15) ----
  dense16<K0, H>(L.S2, KP, L.tw1t, WS, 1, L.tb1, L.C, true, nullptr, 0, 8);
  dense16<K0, H>(L.S, KP, L.w1t, WS, 1, L.b1, L.A, true, nullptr, 8, 8);
  __syncthreads();
  SGK_MSTAMP(2);
  dense16<H, H>(L.C, H, L.tw2t, WS, 1, L.tb2, L.D, true, nullptr, 0, 8);
  dense16<H, H>(L.A, H, L.w2, 1, W2S, L.b2, L.Bq, true, nullptr, 8, 8);  // wt(k, n) = W2[n][k]
  __syncthreads();
  SGK_MSTAMP(3);
  if (t < NBW * 4) {  // heads: one lane per (sample, action), target and Q
    const int b = t >> 2, ac = t & 3;
    float st = L.tb3[ac], sq = L.b3[ac];
    for (int k0 = 0; k0 < H; k0 += 4) {
      const f4 wt_ = *reinterpret_cast<const f4 *>(L.tw3 + ac * H + k0), xt = *reinterpret_cast<const f4 *>(L.D + b * H + k0);
      const f4 wq_ = *reinterpret_cast<const f4 *>(L.w3 + ac * H + k0), xq = *reinterpret_cast<const f4 *>(L.Bq + b * H + k0);
      st = fmaf(xt[0], wt_[0], st); st = fmaf(xt[1], wt_[1], st); st = fmaf(xt[2], wt_[2], st); st = fmaf(xt[3], wt_[3], st);
      sq = fmaf(xq[0], wq_[0], sq); sq = fmaf(xq[1], wq_[1], sq); sq = fmaf(xq[2], wq_[2], sq); sq = fmaf(xq[3], wq_[3], sq);
    }
    L.tq[t] = st;
    L.q[t] = sq;
  }
  __syncthreads();
  SGK_MSTAMP(4);
  // ---- TD targets of this workgroup's samples; the reference's loss needs the mean over ALL of them (B1) ----
  float target = 0.0f, qsa = 0.0f;
  const int bg = wg * NBW + t;  // (meaningful for t < NBW)
  if (t < NBW && bg < B) {
    const float nq = fmaxf(fmaxf(L.tq[t * 4], L.tq[t * 4 + 1]), fmaxf(L.tq[t * 4 + 2], L.tq[t * 4 + 3]));
    target = a.discount * (L.term[t] ? 0.0f : nq) + L.rew[t];
    qsa = L.q[t * 4 + L.act[t]];
  }
  float ybar = 0.0f;
  if (a.loss_mode == 0) {
    if (t < NBW) sh->y[bg] = target;  // (0 for the padding samples b >= B)
    grid_barrier(sh);
  SGK_MSTAMP(5);
    if (t < LB) L.y[t] = sh->y[t];
    __syncthreads();
    if (t < 64) {  // wave 0: the same sum, in the same order, in every workgroup
      ybar = __ockl_wfred_add_f32(L.y[t]) / (float)B;
    }
    if (t == 0) L.scratch[18] = ybar;
    __syncthreads();
    ybar = L.scratch[18];
  }
  float sq = 0.0f;
  if (t < NBW) {
    float g = 0.0f;
    if (bg < B) {
      if (a.loss_mode == 0) {
        const float d = qsa - ybar, e = target - ybar;
        sq = fmaf(d, d, e * e);
        g = 2.0f * d / (float)B;
      } else {
        const float d = qsa - target;
        sq = d * d;
        g = 2.0f * d / (float)B;
      }
    }
    L.y[t] = g;
  }
  const float loss_part = block_sum(sq, L.scratch);  // (has barriers: y[] is visible afterwards)
  if (t < NBW * 4) L.q[t] = ((t & 3) == L.act[t >> 2]) ? L.y[t >> 2] : 0.0f;  // q now holds dL/dq of the 16 samples
  __syncthreads();
  SGK_MSTAMP(6);
  // ---- backward on this workgroup's 16 samples; the partial gradients go to scratch in torch's parameter order ----
  float *mine = partials + (size_t)wg * PM::P;
  dense16<4, H>(L.q, 4, L.w3, H, 1, nullptr, L.C, false, L.Bq, 0, 8);  // dL/dh2 = relu'(h2) * (dq W3)
  if (wave >= 8 && wave - 8 < MT) {  // W3: columns k = 16 (wave - 8) .. of dq^T h2; the lanes of group 0 hold the four actions
    const int k = 16 * (wave - 8) + col;
    const f4 g = wgrad16(L.q, 4, 0, L.Bq, H, 16 * (wave - 8), lane, false);
    if (grp == 0 && k < H) {
#pragma unroll
      for (int r = 0; r < 4; ++r) mine[PM::o_w3 + r * H + k] = g[r];
    }
  } else if (wave == 8 + MT) {
    const f4 g = wgrad16(L.q, 4, 0, L.Bq, H, 0, lane, true);
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) mine[PM::o_b3 + r] = g[r];
    }
  }
  __syncthreads();
  SGK_MSTAMP(7);
  dense16<H, H>(L.C, H, L.w2, W2S, 1, nullptr, L.D, false, L.A, 0, 8);  // dL/dh1 = relu'(h1) * (dh2 W2): wt(k = j, n) = W2[j][n]
  // meanwhile, on the other eight waves: W2 / b2 gradient tiles (they need dh2 = C and h1 = A only)
  {
    constexpr int T2 = MT * MT + MT;
    for (int tile = wave - 8; wave >= 8 && tile < T2; tile += 8) {
      if (tile < MT * MT) {
        const int j0 = 16 * (tile / MT), k0 = 16 * (tile % MT);
        const f4 g = wgrad16(L.C, H, j0, L.A, H, k0, lane, false);
        const int j = j0 + 4 * grp, k = k0 + col;
        if (k < H) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (j + r < H) mine[PM::o_w2 + (j + r) * H + k] = g[r];
        }
      } else {
        const int j = 16 * (tile - MT * MT) + 4 * grp;
        const f4 g = wgrad16(L.C, H, 16 * (tile - MT * MT), L.A, H, 0, lane, true);
        if (col == 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (j + r < H) mine[PM::o_b2 + j + r] = g[r];
        }
      }
    }
  }
  __syncthreads();
  SGK_MSTAMP(8);
  {
    constexpr int T1 = MT * KT1 + MT;
    for (int tile = wave; tile < T1; tile += LWG / 64) {
      if (tile < MT * KT1) {
        const int j0 = 16 * (tile / KT1), k0 = 16 * (tile % KT1);
        const f4 g = wgrad16(L.D, H, j0, L.S, KP, k0, lane, false);
        const int j = j0 + 4 * grp, k = k0 + col;
        if (k < K0) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (j + r < H) mine[PM::o_w1 + (j + r) * K0 + k] = g[r];
        }
      } else {
        const int j = 16 * (tile - MT * KT1) + 4 * grp;
        const f4 g = wgrad16(L.D, H, 16 * (tile - MT * KT1), L.S, KP, 0, lane, true);
        if (col == 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (j + r < H) mine[PM::o_b1 + j + r] = g[r];
        }
      }
    }
  }
  SGK_MSTAMP(9);
  if (t == 0) sh->loss[wg] = loss_part;
  grid_barrier(sh);  // B2
  SGK_MSTAMP(10);
  // ---- this workgroup's quarter of the flat parameter vector: the gradient (partials summed in the fixed order 0..3), its share
  // of the squared norm; Adam's state for those elements is requested before the norm's barrier and used behind it ----
  constexpr int NE = (PM::P + GW * LWG - 1) / (GW * LWG);
  float g[NE], pw[NE], pm[NE], pv[NE], px[NE];
  float ss = 0.0f;
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const int e = wg * LWG + t + i * (GW * LWG);
    g[i] = 0.0f;
    pw[i] = pm[i] = pv[i] = px[i] = 0.0f;
    if (e < PM::P) {
      float s_ = partials[e];
#pragma unroll
      for (int w = 1; w < GW; ++w) s_ += partials[(size_t)w * PM::P + e];
      g[i] = s_;
      ss = fmaf(s_, s_, ss);
      const int ten = e < PM::o_b1 ? 0 : e < PM::o_w2 ? 1 : e < PM::o_b2 ? 2 : e < PM::o_w3 ? 3 : e < PM::o_b3 ? 4 : 5;
      const int off = ten == 0 ? PM::o_w1 : ten == 1 ? PM::o_b1 : ten == 2 ? PM::o_w2 : ten == 3 ? PM::o_b2 : ten == 4 ? PM::o_w3 : PM::o_b3;
      const float *wp = ten == 0 ? a.w1 : ten == 1 ? a.b1 : ten == 2 ? a.w2 : ten == 3 ? a.b2 : ten == 4 ? a.w3 : a.b3;
      pw[i] = wp[e - off];
      pm[i] = a.m[ten][e - off];
      pv[i] = a.v[ten][e - off];
      px[i] = a.vmax[ten][e - off];
    }
  }
  const float ss_part = block_sum(ss, L.scratch);
  SGK_MSTAMP(11);
  if (t == 0) sh->ss[wg] = ss_part;
  grid_barrier(sh);  // B3
  SGK_MSTAMP(12);
  float total = 0.0f;
#pragma unroll
  for (int w = 0; w < GW; ++w) total += sh->ss[w];
  const float coef = fminf(a.max_norm / (sqrtf(total) + 1e-6f), 1.0f);
  AdamCoef ac;
  ac.lr_bc1 = L.scratch[16];
  ac.inv_bc2_sqrt = L.scratch[17];
  ac.beta1 = a.beta1; ac.beta2 = a.beta2; ac.eps = a.eps;
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const int e = wg * LWG + t + i * (GW * LWG);
    if (e < PM::P) {
      const int ten = e < PM::o_b1 ? 0 : e < PM::o_w2 ? 1 : e < PM::o_b2 ? 2 : e < PM::o_w3 ? 3 : e < PM::o_b3 ? 4 : 5;
      const int off = ten == 0 ? PM::o_w1 : ten == 1 ? PM::o_b1 : ten == 2 ? PM::o_w2 : ten == 3 ? PM::o_b2 : ten == 4 ? PM::o_w3 : PM::o_b3;
      float *wp = ten == 0 ? a.w1 : ten == 1 ? a.b1 : ten == 2 ? a.w2 : ten == 3 ? a.b2 : ten == 4 ? a.w3 : a.b3;
      const int le = e - off;
      float m = pm[i], v = pv[i], x = px[i];
      const float nw = adam_scalar(pw[i], m, v, x, g[i] * coef, ac);
      wp[le] = nw;
      a.m[ten][le] = m; a.v[ten][le] = v; a.vmax[ten][le] = x;
      // the transposed copies the policy kernels (W1^T, W3^T) and the one-workgroup learner (W2^T) read
      if (ten == 0) { const int j = le / K0, k = le - j * K0; a.w1t[(size_t)k * H + j] = nw; }
      else if (ten == 2) { const int j = le / H, k = le - j * H; a.w2t[(size_t)k * H + j] = nw; }
      else if (ten == 4) { const int ai = le / H, k = le - ai * H; a.w3t[4 * k + ai] = nw; }
    }
  }
#ifdef SGK_LEARN_TIMELINE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  SGK_MSTAMP(13);
#endif
  if (wg == 0 && t == 0) {
    *a.step = step0 + 1;
    if (a.loss_out) {
      float l = 0.0f;
#pragma unroll
      for (int w = 0; w < GW; ++w) l += sh->loss[w];
      *a.loss_out = l / (float)B;
    }
  }
}

#endif  // SGK_DQN_MULTI_WG

// ------------------------------------------------------------------------------------------------
// PPOBaseAgent.learn (reference policy_base.py:64-131) for PPOMLPAgent's default topology, ALL epochs in one launch: per
// epoch a minibatch of rows drawn with replacement from the gathered rollout, the current network (trunk, actor, critic)
// and the old policy (trunk, actor) forward, the clipped surrogate with batch-normalised advantages, critic MSE and entropy
// bonus, the backward pass (including the path of the policy loss into the critic through the normalised advantage, which
// the reference's autograd graph has) and Adam. Same structure as dqn_sgd_kernel: one workgroup, activations in LDS, weight
// matrices staged per phase, dense layers and weight gradients on fp32 MFMA.
// ------------------------------------------------------------------------------------------------
struct PpoArgs {
  // rollout: states int8 [T][N][K0], actions uint8 [T][N], returns float [N][T], lengths int32 [N]
  const int8_t *states;
  const uint8_t *actions;
  const float *returns;
  const int32_t *lengths;
  int32_t T;
  int64_t N;
  // current network (torch layouts), updated in place; transposed trunk copies kept current; Adam state per tensor
  float *w1, *b1, *w2, *b2, *wa, *ba, *wc, *bc;  // wa [4][H], ba [4], wc [1][H], bc [1]
  float *w1t, *w2t;
  float *m[8], *v[8];  // order: w1, b1, w2, b2, wa, ba, wc, bc
  // old policy: transposed trunk weights, actor as it is
  const float *ow1t, *ob1, *ow2t, *ob2, *owa, *oba;
  long long *step;   // Adam step counter (device); also the key of the minibatch draws
  float *stats_out;  // [n_epochs][3]: policy loss, value loss, entropy; or null
  const long long *rows;  // [n_epochs][batch] rows t * N + env to use instead of the random draws; or null
  long long *rows_out;    // [n_epochs][batch] the rows each epoch used; or null
  int32_t batch, n_epochs;
  uint64_t seed;
  float lr, beta1, beta2, eps, clipping, critic_coeff, entropy_bonus;
};

struct PpoLds {
  float *A, *Bq, *C, *D, *ST, *wh, *owa, *b1, *b2, *ob1, *ob2, *bh, *oba, *out, *oout, *dout, *ret, *scratch;
  int8_t *S;
  int *act;
};

template <int KP, int H>
__device__ __forceinline__ PpoLds carve_ppo(unsigned char *base) {
  PpoLds L;
  float *f = reinterpret_cast<float *>(base);
  L.A = f; f += LB * H;      // current hidden 1
  L.Bq = f; f += LB * H;     // current hidden 2
  L.C = f; f += LB * H;      // old hidden 1, then dL/dh2
  L.D = f; f += LB * H;      // old hidden 2, then dL/dh1
  L.ST = f; f += H * weight_stride(H);  // the weight matrix of the running phase (padded rows)
  L.wh = f; f += 8 * H;      // current heads: rows 0..3 actor, row 4 critic, rows 5..7 zero
  L.owa = f; f += 4 * H;     // old actor
  L.b1 = f; f += H;
  L.b2 = f; f += H;
  L.ob1 = f; f += H;
  L.ob2 = f; f += H;
  L.bh = f; f += 8;          // actor biases, critic bias, zeros
  L.oba = f; f += 4;
  L.out = f; f += LB * 8;    // current logits [0..3], value [4]
  L.oout = f; f += LB * 4;   // old logits
  L.dout = f; f += LB * 8;   // dL/d(logits, value), zero-padded to 8
  L.ret = f; f += LB;
  L.scratch = f; f += 32;
  L.act = reinterpret_cast<int *>(f); f += LB;
  L.S = reinterpret_cast<int8_t *>(f);
  return L;
}

constexpr size_t ppo_lds_bytes(int KP, int H) {
  return sizeof(float) * ((size_t)4 * LB * H + (size_t)H * weight_stride(H) + 12 * H + 4 * H + 12 + LB * 8 + LB * 4 + LB * 8 + LB + 32 + LB) +
         (size_t)LB * KP + 64;
}

// heads: out[b][o] = bh[o] + sum_k h[b][k] * wh[o][k] for o < NOUT (all LDS): one lane per (b, o)
template <int NOUT, int OS>
__device__ __forceinline__ void heads_forward(const float *h, int H, const float *wh, const float *bh, float *out) {
  if (threadIdx.x < LB * NOUT) {
    const int b = threadIdx.x / NOUT, o = threadIdx.x % NOUT;
    float acc = bh[o];
    for (int k0 = 0; k0 < H; k0 += 4) {
      const f4 w = *reinterpret_cast<const f4 *>(wh + o * H + k0);
      const f4 x = *reinterpret_cast<const f4 *>(h + b * H + k0);
      acc = fmaf(x[0], w[0], acc);
      acc = fmaf(x[1], w[1], acc);
      acc = fmaf(x[2], w[2], acc);
      acc = fmaf(x[3], w[3], acc);
    }
    out[b * OS + o] = acc;
  }
}

__device__ __forceinline__ float adam_plain(float p, float &m, float &v, float g, const AdamCoef &c) {
  m = m + (1.0f - c.beta1) * (g - m);
  v = c.beta2 * v + (1.0f - c.beta2) * g * g;
  return p - c.lr_bc1 * (m * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v) * c.inv_bc2_sqrt + c.eps));  // (one-ulp sqrt / rcp: adam_scalar)
}

template <int K0, int H>
__global__ __launch_bounds__(LWG) void ppo_epochs_kernel(PpoArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char learn_smem[];
  constexpr int KP = (K0 + 3) & ~3;
  const PpoLds L = carve_ppo<KP, H>(learn_smem);
  const int t0 = threadIdx.x, B = a.batch;
  constexpr int MT = (H + 15) / 16, KT1 = (K0 + 15) / 16;
  constexpr int T2 = MT * MT + MT, T1 = MT * KT1 + MT;  // weight tiles + the bias tiles (column sums)
  constexpr int N2 = (T2 + LWG / 64 - 1) / (LWG / 64), N1 = (T1 + LWG / 64 - 1) / (LWG / 64);
  const long long step0 = *a.step;
  {
    const int t = t0;
  // the old policy does not change during the call: its small tensors are staged once
  stage(L.owa, a.owa, 4 * H);
  stage(L.ob1, a.ob1, H);
  stage(L.ob2, a.ob2, H);
  if (t < 4) L.oba[t] = a.oba[t];
  for (int i = t; i < 3 * H; i += LWG) L.wh[5 * H + i] = 0.0f;  // rows 5..7 of the padded head matrix
  if (t < 3) L.bh[5 + t] = 0.0f;
  }

  // The rows of an epoch depend only on (seed, Adam step, lengths): they are drawn one epoch AHEAD -- behind the backward pass,
  // in front of the Adam stores -- so that the three dependent memory round trips (lengths, then action / return, then the
  // board gather's addresses) are off the critical path. `t`, `lane`: this lane's ids (laundered per epoch, see below).
  auto draw_rows = [&](int e_ix, long long e_step, int t, int lane) {
  // ---- minibatch: a row = a (t, env) pair inside that env's episode, uniform with replacement. Each sample tries 16
  // candidates per round (16 lanes), the first valid one in lane order wins; another round only if all 16 miss. ----
  {
    const int b = t >> 4, c = t & 15;  // 64 samples x 16 candidate lanes
    int tt = 0;
    long long nn = 0;
    bool found = a.rows != nullptr;
    if (found && b < B) {
      long long row = a.rows[(long long)e_ix * B + b];
      row = row < 0 ? 0 : (row >= (long long)a.T * a.N ? (long long)a.T * a.N - 1 : row);  // a bad row must not leave the rollout
      tt = (int)(row / a.N);
      nn = row - (long long)tt * a.N;
    }
    for (int round = 0; round < 64 && !found; ++round) {
      uint32_t x[4];
      philox4x32_10((uint32_t)(b * 16 + c), (uint32_t)round, (uint32_t)e_step, 5u, (uint32_t)a.seed, (uint32_t)(a.seed >> 32), x);
      const long long n_c = (long long)__umul64hi(((unsigned long long)x[0] << 32) | x[1], (unsigned long long)a.N);
      const int t_c = (int)__umulhi(x[2], (uint32_t)a.T);
      const bool ok = t_c < a.lengths[n_c];
      const unsigned long long mask = __ballot(ok);
      const unsigned int mine = (unsigned int)((mask >> (lane & 48)) & 0xffffull);  // this sample's 16 candidate lanes
      if (mine) {
        const int win = (lane & 48) + __ffs(mine) - 1;
        tt = __shfl(t_c, win, 64);
        nn = __shfl((int)n_c, win, 64);  // N < 2^31 (checked by the launcher)
        found = true;
      }
    }
    if (c == 0) {
      const long long row = (long long)tt * a.N + nn;
      L.act[b] = b < B ? (int)(a.actions[row] & 3) : 0;
      L.ret[b] = b < B ? a.returns[nn * a.T + tt] : 0.0f;
      reinterpret_cast<long long *>(L.oout)[b] = row;  // parked for the gather at the top of that epoch (oout is free until then)
      if (a.rows_out && b < B) a.rows_out[(long long)e_ix * B + b] = row;
    }
  }
  };
  draw_rows(0, step0, t0, t0 & 63);

  for (int epoch = 0; epoch < a.n_epochs; ++epoch) {
    const long long step = step0 + epoch;  // Adam steps done before this epoch; keys the minibatch draws
    // the lane id is re-derived through an opaque move every epoch: with a loop-invariant id the compiler hoists the ~100
    // per-lane 64-bit addresses of the Adam phase out of the loop and keeps them live across it (261 VGPRs spilled)
    int t = t0;
    asm volatile("" : "+v"(t));
    const int lane = t & 63, wave = t >> 6, col = lane & 15, grp = lane >> 4;
    __syncthreads();                       // the previous epoch's update is complete before anything of it is re-read
    stage_rows<H>(L.ST, a.ow1t, K0);
    stage(L.wh, a.wa, 4 * H);
    stage(L.wh + 4 * H, a.wc, H);
    stage(L.b1, a.b1, H);
    stage(L.b2, a.b2, H);
    if (t < 4) L.bh[t] = a.ba[t];
    if (t == 4) L.bh[4] = a.bc[0];
    __syncthreads();
    for (int i = t; i < LB * KP; i += LWG) {
      const int b = i / KP, k = i - b * KP;
      const long long row = reinterpret_cast<const long long *>(L.oout)[b];
      L.S[i] = (b < B && k < K0) ? a.states[row * K0 + k] : (int8_t)0;
    }
    __syncthreads();
    // ---- old policy, then current network, on the same states ----
    dense_layer<K0, H, weight_stride(H)>(L.S, KP, L.ST, L.ob1, L.C, true, nullptr);
    __syncthreads();
    stage_rows<H>(L.ST, a.ow2t, H);
    __syncthreads();
    dense_layer<H, H, weight_stride(H)>(L.C, H, L.ST, L.ob2, L.D, true, nullptr);
    __syncthreads();
    heads_forward<4, 4>(L.D, H, L.owa, L.oba, L.oout);
    stage_rows<H>(L.ST, a.w1t, K0);
    __syncthreads();
    dense_layer<K0, H, weight_stride(H)>(L.S, KP, L.ST, L.b1, L.A, true, nullptr);
    __syncthreads();
    stage_rows<H>(L.ST, a.w2t, H);
    __syncthreads();
    dense_layer<H, H, weight_stride(H)>(L.A, H, L.ST, L.b2, L.Bq, true, nullptr);
    __syncthreads();
    heads_forward<5, 8>(L.Bq, H, L.wh, L.bh, L.out);
    stage_rows<H>(L.ST, a.w2, H);  // W2 as it is, for the back-propagation
    __syncthreads();
    // ---- losses and dL/d(logits, value): the 64 samples are the 64 lanes of wave 0, batch statistics by wave reductions ----
    if (wave == 0) {
      const int b = lane;
      const bool live = b < B;
      const float invB = 1.0f / (float)B;
      float l[4], lo[4], p[4], logp[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { l[k] = L.out[b * 8 + k]; lo[k] = L.oout[b * 4 + k]; }
      const float v = L.out[b * 8 + 4], r = L.ret[b];
      const int ac = L.act[b];
      const float mx = fmaxf(fmaxf(l[0], l[1]), fmaxf(l[2], l[3])), mxo = fmaxf(fmaxf(lo[0], lo[1]), fmaxf(lo[2], lo[3]));
      const float lse = mx + logf(expf(l[0] - mx) + expf(l[1] - mx) + expf(l[2] - mx) + expf(l[3] - mx));
      const float lseo = mxo + logf(expf(lo[0] - mxo) + expf(lo[1] - mxo) + expf(lo[2] - mxo) + expf(lo[3] - mxo));
      float ent = 0.0f;
#pragma unroll
      for (int k = 0; k < 4; ++k) { logp[k] = l[k] - lse; p[k] = expf(logp[k]); ent -= p[k] * logp[k]; }
      const float logp_a = ac == 0 ? logp[0] : ac == 1 ? logp[1] : ac == 2 ? logp[2] : logp[3];
      const float lo_a = (ac == 0 ? lo[0] : ac == 1 ? lo[1] : ac == 2 ? lo[2] : lo[3]) - lseo;
      const float ratio = expf(logp_a - lo_a);
      // advantages normalised over the minibatch (unbiased std, as torch.std)
      const float adv = live ? r - v : 0.0f;
      const float mu = __ockl_wfred_add_f32(adv) * invB;
      const float dev = live ? adv - mu : 0.0f;
      const float sigma = sqrtf(__ockl_wfred_add_f32(dev * dev) / (float)(B - 1));
      const float advn = dev / sigma;
      const float lo_c = 1.0f - a.clipping, hi_c = 1.0f + a.clipping;
      const float rc = fminf(fmaxf(ratio, lo_c), hi_c);
      const float s1 = advn * ratio, s2 = advn * rc;
      const float w1 = s1 < s2 ? 1.0f : (s1 == s2 ? 0.5f : 0.0f), w2 = 1.0f - w1;  // torch.min's gradient: halves on a tie
      const float inrange = (ratio >= lo_c && ratio <= hi_c) ? 1.0f : 0.0f;
      const float g_advn = live ? -invB * (w1 * ratio + w2 * rc) : 0.0f;
      const float g_ratio = live ? -invB * advn * (w1 + w2 * inrange) : 0.0f;
      const float g_logp = g_ratio * ratio;
      // back through the normalisation: dL/dadv_j = (g_j - mean g) / sigma - (adv_j - mu) * S / ((B - 1) sigma^3)
      const float gbar = __ockl_wfred_add_f32(g_advn) * invB;
      const float S = __ockl_wfred_add_f32(g_advn * dev);
      const float g_adv = live ? (g_advn - gbar) / sigma - dev * S / ((float)(B - 1) * sigma * sigma * sigma) : 0.0f;
      const float dv = live ? a.critic_coeff * 2.0f * (v - r) * invB - g_adv : 0.0f;
      float dl[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float onehot = k == ac ? 1.0f : 0.0f;
        dl[k] = live ? g_logp * (onehot - p[k]) + a.entropy_bonus * invB * p[k] * (logp[k] + ent) : 0.0f;
      }
      *reinterpret_cast<f4 *>(L.dout + b * 8) = (f4){dl[0], dl[1], dl[2], dl[3]};
      *reinterpret_cast<f4 *>(L.dout + b * 8 + 4) = (f4){dv, 0.0f, 0.0f, 0.0f};
      if (a.stats_out) {
        const float pl = -__ockl_wfred_add_f32(live ? fminf(s1, s2) : 0.0f) * invB;
        const float vl = __ockl_wfred_add_f32(live ? (v - r) * (v - r) : 0.0f) * invB;
        const float en = __ockl_wfred_add_f32(live ? ent : 0.0f) * invB;
        if (lane == 0) { a.stats_out[epoch * 3] = pl; a.stats_out[epoch * 3 + 1] = vl; a.stats_out[epoch * 3 + 2] = en; }
      }
    }
    __syncthreads();
    // ---- backward ----
    dense_layer<8, H>(L.dout, 8, L.wh, nullptr, L.C, false, L.Bq);  // dL/dh2 = relu'(h2) * (dout W_heads) -> C
    // head weights [5][H] on the matrix cores: wave w < MT takes the columns k = 16 w ..; A = dout [b][8], so the tile's rows
    // 0..3 are the actor (registers of group 0), row 4 the critic (register 0 of group 1), rows 8..15 the next sample's values
    // (dropped). Head biases = column sums of dout: wave MT (lane 0: actor, lane 16: critic).
    f4 gh = {0.0f, 0.0f, 0.0f, 0.0f}, gbh = {0.0f, 0.0f, 0.0f, 0.0f};
    if (wave < MT) gh = weight_grad_mfma(L.dout, 8, 0, L.Bq, H, 16 * wave, lane);
    else if (wave == MT) gbh = column_sum_mfma(L.dout, 8, 0, lane);
    __syncthreads();
    dense_layer<H, H, weight_stride(H)>(L.C, H, L.ST, nullptr, L.D, false, L.A);  // dL/dh1 -> D
    // W2 tiles, then MT more tiles with b2's column sums of dL/dh2 (owned by the lanes of column 0); W1 / b1 the same way
    f4 gw2[N2];
#pragma unroll
    for (int i = 0; i < N2; ++i) {
      const int tile = wave + i * (LWG / 64);
      gw2[i] = (f4){0.0f, 0.0f, 0.0f, 0.0f};
      if (tile < MT * MT) gw2[i] = weight_grad_mfma(L.C, H, 16 * (tile / MT), L.A, H, 16 * (tile % MT), lane);
      else if (tile < T2) gw2[i] = column_sum_mfma(L.C, H, 16 * (tile - MT * MT), lane);
    }
    __syncthreads();
    f4 gw1[N1];
#pragma unroll
    for (int i = 0; i < N1; ++i) {
      const int tile = wave + i * (LWG / 64);
      gw1[i] = (f4){0.0f, 0.0f, 0.0f, 0.0f};
      if (tile < MT * KT1) gw1[i] = weight_grad_mfma(L.D, H, 16 * (tile / KT1), L.S, KP, 16 * (tile % KT1), lane);
      else if (tile < T1) gw1[i] = column_sum_mfma(L.D, H, 16 * (tile - MT * KT1), lane);
    }
    if (epoch + 1 < a.n_epochs) draw_rows(epoch + 1, step + 1, t, lane);  // act / ret / oout are free from here on
    // ---- Adam (torch defaults: no amsgrad, no gradient clipping) ----
    if (t == 0) {
      L.scratch[16] = a.lr / (float)(1.0 - pow((double)a.beta1, (double)(step + 1)));
      L.scratch[17] = 1.0f / sqrtf((float)(1.0 - pow((double)a.beta2, (double)(step + 1))));
    }
    __syncthreads();
    AdamCoef ac;
    ac.lr_bc1 = L.scratch[16];
    ac.inv_bc2_sqrt = L.scratch[17];
    ac.beta1 = a.beta1; ac.beta2 = a.beta2; ac.eps = a.eps;
    if (wave < MT && 16 * wave + col < H) {
      const int k = 16 * wave + col;
      if (grp == 0) {
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          const int e = o * H + k;
          float m = a.m[4][e], v = a.v[4][e];
          a.wa[e] = adam_plain(a.wa[e], m, v, gh[o], ac);
          a.m[4][e] = m; a.v[4][e] = v;
        }
      } else if (grp == 1) {
        float m = a.m[6][k], v = a.v[6][k];
        a.wc[k] = adam_plain(a.wc[k], m, v, gh[0], ac);
        a.m[6][k] = m; a.v[6][k] = v;
      }
    } else if (wave == MT && lane == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float m = a.m[5][e], v = a.v[5][e];
        a.ba[e] = adam_plain(a.ba[e], m, v, gbh[e], ac);
        a.m[5][e] = m; a.v[5][e] = v;
      }
    } else if (wave == MT && lane == 16) {
      float m = a.m[7][0], v = a.v[7][0];
      a.bc[0] = adam_plain(a.bc[0], m, v, gbh[0], ac);
      a.m[7][0] = m; a.v[7][0] = v;
    }
#pragma unroll
    for (int i = 0; i < N2; ++i) {
      const int tile = wave + i * (LWG / 64);
      const int j = 16 * (tile / MT) + 4 * grp, k = 16 * (tile % MT) + col;
      if (tile >= MT * MT && tile < T2) {
        const int jb = 16 * (tile - MT * MT) + 4 * grp;
        if (col == 0 && jb < H) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float m = a.m[3][jb + r], v = a.v[3][jb + r];
            a.b2[jb + r] = adam_plain(a.b2[jb + r], m, v, gw2[i][r], ac);
            a.m[3][jb + r] = m; a.v[3][jb + r] = v;
          }
        }
      } else if (tile < MT * MT && j < H && k < H) {
        f4 nw;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int e = (j + r) * H + k;
          float m = a.m[2][e], v = a.v[2][e];
          nw[r] = adam_plain(a.w2[e], m, v, gw2[i][r], ac);
          a.w2[e] = nw[r]; a.m[2][e] = m; a.v[2][e] = v;
        }
        *reinterpret_cast<f4 *>(a.w2t + (size_t)k * H + j) = nw;
      }
    }
#pragma unroll
    for (int i = 0; i < N1; ++i) {
      const int tile = wave + i * (LWG / 64);
      const int j = 16 * (tile / KT1) + 4 * grp, k = 16 * (tile % KT1) + col;
      if (tile >= MT * KT1 && tile < T1) {
        const int jb = 16 * (tile - MT * KT1) + 4 * grp;
        if (col == 0 && jb < H) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float m = a.m[1][jb + r], v = a.v[1][jb + r];
            a.b1[jb + r] = adam_plain(a.b1[jb + r], m, v, gw1[i][r], ac);
            a.m[1][jb + r] = m; a.v[1][jb + r] = v;
          }
        }
      } else if (tile < MT * KT1 && j < H && k < K0) {
        f4 nw;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int e = (j + r) * K0 + k;
          float m = a.m[0][e], v = a.v[0][e];
          nw[r] = adam_plain(a.w1[e], m, v, gw1[i][r], ac);
          a.w1[e] = nw[r]; a.m[0][e] = m; a.v[0][e] = v;
        }
        *reinterpret_cast<f4 *>(a.w1t + (size_t)k * H + j) = nw;
      }
    }
    // no device-scope fence: the next epoch's staging loads are ordered behind these stores by the workgroup barrier at the top
    // of the loop (same CU, one L1), and nobody else reads the parameters before the kernel ends (it cost ~4 us per epoch)
  }
  __syncthreads();
  if (t0 == 0) *a.step = step0 + a.n_epochs;
}

// ReplayBuffer.add for every env (reference contain.py:15-17, called from value.py:114): phase 0, before env.step, stores the
// boards as the transitions' `state`; phase 1, after it, stores the boards as `successor` and the action / reward /
// terminal flag from the step records. One launch per phase instead of seven tensor copies per lockstep step.
__global__ __launch_bounds__(WG) void replay_store_kernel(const int8_t *__restrict__ boards, int pitch, int nc, int64_t n,
                                                          const uint32_t *__restrict__ rec, const uint8_t *__restrict__ actions,
                                                          int phase, int cheat, int64_t head, const long long *__restrict__ head_dev,
                                                          int8_t *__restrict__ states, int8_t *__restrict__ successors,
                                                          uint8_t *__restrict__ r_actions, int8_t *__restrict__ r_rewards,
                                                          uint8_t *__restrict__ r_terminals) {
  const int64_t slice = head_dev ? (int64_t)*head_dev : head;
  int8_t *dst = (phase == 0 ? states : successors) + slice * n * nc;
  const int64_t gtid = (int64_t)blockIdx.x * WG + threadIdx.x, gsz = (int64_t)gridDim.x * WG;
  if (pitch == nc && ((n * nc) & 3) == 0) {  // rows back to back: copy dwords
    const uint32_t *s32 = reinterpret_cast<const uint32_t *>(boards);
    uint32_t *d32 = reinterpret_cast<uint32_t *>(dst);
    for (int64_t i = gtid; i < n * nc / 4; i += gsz) d32[i] = s32[i];
  } else {
    for (int64_t i = gtid; i < n * nc; i += gsz) dst[i] = boards[(i / nc) * pitch + i % nc];
  }
  if (phase == 1) {
    for (int64_t env = gtid; env < n; env += gsz) {
      const uint32_t r = rec[env];  // reward | hidden << 8 | done << 16 | actual action << 24
      // (& 3: an executed "stay", action 4 under a non-default SGK_INTERRUPT_FORCED_ACTION, has no Q column; it is stored as UP,
      // the default reading of that switch -- same trajectory in the level's corridor)
      r_actions[slice * n + env] = cheat ? (uint8_t)((r >> 24) & 3u) : actions[env];
      r_rewards[slice * n + env] = cheat ? (int8_t)(r >> 8) : (int8_t)r;
      r_terminals[slice * n + env] = (uint8_t)((r >> 16) & 1u);
    }
  }
}

hipError_t launch_replay_store(const Shard &sh, int phase, const uint8_t *actions, int cheat, int64_t head, const long long *head_dev,
                               int8_t *states, int8_t *successors, uint8_t *r_actions, int8_t *r_rewards, uint8_t *r_terminals,
                               hipStream_t st) {
  (void)hipGetLastError();
  int grid = grid_for((sh.n * sh.n_cells / 4 + WG - 1) / WG, sh.max_grid);
  replay_store_kernel<<<dim3(grid), dim3(WG), 0, st>>>(sh.boards, sh.pitch, sh.n_cells, sh.n, sh.rec, actions, phase, cheat, head,
                                                      head_dev, states, successors, r_actions, r_rewards, r_terminals);
  return hipGetLastError();
}

size_t dqn_sgd_lds_bytes(int n_cells, int n_hidden) {
  const size_t kp = (size_t)((n_cells + 3) & ~3), h = (size_t)n_hidden;
  return sizeof(float) * (4 * LB * h + h * (size_t)weight_stride((int)h) + 8 * h + 4 * h + 8 + 2 * LB * 4 + LB + 32 + LB + 3 * LB) + 2 * LB * kp + 64;
}

// the handle's scratch block: AdamHeader + the flat gradient (two-launch Adam); in the experiment build also the four-workgroup
// kernel's barrier words and partial gradients, whichever is larger
size_t dqn_sgd_scratch_bytes(int n_cells, int n_hidden) {
  const size_t p = (size_t)n_hidden * n_cells + n_hidden + (size_t)n_hidden * n_hidden + n_hidden + 4 * (size_t)n_hidden + 4;
  size_t need = sizeof(AdamHeader) + sizeof(float) * p;
#ifdef SGK_DQN_MULTI_WG
  need = std::max(need, sizeof(LearnShared) + sizeof(float) * GW * p);
#endif
  return (need + 255) & ~(size_t)255;
}

#ifdef SGK_DQN_MULTI_WG
// the four-workgroup form: LDS its workgroups need (0: no instantiation for this shape)
template <int K0, int H>
static hipError_t launch_multi(const LearnArgs &a, void *scratch, int device, hipStream_t st) {
  constexpr size_t lds = multi_lds_bytes<K0, H>();
  static std::atomic<unsigned long long> opted_in{0};
  if (!((opted_in.load() >> (device & 63)) & 1ull)) {
    hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void *>(&dqn_sgd_multi_kernel<K0, H>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (ae != hipSuccess) return ae;
    opted_in.fetch_or(1ull << (device & 63));
  }
  LearnShared *sh = reinterpret_cast<LearnShared *>(scratch);
  float *partials = reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(scratch) + sizeof(LearnShared));
  dqn_sgd_multi_kernel<K0, H><<<dim3(GW), dim3(LWG), lds, st>>>(a, sh, partials);
  return hipGetLastError();
}
#endif

hipError_t launch_dqn_sgd(const Shard &sh, const DqnLearner &L, hipStream_t st) {
  (void)hipGetLastError();
  if (L.n_hidden > 128 || L.n_hidden < 4 || (L.n_hidden & 3) || L.batch < 1 || L.batch > LB || sh.n_cells > 64) return hipErrorInvalidValue;
  if ((L.n_hidden / 4) * (L.n_hidden / 4) + L.n_hidden > LWG) return hipErrorInvalidValue;  // tile lanes + bias lanes
  const size_t lds = dqn_sgd_lds_bytes(sh.n_cells, L.n_hidden);
  if (lds > 160u * 1024u) return hipErrorInvalidValue;
  LearnArgs a;
  a.states = L.states; a.successors = L.successors; a.actions = L.actions; a.rewards = L.rewards; a.terminals = L.terminals;
  a.n_envs = sh.n; a.total = (int64_t)L.slices_filled * sh.n; a.n_cells = sh.n_cells;
  a.w1 = L.w1; a.b1 = L.b1; a.w2 = L.w2; a.b2 = L.b2; a.w3 = L.w3; a.b3 = L.b3;
  a.w1t = L.w1t; a.w2t = L.w2t; a.w3t = L.w3t;
  for (int i = 0; i < 6; ++i) { a.m[i] = L.m[i]; a.v[i] = L.v[i]; a.vmax[i] = L.vmax[i]; }
  a.tw1t = L.tw1t; a.tb1 = L.tb1; a.tw2t = L.tw2t; a.tb2 = L.tb2; a.tw3 = L.tw3; a.tb3 = L.tb3;
  a.step = L.step; a.loss_out = L.loss_out; a.n_hidden = L.n_hidden; a.batch = L.batch;
  a.loss_mode = L.loss_mode; a.rows = L.rows; a.rows_out = L.rows_out;
  a.adam_scratch = (L.scratch && !L.multi_wg) ? reinterpret_cast<float *>(L.scratch) : nullptr;
  a.reward_scale = sh.rules_host.reward_scale;
  a.seed = sh.seed;
  a.lr = (float)L.lr; a.beta1 = (float)L.beta1; a.beta2 = (float)L.beta2; a.eps = (float)L.eps; a.discount = (float)L.discount;
  a.max_norm = (float)L.max_grad_norm;
#ifdef SGK_DQN_MULTI_WG
  /* (a shape whose four matrices do not fit a CU's LDS at once -- 7 x 9 boards with 100 units -- keeps the one-workgroup kernel) */
#define SGK_SGD_TRY_MULTI(K0V, HV)                                                                                         \
  do {                                                                                                                     \
    if constexpr (multi_lds_bytes<K0V, HV>() <= 160u * 1024u) {                                                            \
      if (L.scratch && L.multi_wg) return launch_multi<K0V, HV>(a, L.scratch, sh.device, st);                             \
    }                                                                                                                      \
  } while (0)
#else
#define SGK_SGD_TRY_MULTI(K0V, HV) do { } while (0)
#endif
#define SGK_SGD_LAUNCH(K0V, HV)                                                                                            \
  do {                                                                                                                     \
    SGK_SGD_TRY_MULTI(K0V, HV);                                                                                            \
    static std::atomic<unsigned long long> opted_in{0};                                                                                \
    if (!((opted_in.load() >> (sh.device & 63)) & 1ull)) {                                                                        \
      hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void *>(&dqn_sgd_kernel<K0V, HV>),                        \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                         \
      if (ae != hipSuccess) return ae;                                                                                     \
      opted_in.fetch_or(1ull << (sh.device & 63));                                                                         \
    }                                                                                                                      \
    dqn_sgd_kernel<K0V, HV><<<dim3(1), dim3(LWG), lds, st>>>(a);                                                           \
    if (a.adam_scratch && !L.reset_store) {                                                                                \
      hipError_t le = hipGetLastError();                                                                                   \
      if (le != hipSuccess) return le;                                                                                     \
      dqn_adam_kernel<K0V, HV><<<dim3((ParamMap<K0V, HV>::P + 255) / 256), dim3(256), 0, st>>>(a);                         \
    }                                                                                                                      \
  } while (0)
#define SGK_SGD_LAUNCH_K(K0V)                                                                                              \
  do {                                                                                                                     \
    if (L.n_hidden == 100) SGK_SGD_LAUNCH(K0V, 100);                                                                       \
    else if (L.n_hidden == 64) SGK_SGD_LAUNCH(K0V, 64);                                                                    \
    else return hipErrorInvalidValue;                                                                                      \
  } while (0)
  switch (sh.n_cells) {  // the shapes with an instantiation: the four levels x {64, 100} hidden units
  case 25: SGK_SGD_LAUNCH_K(25); break;
  case 30: SGK_SGD_LAUNCH_K(30); break;
  case 36: SGK_SGD_LAUNCH_K(36); break;
  case 48: SGK_SGD_LAUNCH_K(48); break;
  case 49: SGK_SGD_LAUNCH_K(49); break;
  case 56: SGK_SGD_LAUNCH_K(56); break;
  case 63: SGK_SGD_LAUNCH_K(63); break;
  default: return hipErrorInvalidValue;
  }
#undef SGK_SGD_LAUNCH_K
#undef SGK_SGD_LAUNCH
#undef SGK_SGD_TRY_MULTI
  if (L.reset_store) {  // Adam + reset_done + the next transitions' states in ONE launch behind the SGD kernel
    if (!a.adam_scratch) return hipErrorInvalidValue;  // (the two-launch form only: the gradient must leave the SGD kernel)
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) return le;
    ResetStoreArgs r;
    r.rules = sh.rules_dev; r.state = sh.state; r.boards = sh.boards; r.n_resets = sh.n_resets; r.aux = sh.aux;
    r.n = sh.n; r.seed = sh.seed; r.env_base = sh.env_base;
    r.mode_flags = 1 | ((L.rs_flags & SGK_F_NO_BOARDS) ? 4 : 0);
    r.st = make_store(sh, L.rs_states_ring, nullptr, nullptr, nullptr, L.rs_slice, L.rs_slice_dev, L.rs_ring, 0);
    const int reset_blocks = grid_for((sh.n + WG - 1) / WG, sh.max_grid);
#define SGK_ADAM_RESET(HV)                                                                                                 \
  do {                                                                                                                     \
    const int adam_blocks = (ParamMap<Geom<E>::NC, HV>::P + 255) / 256;                                                    \
    dqn_adam_reset_kernel<HV, E, L><<<dim3(adam_blocks + reset_blocks), dim3(256), 0, st>>>(a, adam_blocks, r);            \
  } while (0)
    const int nh = a.n_hidden;
    SGK_DISPATCH_ENV_LAYOUT(sh.env_id, sh.layout, {
      if (Geom<E>::NC != sh.n_cells) return hipErrorInvalidValue;
      if (nh == 100) SGK_ADAM_RESET(100);
      else SGK_ADAM_RESET(64);
    });
#undef SGK_ADAM_RESET
  }
  return hipGetLastError();
}

#ifdef SGK_LEARN_TIMELINE
extern "C" __attribute__((visibility("default"))) int sgk_debug_learn_stamps(unsigned long long *out32) {
  (void)hipDeviceSynchronize();
  return (int)hipMemcpyFromSymbol(out32, HIP_SYMBOL(learn_stamps), sizeof(unsigned long long) * 32);
}
#endif

size_t ppo_epochs_lds_bytes(int n_cells, int n_hidden) { return ppo_lds_bytes((n_cells + 3) & ~3, n_hidden); }

hipError_t launch_ppo_epochs(const Shard &sh, const PpoLearner &P, hipStream_t st) {
  (void)hipGetLastError();
  if (P.batch < 2 || P.batch > LB || P.n_epochs < 1 || P.horizon < 1 || P.n_trajectories < 1 || P.n_trajectories >= (1ll << 31))
    return hipErrorInvalidValue;
  const size_t lds = ppo_epochs_lds_bytes(sh.n_cells, P.n_hidden);
  if (lds > 160u * 1024u) return hipErrorInvalidValue;
  PpoArgs a;
  a.states = P.states; a.actions = P.actions; a.returns = P.returns; a.lengths = P.lengths;
  a.T = P.horizon; a.N = P.n_trajectories;
  a.w1 = P.w1; a.b1 = P.b1; a.w2 = P.w2; a.b2 = P.b2; a.wa = P.wa; a.ba = P.ba; a.wc = P.wc; a.bc = P.bc;
  a.w1t = P.w1t; a.w2t = P.w2t;
  for (int i = 0; i < 8; ++i) { a.m[i] = P.m[i]; a.v[i] = P.v[i]; }
  a.ow1t = P.ow1t; a.ob1 = P.ob1; a.ow2t = P.ow2t; a.ob2 = P.ob2; a.owa = P.owa; a.oba = P.oba;
  a.step = P.step; a.stats_out = P.stats_out; a.rows = P.rows; a.rows_out = P.rows_out;
  a.batch = P.batch; a.n_epochs = P.n_epochs; a.seed = sh.seed;
  a.lr = (float)P.lr; a.beta1 = (float)P.beta1; a.beta2 = (float)P.beta2; a.eps = (float)P.eps;
  a.clipping = (float)P.clipping; a.critic_coeff = (float)P.critic_coeff; a.entropy_bonus = (float)P.entropy_bonus;
#define SGK_PPO_LAUNCH(K0V, HV)                                                                                            \
  do {                                                                                                                     \
    static std::atomic<unsigned long long> opted_in{0};                                                                                \
    if (!((opted_in.load() >> (sh.device & 63)) & 1ull)) {                                                                        \
      hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void *>(&ppo_epochs_kernel<K0V, HV>),                     \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                         \
      if (ae != hipSuccess) return ae;                                                                                     \
      opted_in.fetch_or(1ull << (sh.device & 63));                                                                         \
    }                                                                                                                      \
    ppo_epochs_kernel<K0V, HV><<<dim3(1), dim3(LWG), lds, st>>>(a);                                                        \
  } while (0)
#define SGK_PPO_LAUNCH_K(K0V)                                                                                              \
  do {                                                                                                                     \
    if (P.n_hidden == 100) SGK_PPO_LAUNCH(K0V, 100);                                                                       \
    else if (P.n_hidden == 64) SGK_PPO_LAUNCH(K0V, 64);                                                                    \
    else return hipErrorInvalidValue;                                                                                      \
  } while (0)
  switch (sh.n_cells) {
  case 25: SGK_PPO_LAUNCH_K(25); break;
  case 30: SGK_PPO_LAUNCH_K(30); break;
  case 36: SGK_PPO_LAUNCH_K(36); break;
  case 48: SGK_PPO_LAUNCH_K(48); break;
  case 49: SGK_PPO_LAUNCH_K(49); break;
  case 56: SGK_PPO_LAUNCH_K(56); break;
  case 63: SGK_PPO_LAUNCH_K(63); break;
  default: return hipErrorInvalidValue;
  }
#undef SGK_PPO_LAUNCH_K
#undef SGK_PPO_LAUNCH
  return hipGetLastError();
}


}  // namespace sgk
