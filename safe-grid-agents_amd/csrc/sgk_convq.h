// sgk_convq.h -- the pieces shared by the conv body's kernels: sgk_convq.hip (forward + draw of one lockstep step, the design notes are
// there) and sgk_convq_rollout.hip (n_steps of forward + draw + env.step in one launch).
#pragma once
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
  // (the activations first: their LDS address is then the lane's own base register alone, and a row's constant offset fits the 8-bit
  // dword offsets of ds_read2_b32 -- behind 3 KB of linear weights every row read needed a v_add_u32 first)
  static constexpr int O_ACT = 0, O_WL = (ENVS * ENV_F + 3) & ~3;
  static constexpr int PG = (255 - 2 * PW - 2) / PL + 1, NPG = (PLANES + PG - 1) / PG;  // planes per base register (ds_read2's offsets reach 255 dwords), base registers
  static constexpr int CENTRE = PW + 1;                  // slot r's own cell relative to its window's top-left corner
  static constexpr size_t lds_bytes = sizeof(float) * (size_t)(O_WL + SE * WLR);
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
__device__ __forceinline__ void cq_taps(const float *act, const int (&base)[G::GPW][G::NPG], const float (&a)[G::MT][KC], const float *__restrict__ wrem,
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
      // (addressed from the base register of the plane's group of four: the offset then fits ds_read2_b32's 8 bits of dwords and no
      // address is computed inside the loop -- 34 + 29 v_add_u32 per pass otherwise)
      constexpr int pg = (IN_PLANE + ci) / G::PG, off = ((IN_PLANE + ci) % G::PG) * G::PL + dy * G::PW;
      static_assert(off + 2 < 256, "ds_read2 offset");
#pragma unroll
      for (int j = 0; j < G::GPW; ++j) {
        pr[ci & 1][j][dy] = cq_f2{act[base[j][pg] + off], act[base[j][pg] + off + 1]};
        sg[ci & 1][j][dy] = act[base[j][pg] + off + 2];
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


// what a lane keeps for the whole launch: the three convolutions' A operands and its slots
template <class G>
struct CqLane {
  float a1[G::MT][1], a2[G::MT][G::KC2], ah[G::MT][G::KC2];
  int base[G::GPW], wlrow[G::GPW];
  int pbase[G::GPW][G::NPG];  // base + the first plane of every group of four
  bool interior[G::GPW];
};

// weights into A-operand registers, the linear head's per-slot rows into LDS, the activation planes zeroed, this lane's slots
template <class G, int WW, int C>
__device__ __forceinline__ void cq_setup(CqLane<G> &L, float *WL, float *act, const float *__restrict__ w1r, const float *__restrict__ w2r,
                                         const float *__restrict__ whr, const float *__restrict__ wlr) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  {
    const int i = lane & 3, kk = lane >> 2;
#pragma unroll
    for (int m = 0; m < G::MT; ++m) {
      const int c = 4 * m + i;
      L.a1[m][0] = (c < G::CM && kk < G::K1) ? w1r[c * 9 + kk] : 0.0f;
#pragma unroll
      for (int q = 0; q < G::KC2; ++q) {
        const int k = 16 * q + kk;
        const bool live = c < G::CM && k < G::K2;
        L.a2[m][q] = live ? w2r[c * G::K2 + k] : 0.0f;  // [c][ci][dy][dx] flattened = c * 9 C + k
        L.ah[m][q] = live ? whr[c * G::K2 + k] : 0.0f;
      }
    }
  }
  // the linear head's weights per slot: WL[r][action][channel], zero rows for the border slots; the planes zeroed (borders stay
  // zero: nothing is ever stored there)
  for (int i = t; i < G::SE * G::WLR; i += CQ_WG) {
    const int r = i / G::WLR, ac = i - r * G::WLR, a = ac / C, c = ac - a * C;
    const int y = r / G::PW, x = r - y * G::PW;
    WL[i] = x < WW ? wlr[a * G::NF + c * G::NC + y * WW + x] : 0.0f;
  }
  for (int i = t; i < G::ENVS * G::ENV_F; i += CQ_WG) act[i] = 0.0f;
  // this lane's slots (the same in every pass): group g = wave + 4 j, slot s = 64 g + lane = (env e, r)
#pragma unroll
  for (int j = 0; j < G::GPW; ++j) {
    const int s = 64 * (wave + 4 * j) + lane;
    const bool valid = s < G::ENVS * G::SE;
    const int e = valid ? s / G::SE : 0, r = valid ? s - e * G::SE : 0;
    L.base[j] = e * G::ENV_F + r;  // the window's top-left corner in plane 0
#pragma unroll
    for (int g = 0; g < G::NPG; ++g) {
      L.pbase[j][g] = L.base[j] + g * G::PG * G::PL;
      asm volatile("" : "+v"(L.pbase[j][g]));  // (opaque: otherwise the compiler folds it back into base + constant and re-adds it per row)
    }
    L.wlrow[j] = r * G::WLR;
    L.interior[j] = valid && (r % G::PW) < WW;
  }
}

// the network on the boards in plane 0: three convolutions, the linear head's per-slot products left in planes 1 .. 4. Starts behind a
// barrier that made plane 0 visible, ends with the barrier that makes the products visible. hz: always 0, but not to the compiler (the
// fifth channel's scalar weights are then loaded per pass into scalar registers instead of being hoisted into ~100 vector registers
// for the whole launch, which halves the occupancy).
template <class G, int C>
__device__ __forceinline__ void cq_network(const CqLane<G> &L, float *act, const float *WL, int hz, const float *__restrict__ w1r,
                                           const float *__restrict__ b1r, const float *__restrict__ w2r, const float *__restrict__ b2r,
                                           const float *__restrict__ wbr, const float *__restrict__ bbr, const float *__restrict__ whr,
                                           const float *__restrict__ bhr) {
  // ---- conv3x3 1 -> C, ReLU: planes 1 .. C (border slots are never written: they stay zero) ----
  {
    cq_f4 acc[G::GPW][G::MT] = {};
    float racc[G::GPW] = {};
    cq_taps<G, G::K1, 1, 0>(act, L.pbase, L.a1, w1r + G::CM * G::K1 + hz, acc, racc);
#pragma unroll
    for (int j = 0; j < G::GPW; ++j)
      if (L.interior[j]) {
#pragma unroll
        for (int c = 0; c < C; ++c) act[L.base[j] + (1 + c) * G::PL + G::CENTRE] = fmaxf(cq_channel<G>(acc[j], racc[j], c) + b1r[c + hz], 0.0f);
      }
  }
  __syncthreads();
  // ---- conv3x3 C -> C, ReLU, + the 1 x 1 bottleneck of the board: planes C + 1 .. 2 C (the trunk) ----
  {
    cq_f4 acc[G::GPW][G::MT] = {};
    float racc[G::GPW] = {};
    cq_taps<G, G::K2, G::KC2, 1>(act, L.pbase, L.a2, w2r + G::CM * G::K2 + hz, acc, racc);
#pragma unroll
    for (int j = 0; j < G::GPW; ++j)
      if (L.interior[j]) {
        const float xin = act[L.base[j] + G::CENTRE];
#pragma unroll
        for (int c = 0; c < C; ++c)
          act[L.base[j] + (1 + C + c) * G::PL + G::CENTRE] =
              fmaxf(cq_channel<G>(acc[j], racc[j], c) + b2r[c + hz], 0.0f) + fmaf(wbr[c + hz], xin, bbr[c + hz]);
      }
  }
  __syncthreads();
  // ---- head conv3x3 C -> C, ReLU, times this slot's rows of the linear layer: four per-slot products into planes 1 .. 4 ----
  {
    cq_f4 acc[G::GPW][G::MT] = {};
    float racc[G::GPW] = {};
    cq_taps<G, G::K2, G::KC2, 1 + C>(act, L.pbase, L.ah, whr + G::CM * G::K2 + hz, acc, racc);
#pragma unroll
    for (int j = 0; j < G::GPW; ++j)
      if (L.interior[j]) {
        float wr[G::WLR];
        const cq_f4 *wp = reinterpret_cast<const cq_f4 *>(WL + L.wlrow[j]);
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
          act[L.base[j] + (1 + a) * G::PL + G::CENTRE] = sacc;
        }
      }
  }
  __syncthreads();
}

// wave 0, lane = (env e = idx / 4, action a = idx % 4) of the pass: the env's four outputs (every lane of the env's quad gets all four),
// summed from its per-slot products in slot order (border slots hold zeros)
template <class G>
__device__ __forceinline__ void cq_outputs(const float *act, const float *__restrict__ blr, int idx, float &q0, float &q1, float &q2, float &q3) {
  const int lane = threadIdx.x & 63;
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
  q0 = __shfl(tot, l0);
  q1 = __shfl(tot, l0 + 1);
  q2 = __shfl(tot, l0 + 2);
  q3 = __shfl(tot, l0 + 3);
}

}  // namespace sgk
