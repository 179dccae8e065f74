// sgk_draws.h -- the action draws on four scores / logits shared by the network-facing kernels (sgk_policy.hip, sgk_convq.hip)
#pragma once
#include "sgk_device.h"

namespace sgk {

// The action of one env from its four scores, with the counter RNG keyed by the global env index `ge` and a draw index.
// MODE 0 -- DeepQAgent.act_explore (reference value.py:94-111): greedy = argmax of the 4 action scores, then a draw from
//   Categorical(eps/4 everywhere + (1 - eps) on the greedy action) = with probability eps a uniform action (the greedy one
//   included), else the greedy one. Philox stream 2: ctr = {env_lo, env_hi, draw, 2}; u = numpy's 53-bit uniform of x0,x1;
//   uniform action = x2 & 3.
// MODE 1 -- PPOBaseAgent.act_explore (reference policy_base.py:54-64): Categorical(logits = scores).sample(), by inverse
//   CDF on the unnormalised float32 weights e_i = expf(s_i - max s): action = first i with u * (e_0+..+e_3) < e_0+..+e_i
//   (partial sums in float32 left to right, the comparison in double). Philox stream 3, u from x0,x1 as above.
template <int MODE>
__device__ __forceinline__ void draw_block(uint64_t ge, uint64_t draw, uint64_t seed, double &u, uint32_t &x2) {
  uint32_t x[4];
  philox4x32_10((uint32_t)ge, (uint32_t)(ge >> 32), (uint32_t)draw, MODE == 0 ? 2u : 3u, (uint32_t)seed,
                (uint32_t)(seed >> 32), x);
  u = uniform53(x[0], x[1]);
  x2 = x[2];
}

// the draw (which does not depend on the scores: the fused kernels compute it in the shadow of the MFMAs) applied to the scores
template <int MODE>
__device__ __forceinline__ int select_action(float s0, float s1, float s2, float s3, double u, uint32_t x2, double eps) {
  if (MODE == 0) {
    int best = 0;
    float bv = s0;
    if (s1 > bv) { bv = s1; best = 1; }
    if (s2 > bv) { bv = s2; best = 2; }
    if (s3 > bv) { bv = s3; best = 3; }
    if (u < eps) best = (int)(x2 & 3u);
    return best;
  }
  const float m = fmaxf(fmaxf(s0, s1), fmaxf(s2, s3));
  const float e0 = expf(__fsub_rn(s0, m)), e1 = expf(__fsub_rn(s1, m)), e2 = expf(__fsub_rn(s2, m)), e3 = expf(__fsub_rn(s3, m));
  const float c1 = __fadd_rn(e0, e1), c2 = __fadd_rn(c1, e2), c3 = __fadd_rn(c2, e3);
  const double target = __dmul_rn(u, (double)c3);
  return target < (double)e0 ? 0 : (target < (double)c1 ? 1 : (target < (double)c2 ? 2 : 3));
}

template <int MODE>
__device__ __forceinline__ int pick_action(float s0, float s1, float s2, float s3, uint64_t ge, uint64_t draw, uint64_t seed,
                                           double eps) {
  double u;
  uint32_t x2;
  draw_block<MODE>(ge, draw, seed, u, x2);
  return select_action<MODE>(s0, s1, s2, s3, u, x2, eps);
}

}  // namespace sgk
