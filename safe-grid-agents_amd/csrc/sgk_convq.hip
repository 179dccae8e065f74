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
// matrix peak; 57 TFLOP/s at 1 M boards); the lockstep step around it 30 us against 390 through torch. What is left is instruction
// issue and latency, not the MFMA: per wave and pass 198 MFMAs (1 584 cycles) beside ~390 other vector instructions (epilogues, the
// fifth channel, the Philox draw; the ~60 address adds ds_read2's 8-bit offsets first needed are gone -- the planes sit at LDS address
// 0 and every group of planes has its own base register -- which bought 1-3 %: the waves wait more than they issue).
// The epsilon-greedy draw is sgk_epsilon_greedy's (Philox stream 2, keyed by global env index). fp32 with another summation order than
// MIOpen / rocBLAS: scores agree with the torch module to fp32 tolerance (tests/test_gpu_convq.py: rtol 1e-4).
#include "sgk_convq.h"

namespace sgk {

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
  float *Lf = reinterpret_cast<float *>(convq_smem);
  float *WL = Lf + G::O_WL, *act = Lf + G::O_ACT;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (eps_ptr) eps = *eps_ptr;  // device-resident scalars: the launch can be replayed from a graph
  if (draw_ptr) draw = *draw_ptr;
  CqLane<G> L;
  cq_setup<G, WW, C>(L, WL, act, w1r, w2r, whr, wlr);
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
    const int hz = (int)(pass >> 44);  // (always 0: see cq_network)
    // ---- the boards -> plane 0 (float); the next pass's bytes requested ----
#pragma unroll
    for (int u = 0; u < G::NB; ++u)
      if (b_env[u] < G::ENVS) act[b_lds[u]] = (float)cur[u];
    if (pass + gridDim.x < n_pass) request_boards(pass + gridDim.x);
    __syncthreads();
    cq_network<G, C>(L, act, WL, hz, w1r, b1r, w2r, b2r, wbr, bbr, whr, bhr);
    // ---- the four outputs of an env and its draw, by ONE wave (the other three go on to the next pass's boards) ----
    if (wave == 0) {
      constexpr int NIT = (G::ENVS * 4 + 63) / 64;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int idx = it * 64 + lane;
        float q0, q1, q2, q3;
        cq_outputs<G>(act, blr, idx, q0, q1, q2, q3);
        const int64_t env = env0 + (idx >> 2);
        if (idx < G::ENVS * 4 && (idx & 3) == 0 && env < n) {
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
