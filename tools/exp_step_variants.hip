// exp_step_variants.hip -- standalone probe (not part of the product): the product's per-step kernel (BoatRace, COMPACT, random
// actions; the body of sgk::step_kernel with the product's own device code from sgk_device.h) with pieces switched off one at a
// time, replayed as a hipGraph chain of 100 dependent launches like sgk_step_random does. Says where the launch's time goes
// beyond the synthetic model of tools/exp_step_latency.hip.
//   knob bits: 1 no episode bookkeeping (metrics accumulators, last_return / last_perf / n_episodes)   2 no step-record store
//              4 no board tile   8 no state-word store   16 no Philox (action = lane & 3)   32 no auto-reset branch at all
//              64 record: plain store (product: sc1)   1024 record: non-temporal   128 board tile: plain stores (product: sc1)
//              256 board tile: non-temporal   512 state word: sc1 (product: plain)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I safe-grid-agents_amd/csrc tools/exp_step_variants.hip \
//         safe-grid-agents_amd/csrc/sgk_rules.cpp -o /tmp/sv && /tmp/sv [n_envs]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "sgk_step.hip"  // the product's own kernels (step_kernel<...>), for the "library kernel" rows

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

using namespace sgk;

template <int KN, int WGT, int ENV = SGK_BOAT_RACE>
__global__ __launch_bounds__(WGT) void variant(StepArgs a) {
  constexpr int NC = Geom<ENV>::NC;
  __shared__ WaveRulesImage rules_images[WGT / 64];
  __shared__ __attribute__((aligned(16))) uint8_t tile_images[WGT / 64][64 * NC];
  const int lane = threadIdx.x & 63, wave = wave_index();
  const int64_t n_wt = (a.n + 63) / 64;
  const int64_t wt0 = (int64_t)blockIdx.x * (WGT / 64) + wave, wstride = (int64_t)gridDim.x * (WGT / 64);
  keep_in_sgprs(a.rec, a.boards, a.last_return, a.last_perf);
  keep_in_sgprs(a.n_episodes, a.n_resets, a.metrics, a.aux);
  keep_in_sgprs(a.seed, a.env_base, a.t, a.flags);
  const uint64_t *t_word = a.t_ptr ? a.t_ptr : reinterpret_cast<const uint64_t *>(a.rules);
  const uint64_t t_base = *t_word;
  uint64_t w_cur = 0;
  {
    const int64_t e0 = wt0 * 64 + lane;
    const int64_t e0c = e0 < a.n ? e0 : a.n - 1;
    w_cur = a.state[e0c];
  }
  WaveRulesLoad rules_load;
  rules_load.request(a.rules);
  WaveTileLds<ENV, NC> W;
  W.bind(tile_images[wave]);
  typename WaveTileLds<ENV, NC>::Blank blank;
  W.request_blank(blank, a.rules);
  rules_load.commit(rules_images[wave]);
  const SgkRules &R = rules_images[wave].r;
  const uint64_t t_now = a.t + (a.t_ptr ? t_base : 0ull);
  EpisodeAcc acc;
  acc_init(acc);
  for (int64_t wt = wt0; wt < n_wt; wt += wstride) {
    const int64_t env = wt * 64 + lane;
    const bool valid = env < a.n;
    EnvState s = unpack_state(w_cur);
    {
      const int64_t nt = wt + wstride;
      const int64_t ne = nt * 64 + lane;
      const bool nv = nt < n_wt && ne < a.n;
      w_cur = nv ? a.state[ne] : 0;
    }
    if (!valid) s = initial_state(R);
    int action = lane & 3;
    if (!(KN & 16)) {
      uint64_t ge = a.env_base + (uint64_t)env;
      uint32_t x[4];
      philox4x32_10((uint32_t)ge, (uint32_t)(ge >> 32), (uint32_t)(t_now >> 6), 0u, (uint32_t)a.seed, (uint32_t)(a.seed >> 32), x);
      action = action_from_block(x, t_now);
    }
    uint32_t rec;
    if (KN & 1) {
      int r_obs = 0, r_hid = 0, term = 0;
      bool finished = false;
      if (valid && !s.over) {
        transition<ENV>(R, s, action, r_obs, r_hid, term, nullptr);
        s.frame += 1;
        s.ret += r_obs;
        s.hid += r_hid;
        finished = term || s.frame >= R.max_iterations;
      }
      rec = pack_rec(r_obs, r_hid, (valid && (s.over || finished)) ? 1 : 0, action);
      if (!(KN & 32) && finished) s = initial_state(R);
    } else {
      step_one<ENV>(R, a, env, valid, action, s, rec, acc);
    }
    if (valid) {
      if (!(KN & 8)) {
        if (KN & 512) __hip_atomic_store(&a.state[env], pack_state(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else a.state[env] = pack_state(s);
      }
      if (!(KN & 2)) {
        if (KN & 64) a.rec[env] = rec;
        else if (KN & 1024) __builtin_nontemporal_store(rec, &a.rec[env]);
        else __hip_atomic_store(&a.rec[env], rec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (!(KN & 4)) {
      W.draw_from_blank(blank, R, sprite_info<ENV>(R, s));
      if (KN & 128) W.template flush<0>(a.boards + wt * 64 * NC);
      else if (KN & 256) W.template flush<2>(a.boards + wt * 64 * NC);
      else W.flush(a.boards + wt * 64 * NC);
    }
  }
  if (!(KN & 1)) acc_flush(acc, a.metrics);
}

__global__ void bump_counter(uint64_t *ctr, uint64_t v) { *ctr += v; }  // (as sgk_step_random's graphs end: the next replay draws new actions)

template <int KN, int WGT, int ENV = SGK_BOAT_RACE>
static void run(const char *label, StepArgs a, hipStream_t st, uint64_t *t_dev) {
  const int chain = 100, reps = 30;
  const int grid = (int)((a.n + WGT - 1) / WGT);
  hipGraph_t g = nullptr;
  hipGraphExec_t ge = nullptr;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int i = 0; i < chain; ++i) {
    a.t = (uint64_t)i;
    a.t_ptr = t_dev;
    hipLaunchKernelGGL((variant<KN, WGT, ENV>), dim3(grid), dim3(WGT), 0, st, a);
  }
  hipLaunchKernelGGL(bump_counter, dim3(1), dim3(1), 0, st, t_dev, (uint64_t)chain);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<float> ms;
  for (int r = 0; r < reps + 2; ++r) {
    CK(hipEventRecord(e0, st));
    CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float t;
    CK(hipEventElapsedTime(&t, e0, e1));
    if (r >= 2) ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  printf("%-64s wg %3d  median %6.3f us  best %6.3f us per launch\n", label, WGT, ms[ms.size() / 2] * 1e3 / chain, ms[0] * 1e3 / chain);
  fflush(stdout);
  CK(hipGraphExecDestroy(ge));
  CK(hipGraphDestroy(g));
}

template <int ENV, bool SMALL>
static void run_product(const char *label, StepArgs a, hipStream_t st, uint64_t *t_dev) {
  const int chain = 100, reps = 30;
  const int wgt = SMALL ? 64 : 256;
  const int grid = (int)((a.n + wgt - 1) / wgt);
  hipGraph_t g = nullptr;
  hipGraphExec_t ge = nullptr;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int i = 0; i < chain; ++i) {
    a.t = (uint64_t)i;
    a.t_ptr = t_dev;
    hipLaunchKernelGGL((sgk::step_kernel<ENV, SGK_LAYOUT_COMPACT, true, SMALL>), dim3(grid), dim3(wgt), 0, st, a, sgk::StepStore{});
  }
  hipLaunchKernelGGL(bump_counter, dim3(1), dim3(1), 0, st, t_dev, (uint64_t)chain);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<float> ms;
  for (int r = 0; r < reps + 2; ++r) {
    CK(hipEventRecord(e0, st));
    CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float t;
    CK(hipEventElapsedTime(&t, e0, e1));
    if (r >= 2) ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  printf("%-64s wg %3d  median %6.3f us  best %6.3f us per launch\n", label, wgt, ms[ms.size() / 2] * 1e3 / chain, ms[0] * 1e3 / chain);
  fflush(stdout);
  CK(hipGraphExecDestroy(ge));
  CK(hipGraphDestroy(g));
}

__global__ void read_slab(const long long *slab, long long *out) {
  long long s = 0;
  for (int i = 0; i < SGK_METRIC_SLOTS; ++i) s += slab[(size_t)i * SGK_METRICS_LEN + SGK_M_EPISODES];
  *out = s;
}

__global__ void init_state(uint64_t *state, int64_t n, uint64_t w) {
  for (int64_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) state[i] = w;
}

int main(int argc, char **argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 65536;
  const int64_t n_pad = (n + 255) / 256 * 256;
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  SgkRules R;
  if (sgk_build_rules(SGK_BOAT_RACE, &R) != 0) return 1;
  std::vector<uint8_t> image(SGK_RULES_DEV_BYTES, 0);
  memcpy(image.data(), &R, sizeof(R));
  for (int e = 0; e < 64; ++e) memcpy(image.data() + SGK_RULES_IMAGE_BYTES + e * R.n_cells, R.templ, (size_t)R.n_cells);
  StepArgs a;
  memset(&a, 0, sizeof(a));
  void *rd;
  CK(hipMalloc(&rd, SGK_RULES_DEV_BYTES));
  CK(hipMemcpy(rd, image.data(), image.size(), hipMemcpyHostToDevice));
  a.rules = (const SgkRules *)rd;
  CK(hipMalloc(&a.state, n_pad * 8));
  CK(hipMalloc(&a.rec, n_pad * 4));
  CK(hipMalloc(&a.boards, n_pad * R.n_cells));
  CK(hipMalloc(&a.last_return, n_pad * 4));
  CK(hipMalloc(&a.last_perf, n_pad * 4));
  CK(hipMalloc(&a.n_episodes, n_pad * 4));
  CK(hipMalloc(&a.n_resets, n_pad * 4));
  CK(hipMemset(a.n_episodes, 0, n_pad * 4));
  CK(hipMalloc(&a.metrics, sizeof(long long) * SGK_METRICS_LEN * SGK_METRIC_SLOTS));
  CK(hipMemset(a.metrics, 0, sizeof(long long) * SGK_METRICS_LEN * SGK_METRIC_SLOTS));
  EnvState s0 = initial_state(R);
  hipLaunchKernelGGL(init_state, dim3(256), dim3(256), 0, st, a.state, n_pad, pack_state(s0));
  uint64_t *t_dev;
  CK(hipMalloc(&t_dev, 8));
  CK(hipMemset(t_dev, 0, 8));
  a.n = n;
  a.seed = 0x5AFE;
  a.flags = SGK_F_AUTO_RESET;
  CK(hipStreamSynchronize(st));
  printf("n = %lld BoatRace envs, the product's step kernel body with pieces switched off; chains of 100 dependent launches\n", (long long)n);
  for (int pass = 0; pass < 2; ++pass) {
    run<0, 256>("product body", a, st, t_dev);
    run<0, 64>("product body", a, st, t_dev);
    run<1, 256>("- episode bookkeeping", a, st, t_dev);
    run<1 | 32, 256>("- episode bookkeeping - reset branch", a, st, t_dev);
    run<2, 256>("- record store", a, st, t_dev);
    run<4, 256>("- board tile", a, st, t_dev);
    run<8, 256>("- state store", a, st, t_dev);
    run<16, 256>("- philox", a, st, t_dev);
    run<1 | 4, 256>("- bookkeeping - board", a, st, t_dev);
    run<1 | 2 | 4, 256>("- bookkeeping - board - record", a, st, t_dev);
    run<1 | 2 | 4 | 16, 256>("- bookkeeping - board - record - philox", a, st, t_dev);
    run<1 | 2 | 4 | 8 | 16 | 32, 256>("- everything (rules + state load + transition only)", a, st, t_dev);
    run<64, 256>("record as a plain store", a, st, t_dev);
    run<1024, 256>("record as a non-temporal store", a, st, t_dev);
    run<128, 256>("board tile as plain stores", a, st, t_dev);
    run<256, 256>("board tile as non-temporal stores", a, st, t_dev);
    run<64 | 128, 256>("record + board tile plain", a, st, t_dev);
    run<512, 256>("state word sc1", a, st, t_dev);
    run<512 | 16, 256>("state word sc1 - philox", a, st, t_dev);
    run<64 | 128 | 1, 256>("record + board tile plain - bookkeeping", a, st, t_dev);
    run<64 | 128, 64>("record + board tile plain", a, st, t_dev);
    run<64 | 128, 128>("record + board tile plain", a, st, t_dev);
    run<64, 64>("record as a plain store", a, st, t_dev);
    run<0, 128>("product body", a, st, t_dev);
    run<0, 256>("product body (again)", a, st, t_dev);
  }
  // the same body on IslandNavigation (48-cell boards, episodes ending all the time under random actions): what the episode-end
  // bookkeeping costs where it actually runs
  {
    SgkRules RI;
    if (sgk_build_rules(SGK_ISLAND_NAVIGATION, &RI) != 0) return 1;
    std::vector<uint8_t> img(SGK_RULES_DEV_BYTES, 0);
    memcpy(img.data(), &RI, sizeof(RI));
    for (int e = 0; e < 64; ++e) memcpy(img.data() + SGK_RULES_IMAGE_BYTES + e * RI.n_cells, RI.templ, (size_t)RI.n_cells);
    CK(hipMemcpy(rd, img.data(), img.size(), hipMemcpyHostToDevice));
    CK(hipFree(a.boards));
    CK(hipMalloc(&a.boards, n_pad * RI.n_cells));
    hipLaunchKernelGGL(init_state, dim3(256), dim3(256), 0, st, a.state, n_pad, pack_state(initial_state(RI)));
    CK(hipStreamSynchronize(st));
    printf("IslandNavigation, same body\n");
    for (int pass = 0; pass < 2; ++pass) {
      run<0, 256, SGK_ISLAND_NAVIGATION>("product body", a, st, t_dev);
      run<0, 64, SGK_ISLAND_NAVIGATION>("product body", a, st, t_dev);
      run<1, 256, SGK_ISLAND_NAVIGATION>("- episode bookkeeping", a, st, t_dev);
      run<1 | 32, 256, SGK_ISLAND_NAVIGATION>("- episode bookkeeping - reset branch", a, st, t_dev);
      run<4, 256, SGK_ISLAND_NAVIGATION>("- board tile", a, st, t_dev);
      run<1 | 4, 256, SGK_ISLAND_NAVIGATION>("- bookkeeping - board", a, st, t_dev);
      run<64 | 128, 64, SGK_ISLAND_NAVIGATION>("record + board tile plain", a, st, t_dev);
      run<64 | 128 | 1, 64, SGK_ISLAND_NAVIGATION>("record + board tile plain - bookkeeping", a, st, t_dev);
      run_product<SGK_ISLAND_NAVIGATION, true>("the library's kernel (SMALL)", a, st, t_dev);
      run_product<SGK_ISLAND_NAVIGATION, false>("the library's kernel", a, st, t_dev);
    }
    long long *cnt;
    CK(hipMalloc(&cnt, 8));
    hipLaunchKernelGGL(read_slab, dim3(1), dim3(1), 0, st, (const long long *)a.metrics, cnt);
    CK(hipStreamSynchronize(st));
    long long h = 0;
    CK(hipMemcpy(&h, cnt, 8, hipMemcpyDeviceToHost));
    printf("episodes booked in the metrics slab so far: %lld\n", h);
  }
  return 0;
}
