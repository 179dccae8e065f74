// exp_step_latency.hip -- standalone probe (not part of the product): where do the ~3 us of ONE per-step launch at 65 536 BoatRace
// envs go (BASELINE config 2)? Synthetic kernels with the product's memory shape -- 8-byte state word in,
// state word + 4-byte record + 25-byte board row out -- built up a piece at a time and replayed as a hipGraph chain of 100
// dependent launches, like sgk_step_random does:
//   A  empty kernel
//   B  state word load -> store + record store                                   (one memory round trip + stores)
//   C  B + Philox-4x32-10 for the action                                         (the arithmetic)
//   D  C + 1.8 KB rule table staged to LDS by the workgroup (barrier) + a lookup (the product's table staging)
//   E  D + rotation table (second barrier) + LDS tile image + 16-byte sc1 buffer stores of the board tile  (~ step_kernel)
//   F  wave-private form: transition table held in 2 VGPRs per lane (loaded at entry, ds_bpermute lookups), blank tile image
//      loaded from a precomputed 64 x NC image at entry; every global load is issued before the first wait; no barrier
//   wg = lanes per workgroup (64 / 256); args = +0 / +2048 bytes of kernel arguments (is the kernarg fetch on the critical path?)
//   hipcc --offload-arch=gfx950 -O3 tools/exp_step_latency.hip -o /tmp/sl && /tmp/sl [n_envs]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int NC = 25;

struct Rules {  // the product's SgkRules is 1.8 KB: same size here
  uint32_t trans[256];
  uint8_t templ[64], agent_value[64];
  uint32_t pad[160];
};
static_assert(sizeof(Rules) == 1024 + 128 + 640, "1.8 KB");

struct Args {
  const Rules *rules;
  uint64_t *state;
  uint32_t *rec;
  int8_t *boards;
  const u32x4 *blank;  // [4 * NC] 16-byte chunks of a blank 64-env tile
  const uint64_t *t_ptr;
  int64_t n;
  uint64_t seed, t;
};
struct Fat { uint32_t w[512]; };

__device__ __forceinline__ void philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t x[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0, h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
    uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
    c0 = n0; c1 = l1; c2 = n2; c3 = l0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  x[0] = c0; x[1] = c1; x[2] = c2; x[3] = c3;
}

template <int V, int WG, bool FAT>
__global__ __launch_bounds__(WG) void k(Args a, Fat fat) {
  __shared__ Rules R;
  __shared__ __attribute__((aligned(16))) uint8_t rot[NC][16];
  __shared__ __attribute__((aligned(16))) uint8_t tiles[WG / 64][64 * NC];
  if (V == 0) return;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t wt = (int64_t)blockIdx.x * (WG / 64) + wave;
  const int64_t env = wt * 64 + lane;
  if (wt * 64 >= a.n) return;
  uint64_t w = a.state[env];
  uint32_t tab0 = 0, tab1 = 0, av = 0;
  u32x4 b0, b1;
  if (V == 5) {  // everything the wave needs, requested before the first wait
    const uint32_t *tr = a.rules->trans;
    tab0 = tr[lane];
    tab1 = tr[64 + lane];
    av = a.rules->agent_value[lane];
    b0 = a.blank[lane];
    b1 = a.blank[lane + 64 < 4 * NC ? lane + 64 : 4 * NC - 1];
  }
  const uint64_t t_now = *a.t_ptr + a.t;
  if (V == 3 || V == 4) {
    const uint32_t *s = reinterpret_cast<const uint32_t *>(a.rules);
    uint32_t *d = reinterpret_cast<uint32_t *>(&R);
    for (int i = threadIdx.x; i < (int)(sizeof(Rules) / 4); i += WG) d[i] = s[i];
    __syncthreads();
  }
  if (V == 4) {
    for (int i = threadIdx.x; i < NC * 16; i += WG) rot[i >> 4][i & 15] = R.templ[((i >> 4) + (i & 15)) % NC];
    __syncthreads();
  }
  uint32_t pos = (uint32_t)w & 0xff, frame = (uint32_t)(w >> 16) & 0xff;
  int ret = (int16_t)(w >> 32);
  uint32_t action = (uint32_t)(w >> 8) & 3;
  if (V >= 2) {
    uint32_t x[4];
    philox((uint32_t)env, (uint32_t)((uint64_t)env >> 32), (uint32_t)(t_now >> 6), 0u, (uint32_t)a.seed, (uint32_t)(a.seed >> 32), x);
    action = (x[(t_now >> 4) & 3] >> (2 * (t_now & 15))) & 3u;
  }
  uint32_t e = 0;
  if (V == 3 || V == 4) e = R.trans[(pos % NC) * 4 + action];
  if (V == 5) {
    const uint32_t idx = (pos % NC) * 4 + action;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((idx & 63) << 2), (int)tab0);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((idx & 63) << 2), (int)tab1);
    e = idx < 64 ? lo : hi;
  }
  if (V >= 3) pos = e & 0xff, ret += (int)(int8_t)(e >> 8);
  else pos = (pos + action) % NC;
  frame += 1;
  if (frame >= 100) frame = 0, ret = 0, pos = 6;
  if (FAT) ret += (int)fat.w[(pos + lane) & 511];
  w = (uint64_t)pos | ((uint64_t)action << 8) | ((uint64_t)frame << 16) | ((uint64_t)(uint16_t)ret << 32);
  a.state[env] = w;
  __hip_atomic_store(&a.rec[env], (uint32_t)action << 24 | (frame == 0 ? 0x10000u : 0u) | 0xffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (V >= 4) {
    uint8_t *tile = tiles[wave];
    if (V == 4) {
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int j = lane + 64 * it;
        if (j < 4 * NC) *reinterpret_cast<uint4 *>(tile + 16 * j) = *reinterpret_cast<const uint4 *>(&rot[(16 * j) % NC][0]);
      }
    } else {
      *reinterpret_cast<u32x4 *>(tile + 16 * lane) = b0;
      if (lane + 64 < 4 * NC) *reinterpret_cast<u32x4 *>(tile + 16 * (lane + 64)) = b1;
    }
    __builtin_amdgcn_wave_barrier();
    uint32_t aval = 2;
    if (V == 4) aval = R.agent_value[pos];
    if (V == 5) aval = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(pos << 2), (int)av) & 0xff;
    tile[lane * NC + pos] = (uint8_t)aval;
    __builtin_amdgcn_wave_barrier();
    int8_t *dst = a.boards + wt * 64 * NC;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)dst, 0, 64 * NC, 0x00020000);
    uint4 v0 = *reinterpret_cast<const uint4 *>(tile + 16 * lane);
    uint4 v1 = *reinterpret_cast<const uint4 *>(tile + 16 * (lane + 64 < 4 * NC ? lane + 64 : 4 * NC - 1));
    u32x4 q0 = {v0.x, v0.y, v0.z, v0.w}, q1 = {v1.x, v1.y, v1.z, v1.w};
    __builtin_amdgcn_raw_buffer_store_b128(q0, rsrc, lane * 16, 0, 16);
    __builtin_amdgcn_raw_buffer_store_b128(q1, rsrc, (lane + 64) * 16, 0, 16);
  }
}

template <int V, int WG, bool FAT>
static double run(const char *label, Args a, int chain, int reps, hipStream_t st, bool use_graph) {
  const int grid = (int)((a.n + WG - 1) / WG);
  Fat fat;
  memset(&fat, 0, sizeof(fat));
  hipGraph_t g = nullptr;
  hipGraphExec_t ge = nullptr;
  auto chain_launch = [&]() {
    for (int i = 0; i < chain; ++i) {
      a.t = (uint64_t)i;
      hipLaunchKernelGGL((k<V, WG, FAT>), dim3(grid), dim3(WG), 0, st, a, fat);
    }
  };
  if (use_graph) {
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    chain_launch();
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<float> ms;
  for (int r = 0; r < reps + 2; ++r) {
    CK(hipEventRecord(e0, st));
    if (use_graph) CK(hipGraphLaunch(ge, st));
    else chain_launch();
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float t;
    CK(hipEventElapsedTime(&t, e0, e1));
    if (r >= 2) ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  const double med = ms[ms.size() / 2] * 1e3 / chain, best = ms[0] * 1e3 / chain;
  printf("%-58s wg %3d %s %s  median %6.3f us  best %6.3f us per launch\n", label, WG, FAT ? "args+2K" : "args   ",
         use_graph ? "graph " : "stream", med, best);
  fflush(stdout);
  if (ge) CK(hipGraphExecDestroy(ge));
  if (g) CK(hipGraphDestroy(g));
  return med;
}

int main(int argc, char **argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 65536;
  const int chain = 100, reps = 30;
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  Args a;
  Rules hr;
  memset(&hr, 0, sizeof(hr));
  for (int c = 0; c < NC; ++c)
    for (int d = 0; d < 4; ++d) hr.trans[c * 4 + d] = (uint32_t)((c + d + 1) % NC) | (0xffu << 8);
  for (int c = 0; c < 64; ++c) hr.templ[c] = (uint8_t)(c % 3), hr.agent_value[c] = 2;
  Rules *dr;
  CK(hipMalloc(&dr, sizeof(Rules)));
  CK(hipMemcpy(dr, &hr, sizeof(Rules), hipMemcpyHostToDevice));
  a.rules = dr;
  const int64_t n_pad = (n + 255) / 256 * 256;
  CK(hipMalloc(&a.state, n_pad * 8));
  CK(hipMemset(a.state, 0, n_pad * 8));
  CK(hipMalloc(&a.rec, n_pad * 4));
  CK(hipMalloc(&a.boards, n_pad * NC));
  std::vector<uint8_t> blank(64 * NC);
  for (int i = 0; i < 64 * NC; ++i) blank[i] = hr.templ[i % NC];
  u32x4 *db;
  CK(hipMalloc(&db, 64 * NC));
  CK(hipMemcpy(db, blank.data(), 64 * NC, hipMemcpyHostToDevice));
  a.blank = db;
  uint64_t *tp;
  CK(hipMalloc(&tp, 8));
  CK(hipMemset(tp, 0, 8));
  a.t_ptr = tp;
  a.n = n;
  a.seed = 0x5AFE;
  a.t = 0;
  printf("n = %lld envs, chains of %d dependent launches, %d repetitions\n", (long long)n, chain, reps);
  for (int pass = 0; pass < 2; ++pass) {
    const bool gr = pass == 0;
    run<0, 256, false>("A empty", a, chain, reps, st, gr);
    run<0, 64, false>("A empty", a, chain, reps, st, gr);
    run<0, 256, true>("A empty", a, chain, reps, st, gr);
    run<1, 256, false>("B state word in/out + record", a, chain, reps, st, gr);
    run<1, 64, false>("B state word in/out + record", a, chain, reps, st, gr);
    run<1, 256, true>("B state word in/out + record", a, chain, reps, st, gr);
    run<2, 256, false>("C B + philox", a, chain, reps, st, gr);
    run<3, 256, false>("D C + rules staged to LDS (barrier) + lookup", a, chain, reps, st, gr);
    run<3, 64, false>("D C + rules staged to LDS (barrier) + lookup", a, chain, reps, st, gr);
    run<4, 256, false>("E D + rotations (barrier) + LDS tile + board stores", a, chain, reps, st, gr);
    run<4, 64, false>("E D + rotations (barrier) + LDS tile + board stores", a, chain, reps, st, gr);
    run<5, 256, false>("F wave-private: tables in VGPRs, blank tile preloaded", a, chain, reps, st, gr);
    run<5, 64, false>("F wave-private: tables in VGPRs, blank tile preloaded", a, chain, reps, st, gr);
    run<5, 128, false>("F wave-private: tables in VGPRs, blank tile preloaded", a, chain, reps, st, gr);
  }
  return 0;
}
