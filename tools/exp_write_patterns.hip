// exp_write_patterns.hip -- standalone probe (not part of the product): what DRAM write rate does MI355X give to the STORE
// PATTERNS a streamed rollout can choose between when every step's board tile + step records are kept in a trajectory ring?
// No env arithmetic here: waves only issue the stores (16 B per lane, up to 1 KiB contiguous per wave-instruction, like the
// product's tile writer), so the table isolates the memory side of VERDICT r02 item 2:
//   layout 0  slice-major  boards [ring][n_tiles][piece]            (step k of every tile lands in slice k: pieces 26 MB apart)
//   layout 1  tile-major   boards [n_waves][ring][M * piece]        (a wave's steps are adjacent)
//   M         adjacent 64-env tiles owned by one wave (piece per store burst = M * 1600 B for BoatRace)
//   B         steps staged (in the product: in LDS) before they are flushed as one run of B * M * piece bytes (tile-major) or as B
//             separate pieces (slice-major)
//   aux       cache policy of the stores (raw buffer store aux bits on gfx950: 0 plain, 1 sc0, 2 nt, 16 sc1)
//   wgs/cu    resident 256-lane workgroups per CU (limited through the dynamic LDS size)
//   delay     s_sleep(1) units between two steps of a wave (a stand-in for the step's arithmetic)
//   hipcc --offload-arch=gfx950 -O3 tools/exp_write_patterns.hip -o /tmp/wp && /tmp/wp
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));     \
      exit(1);                                                                      \
    }                                                                               \
  } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct P {
  char *boards, *recs;
  int n_waves;  // waves that own tiles (n_tiles / M)
  int steps, ring;
  int piece;   // board bytes per WAVE per step (M * 1600)
  int rpiece;  // record bytes per WAVE per step (M * 256); 0 = no record ring
  int burst, layout, delay;
  int s0;  // first slice of this launch (chunked launches: a launch covers `steps` consecutive slices from here)
  int rec_dword;  // 1: the record ring is written as one plain dword per lane (256 B per wave-instruction), like the product's step records
  int xcd;  // 1: workgroup b (dealt to XCD b % 8) owns tiles of the b % 8-th CONTIGUOUS eighth of the batch
  long long pad_b, pad_r;  // bytes added to the slice stride of the board / record ring (slice-major layout): WP_SLICE_PAD
  int sync;  // 1: the four waves of a workgroup meet at a barrier before every step's stores (6.4 KB contiguous issued together)
  int wait;  // >= 0: after every step the wave waits until at most this many of its stores are outstanding (s_waitcnt vmcnt)
};

template <int AUX>
__device__ __forceinline__ void put(char *dst, int len, uint32_t v) {  // dst wave-uniform, 16-byte aligned; len % 16 == 0
  const int lane = threadIdx.x & 63;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)dst, 0, len, 0x00020000);
  u32x4 x = {v, v + 1, v + 2, v + 3};
  for (int j = lane; j < len / 16; j += 64) __builtin_amdgcn_raw_buffer_store_b128(x, rsrc, j * 16, 0, AUX);
}

template <int AUX, int WGT = 256>
__global__ __launch_bounds__(WGT) void wr(P p) {
  extern __shared__ char lds_pad[];
  int w = blockIdx.x * (WGT / 64) + (threadIdx.x >> 6);
  if (p.xcd) {  // n_waves % 32 == 0 here
    const int per = p.n_waves / 8;
    w = (blockIdx.x & 7) * per + (blockIdx.x >> 3) * 4 + (threadIdx.x >> 6);
  }
  if (w >= p.n_waves && !p.sync) return;  // (sync: n_waves % 4 == 0, nobody leaves)
  if (p.delay < 0) lds_pad[threadIdx.x] = 1;  // keeps the dynamic LDS allocation alive
  uint32_t v = (uint32_t)w;
  for (int k0 = 0; k0 < p.steps; k0 += p.burst) {
    for (int i = 0; i < p.delay * p.burst; ++i) __builtin_amdgcn_s_sleep(1);
    if (p.sync) __syncthreads();
    const int nb = min(p.burst, p.steps - k0);
    if (p.layout == 1) {
      // the wave's steps are adjacent: one run of nb * piece bytes (rings wrap at whole bursts: ring % burst == 0)
      const int s0 = (p.s0 + k0) % p.ring;
      put<AUX>(p.boards + ((size_t)w * p.ring + s0) * p.piece, nb * p.piece, v);
      if (p.rpiece) put<AUX>(p.recs + ((size_t)w * p.ring + s0) * p.rpiece, nb * p.rpiece, v);
    } else {
      for (int b = 0; b < nb; ++b) {
        const int s = (p.s0 + k0 + b) % p.ring;
        put<AUX>(p.boards + (size_t)s * ((size_t)p.n_waves * p.piece + p.pad_b) + (size_t)w * p.piece, p.piece, v);
        char *rbase = p.recs + (size_t)s * ((size_t)p.n_waves * p.rpiece + p.pad_r) + (size_t)w * p.rpiece;
        if (p.rpiece && p.rec_dword) {
          uint32_t *r = reinterpret_cast<uint32_t *>(rbase);
          for (int j = threadIdx.x & 63; j < p.rpiece / 4; j += 64) r[j] = v;
        } else if (p.rpiece) put<AUX>(rbase, p.rpiece, v);
      }
    }
    v += 7;
    // bound how far a wave's ISSUED stores run ahead of the memory system (3 store instructions per step here)
    switch (p.wait) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
      case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
      case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
      default: break;
    }
  }
}

__global__ __launch_bounds__(256) void fill16(u32x4 *dst, size_t n16) {
  u32x4 x = {1, 2, 3, 4};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
    __builtin_nontemporal_store(x, &dst[i]);
}
__global__ __launch_bounds__(256) void fill16_plain(u32x4 *dst, size_t n16) {
  u32x4 x = {1, 2, 3, 4};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = x;
}

struct Cfg {
  int M, B, layout, aux, wgs_per_cu, delay, ring, boards_on, recs_on, merged;
  int xcd = 0, rec_dword = 0;
  int wait = -1;  // >= 0: s_waitcnt vmcnt(wait) after every step
  int sync = 0;  // 1: barrier before every step's stores
  int big = 0;   // 1: 1024-lane workgroups (16 tiles = 25.6 KB contiguous per step), with sync
  int oneshot = 0;  // 1: a workgroup per (step, four tiles), step-major; 2: + flag chain; 3: + 24-byte hand-off per lane
  int chunk = 0;  // > 0: the `steps` steps are issued as steps / chunk launches of `chunk` steps each (bounds the waves' drift)
};

static char *g_boards, *g_recs;
static size_t g_cap_b, g_cap_r;
static int n_cus;

// ONE-SHOT shape: a workgroup per (step, four tiles), dispatched step-major -- the order a launch per step would write in, inside
// one launch. chain = 1: workgroup (s, j) first waits for a flag that workgroup (s - 1, j) sets when it is done (what a K-step
// env kernel of this shape would need: the state is handed from step to step through memory); chain = 2 additionally reads and
// rewrites 24 bytes per lane as that hand-off.
template <int AUX>
__global__ __launch_bounds__(256) void wr_oneshot(P p, int groups, int chain, uint32_t *flags, uint32_t epoch, u32x4 *handoff) {
  const int s_idx = blockIdx.x / groups, j = blockIdx.x % groups;
  const int w = j * 4 + (threadIdx.x >> 6);
  uint32_t v = (uint32_t)w + 7u * s_idx;
  if (chain) {
    if (s_idx > 0) {
      if (threadIdx.x == 0)
        for (int spin = 0; spin < 2000000 && __hip_atomic_load(&flags[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch + (uint32_t)s_idx; ++spin)
          __builtin_amdgcn_s_sleep(1);  // (bounded: a probe must not be able to hang the box)
      __syncthreads();
    }
    if (chain == 2 && w < p.n_waves) {
      const u32x4 h = handoff[(size_t)w * 64 + (threadIdx.x & 63)];
      v += h.x & 1u;
    }
  }
  for (int i = 0; i < p.delay; ++i) __builtin_amdgcn_s_sleep(1);
  if (w < p.n_waves) {
    const int s = (p.s0 + s_idx) % p.ring;
    put<AUX>(p.boards + ((size_t)s * p.n_waves + w) * p.piece, p.piece, v);
    if (p.rpiece) put<AUX>(p.recs + ((size_t)s * p.n_waves + w) * p.rpiece, p.rpiece, v);
    if (chain == 2) {
      u32x4 h = {v, v, v, v};
      handoff[(size_t)w * 64 + (threadIdx.x & 63)] = h;
    }
  }
  if (chain) {
    __syncthreads();  // (the hand-off stores of all four waves are issued; same-XCD successor: groups % 8 == 0)
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_store(&flags[j], epoch + (uint32_t)s_idx + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

static uint32_t *g_flags;
static u32x4 *g_handoff;
static uint32_t g_epoch = 0;

static double run(const Cfg &c, int n_tiles, int steps, int piece1, int rpiece1) {
  P p;
  p.boards = g_boards;
  p.recs = g_recs;
  p.n_waves = n_tiles / c.M;
  p.steps = c.chunk > 0 ? c.chunk : steps;
  p.s0 = 0;
  const int n_launch = c.chunk > 0 ? steps / c.chunk : 1;
  p.ring = c.ring;
  p.piece = c.M * (c.merged ? piece1 + rpiece1 : piece1);
  p.rpiece = (c.recs_on && !c.merged) ? c.M * rpiece1 : 0;
  if (!c.boards_on) {  // records only: they take the "boards" slot
    p.piece = c.M * rpiece1;
    p.rpiece = 0;
  }
  p.burst = c.B;
  p.layout = c.layout;
  p.delay = c.delay;
  p.xcd = c.xcd;
  p.rec_dword = c.rec_dword;
  p.wait = c.wait;
  {  // WP_SLICE_PAD="<board bytes>,<record bytes>": slices of 25 MiB / 4 MiB put every slice's tile w on the same channel and bank
    const char *sp = getenv("WP_SLICE_PAD");
    p.pad_b = p.pad_r = 0;
    if (sp) sscanf(sp, "%lld,%lld", &p.pad_b, &p.pad_r);
  }
  p.sync = c.sync;
  if (((size_t)p.n_waves * p.piece + p.pad_b) * p.ring > g_cap_b || ((size_t)p.n_waves * (size_t)p.rpiece + p.pad_r) * p.ring > g_cap_r) return -1;
  const int lds = c.wgs_per_cu >= 8 ? 0 : (160 * 1024 / c.wgs_per_cu) - 512;
  const int grid = (p.n_waves + 3) / 4;
  auto launch = [&]() {
   if (c.oneshot) {
     const int groups = (p.n_waves + 3) / 4;
     if (!g_flags) {
       CK(hipMalloc(&g_flags, sizeof(uint32_t) * 65536));
       CK(hipMemset(g_flags, 0, sizeof(uint32_t) * 65536));
       CK(hipMalloc(&g_handoff, sizeof(u32x4) * 64 * 65536 * 4));
       CK(hipMemset(g_handoff, 0, sizeof(u32x4) * 64 * 65536 * 4));
     }
     p.s0 = 0;
     const int chain = c.oneshot - 1;
     if (c.aux == 18) hipLaunchKernelGGL(wr_oneshot<18>, dim3(groups * steps), dim3(256), lds, 0, p, groups, chain, g_flags, g_epoch, g_handoff);
     else if (c.aux == 0) hipLaunchKernelGGL(wr_oneshot<0>, dim3(groups * steps), dim3(256), lds, 0, p, groups, chain, g_flags, g_epoch, g_handoff);
     else hipLaunchKernelGGL(wr_oneshot<16>, dim3(groups * steps), dim3(256), lds, 0, p, groups, chain, g_flags, g_epoch, g_handoff);
     g_epoch += (uint32_t)steps;
     return;
   }
   if (c.big) {
     p.s0 = 0;
     const int g16 = (p.n_waves + 15) / 16;
     if (c.aux == 18) hipLaunchKernelGGL((wr<18, 1024>), dim3(g16), dim3(1024), lds, 0, p);
     else hipLaunchKernelGGL((wr<16, 1024>), dim3(g16), dim3(1024), lds, 0, p);
     return;
   }
   for (int l = 0; l < n_launch; ++l) {
    p.s0 = l * p.steps;
    switch (c.aux) {
    case 0: hipLaunchKernelGGL(wr<0>, dim3(grid), dim3(256), lds, 0, p); break;
    case 2: hipLaunchKernelGGL(wr<2>, dim3(grid), dim3(256), lds, 0, p); break;
    case 18: hipLaunchKernelGGL(wr<18>, dim3(grid), dim3(256), lds, 0, p); break;
    case 17: hipLaunchKernelGGL(wr<17>, dim3(grid), dim3(256), lds, 0, p); break;
    default: hipLaunchKernelGGL(wr<16>, dim3(grid), dim3(256), lds, 0, p); break;
    }
   }
  };
  launch();
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<float> ms;
  for (int r = 0; r < 4; ++r) {
    CK(hipEventRecord(e0, 0));
    launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float t;
    CK(hipEventElapsedTime(&t, e0, e1));
    ms.push_back(t);
  }
  CK(hipGetLastError());
  std::sort(ms.begin(), ms.end());
  return (ms[1] + ms[2]) * 0.5 * 1e3;  // us per launch, median of four
}

int main(int argc, char **argv) {
  int n_tiles = 16384, steps = 100, piece = 1600, rpiece = 256;  // BoatRace at 1 M envs
  const char *only = argc > 1 ? argv[1] : "";
  if (argc > 2) piece = atoi(argv[2]);  // 3072 = IslandNavigation
  if (argc > 3) n_tiles = atoi(argv[3]);  // 2048 = the per-GPU share of the 1 M batch at 8 GPUs
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  n_cus = prop.multiProcessorCount;
  g_cap_b = (size_t)n_tiles * 100 * (piece + rpiece) + (256 << 20);
  g_cap_r = (size_t)n_tiles * 100 * rpiece + (256 << 20);
  // WP_ALLOC=vmm: the two rings through HIP's virtual memory management API -- one physical allocation each, mapped at a
  // 1-GiB-aligned virtual address -- instead of hipMalloc: does the driver then cover them with larger page-table fragments?
  const char *alloc = getenv("WP_ALLOC");
  if (alloc && !strcmp(alloc, "vmm")) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    printf("# vmm: recommended granularity %zu\n", gran);
    const size_t GB = (size_t)1 << 30;
    for (int which = 0; which < 2; ++which) {
      size_t &cap = which ? g_cap_r : g_cap_b;
      cap = (cap + GB - 1) / GB * GB;
      hipMemGenericAllocationHandle_t h;
      CK(hipMemCreate(&h, cap, &prop, 0));
      void *ptr = nullptr;
      CK(hipMemAddressReserve(&ptr, cap, GB, nullptr, 0));
      CK(hipMemMap(ptr, cap, 0, h, 0));
      hipMemAccessDesc acc = {};
      acc.location = prop.location;
      acc.flags = hipMemAccessFlagsProtReadWrite;
      CK(hipMemSetAccess(ptr, cap, &acc, 1));
      (which ? g_recs : g_boards) = (char *)ptr;
      printf("# vmm: ring %d at %p, %zu bytes\n", which, ptr, cap);
    }
  } else if (alloc && !strcmp(alloc, "contig")) {  // physically contiguous VRAM: the largest page-table fragments the driver has
    CK(hipExtMallocWithFlags((void **)&g_boards, g_cap_b, hipDeviceMallocContiguous));
    CK(hipExtMallocWithFlags((void **)&g_recs, g_cap_r, hipDeviceMallocContiguous));
    printf("# hipExtMallocWithFlags(hipDeviceMallocContiguous): rings at %p and %p\n", (void *)g_boards, (void *)g_recs);
  } else {
    CK(hipMalloc(&g_boards, g_cap_b));
    CK(hipMalloc(&g_recs, g_cap_r));
    printf("# hipMalloc: rings at %p and %p\n", (void *)g_boards, (void *)g_recs);
  }
  printf("# %s, %d CUs; %d tiles x %d steps, %d + %d bytes per tile-step (%.1f MB per step)\n", prop.name, n_cus, n_tiles, steps, piece,
         rpiece, n_tiles * (piece + rpiece) / 1e6);
  // --- baselines: a sequential fill of the same byte count
  for (int rep = 0; rep < ((getenv("WP_INDEX") || getenv("WP_LIST")) ? 0 : 2); ++rep) {
    for (size_t mb : {240, 1000, 3000}) {
      size_t n16 = mb * 1000 * 1000 / 16;
      if (n16 * 16 > g_cap_b) continue;
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0));
      CK(hipEventCreate(&e1));
      for (int nt = 0; nt < 2; ++nt) {
        float best = 1e9;
        for (int r = 0; r < 4; ++r) {
          CK(hipEventRecord(e0, 0));
          if (nt) hipLaunchKernelGGL(fill16, dim3(n_cus * 8), dim3(256), 0, 0, (u32x4 *)g_boards, n16);
          else hipLaunchKernelGGL(fill16_plain, dim3(n_cus * 8), dim3(256), 0, 0, (u32x4 *)g_boards, n16);
          CK(hipEventRecord(e1, 0));
          CK(hipEventSynchronize(e1));
          float t;
          CK(hipEventElapsedTime(&t, e0, e1));
          if (r) best = std::min(best, t);
        }
        printf("fill %-5s %5zu MB: %8.1f us  %.2f TB/s\n", nt ? "nt" : "plain", mb, best * 1e3, n16 * 16 / (best * 1e-3) / 1e12);
      }
    }
  }
  std::vector<std::pair<const char *, Cfg>> T;
  //                    M  B  lay aux wg dly ring b  r  merged
  T.push_back({"base slice sc1", {1, 1, 0, 16, 8, 0, 100, 1, 1, 0}});
  T.push_back({"base slice plain", {1, 1, 0, 0, 8, 0, 100, 1, 1, 0}});
  T.push_back({"base slice nt", {1, 1, 0, 2, 8, 0, 100, 1, 1, 0}});
  T.push_back({"base slice nt+sc1", {1, 1, 0, 18, 8, 0, 100, 1, 1, 0}});
  T.push_back({"base slice sc0sc1", {1, 1, 0, 17, 8, 0, 100, 1, 1, 0}});
  T.push_back({"slice boards only", {1, 1, 0, 16, 8, 0, 100, 1, 0, 0}});
  T.push_back({"slice recs only", {1, 1, 0, 16, 8, 0, 100, 0, 1, 0}});
  T.push_back({"slice merged rec+board", {1, 1, 0, 16, 8, 0, 100, 1, 1, 1}});
  for (int d : {4, 16, 64}) T.push_back({"slice sc1 delay", {1, 1, 0, 16, 8, d, 100, 1, 1, 0}});
  for (int wg : {1, 2, 4, 6}) T.push_back({"slice sc1 wgs/cu", {1, 1, 0, 16, wg, 0, 100, 1, 1, 0}});
  for (int m : {2, 4, 8}) T.push_back({"slice M tiles/wave", {m, 1, 0, 16, 8, 0, 100, 1, 1, 0}});
  for (int m : {2, 4, 8}) T.push_back({"slice M tiles/wave nt", {m, 1, 0, 2, 8, 0, 100, 1, 1, 0}});
  for (int r : {4, 8, 16, 32, 64}) T.push_back({"slice ring size", {1, 1, 0, 16, 8, 0, r, 1, 1, 0}});
  T.push_back({"tile-major sc1", {1, 1, 1, 16, 8, 0, 100, 1, 1, 0}});
  T.push_back({"tile-major nt", {1, 1, 1, 2, 8, 0, 100, 1, 1, 0}});
  T.push_back({"tile-major plain", {1, 1, 1, 0, 8, 0, 100, 1, 1, 0}});
  T.push_back({"tile-major merged", {1, 1, 1, 16, 8, 0, 100, 1, 1, 1}});
  for (int b : {2, 4, 5, 10, 20, 50}) T.push_back({"tile-major burst sc1", {1, b, 1, 16, 8, 0, 100, 1, 1, 0}});
  for (int b : {2, 4, 5, 10, 20, 50}) T.push_back({"tile-major burst nt", {1, b, 1, 2, 8, 0, 100, 1, 1, 0}});
  for (int b : {4, 10, 20}) T.push_back({"tile-major burst plain", {1, b, 1, 0, 8, 0, 100, 1, 1, 0}});
  for (int b : {4, 10, 20}) T.push_back({"tile-major burst merged sc1", {1, b, 1, 16, 8, 0, 100, 1, 1, 1}});
  for (int b : {4, 10}) for (int wg : {2, 3, 4, 6}) T.push_back({"tile-major burst sc1 wgs/cu", {1, b, 1, 16, wg, 0, 100, 1, 1, 0}});
  for (int b : {4, 10}) for (int d : {4, 16, 64}) T.push_back({"tile-major burst sc1 delay", {1, b, 1, 16, 8, d, 100, 1, 1, 0}});
  for (int m : {2, 4}) for (int b : {1, 4, 10}) T.push_back({"tile-major M x burst sc1", {m, b, 1, 16, 8, 0, 100, 1, 1, 0}});
  for (int b : {4, 10, 20}) T.push_back({"slice burst-in-time sc1", {1, b, 0, 16, 8, 0, 100, 1, 1, 0}});
  for (int r : {8, 20, 40}) T.push_back({"tile-major burst4 ring size", {1, 4, 1, 16, 8, 0, r, 1, 1, 0}});
  for (int wg : {8, 5}) {
    Cfg c{1, 1, 0, 16, wg, 0, 100, 1, 1, 0};
    c.rec_dword = 1;
    T.push_back({"slice, records as plain dwords", c});
  }
  for (int d : {2, 8}) {
    Cfg c{1, 1, 0, 16, 5, d, 100, 1, 1, 0};
    c.rec_dword = 1;
    T.push_back({"slice, dword records, 5 wgs/cu, delay", c});
  }
  for (int lay : {0, 1})
    for (int r : {32, 48, 64, 100}) {
      Cfg c{1, 1, lay, 16, 8, 0, r, 1, 1, 0};
      c.xcd = 1;
      T.push_back({lay ? "XCD-contiguous tile-major" : "XCD-contiguous slice", c});
    }
  for (int m : {2, 4}) {
    Cfg c{m, 1, 0, 16, 8, 0, 100, 1, 1, 0};
    c.xcd = 1;
    T.push_back({"XCD-contiguous slice M", c});
  }
  for (int aux : {0, 2, 18}) {
    Cfg c{1, 1, 0, aux, 8, 0, 100, 1, 1, 0};
    c.xcd = 1;
    T.push_back({"XCD-contiguous slice aux", c});
  }
  for (int wg : {2, 4}) {
    Cfg c{1, 1, 0, 16, wg, 0, 100, 1, 1, 0};
    c.xcd = 1;
    T.push_back({"XCD-contiguous slice wgs/cu", c});
  }
  for (int ck : {5, 10, 20, 25, 50}) {
    Cfg c{1, 1, 0, 16, 8, 0, 100, 1, 1, 0};
    c.chunk = ck;
    T.push_back({"slice chunked launches", c});
  }
  for (int ck : {10, 25}) {
    Cfg c{1, 1, 1, 16, 8, 0, 100, 1, 1, 0};
    c.chunk = ck;
    T.push_back({"tile-major chunked launches", c});
  }
  for (int r : {36, 40, 48, 56, 64, 72, 80, 90}) T.push_back({"slice ring size fine", {1, 1, 0, 16, 8, 0, r, 1, 1, 0}});
  for (int ring : {100, 32})
    for (int wg : {8, 5, 2})
      for (int aux : {16, 18}) {
        Cfg c{1, 1, 0, aux, wg, 0, ring, 1, 1, 0};
        c.sync = 1;
        T.push_back({"slice, workgroup barrier per step", c});
      }
  for (int ring : {100, 32})
    for (int wg : {1, 2})
      for (int sync : {1, 0})
        for (int aux : {16, 18}) {
          Cfg c{1, 1, 0, aux, wg, 0, ring, 1, 1, 0};
          c.sync = sync;
          c.big = 1;
          T.push_back({sync ? "slice, 1024-lane groups + barrier" : "slice, 1024-lane groups", c});
        }
  for (int os : {1, 2, 3})
    for (int ring : {100, 32})
      for (int aux : {16, 0, 18})
        for (int dly : {0, 4, 16}) {
          if ((aux != 16 || ring != 100) && dly) continue;
          Cfg c{1, 1, 0, aux, 8, dly, ring, 1, 1, 0};
          c.oneshot = os;
          T.push_back({os == 1 ? "ONE-SHOT per (step, 4 tiles)" : (os == 2 ? "ONE-SHOT + flag chain" : "ONE-SHOT + chain + hand-off"), c});
        }
  for (int ring : {100, 32})
    for (int wg : {8, 5})
      for (int wt : {0, 3, 6, 9, 12, 24}) {
        Cfg c{1, 1, 0, 16, wg, 0, ring, 1, 1, 0};
        c.wait = wt;
        T.push_back({"slice sc1, stores in flight capped", c});
      }
  for (int wt : {3, 6, 12}) {
    Cfg c{1, 1, 1, 16, 8, 0, 100, 1, 1, 0};
    c.wait = wt;
    T.push_back({"tile-major sc1, stores in flight capped", c});
  }
  printf("%-30s %2s %3s %3s %3s %3s %4s %4s | %9s %8s %7s\n", "variant", "M", "B", "lay", "aux", "wg", "dly", "ring", "us/100st.", "us/step",
         "TB/s");
  const char *ring_only = getenv("WP_RING"), *index_only = getenv("WP_INDEX");
  int index = -1;
  for (auto &t : T) {
    ++index;
    if (index_only && atoi(index_only) != index) continue;
    if (*only && !strstr(t.first, only)) continue;
    const Cfg &c = t.second;
    if (ring_only && atoi(ring_only) != c.ring) continue;
    if (getenv("WP_LIST")) {
      printf("[%3d] %s M=%d B=%d lay=%d aux=%d wg=%d dly=%d ring=%d xcd=%d chunk=%d recdw=%d\n", index, t.first, c.M, c.B, c.layout, c.aux,
             c.wgs_per_cu, c.delay, c.ring, c.xcd, c.chunk, c.rec_dword);
      if (c.wait >= 0) printf("      (vmcnt cap %d)\n", c.wait);
      continue;
    }
    printf("[%3d] ", index);
    for (int rep = 0; rep < 2; ++rep) {
      double us = run(c, n_tiles, steps, piece, rpiece);
      const double bytes = (double)n_tiles * steps * ((c.boards_on ? piece : 0) + (c.recs_on ? rpiece : 0));
      printf("%s%-30s %2d %3d %3d %3d %3d %4d %4d | %9.1f %8.3f %7.2f\n", rep ? "      " : "", t.first, c.M, c.B, c.layout, c.aux, c.wgs_per_cu, c.delay, c.ring,
             us, us / steps, bytes / (us * 1e-6) / 1e12);
      if (c.wait >= 0 && !rep) printf("      (at most %d store instructions of a wave outstanding after each step)\n", c.wait);
    }
    fflush(stdout);
  }
  return 0;
}
