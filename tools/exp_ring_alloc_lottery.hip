// exp_ring_alloc_lottery.hip -- standalone probe (not part of the product): the trajectory ring's write rate differs from
// allocation to allocation. Is it a property of the ALLOCATION (physical placement), stable over time, or of
// where in its address range the ring starts? The ring's store pattern (16 384 waves, 1 600 + 256 bytes per wave-step, 100
// slices, persistent) into ten fresh hipMalloc pairs kept alive side by side, each timed in two rounds; then into the first
// pair at several byte offsets.
//   hipcc --offload-arch=gfx950 -O3 tools/exp_ring_alloc_lottery.hip -o /tmp/lottery && /tmp/lottery
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
static const int TILES = 16384, STEPS = 100, PIECE = 1600, RPIECE = 256;

__device__ __forceinline__ void put(char *dst, int len, uint32_t v) {
  const int lane = threadIdx.x & 63;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)dst, 0, len, 0x00020000);
  u32x4 x = {v, v + 1, v + 2, v + 3};
  for (int j = lane; j < len / 16; j += 64) __builtin_amdgcn_raw_buffer_store_b128(x, rsrc, j * 16, 0, 16);
}

// mul / pad: step s goes to slice position (s * mul) % STEPS, every slice shifted by position * pad bytes (mul = 1, pad = 0: the
// product's layout). tmul: tile w sits at position (w * tmul) % TILES inside its slice.
__global__ __launch_bounds__(256) void ring_writer(char *boards, char *recs, int mul = 1, size_t pad = 0, int tmul = 1) {
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
  const size_t wp = (size_t)(((long long)w * tmul) % TILES);
  uint32_t v = (uint32_t)w;
  for (int s = 0; s < STEPS; ++s) {
    const size_t pos = (size_t)((s * mul) % STEPS);
    put(boards + (pos * TILES + wp) * PIECE + pos * pad, PIECE, v);
    put(recs + (pos * TILES + wp) * RPIECE + pos * pad, RPIECE, v);
    v += 7;
  }
}

// XCD skew: workgroup b runs on XCD b % 8; the even XCDs get `even` tile groups each, the odd ones (1024 - even): does giving the
// XCDs that write faster more tiles shorten the launch?
__global__ __launch_bounds__(256) void ring_writer_skew(char *boards, char *recs, int even) {
  const int x = blockIdx.x & 7, i = blockIdx.x >> 3, odd = 1024 - even;
  const int quota = (x & 1) ? odd : even;
  if (i >= quota) return;
  int off = 0;
  for (int y = 0; y < x; ++y) off += (y & 1) ? odd : even;
  const int w = (off + i) * 4 + (threadIdx.x >> 6);
  uint32_t v = (uint32_t)w;
  for (int s = 0; s < STEPS; ++s) {
    put(boards + ((size_t)s * TILES + w) * PIECE, PIECE, v);
    put(recs + ((size_t)s * TILES + w) * RPIECE, RPIECE, v);
    v += 7;
  }
}

// the ring as NCH separately allocated chunks of STEPS / NCH slices each (boards and records of a chunk in one allocation)
struct Chunks { char *b[20]; char *r[20]; int per; };
__global__ __launch_bounds__(256) void ring_writer_chunks(Chunks c) {
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
  uint32_t v = (uint32_t)w;
  for (int s = 0; s < STEPS; ++s) {
    const int ch = s / c.per, k = s - ch * c.per;
    put(c.b[ch] + ((size_t)k * TILES + w) * PIECE, PIECE, v);
    put(c.r[ch] + ((size_t)k * TILES + w) * RPIECE, RPIECE, v);
    v += 7;
  }
}
static double us_per_step_chunks(const Chunks &c) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(ring_writer_chunks, dim3(TILES / 4), dim3(256), 0, 0, c);
  std::vector<float> ms;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(ring_writer_chunks, dim3(TILES / 4), dim3(256), 0, 0, c);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float t;
    CK(hipEventElapsedTime(&t, e0, e1));
    ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  CK(hipEventDestroy(e0));
  CK(hipEventDestroy(e1));
  return ms[1] * 1e3 / STEPS;
}

static double us_per_step(char *b, char *r, int mul = 1, size_t pad = 0, int tmul = 1) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(ring_writer, dim3(TILES / 4), dim3(256), 0, 0, b, r, mul, pad, tmul);
  std::vector<float> ms;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(ring_writer, dim3(TILES / 4), dim3(256), 0, 0, b, r, mul, pad, tmul);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float t;
    CK(hipEventElapsedTime(&t, e0, e1));
    ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  CK(hipEventDestroy(e0));
  CK(hipEventDestroy(e1));
  return ms[1] * 1e3 / STEPS;
}

int main(int argc, char **argv) {
  const size_t bb = (size_t)TILES * STEPS * PIECE, rb = (size_t)TILES * STEPS * RPIECE, slack = (size_t)1 << 30;
  if (argc > 1 && !strcmp(argv[1], "flags")) {
    // other kinds of device memory: fine-grained (coherent) and uncached allocations, next to plain hipMalloc
    for (int trial = 0; trial < 3; ++trial) {
      printf("trial %d:", trial);
      for (int kind = 0; kind < 4; ++kind) {
        char *b = nullptr, *r = nullptr;
        hipError_t e1, e2;
        if (kind == 0) { e1 = hipMalloc(&b, bb); e2 = hipMalloc(&r, rb); }
        else {
          const unsigned flag = kind == 1 ? hipDeviceMallocFinegrained : (kind == 2 ? hipDeviceMallocUncached : hipDeviceMallocContiguous);
          e1 = hipExtMallocWithFlags((void **)&b, bb, flag);
          e2 = hipExtMallocWithFlags((void **)&r, rb, flag);
        }
        const char *name = kind == 0 ? "hipMalloc" : (kind == 1 ? "fine-grained" : (kind == 2 ? "uncached" : "contiguous"));
        if (e1 != hipSuccess || e2 != hipSuccess) { (void)hipGetLastError(); printf(" %s (failed)", name); continue; }
        printf(" %s %.2f", name, us_per_step(b, r));
        fflush(stdout);
      }
      printf("\n");
    }
    return 0;
  }
  if (argc > 1 && !strcmp(argv[1], "skew")) {
    for (int a = 0; a < 3; ++a) {
      char *b, *r;
      CK(hipMalloc(&b, bb));
      CK(hipMalloc(&r, rb));
      printf("pair %d: uniform %.2f |", a, us_per_step(b, r));
      for (int even : {512, 480, 544, 560, 576, 592, 608, 640, 704}) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        const int grid = 8 * (even > 512 ? even : 1024 - even);  // every XCD gets enough workgroup ids for the larger quota
        std::vector<float> ms;
        for (int rep = 0; rep < 4; ++rep) {
          CK(hipEventRecord(e0, 0));
          hipLaunchKernelGGL(ring_writer_skew, dim3(grid), dim3(256), 0, 0, b, r, even);
          CK(hipEventRecord(e1, 0));
          CK(hipEventSynchronize(e1));
          float t;
          CK(hipEventElapsedTime(&t, e0, e1));
          if (rep) ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        printf(" even=%d %.2f", even, ms[1] * 1e3 / STEPS);
      }
      printf("\n");
      fflush(stdout);
    }
    return 0;
  }
  if (argc > 1 && !strcmp(argv[1], "vmm")) {
    // A contiguous VIRTUAL ring over scattered PHYSICAL memory (HIP virtual memory management): physical chunks of `chunk` bytes
    // created one after another, mapped into the ring's address range in order, in a shuffled order, or with every other created
    // chunk given back first (the kept ones lie 2 x chunk apart). Does the ring's rate follow the physical arrangement?
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    printf("# granularity %zu bytes; columns: in order | shuffled | every other chunk kept (in order) | every other kept + shuffled\n", gran);
    for (size_t chunk : {(size_t)64 << 10, (size_t)2 << 20, (size_t)32 << 20, (size_t)256 << 20, (size_t)512 << 20, (size_t)1 << 30, (size_t)2 << 30, (size_t)4 << 30}) {
      if (argc > 2 && (size_t)atoll(argv[2]) != (chunk >> 10)) continue;  // one chunk size, in KiB
      for (int trial = 0; trial < 3; ++trial) {
        printf("chunk %7zu KiB, trial %d:", chunk >> 10, trial);
        for (int mode = 0; mode < 4; ++mode) {
          const bool shuffle = mode & 1, sparse = mode & 2;
          if (chunk < ((size_t)1 << 20) && mode) continue;  // (47 000 handles: once)
          if (sparse && chunk > ((size_t)256 << 20)) continue;
          char *base[2];
          std::vector<hipMemGenericAllocationHandle_t> hs[2], dropped;
          size_t sizes[2] = {(bb + chunk - 1) / chunk * chunk, (rb + chunk - 1) / chunk * chunk};
          for (int w = 0; w < 2; ++w) {
            const size_t nch = sizes[w] / chunk;
            for (size_t i = 0; i < nch; ++i) {
              hipMemGenericAllocationHandle_t h;
              CK(hipMemCreate(&h, chunk, &prop, 0));
              hs[w].push_back(h);
              if (sparse) {  // a second chunk right behind it, given back below: the kept ones are 2 x chunk apart
                hipMemGenericAllocationHandle_t d;
                CK(hipMemCreate(&d, chunk, &prop, 0));
                dropped.push_back(d);
              }
            }
            void *ptr = nullptr;
            CK(hipMemAddressReserve(&ptr, sizes[w], 0, nullptr, 0));
            base[w] = (char *)ptr;
            std::vector<size_t> order(nch);
            for (size_t i = 0; i < nch; ++i) order[i] = i;
            if (shuffle) {
              uint64_t x = 88172645463325252ull + trial;
              for (size_t i = nch - 1; i > 0; --i) {
                x ^= x << 13; x ^= x >> 7; x ^= x << 17;
                std::swap(order[i], order[x % (i + 1)]);
              }
            }
            for (size_t i = 0; i < nch; ++i) CK(hipMemMap(base[w] + i * chunk, chunk, 0, hs[w][order[i]], 0));
            hipMemAccessDesc acc = {};
            acc.location = prop.location;
            acc.flags = hipMemAccessFlagsProtReadWrite;
            CK(hipMemSetAccess(ptr, sizes[w], &acc, 1));
          }
          for (auto d : dropped) CK(hipMemRelease(d));
          printf(" %.2f", us_per_step(base[0], base[1]));
          fflush(stdout);
          for (int w = 0; w < 2; ++w) {
            CK(hipMemUnmap(base[w], sizes[w]));
            CK(hipMemAddressFree(base[w], sizes[w]));
            for (auto h : hs[w]) CK(hipMemRelease(h));
          }
        }
        printf("\n");
      }
    }
    return 0;
  }
  if (argc > 1 && !strcmp(argv[1], "spread")) {
    // Is it WHERE in physical memory the slices lie relative to each other? The ring as NCH chunks, allocated back to back or
    // with spacer allocations of `gap` GiB between them (kept while measuring), several trials each.
    printf("# chunks x spacer GiB : us per step over trials\n");
    for (int nch : {1, 4, 10, 20})
      for (int gap : {0, 8, 24}) {
        if (nch == 1 && gap) continue;
        if ((size_t)nch * gap > 230) continue;
        printf("%2d chunks, %2d GiB spacers:", nch, gap);
        for (int trial = 0; trial < 5; ++trial) {
          Chunks c;
          c.per = STEPS / nch;
          std::vector<char *> spacers;
          bool ok = true;
          for (int k = 0; k < nch && ok; ++k) {
            ok = hipMalloc(&c.b[k], (size_t)c.per * TILES * PIECE) == hipSuccess && hipMalloc(&c.r[k], (size_t)c.per * TILES * RPIECE) == hipSuccess;
            if (gap && k + 1 < nch) {
              char *sp = nullptr;
              if (hipMalloc(&sp, (size_t)gap << 30) == hipSuccess) spacers.push_back(sp); else ok = false;
            }
          }
          if (!ok) { (void)hipGetLastError(); printf(" (alloc failed)"); break; }
          printf(" %.2f", us_per_step_chunks(c));
          // (nothing is freed between trials on purpose for gap = 0: the next trial lands elsewhere; spacers are returned)
          for (char *sp : spacers) CK(hipFree(sp));
          fflush(stdout);
        }
        printf("\n");
      }
    return 0;
  }
  if (argc > 1 && !strcmp(argv[1], "pmc")) {
    // for rocprofv3 --pmc: sixteen pairs, three launches each, nothing else -- counters per dispatch against its duration
    for (int a = 0; a < 16; ++a) {
      char *b, *r;
      CK(hipMalloc(&b, bb));
      CK(hipMalloc(&r, rb));
      for (int l = 0; l < 3; ++l) hipLaunchKernelGGL(ring_writer, dim3(TILES / 4), dim3(256), 0, 0, b, r, 1, (size_t)0, 1);
      CK(hipDeviceSynchronize());
    }
    return 0;
  }
  if (argc > 1 && !strcmp(argv[1], "perm")) {
    // the same allocations, other ORDERS of the same bytes: step s at slice position (s * mul) % 100 (neighbours in time far apart
    // in memory), slices shifted by irregular pads, tiles permuted inside a slice
    const size_t extra = (size_t)512 << 20;
    printf("# pair: identity | slice order x37 | x51 | pad 1 MiB+4 KB per slice | pad 3.3 MB | tiles x4099 | tiles x4099 + slices x37\n");
    for (int a = 0; a < 12; ++a) {
      char *b, *r;
      CK(hipMalloc(&b, bb + extra));
      CK(hipMalloc(&r, rb + extra));
      printf("pair %2d: %.2f | %.2f | %.2f | %.2f | %.2f | %.2f | %.2f\n", a, us_per_step(b, r), us_per_step(b, r, 37), us_per_step(b, r, 51),
             us_per_step(b, r, 1, ((size_t)1 << 20) + 4096), us_per_step(b, r, 1, 3300000 / 16 * 16), us_per_step(b, r, 1, 0, 4099),
             us_per_step(b, r, 37, 0, 4099));
      fflush(stdout);
    }
    return 0;
  }
  if (argc > 1 && !strncmp(argv[1], "map", 3)) {
    // "map": the rings' own sizes (3.05 GB + 0.42 GB); "map4g": each allocation rounded up to a power of two (4 GiB + 512 MiB):
    // does a buddy-allocated, naturally aligned block always land on the fast side?  "mapbig": ONE allocation of 8 GiB per pair,
    // boards at +0, records at +4 GiB
    const bool pow2 = !strcmp(argv[1], "map4g"), big = !strcmp(argv[1], "mapbig");
    const size_t ab = pow2 ? (size_t)4 << 30 : (big ? (size_t)8 << 30 : bb), ar = pow2 ? (size_t)512 << 20 : rb;
    // the whole memory: pairs allocated until ~250 GB are held, each timed once; then everything freed and the same again
    for (int pass = 0; pass < 2; ++pass) {
      std::vector<char *> Bs, Rs;
      printf("# pass %d: pair, GB allocated before it, us per step\n", pass);
      double before = 0;
      for (int a = 0; a < (pow2 || big ? 30 : 72); ++a) {
        char *b = nullptr, *r = nullptr;
        if (hipMalloc(&b, ab) != hipSuccess) { (void)hipGetLastError(); break; }
        if (big) r = b + ((size_t)4 << 30);
        else if (hipMalloc(&r, ar) != hipSuccess) { (void)hipGetLastError(); break; }
        Bs.push_back(b);
        Rs.push_back(r);
        printf("%d %3d %6.1f %p %.2f\n", pass, a, before, (void *)b, us_per_step(b, r));
        before += (big ? ab : ab + ar) / 1e9;
        fflush(stdout);
      }
      // a second look at a few of them, in reverse order
      for (int a = (int)Bs.size() - 1; a >= 0; a -= 9) printf("%d again %3d %.2f\n", pass, a, us_per_step(Bs[a], Rs[a]));
      for (size_t a = 0; a < Bs.size(); ++a) { CK(hipFree(Bs[a])); if (!big) CK(hipFree(Rs[a])); }
    }
    return 0;
  }
  const int N = 10;
  char *B[N], *R[N];
  printf("# ring pattern, us per step (median of 3 launches); pairs allocated one after another and all kept\n");
  for (int a = 0; a < N; ++a) {
    CK(hipMalloc(&B[a], bb + (a == 0 ? slack : 0)));
    CK(hipMalloc(&R[a], rb + (a == 0 ? slack : 0)));
  }
  for (int round = 0; round < 2; ++round)
    for (int a = 0; a < N; ++a)
      printf("round %d  pair %2d  boards %p  recs %p : %.2f us per step\n", round, a, (void *)B[a], (void *)R[a], us_per_step(B[a], R[a]));
  printf("# pair 0 at byte offsets into its (1 GiB larger) allocations\n");
  const size_t offs[] = {0, 4096, 65536, (size_t)2 << 20, (size_t)32 << 20, (size_t)256 << 20, (size_t)512 << 20, (size_t)1 << 30, 0};
  for (size_t o : offs) printf("offset %10zu : %.2f us per step\n", o, us_per_step(B[0] + o, R[0] + o));
  printf("# boards of pair i with records of pair j\n");
  for (int i = 0; i < 4; ++i) {
    for (int j = 0; j < 4; ++j) printf(" %.2f", us_per_step(B[i], R[j]));
    printf("\n");
  }
  printf("# freed and allocated again\n");
  for (int a = 0; a < N; ++a) {
    CK(hipFree(B[a]));
    CK(hipFree(R[a]));
  }
  for (int a = 0; a < 4; ++a) {
    CK(hipMalloc(&B[a], bb));
    CK(hipMalloc(&R[a], rb));
    printf("new pair %d  boards %p  recs %p : %.2f us per step\n", a, (void *)B[a], (void *)R[a], us_per_step(B[a], R[a]));
  }
  return 0;
}
