// exp_fill_segments.hip -- standalone probe (not part of the product): is the write rate a property of WHERE in an allocation the
// bytes go? One-shot 4 KB workgroups (the shape that writes 6.9-7.0 TB/s, tools/exp_fill_shapes.hip) over 512 MB segments of a
// few allocations, each segment filled 6 times back to back; then the same over the whole allocation. A ring whose rate differs
// from allocation to allocation would show as slow segments here if physical placement were the cause.
//   hipcc --offload-arch=gfx950 -O3 tools/exp_fill_segments.hip -o /tmp/fill_segments && /tmp/fill_segments
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void fill_oneshot(u32x4 *dst, size_t n16) {
  u32x4 x = {1, 2, 3, 4};
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n16) dst[i] = x;
}

// the persistent shape over the same bytes (8 workgroups per CU, grid-stride)
__global__ __launch_bounds__(256) void fill_stride(u32x4 *dst, size_t n16) {
  u32x4 x = {1, 2, 3, 4};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = x;
}

static double time_fill(char *p, size_t bytes, int shape, int cus) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const size_t n16 = bytes / 16;
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, 0));
    for (int l = 0; l < 6; ++l) {
      if (shape == 0) hipLaunchKernelGGL(fill_oneshot, dim3((n16 + 255) / 256), dim3(256), 0, 0, (u32x4 *)p, n16);
      else hipLaunchKernelGGL(fill_stride, dim3(cus * 8), dim3(256), 0, 0, (u32x4 *)p, n16);
    }
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float t;
    CK(hipEventElapsedTime(&t, e0, e1));
    if (rep && t < best) best = t;
  }
  CK(hipEventDestroy(e0));
  CK(hipEventDestroy(e1));
  return bytes * 6.0 / (best * 1e-3) / 1e12;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const size_t seg = (size_t)512 << 20;
  printf("# %s; TB/s written, one-shot 4 KB workgroups (and the persistent grid-stride shape), 512 MiB segments\n", prop.name);
  const size_t sizes[] = {(size_t)3 << 30, (size_t)3 << 30, (size_t)8 << 30, (size_t)3 << 30};
  char *keep[4];
  for (int a = 0; a < 4; ++a) {
    char *p;
    CK(hipMalloc(&p, sizes[a]));
    keep[a] = p;
    printf("allocation %d: %zu GiB at %p\n", a, sizes[a] >> 30, (void *)p);
    printf("  segment  :");
    for (size_t s = 0; s * seg < sizes[a]; ++s) printf(" %5zu", s);
    printf(" | whole\n  one-shot :");
    for (size_t s = 0; s * seg < sizes[a]; ++s) printf(" %5.2f", time_fill(p + s * seg, seg, 0, cus));
    printf(" | %5.2f\n  persist. :", time_fill(p, sizes[a], 0, cus));
    for (size_t s = 0; s * seg < sizes[a]; ++s) printf(" %5.2f", time_fill(p + s * seg, seg, 1, cus));
    printf(" | %5.2f\n", time_fill(p, sizes[a], 1, cus));
    // a window of 3 GiB sliding over the 8 GiB allocation
    if (sizes[a] > ((size_t)3 << 30)) {
      printf("  3 GiB windows at +0, +1, ... GiB: one-shot");
      for (size_t o = 0; o + ((size_t)3 << 30) <= sizes[a]; o += (size_t)1 << 30) printf(" %5.2f", time_fill(p + o, (size_t)3 << 30, 0, cus));
      printf(" | persistent");
      for (size_t o = 0; o + ((size_t)3 << 30) <= sizes[a]; o += (size_t)1 << 30) printf(" %5.2f", time_fill(p + o, (size_t)3 << 30, 1, cus));
      printf("\n");
    }
    fflush(stdout);
  }
  for (int a = 0; a < 4; ++a) CK(hipFree(keep[a]));
  return 0;
}
