// exp_fill_shapes.hip -- standalone probe (not part of the product): why does torch's fill_ write 6.3-6.9 TB/s where the grid-stride
// fill of exp_write_patterns.hip writes 4.4-4.9? The same bytes (16-byte stores, every byte once) issued in different SHAPES:
// persistent workgroups striding over the buffer, one short-lived workgroup per 4 / 16 / 64 KB, a wave owning a contiguous run.
// Timed over several back-to-back launches (one event pair around them), buffers of 256 MB, 1 GB and 3 GB.
//   hipcc --offload-arch=gfx950 -O3 tools/exp_fill_shapes.hip -o /tmp/fill_shapes && /tmp/fill_shapes
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

// persistent workgroups, consecutive lanes consecutive 16-byte chunks, stride = the whole grid
__global__ __launch_bounds__(256) void fill_stride(u32x4 *dst, size_t n16) {
  u32x4 x = {1, 2, 3, 4};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = x;
}

// the same, but a wave never has more than W of its stores outstanding (s_waitcnt vmcnt(W) after each)
template <int W>
__global__ __launch_bounds__(256) void fill_stride_capped(u32x4 *dst, size_t n16) {
  u32x4 x = {1, 2, 3, 4};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
    dst[i] = x;
    if (W == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (W == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    if (W == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    if (W == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    if (W == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
}

// persistent, but the waves of a workgroup issue each iteration's stores TOGETHER (a barrier per iteration): T * 16 contiguous bytes at a time
template <int T>
__global__ __launch_bounds__(T) void fill_stride_sync(u32x4 *dst, size_t n16) {
  u32x4 x = {1, 2, 3, 4};
  const size_t iters = (n16 + (size_t)gridDim.x * T - 1) / ((size_t)gridDim.x * T);
  for (size_t it = 0; it < iters; ++it) {
    const size_t i = (it * gridDim.x + blockIdx.x) * T + threadIdx.x;
    __syncthreads();
    if (i < n16) dst[i] = x;
  }
}

// one workgroup per PER * 4 KB: PER stores per lane, the workgroup's region contiguous
template <int PER>
__global__ __launch_bounds__(256) void fill_oneshot(u32x4 *dst, size_t n16) {
  u32x4 x = {1, 2, 3, 4};
  const size_t base = (size_t)blockIdx.x * 256 * PER + threadIdx.x;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const size_t i = base + (size_t)k * 256;
    if (i < n16) dst[i] = x;
  }
}

// persistent workgroups, but each takes a contiguous RUN of the buffer (its share), walking through it 4 KB at a time
__global__ __launch_bounds__(256) void fill_runs(u32x4 *dst, size_t n16) {
  u32x4 x = {1, 2, 3, 4};
  const size_t per = (n16 + gridDim.x - 1) / gridDim.x;
  const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < n16 ? lo + per : n16;
  for (size_t i = lo + threadIdx.x; i < hi; i += 256) dst[i] = x;
}

// persistent workgroups that pick the next 16 KB piece from an atomic counter (dynamic, in address order)
__global__ __launch_bounds__(256) void fill_ticket(u32x4 *dst, size_t n16, unsigned long long *ticket) {
  u32x4 x = {1, 2, 3, 4};
  __shared__ unsigned long long piece;
  for (;;) {
    if (threadIdx.x == 0) piece = atomicAdd(ticket, 1ull);
    __syncthreads();
    const size_t base = (size_t)piece * 1024;
    __syncthreads();
    if (base >= n16) break;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const size_t i = base + (size_t)k * 256 + threadIdx.x;
      if (i < n16) dst[i] = x;
    }
  }
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const size_t cap = (size_t)3000 * 1000 * 1000;
  char *buf;
  CK(hipMalloc(&buf, cap));
  unsigned long long *ticket;
  CK(hipMalloc(&ticket, 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  printf("# %s, %d CUs; 16-byte plain stores, every byte of the buffer once per launch; 6 launches per timing, best of 3\n", prop.name, cus);
  printf("%-44s %8s | %10s %8s\n", "shape", "MB", "us/launch", "TB/s");
  for (size_t mb : {256, 1000, 3000}) {
    const size_t n16 = mb * 1000 * 1000 / 16;
    for (int shape = 0; shape < 26; ++shape) {
      if (shape == 9 || shape == 10) continue;  // 16 KB tickets: 1.2-1.35 TB/s (same-address atomics serialise), measured once
      float best = 1e9;
      const char *name = "";
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0, 0));
        for (int l = 0; l < 6; ++l) {
          switch (shape) {
            case 0: name = "grid-stride, 8 workgroups per CU"; hipLaunchKernelGGL(fill_stride, dim3(cus * 8), dim3(256), 0, 0, (u32x4 *)buf, n16); break;
            case 1: name = "grid-stride, 2 workgroups per CU"; hipLaunchKernelGGL(fill_stride, dim3(cus * 2), dim3(256), 0, 0, (u32x4 *)buf, n16); break;
            case 2: name = "grid-stride, 32 workgroups per CU"; hipLaunchKernelGGL(fill_stride, dim3(cus * 32), dim3(256), 0, 0, (u32x4 *)buf, n16); break;
            case 3: name = "one workgroup per 4 KB"; hipLaunchKernelGGL(fill_oneshot<1>, dim3((n16 + 255) / 256), dim3(256), 0, 0, (u32x4 *)buf, n16); break;
            case 4: name = "one workgroup per 16 KB"; hipLaunchKernelGGL(fill_oneshot<4>, dim3((n16 + 1023) / 1024), dim3(256), 0, 0, (u32x4 *)buf, n16); break;
            case 5: name = "one workgroup per 64 KB"; hipLaunchKernelGGL(fill_oneshot<16>, dim3((n16 + 4095) / 4096), dim3(256), 0, 0, (u32x4 *)buf, n16); break;
            case 6: name = "one workgroup per 256 KB"; hipLaunchKernelGGL(fill_oneshot<64>, dim3((n16 + 16383) / 16384), dim3(256), 0, 0, (u32x4 *)buf, n16); break;
            case 7: name = "contiguous run per workgroup, 8 per CU"; hipLaunchKernelGGL(fill_runs, dim3(cus * 8), dim3(256), 0, 0, (u32x4 *)buf, n16); break;
            case 8: name = "contiguous run per workgroup, 64 per CU"; hipLaunchKernelGGL(fill_runs, dim3(cus * 64), dim3(256), 0, 0, (u32x4 *)buf, n16); break;
            case 9: name = "16 KB tickets, 8 workgroups per CU"; CK(hipMemsetAsync(ticket, 0, 8, 0)); hipLaunchKernelGGL(fill_ticket, dim3(cus * 8), dim3(256), 0, 0, (u32x4 *)buf, n16, ticket); break;
            case 10: name = "16 KB tickets, 4 workgroups per CU"; CK(hipMemsetAsync(ticket, 0, 8, 0)); hipLaunchKernelGGL(fill_ticket, dim3(cus * 4), dim3(256), 0, 0, (u32x4 *)buf, n16, ticket); break;
            case 11: name = "hipMemsetAsync"; CK(hipMemsetAsync(buf, 3, n16 * 16, 0)); break;
            case 20: name = "grid-stride + barrier per iteration, 8/CU"; hipLaunchKernelGGL(fill_stride_sync<256>, dim3(cus * 8), dim3(256), 0, 0, (u32x4 *)buf, n16); break;
            case 21: name = "grid-stride + barrier per iteration, 2/CU"; hipLaunchKernelGGL(fill_stride_sync<256>, dim3(cus * 2), dim3(256), 0, 0, (u32x4 *)buf, n16); break;
            case 22: name = "grid-stride + barrier, 1024-lane groups, 2/CU"; hipLaunchKernelGGL(fill_stride_sync<1024>, dim3(cus * 2), dim3(1024), 0, 0, (u32x4 *)buf, n16); break;
            case 23: name = "grid-stride + barrier, 1024-lane groups, 1/CU"; hipLaunchKernelGGL(fill_stride_sync<1024>, dim3(cus), dim3(1024), 0, 0, (u32x4 *)buf, n16); break;
            case 24: name = "grid-stride + barrier, 64-lane groups, 8/CU"; hipLaunchKernelGGL(fill_stride_sync<64>, dim3(cus * 8), dim3(64), 0, 0, (u32x4 *)buf, n16); break;
            case 25: name = "grid-stride + barrier, 64-lane groups, 32/CU"; hipLaunchKernelGGL(fill_stride_sync<64>, dim3(cus * 32), dim3(64), 0, 0, (u32x4 *)buf, n16); break;
            case 12: name = "grid-stride 8/CU, <= 0 stores outstanding"; hipLaunchKernelGGL(fill_stride_capped<0>, dim3(cus * 8), dim3(256), 0, 0, (u32x4 *)buf, n16); break;
            case 13: name = "grid-stride 8/CU, <= 1 outstanding"; hipLaunchKernelGGL(fill_stride_capped<1>, dim3(cus * 8), dim3(256), 0, 0, (u32x4 *)buf, n16); break;
            case 14: name = "grid-stride 8/CU, <= 2 outstanding"; hipLaunchKernelGGL(fill_stride_capped<2>, dim3(cus * 8), dim3(256), 0, 0, (u32x4 *)buf, n16); break;
            case 15: name = "grid-stride 8/CU, <= 4 outstanding"; hipLaunchKernelGGL(fill_stride_capped<4>, dim3(cus * 8), dim3(256), 0, 0, (u32x4 *)buf, n16); break;
            case 16: name = "grid-stride 8/CU, <= 8 outstanding"; hipLaunchKernelGGL(fill_stride_capped<8>, dim3(cus * 8), dim3(256), 0, 0, (u32x4 *)buf, n16); break;
            case 17: name = "grid-stride 5/CU, <= 2 outstanding"; hipLaunchKernelGGL(fill_stride_capped<2>, dim3(cus * 5), dim3(256), 0, 0, (u32x4 *)buf, n16); break;
            case 18: name = "grid-stride 5/CU, <= 4 outstanding"; hipLaunchKernelGGL(fill_stride_capped<4>, dim3(cus * 5), dim3(256), 0, 0, (u32x4 *)buf, n16); break;
            case 19: name = "grid-stride 16/CU, <= 1 outstanding"; hipLaunchKernelGGL(fill_stride_capped<1>, dim3(cus * 16), dim3(256), 0, 0, (u32x4 *)buf, n16); break;
          }
        }
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        if (rep && t < best) best = t;
      }
      CK(hipGetLastError());
      printf("%-44s %8zu | %10.1f %8.2f\n", name, mb, best * 1e3 / 6, n16 * 16.0 * 6 / (best * 1e-3) / 1e12);
      fflush(stdout);
    }
  }
  return 0;
}
