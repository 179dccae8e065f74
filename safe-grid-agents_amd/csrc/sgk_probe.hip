// sgk_probe.hip -- what the chip can do NOW, measured in the caller's own process: the instruction-issue ceilings the
// outputs-once rollout kernel (sgk_rollout_random) is held against. That kernel stores nothing per step; its bound is the rate
// at which a SIMD issues vector-ALU and scalar-ALU instructions, and that rate moves with the clock the box happens to run at
// -- a committed figure from another box made the fraction in bench.py's line irreproducible (round 3: 0.51 measured against a
// claimed 0.62). Register-only loops of independent instructions, no memory.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sgk_host_core.h"

using sgk::host::fail;
using sgk::host::hip_fail;

namespace {

// 32 VALU instructions per iteration, eight independent chains
__global__ __launch_bounds__(256) void valu_issue_loop(uint32_t *out, int iters) {
  uint32_t a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3, e = a + 4, f = a + 5, g = a + 6, h = a + 7;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
      asm volatile(
          "v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
          "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
          : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h)
          : "v"(i));
  }
  if (a + b + c + d + e + f + g + h == 0x12345) out[0] = a;
}

// 32 SALU instructions per iteration, eight independent chains
__global__ __launch_bounds__(256) void salu_issue_loop(uint32_t *out, int iters) {
  uint32_t a = blockIdx.x, b = a + 1, c = a + 2, d = a + 3, e = a + 4, f = a + 5, g = a + 6, h = a + 7;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
      asm volatile(
          "s_xor_b32 %0, %0, %8\n s_xor_b32 %1, %1, %8\n s_xor_b32 %2, %2, %8\n s_xor_b32 %3, %3, %8\n"
          "s_xor_b32 %4, %4, %8\n s_xor_b32 %5, %5, %8\n s_xor_b32 %6, %6, %8\n s_xor_b32 %7, %7, %8\n"
          : "+s"(a), "+s"(b), "+s"(c), "+s"(d), "+s"(e), "+s"(f), "+s"(g), "+s"(h)
          : "s"(i)
          : "scc");
  }
  if (a + b + c + d + e + f + g + h == 0x12345) out[0] = a;
}

}  // namespace

extern "C" int sgk_issue_peak(int32_t device, int32_t waves_per_simd, double *valu_wave_instr_per_s, double *salu_wave_instr_per_s) try {
  if (!valu_wave_instr_per_s || !salu_wave_instr_per_s) return fail(SGK_ERR_INVALID, "NULL output");
  if (waves_per_simd < 1 || waves_per_simd > 8) return fail(SGK_ERR_INVALID, "waves_per_simd must be 1 .. 8");
  // takes no handle and runs in the middle of a caller's program (bench.py): the thread's current device is put back on return.
  // The call synchronises (it times its own launches on a stream of its own).
  sgk::host::DeviceGuard keep_current_device;
  hipError_t err = hipSetDevice(device);
  if (err != hipSuccess) return hip_fail(err, "hipSetDevice");
  hipDeviceProp_t prop;
  err = hipGetDeviceProperties(&prop, device);
  if (err != hipSuccess) return hip_fail(err, "hipGetDeviceProperties");
  const int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  const int grid = cus * waves_per_simd, iters = 10000;  // w 256-lane workgroups per CU = w waves on every SIMD
  uint32_t *out = nullptr;
  hipStream_t st = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  err = hipMalloc(&out, 64);
  if (err == hipSuccess) err = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  if (err == hipSuccess) err = hipEventCreate(&e0);
  if (err == hipSuccess) err = hipEventCreate(&e1);
  double rate[2] = {0.0, 0.0};
  for (int kind = 0; kind < 2 && err == hipSuccess; ++kind) {
    float best = 1e30f;
    for (int r = 0; r < 4 && err == hipSuccess; ++r) {  // the first pass warms up (code upload, clocks); the best of three counts
      err = hipEventRecord(e0, st);
      if (err != hipSuccess) break;
      if (kind == 0) hipLaunchKernelGGL(valu_issue_loop, dim3(grid), dim3(256), 0, st, out, iters);
      else hipLaunchKernelGGL(salu_issue_loop, dim3(grid), dim3(256), 0, st, out, iters);
      err = hipGetLastError();
      if (err == hipSuccess) err = hipEventRecord(e1, st);
      if (err == hipSuccess) err = hipEventSynchronize(e1);
      float ms = 0;
      if (err == hipSuccess) err = hipEventElapsedTime(&ms, e0, e1);
      if (r > 0 && ms > 0 && ms < best) best = ms;
    }
    rate[kind] = (double)grid * 4.0 * iters * 32.0 / ((double)best * 1e-3);  // waves x iterations x instructions per iteration
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (st) (void)hipStreamDestroy(st);
  (void)hipFree(out);
  if (err != hipSuccess) return hip_fail(err, "sgk_issue_peak");
  *valu_wave_instr_per_s = rate[0];
  *salu_wave_instr_per_s = rate[1];
  return SGK_OK;
} SGK_CATCH_STATUS
