// exp_issue_peak.hip -- standalone probe (not part of the product): the chip's instruction ISSUE rates, as wave-instructions per
// second, for the two ports the outputs-once rollout kernel (sgk_rollout_random) is bound by: the vector ALU port and the scalar
// ALU port of a SIMD. Loops of independent register-only instructions, no memory, at the occupancies that kernel runs at
// (6 and 8 waves per SIMD) and at 1 and 2. profiles/issue.json takes its `peak` entries from this tool's output; bench.py's
// `fused_rollout.roofline` divides the kernel's instruction rates (SQ_INSTS_VALU / SQ_INSTS_SALU per wave-step x wave-steps per
// second) by them.
//   hipcc --offload-arch=gfx950 -O3 tools/exp_issue_peak.hip -o /tmp/issue_peak && /tmp/issue_peak
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

// 32 VALU instructions per iteration, eight independent chains
__global__ __launch_bounds__(256) void valu_loop(uint32_t *out, int iters) {
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
__global__ __launch_bounds__(256) void salu_loop(uint32_t *out, int iters) {
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

// both at once: 16 VALU + 16 SALU per iteration, interleaved (what a SIMD can co-issue from one wave's stream and from several)
__global__ __launch_bounds__(256) void mixed_loop(uint32_t *out, int iters) {
  uint32_t a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3;
  uint32_t e = blockIdx.x, f = e + 5, g = e + 6, h = e + 7;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
      asm volatile(
          "v_add_u32 %0, %0, %8\n s_xor_b32 %4, %4, %9\n v_add_u32 %1, %1, %8\n s_xor_b32 %5, %5, %9\n"
          "v_add_u32 %2, %2, %8\n s_xor_b32 %6, %6, %9\n v_add_u32 %3, %3, %8\n s_xor_b32 %7, %7, %9\n"
          : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+s"(e), "+s"(f), "+s"(g), "+s"(h)
          : "v"(i), "s"(i)
          : "scc");
  }
  if (a + b + c + d + e + f + g + h == 0x12345) out[0] = a;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  uint32_t *out;
  CK(hipMalloc(&out, 64));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int iters = 20000;
  printf("# %s, %d CUs (%d SIMDs); loops of %d iterations\n", prop.name, cus, cus * 4, iters);
  printf("%-8s %12s | %18s %22s %16s\n", "loop", "waves/SIMD", "ms", "G wave-instr/s (chip)", "cycles/instr/SIMD @2.4GHz");
  for (int kind = 0; kind < 3; ++kind) {
    for (int wps : {1, 2, 4, 6, 8}) {
      // `wps` waves per SIMD = wps 256-lane workgroups per CU
      const int grid = cus * wps;
      float best = 1e9;
      for (int r = 0; r < 4; ++r) {
        CK(hipEventRecord(e0, 0));
        if (kind == 0) hipLaunchKernelGGL(valu_loop, dim3(grid), dim3(256), 0, 0, out, iters);
        else if (kind == 1) hipLaunchKernelGGL(salu_loop, dim3(grid), dim3(256), 0, 0, out, iters);
        else hipLaunchKernelGGL(mixed_loop, dim3(grid), dim3(256), 0, 0, out, iters);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        if (r && t < best) best = t;
      }
      const double per_iter = kind == 2 ? 32.0 : 32.0;  // wave-instructions of the measured kind(s) per iteration
      const double instr = (double)grid * 4 * iters * per_iter;
      const double rate = instr / (best * 1e-3);
      printf("%-8s %12d | %18.3f %22.1f %16.2f%s\n", kind == 0 ? "VALU" : kind == 1 ? "SALU" : "VALU+SALU", wps, best, rate / 1e9,
             2.4e9 * cus * 4 / rate, kind == 2 ? "  (16 + 16 per iteration: each kind runs at half this rate)" : "");
    }
  }
  return 0;
}
