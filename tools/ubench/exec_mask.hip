// Does a wave64 VALU instruction cost less when part of its lanes are switched off? Every SIMD runs
// W waves of N back-to-back independent v_fma_f32 under an EXEC mask; prints cycles per
// wave-instruction per SIMD (4 = the cost of a wave64 instruction with every lane on).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_masked(float *out, int iters, unsigned long long mask) {
  float a0 = threadIdx.x * 1e-3f + 1, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const float b = 1.0000001f, c = 1e-9f;
  const int lane = threadIdx.x & 63;
  if ((mask >> lane) & 1ull) {
    for (int i = 0; i < iters; i++) {
      asm volatile(
          "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
          "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
          : "v"(b), "v"(c));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

int main() {
  hipDeviceProp_t p;
  CHK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  const int iters = 40000;
  struct M { const char *name; unsigned long long m; } ms[] = {
      {"all 64 lanes", ~0ull},          {"lanes 0-31", 0xffffffffull},       {"lanes 32-63", 0xffffffff00000000ull},
      {"lanes 0-15", 0xffffull},        {"lanes 16-31", 0xffff0000ull},      {"lanes 0-15 and 32-47", 0x0000ffff0000ffffull},
      {"even lanes", 0x5555555555555555ull}, {"lane 0", 1ull},               {"lanes 0-47", 0xffffffffffffull},
      {"lane 0 and lane 63", 0x8000000000000001ull}};
  for (int wpe : {4, 8}) {
    const int blocks = cus * wpe;
    float *out;
    CHK(hipMalloc(&out, sizeof(float) * blocks * 256));
    printf("waves per SIMD: %d\n", wpe);
    for (auto &m : ms) {
      hipEvent_t e0, e1;
      CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
      hipLaunchKernelGGL(k_masked, blocks, 256, 0, 0, out, 1000, m.m);
      CHK(hipDeviceSynchronize());
      CHK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_masked, blocks, 256, 0, 0, out, iters, m.m);
      CHK(hipEventRecord(e1));
      CHK(hipDeviceSynchronize());
      float t;
      CHK(hipEventElapsedTime(&t, e0, e1));
      const double ns = (double)t * 1e6 / ((double)iters * 8 * wpe);
      printf("  %-24s %7.3f ms  %6.3f ns per wave-instruction per SIMD (= %5.2f cycles at 2.4 GHz)\n", m.name, t, ns, ns * 2.4);
    }
    CHK(hipFree(out));
  }
  return 0;
}
