// Issue rates of the VALU instructions the traversal kernels are made of, on one MI355X:
// every SIMD runs W waves of N back-to-back independent instructions of one kind; prints
// cycles per wave-instruction per SIMD (4 = full rate for wave64).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define REP8(S) S S S S S S S S
#define BODY(NAME, ASM8)                                                                         \
  __global__ void __launch_bounds__(256) NAME(double *out, int iters, long long *cyc) {          \
    double a0 = threadIdx.x * 1e-3 + 1, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, \
           a7 = a0 + 7;                                                                          \
    double b = 1.0000001, c = 1e-9;                                                              \
    long long t0 = __builtin_amdgcn_s_memtime();                                                 \
    for (int i = 0; i < iters; i++) { asm volatile(ASM8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)); } \
    long long t1 = __builtin_amdgcn_s_memtime();                                                 \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;          \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                             \
  }
// operands: %0..%7 accumulators (64-bit pairs), %8 = b, %9 = c
#define I8(op3) op3("%0") op3("%1") op3("%2") op3("%3") op3("%4") op3("%5") op3("%6") op3("%7")
#define MULF64(r) "v_mul_f64 " r ", " r ", %8\n"
#define ADDF64(r) "v_add_f64 " r ", " r ", %9\n"
#define FMAF64(r) "v_fma_f64 " r ", " r ", %8, %9\n"
#define PKFMA(r) "v_pk_fma_f32 " r ", " r ", %8, %9\n"
#define PKMUL(r) "v_pk_mul_f32 " r ", " r ", %8\n"
#define PKADD(r) "v_pk_add_f32 " r ", " r ", %9\n"
#define RCPF64(r) "v_rcp_f64 " r ", " r "\n"
#define SQRTF64(r) "v_sqrt_f64 " r ", " r "\n"
BODY(k_mul_f64, I8(MULF64))
BODY(k_add_f64, I8(ADDF64))
BODY(k_fma_f64, I8(FMAF64))
BODY(k_pk_fma_f32, I8(PKFMA))
BODY(k_pk_mul_f32, I8(PKMUL))
BODY(k_pk_add_f32, I8(PKADD))
BODY(k_rcp_f64, I8(RCPF64))
BODY(k_sqrt_f64, I8(SQRTF64))

#define BODY32(NAME, ASM8)                                                                       \
  __global__ void __launch_bounds__(256) NAME(double *out, int iters, long long *cyc) {          \
    float a0 = threadIdx.x * 1e-3f + 1, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, \
          a7 = a0 + 7;                                                                           \
    float b = 1.0000001f, c = 1e-9f;                                                             \
    double d = threadIdx.x + 0.5;                                                                \
    long long t0 = __builtin_amdgcn_s_memtime();                                                 \
    for (int i = 0; i < iters; i++) { asm volatile(ASM8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "v"(d)); } \
    long long t1 = __builtin_amdgcn_s_memtime();                                                 \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;          \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                             \
  }
#define FMAF32(r) "v_fma_f32 " r ", " r ", %8, %9\n"
#define MULF32(r) "v_mul_f32 " r ", " r ", %8\n"
#define MIN3F32(r) "v_min3_f32 " r ", " r ", %8, %9\n"
#define CVT3264(r) "v_cvt_f32_f64 " r ", %10\n"
#define RCPF32(r) "v_rcp_f32 " r ", " r "\n"
#define CVTUB(r) "v_cvt_f32_ubyte1 " r ", " r "\n"
#define CNDM(r) "v_cndmask_b32 " r ", " r ", %8, vcc\n"
#define CMPF(r) "v_cmp_lt_f32 vcc, " r ", %8\n"
BODY32(k_fma_f32, I8(FMAF32))
BODY32(k_mul_f32, I8(MULF32))
BODY32(k_min3_f32, I8(MIN3F32))
BODY32(k_cvt_f32_f64, I8(CVT3264))
BODY32(k_rcp_f32, I8(RCPF32))
BODY32(k_cvt_ubyte, I8(CVTUB))
BODY32(k_cndmask, I8(CNDM))
BODY32(k_cmp_f32, I8(CMPF))

int main() {
  hipDeviceProp_t p;
  CHK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  double *out;
  long long *cyc;
  const int iters = 20000;
  struct K { const char *name; void (*f)(double *, int, long long *); } ks[] = {
      {"v_mul_f64", k_mul_f64}, {"v_add_f64", k_add_f64}, {"v_fma_f64", k_fma_f64}, {"v_rcp_f64", k_rcp_f64},
      {"v_sqrt_f64", k_sqrt_f64}, {"v_fma_f32", k_fma_f32}, {"v_mul_f32", k_mul_f32}, {"v_pk_fma_f32", k_pk_fma_f32},
      {"v_pk_mul_f32", k_pk_mul_f32}, {"v_pk_add_f32", k_pk_add_f32}, {"v_min3_f32", k_min3_f32},
      {"v_cvt_f32_f64", k_cvt_f32_f64}, {"v_rcp_f32", k_rcp_f32}, {"v_cvt_f32_ubyte1", k_cvt_ubyte},
      {"v_cndmask_b32", k_cndmask}, {"v_cmp_lt_f32", k_cmp_f32}};
  for (int wpe : {1, 2, 4}) {  // waves per SIMD
    const int blocks = cus * wpe;  // 256 threads = 4 waves = one per SIMD
    CHK(hipMalloc(&out, sizeof(double) * blocks * 256));
    CHK(hipMalloc(&cyc, sizeof(long long) * blocks));
    printf("waves per SIMD: %d\n", wpe);
    for (auto &k : ks) {
      hipEvent_t e0, e1;
      CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
      hipLaunchKernelGGL(k.f, blocks, 256, 0, 0, out, 1000, cyc);
      CHK(hipDeviceSynchronize());
      CHK(hipEventRecord(e0));
      hipLaunchKernelGGL(k.f, blocks, 256, 0, 0, out, iters, cyc);
      CHK(hipEventRecord(e1));
      CHK(hipDeviceSynchronize());
      float ms;
      CHK(hipEventElapsedTime(&ms, e0, e1));
      std::vector<long long> h(blocks);
      CHK(hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost));
      double avg = 0;
      for (auto v : h) avg += v;
      avg /= blocks;
      // s_memtime ticks at 100 MHz: wall time per block; instructions per SIMD = iters * 8 * wpe
      const double ns_per_inst = (double)ms * 1e6 / ((double)iters * 8 * wpe);
      printf("  %-18s %7.3f ms  %6.3f ns per wave-instruction per SIMD (= %5.2f cycles at 2.4 GHz)\n", k.name, ms, ns_per_inst,
             ns_per_inst * 2.4);
    }
    CHK(hipFree(out)); CHK(hipFree(cyc));
  }
  return 0;
}
