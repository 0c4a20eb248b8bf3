// Issue rates of the VALU instructions the traversal kernels are made of, on one MI355X, in SHADER
// CYCLES. Every SIMD runs W waves of N back-to-back independent instructions of one kind. Each wave
// stamps both clocks around its loop: s_memtime (ticks of the shader clock) and s_memrealtime (a fixed
// 100 MHz), so that
//   cycles per wave-instruction per SIMD = median(d memtime) / (N * W)
//   clock the loop ran at               = d memtime / d memrealtime * 100 MHz
// and neither figure depends on an assumed frequency. How many waves really shared a SIMD is not
// assumed either: every wave records where it ran (HW_ID, XCC_ID) and when (s_memrealtime at its
// start and end), and W is the time-average number of waves resident on the wave's SIMD while it
// ran, computed from those intervals (`res` in the table; a launch of CUs x W workgroups does not
// place exactly W on every CU). The wall-time figure (HIP events, all 4 x CUs SIMDs assumed busy
// throughout) is printed beside it as an upper bound. A wave64 instruction at full rate on a
// SIMD-32 takes 2 cycles.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Stamp { long long cyc, real, start, end; unsigned hw, xcc; };
// HW_REG_HW_ID (id 4) and HW_REG_XCC_ID (id 20), all 32 bits: ((size - 1) << 11) | (offset << 6) | id
#define GETREG_HW_ID ((31 << 11) | 4)
#define GETREG_XCC_ID ((31 << 11) | 20)
#define STAMP_STORE(t0, t1, r0, r1)                                                                \
  if ((threadIdx.x & 63) == 0)                                                                     \
    st[blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{t1 - t0, r1 - r0, r0, r1, __builtin_amdgcn_s_getreg(GETREG_HW_ID), \
                                                    __builtin_amdgcn_s_getreg(GETREG_XCC_ID)};

// mask: lanes that execute the loop (the others skip it; the wave runs it under that EXEC mask)
#define BODY(NAME, TYPE, INIT, ASM8)                                                             \
  __global__ void __launch_bounds__(256) NAME(double *out, int iters, Stamp *st, unsigned long long mask) { \
    TYPE a0 = (TYPE)(threadIdx.x * 1e-3 + 1), a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, \
         a6 = a0 + 6, a7 = a0 + 7;                                                               \
    INIT                                                                                         \
    const long long t0 = __builtin_amdgcn_s_memtime();                                           \
    const long long r0 = __builtin_amdgcn_s_memrealtime();                                       \
    if ((mask >> (threadIdx.x & 63)) & 1ull) {                                                   \
      for (int i = 0; i < iters; i++) { asm volatile(ASM8 ASM8 ASM8 ASM8 ASM8 ASM8 ASM8 ASM8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "v"(d), "v"(w32) : "vcc"); } \
    }                                                                                            \
    const long long t1 = __builtin_amdgcn_s_memtime();                                           \
    const long long r1 = __builtin_amdgcn_s_memrealtime();                                       \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (double)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7); \
    STAMP_STORE(t0, t1, r0, r1)                                                                  \
  }
#define INIT64 const double b = 1.0000001, c = 1e-9; const double d = threadIdx.x + 0.5; const unsigned w32 = threadIdx.x * 2654435761u + 12345u;
#define INIT32 const float b = 1.0000001f, c = 1e-9f; const double d = threadIdx.x + 0.5; const unsigned w32 = threadIdx.x * 2654435761u + 12345u;
// operands: %0..%7 accumulators, %8 = b, %9 = c, %10 = d (a double)
#define I8(op) op("%0") op("%1") op("%2") op("%3") op("%4") op("%5") op("%6") op("%7")
#define MULF64(r) "v_mul_f64 " r ", " r ", %8\n"
#define ADDF64(r) "v_add_f64 " r ", " r ", %9\n"
#define FMAF64(r) "v_fma_f64 " r ", " r ", %8, %9\n"
#define PKFMA(r) "v_pk_fma_f32 " r ", " r ", %8, %9\n"
#define RCPF64(r) "v_rcp_f64 " r ", " r "\n"
#define SQRTF64(r) "v_sqrt_f64 " r ", " r "\n"
BODY(k_mul_f64, double, INIT64, I8(MULF64))
BODY(k_add_f64, double, INIT64, I8(ADDF64))
BODY(k_fma_f64, double, INIT64, I8(FMAF64))
BODY(k_pk_fma_f32, double, INIT64, I8(PKFMA))
#define DIVSCALE(r) "v_div_scale_f64 " r ", vcc, " r ", %8, " r "\n"
#define DIVFMAS(r) "v_div_fmas_f64 " r ", " r ", %8, %9\n"
#define DIVFIXUP(r) "v_div_fixup_f64 " r ", " r ", %8, %9\n"
#define RSQF64(r) "v_rsq_f64 " r ", " r "\n"
#define LDEXPF64(r) "v_ldexp_f64 " r ", " r ", 3\n"
#define CMPF64(r) "v_cmp_lt_f64 vcc, " r ", %8\n"
#define MAXF64(r) "v_max_f64 " r ", " r ", %8\n"
#define CVTF64F32(r) "v_cvt_f64_f32 " r ", %11\n"
#define CVTF64U32(r) "v_cvt_f64_u32 " r ", %11\n"
#define MADU64(r) "v_mad_u64_u32 " r ", vcc, %11, %11, " r "\n"
#define LSHLADD64(r) "v_lshl_add_u64 " r ", " r ", 3, %8\n"
BODY(k_div_scale_f64, double, INIT64, I8(DIVSCALE))
BODY(k_div_fmas_f64, double, INIT64, I8(DIVFMAS))
BODY(k_div_fixup_f64, double, INIT64, I8(DIVFIXUP))
BODY(k_rsq_f64, double, INIT64, I8(RSQF64))
BODY(k_ldexp_f64, double, INIT64, I8(LDEXPF64))
BODY(k_cmp_f64, double, INIT64, I8(CMPF64))
BODY(k_max_f64, double, INIT64, I8(MAXF64))
BODY(k_cvt_f64_f32, double, INIT64, I8(CVTF64F32))
BODY(k_cvt_f64_u32, double, INIT64, I8(CVTF64U32))
BODY(k_mad_u64_u32, double, INIT64, I8(MADU64))
BODY(k_lshl_add_u64, double, INIT64, I8(LSHLADD64))
BODY(k_rcp_f64, double, INIT64, I8(RCPF64))
BODY(k_sqrt_f64, double, INIT64, I8(SQRTF64))
#define FMAF32(r) "v_fma_f32 " r ", " r ", %8, %9\n"
#define MIN3F32(r) "v_min3_f32 " r ", " r ", %8, %9\n"
#define CVT3264(r) "v_cvt_f32_f64 " r ", %10\n"
#define RCPF32(r) "v_rcp_f32 " r ", " r "\n"
#define CVTUB(r) "v_cvt_f32_ubyte1 " r ", " r "\n"
#define CNDM(r) "v_cndmask_b32 " r ", " r ", %8, vcc\n"
#define CMPF(r) "v_cmp_lt_f32 vcc, " r ", %8\n"
#define LSHR(r) "v_lshrrev_b32 " r ", 3, " r "\n"
#define BFE(r) "v_bfe_u32 " r ", " r ", 8, 8\n"
BODY(k_fma_f32, float, INIT32, I8(FMAF32))
BODY(k_min3_f32, float, INIT32, I8(MIN3F32))
BODY(k_cvt_f32_f64, float, INIT32, I8(CVT3264))
BODY(k_rcp_f32, float, INIT32, I8(RCPF32))
BODY(k_cvt_ubyte, float, INIT32, I8(CVTUB))
BODY(k_cndmask, float, INIT32, I8(CNDM))
BODY(k_cmp_f32, float, INIT32, I8(CMPF))
BODY(k_bfe_u32, float, INIT32, I8(BFE))
#define MULLO(r) "v_mul_lo_u32 " r ", " r ", %11\n"
#define MULHI(r) "v_mul_hi_u32 " r ", " r ", %11\n"
#define XORB32(r) "v_xor_b32 " r ", " r ", %11\n"
#define ADDF32(r) "v_add_f32 " r ", " r ", %8\n"
#define MULF32(r) "v_mul_f32 " r ", " r ", %8\n"
#define MAXF32(r) "v_max_f32 " r ", " r ", %8\n"
#define FMACF32(r) "v_fmac_f32 " r ", %8, %9\n"
#define FMAMOD(r) "v_fma_f32 " r ", |" r "|, -%8, %9\n"
#define MAX3F32(r) "v_max3_f32 " r ", " r ", %8, %9\n"
#define MED3F32(r) "v_med3_f32 " r ", " r ", %8, %9\n"
#define ADDU32(r) "v_add_u32 " r ", " r ", %8\n"
#define ANDB32(r) "v_and_b32 " r ", " r ", %8\n"
#define LSHL(r) "v_lshlrev_b32 " r ", 3, " r "\n"
#define MOVB32(r) "v_mov_b32 " r ", %8\n"
#define CVTU32(r) "v_cvt_f32_u32 " r ", " r "\n"
#define LDEXP(r) "v_ldexp_f32 " r ", " r ", 3\n"
#define MADU24(r) "v_mad_u32_u24 " r ", " r ", %8, %9\n"
#define CNDM64(r) "v_cndmask_b32_e64 " r ", " r ", %8, vcc\n"
#define INIT32V INIT32 asm volatile("s_mov_b32 vcc_lo, 0x55555555\n s_mov_b32 vcc_hi, 0x55555555" ::: "vcc");
BODY(k_mul_lo_u32, float, INIT32, I8(MULLO))
BODY(k_mul_hi_u32, float, INIT32, I8(MULHI))
BODY(k_xor_b32, float, INIT32, I8(XORB32))
BODY(k_add_f32, float, INIT32, I8(ADDF32))
BODY(k_mul_f32, float, INIT32, I8(MULF32))
BODY(k_max_f32, float, INIT32, I8(MAXF32))
BODY(k_fmac_f32, float, INIT32, I8(FMACF32))
BODY(k_fma_mod, float, INIT32, I8(FMAMOD))
BODY(k_max3_f32, float, INIT32, I8(MAX3F32))
BODY(k_med3_f32, float, INIT32, I8(MED3F32))
BODY(k_add_u32, float, INIT32, I8(ADDU32))
BODY(k_and_b32, float, INIT32, I8(ANDB32))
BODY(k_lshl_b32, float, INIT32, I8(LSHL))
BODY(k_mov_b32, float, INIT32, I8(MOVB32))
BODY(k_cvt_f32_u32, float, INIT32, I8(CVTU32))
BODY(k_ldexp_f32, float, INIT32, I8(LDEXP))
BODY(k_mad_u32_u24, float, INIT32, I8(MADU24))
// a select among arithmetic, and the compare + select pair the compiler emits for a ternary, in both encodings
#define G_CND32 "v_cndmask_b32_e32 %0, %0, %8, vcc\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n" \
                "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
#define G_CND64 "v_cndmask_b32_e64 %0, %0, %8, vcc\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n" \
                "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
#define G_CMPCND32 "v_cmp_lt_f32_e32 vcc, %1, %8\n v_cndmask_b32_e32 %0, %0, %8, vcc\n v_cmp_lt_f32_e32 vcc, %3, %8\n v_cndmask_b32_e32 %2, %2, %8, vcc\n" \
                   "v_cmp_lt_f32_e32 vcc, %5, %8\n v_cndmask_b32_e32 %4, %4, %8, vcc\n v_cmp_lt_f32_e32 vcc, %7, %8\n v_cndmask_b32_e32 %6, %6, %8, vcc\n"
#define G_CMPCND64 "v_cmp_lt_f32_e64 s[20:21], %1, %8\n v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n v_cmp_lt_f32_e64 s[22:23], %3, %8\n v_cndmask_b32_e64 %2, %2, %8, s[22:23]\n" \
                   "v_cmp_lt_f32_e64 s[20:21], %5, %8\n v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n v_cmp_lt_f32_e64 s[22:23], %7, %8\n v_cndmask_b32_e64 %6, %6, %8, s[22:23]\n"
#define BODYS(NAME, ASM8)                                                                        \
  __global__ void __launch_bounds__(256) NAME(double *out, int iters, Stamp *st, unsigned long long mask) { \
    float a0 = (float)(threadIdx.x * 1e-3 + 1), a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, \
          a6 = a0 + 6, a7 = a0 + 7;                                                              \
    INIT32V                                                                                      \
    const long long t0 = __builtin_amdgcn_s_memtime();                                           \
    const long long r0 = __builtin_amdgcn_s_memrealtime();                                       \
    if ((mask >> (threadIdx.x & 63)) & 1ull) {                                                   \
      for (int i = 0; i < iters; i++) { asm volatile(ASM8 ASM8 ASM8 ASM8 ASM8 ASM8 ASM8 ASM8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "v"(d) : "vcc", "s20", "s21", "s22", "s23"); } \
    }                                                                                            \
    const long long t1 = __builtin_amdgcn_s_memtime();                                           \
    const long long r1 = __builtin_amdgcn_s_memrealtime();                                       \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (double)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7); \
    STAMP_STORE(t0, t1, r0, r1)                                                                  \
  }
BODYS(k_g_cnd32, G_CND32)
BODYS(k_g_cnd64, G_CND64)
BODYS(k_g_cmpcnd32, G_CMPCND32)
BODYS(k_g_cmpcnd64, G_CMPCND64)
BODY(k_cndmask_set, float, INIT32V, I8(CNDM))
BODY(k_cndmask_e64, float, INIT32V, I8(CNDM64))
// candidates for a cheaper box test: mixed-precision FMA with an FP16 operand, byte permutes, three-operand
// logic, integer and packed 16-bit min / max, FP16 and FP8 conversions
#define FMAMIXLO(r) "v_fma_mix_f32 " r ", %11, %8, " r " op_sel_hi:[1,0,0]\n"
#define FMAMIXHI(r) "v_fma_mix_f32 " r ", %11, %8, " r " op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
#define PERMB32(r) "v_perm_b32 " r ", " r ", %11, %8\n"
#define CVTF16(r) "v_cvt_f32_f16 " r ", %11\n"
#define ANDOR(r) "v_and_or_b32 " r ", " r ", %11, %8\n"
#define BFI(r) "v_bfi_b32 " r ", " r ", %11, %8\n"
#define ALIGNBIT(r) "v_alignbit_b32 " r ", " r ", %11, 8\n"
#define LSHLOR(r) "v_lshl_or_b32 " r ", " r ", 3, %11\n"
#define LSHLADD(r) "v_lshl_add_u32 " r ", " r ", 3, %11\n"
#define ADD3(r) "v_add3_u32 " r ", " r ", %11, %8\n"
#define MAXI32(r) "v_max_i32 " r ", " r ", %11\n"
#define MINU32(r) "v_min_u32 " r ", " r ", %11\n"
#define MAX3I32(r) "v_max3_i32 " r ", " r ", %11, %8\n"
#define PKMAXF16(r) "v_pk_max_f16 " r ", " r ", %11\n"
#define PKFMAF16(r) "v_pk_fma_f16 " r ", " r ", %11, %8\n"
#define PKMAXI16(r) "v_pk_max_i16 " r ", " r ", %11\n"
#define PKADDU16(r) "v_pk_add_u16 " r ", " r ", %11\n"
#define CMPU32(r) "v_cmp_lt_u32 vcc, " r ", %11\n"
#define SUBREV(r) "v_subrev_f32 " r ", %8, " r "\n"
#define DOT2(r) "v_dot2_f32_f16 " r ", %11, %11, " r "\n"
#define DOT2C(r) "v_dot2c_f32_f16 " r ", %11, %11\n"
#define CVTPKFP8(r) "v_cvt_pk_f32_fp8 " r ", %11\n"
#define XAD(r) "v_xad_u32 " r ", " r ", %11, %8\n"
#define OR3(r) "v_or3_b32 " r ", " r ", %11, %8\n"
#define SDWAADD(r) "v_add_f32_sdwa " r ", " r ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n"
BODY(k_fma_mix_lo, float, INIT32, I8(FMAMIXLO))
BODY(k_fma_mix_hi, float, INIT32, I8(FMAMIXHI))
BODY(k_perm_b32, float, INIT32, I8(PERMB32))
BODY(k_cvt_f32_f16, float, INIT32, I8(CVTF16))
BODY(k_and_or_b32, float, INIT32, I8(ANDOR))
BODY(k_bfi_b32, float, INIT32, I8(BFI))
BODY(k_alignbit_b32, float, INIT32, I8(ALIGNBIT))
BODY(k_lshl_or_b32, float, INIT32, I8(LSHLOR))
BODY(k_lshl_add_u32, float, INIT32, I8(LSHLADD))
BODY(k_add3_u32, float, INIT32, I8(ADD3))
BODY(k_max_i32, float, INIT32, I8(MAXI32))
BODY(k_min_u32, float, INIT32, I8(MINU32))
BODY(k_max3_i32, float, INIT32, I8(MAX3I32))
BODY(k_pk_max_f16, float, INIT32, I8(PKMAXF16))
BODY(k_pk_fma_f16, float, INIT32, I8(PKFMAF16))
BODY(k_pk_max_i16, float, INIT32, I8(PKMAXI16))
BODY(k_pk_add_u16, float, INIT32, I8(PKADDU16))
BODY(k_cmp_lt_u32, float, INIT32, I8(CMPU32))
BODY(k_subrev_f32, float, INIT32, I8(SUBREV))
BODY(k_dot2_f32_f16, float, INIT32, I8(DOT2))
BODY(k_dot2c_f32_f16, float, INIT32, I8(DOT2C))
BODY(k_cvt_pk_f32_fp8, double, INIT64, I8(CVTPKFP8))
BODY(k_xad_u32, float, INIT32, I8(XAD))
BODY(k_or3_b32, float, INIT32, I8(OR3))
BODY(k_add_f32_sdwa, float, INIT32, I8(SDWAADD))
// half v_fma_f32, half v_fma_f64 (the f64 ones on the double d, throw-away results in a0..a3's place is
// not possible with one register class, so the mix runs 4 + 4 on separate accumulators)
__global__ void __launch_bounds__(256) k_mix_32_64(double *out, int iters, Stamp *st, unsigned long long mask) {
  float a0 = threadIdx.x * 1e-3f + 1, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  double e0 = a0 + 4, e1 = a0 + 5, e2 = a0 + 6, e3 = a0 + 7;
  const float b = 1.0000001f, c = 1e-9f;
  const double bd = 1.0000001, cd = 1e-9;
  const long long t0 = __builtin_amdgcn_s_memtime();
  const long long r0 = __builtin_amdgcn_s_memrealtime();
  if ((mask >> (threadIdx.x & 63)) & 1ull) {
    for (int i = 0; i < iters; i++) {
#define MIX8 "v_fma_f32 %0, %0, %8, %9\n v_fma_f64 %4, %4, %10, %11\n v_fma_f32 %1, %1, %8, %9\n v_fma_f64 %5, %5, %10, %11\n" \
             "v_fma_f32 %2, %2, %8, %9\n v_fma_f64 %6, %6, %10, %11\n v_fma_f32 %3, %3, %8, %9\n v_fma_f64 %7, %7, %10, %11\n"
      asm volatile(MIX8 MIX8 MIX8 MIX8 MIX8 MIX8 MIX8 MIX8
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3)
          : "v"(b), "v"(c), "v"(bd), "v"(cd));
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  const long long r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + e0 + e1 + e2 + e3;
  STAMP_STORE(t0, t1, r0, r1)
}

typedef void (*Kern)(double *, int, Stamp *, unsigned long long);
struct K { const char *name; Kern f; unsigned long long mask; int iters; };

int main(int argc, char **argv) {
  hipDeviceProp_t p;
  CHK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  const bool quick = argc > 1 && !strcmp(argv[1], "quick");
  const unsigned long long ALL = ~0ull;
  const K ks[] = {
      {"v_fma_f32", k_fma_f32, ALL, 40000},
      {"v_fma_f32 lanes 0-31", k_fma_f32, 0xffffffffull, 40000},
      {"v_fma_f32 even lanes", k_fma_f32, 0x5555555555555555ull, 40000},
      {"v_fma_f32 lanes 0-15", k_fma_f32, 0xffffull, 40000},
      {"v_pk_fma_f32", k_pk_fma_f32, ALL, 20000},
      {"v_fma_f64", k_fma_f64, ALL, 20000},
      {"v_fma_f64 lanes 0-31", k_fma_f64, 0xffffffffull, 20000},
      {"v_mul_f64", k_mul_f64, ALL, 20000},
      {"v_add_f64", k_add_f64, ALL, 20000},
      {"v_fma_f32 + v_fma_f64 1:1", k_mix_32_64, ALL, 20000},
      {"v_rcp_f64", k_rcp_f64, ALL, 10000},
      {"v_rsq_f64", k_rsq_f64, ALL, 10000},
      {"v_div_scale_f64", k_div_scale_f64, ALL, 20000},
      {"v_div_fmas_f64", k_div_fmas_f64, ALL, 20000},
      {"v_div_fixup_f64", k_div_fixup_f64, ALL, 20000},
      {"v_ldexp_f64", k_ldexp_f64, ALL, 20000},
      {"v_cmp_lt_f64", k_cmp_f64, ALL, 20000},
      {"v_max_f64", k_max_f64, ALL, 20000},
      {"v_cvt_f64_f32", k_cvt_f64_f32, ALL, 20000},
      {"v_cvt_f64_u32", k_cvt_f64_u32, ALL, 20000},
      {"v_mad_u64_u32", k_mad_u64_u32, ALL, 20000},
      {"v_lshl_add_u64", k_lshl_add_u64, ALL, 20000},
      {"v_mul_lo_u32", k_mul_lo_u32, ALL, 20000},
      {"v_mul_hi_u32", k_mul_hi_u32, ALL, 20000},
      {"v_xor_b32", k_xor_b32, ALL, 40000},
      {"v_sqrt_f64", k_sqrt_f64, ALL, 10000},
      {"v_min3_f32", k_min3_f32, ALL, 40000},
      {"v_cvt_f32_f64", k_cvt_f32_f64, ALL, 20000},
      {"v_rcp_f32", k_rcp_f32, ALL, 20000},
      {"v_cvt_f32_ubyte1", k_cvt_ubyte, ALL, 40000},
      {"v_bfe_u32", k_bfe_u32, ALL, 40000},
      {"v_cndmask_b32", k_cndmask, ALL, 40000},
      {"v_cmp_lt_f32", k_cmp_f32, ALL, 40000},
      {"v_add_f32", k_add_f32, ALL, 40000},
      {"v_mul_f32", k_mul_f32, ALL, 40000},
      {"v_max_f32", k_max_f32, ALL, 40000},
      {"v_fmac_f32", k_fmac_f32, ALL, 40000},
      {"v_fma_f32 |a|, -b", k_fma_mod, ALL, 40000},
      {"v_max3_f32", k_max3_f32, ALL, 40000},
      {"v_med3_f32", k_med3_f32, ALL, 40000},
      {"v_add_u32", k_add_u32, ALL, 40000},
      {"v_and_b32", k_and_b32, ALL, 40000},
      {"v_lshlrev_b32", k_lshl_b32, ALL, 40000},
      {"v_mov_b32", k_mov_b32, ALL, 40000},
      {"v_cvt_f32_u32", k_cvt_f32_u32, ALL, 40000},
      {"v_ldexp_f32", k_ldexp_f32, ALL, 40000},
      {"v_mad_u32_u24", k_mad_u32_u24, ALL, 40000},
      {"v_cndmask_b32 vcc set", k_cndmask_set, ALL, 40000},
      {"v_cndmask_b32_e64 vcc set", k_cndmask_e64, ALL, 40000},
      {"1 cndmask_e32 + 7 fma_f32", k_g_cnd32, ALL, 40000},
      {"1 cndmask_e64 + 7 fma_f32", k_g_cnd64, ALL, 40000},
      {"cmp_e32 vcc + cndmask_e32", k_g_cmpcnd32, ALL, 40000},
      {"cmp_e64 sgpr + cndmask_e64", k_g_cmpcnd64, ALL, 40000},
      {"v_fma_mix_f32 f16 lo", k_fma_mix_lo, ALL, 40000},
      {"v_fma_mix_f32 f16 hi", k_fma_mix_hi, ALL, 40000},
      {"v_perm_b32", k_perm_b32, ALL, 40000},
      {"v_cvt_f32_f16", k_cvt_f32_f16, ALL, 40000},
      {"v_and_or_b32", k_and_or_b32, ALL, 40000},
      {"v_bfi_b32", k_bfi_b32, ALL, 40000},
      {"v_alignbit_b32", k_alignbit_b32, ALL, 40000},
      {"v_lshl_or_b32", k_lshl_or_b32, ALL, 40000},
      {"v_lshl_add_u32", k_lshl_add_u32, ALL, 40000},
      {"v_add3_u32", k_add3_u32, ALL, 40000},
      {"v_max_i32", k_max_i32, ALL, 40000},
      {"v_min_u32", k_min_u32, ALL, 40000},
      {"v_max3_i32", k_max3_i32, ALL, 40000},
      {"v_pk_max_f16", k_pk_max_f16, ALL, 40000},
      {"v_pk_fma_f16", k_pk_fma_f16, ALL, 40000},
      {"v_pk_max_i16", k_pk_max_i16, ALL, 40000},
      {"v_pk_add_u16", k_pk_add_u16, ALL, 40000},
      {"v_cmp_lt_u32", k_cmp_lt_u32, ALL, 40000},
      {"v_subrev_f32", k_subrev_f32, ALL, 40000},
      {"v_dot2_f32_f16", k_dot2_f32_f16, ALL, 40000},
      {"v_dot2c_f32_f16", k_dot2c_f32_f16, ALL, 40000},
      {"v_cvt_pk_f32_fp8", k_cvt_pk_f32_fp8, ALL, 20000},
      {"v_xad_u32", k_xad_u32, ALL, 40000},
      {"v_or3_b32", k_or3_b32, ALL, 40000},
      {"v_add_f32_sdwa", k_add_f32_sdwa, ALL, 40000}};
  const size_t first_new = 53;  // "new": only the entries from here on, at 2 and 5 waves per SIMD
  const bool only_new = argc > 1 && !strcmp(argv[1], "new");
  // warm the clocks up: two seconds of back-to-back launches
  {
    double *out; Stamp *st;
    CHK(hipMalloc(&out, sizeof(double) * cus * 8 * 256));
    CHK(hipMalloc(&st, sizeof(Stamp) * cus * 8 * 4));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float total = 0;
    while (total < (quick ? 300.f : 2000.f)) {
      CHK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_fma_f32, cus * 4, 256, 0, 0, out, 25000, st, ALL);
      CHK(hipEventRecord(e1));
      CHK(hipDeviceSynchronize());
      float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
      total += ms;
    }
    CHK(hipFree(out)); CHK(hipFree(st));
  }
  printf("device: %s, %d CUs\n", p.name, cus);
  printf("%-28s %6s %8s %6s %6s %9s %10s %10s\n", "instruction", "w/SIMD", "ms", "SIMDs", "res", "clock GHz", "cyc/instr", "cyc(wall)");
  for (int wpe : {1, 2, 4, 5, 7, 8}) {
    if (only_new && wpe != 2 && wpe != 5) continue;  // waves per SIMD: 256 threads = 4 waves = one per SIMD of a CU
    const int blocks = cus * wpe;
    double *out; Stamp *st;
    CHK(hipMalloc(&out, sizeof(double) * blocks * 256));
    CHK(hipMalloc(&st, sizeof(Stamp) * blocks * 4));
    for (const K &k : ks) {
      if (quick && k.mask != ALL) continue;
      if (only_new && (size_t)(&k - ks) < first_new) continue;
      const int iters = k.iters / 8;  // 64 instructions per iteration
      hipEvent_t e0, e1;
      CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
      hipLaunchKernelGGL(k.f, blocks, 256, 0, 0, out, 200, st, k.mask);
      CHK(hipDeviceSynchronize());
      CHK(hipEventRecord(e0));
      hipLaunchKernelGGL(k.f, blocks, 256, 0, 0, out, iters, st, k.mask);
      CHK(hipEventRecord(e1));
      CHK(hipDeviceSynchronize());
      float ms;
      CHK(hipEventElapsedTime(&ms, e0, e1));
      std::vector<Stamp> h((size_t)blocks * 4);
      CHK(hipMemcpy(h.data(), st, sizeof(Stamp) * h.size(), hipMemcpyDeviceToHost));
      // waves by SIMD: XCC, then HW_ID's se_id [15:13], sh_id [12], cu_id [11:8], simd_id [5:4]
      std::map<unsigned, std::vector<const Stamp *>> by_simd;
      for (const Stamp &s : h) by_simd[(s.xcc & 0xfu) << 16 | (s.hw & 0xff30u)].push_back(&s);
      const double n_wave = (double)iters * 64.0;  // instructions of one wave
      std::vector<double> cpi, clk, res;
      for (auto &kv : by_simd) {
        for (const Stamp *s : kv.second) {
          // time-average number of waves on this SIMD while s ran (itself included)
          double overlap = 0;
          for (const Stamp *o : kv.second) {
            const long long a = std::max(s->start, o->start), b = std::min(s->end, o->end);
            if (b > a) overlap += (double)(b - a);
          }
          const double k_avg = s->real > 0 ? overlap / (double)s->real : 1.0;
          res.push_back(k_avg);
          cpi.push_back((double)s->cyc / (n_wave * k_avg));
          if (s->real > 0) clk.push_back((double)s->cyc / (double)s->real * 0.1);  // GHz
        }
      }
      auto median = [](std::vector<double> &v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
      const double med_clk = median(clk);
      // upper bound: every SIMD of the chip busy for the whole launch
      const double wall_cyc = (double)ms * 1e6 * med_clk / (n_wave * wpe);
      printf("%-28s %6d %8.3f %6zu %6.2f %9.3f %10.3f %10.3f\n", k.name, wpe, ms, by_simd.size(), median(res), med_clk, median(cpi), wall_cyc);
      CHK(hipEventDestroy(e0)); CHK(hipEventDestroy(e1));
    }
    CHK(hipFree(out)); CHK(hipFree(st));
  }
  return 0;
}
