// FP32 pre-tests of a ray against a triangle or a sphere, with rigorous error bounds.
//
// The reference decides every ray-primitive question with an FP64 test
// (src/scene/triangle.cpp:25-44, src/scene/sphere.cpp:11-72). The traversal kernels
// are bound by instruction issue and an FP64 test without FMA contraction costs about
// 150 instructions, so the walk asks a cheaper question first: evaluated in FP32, is
// the answer already certain?
//
//   PT32_MISS    the reference's test certainly rejects the primitive, or its hit lies
//                beyond `best_hi`, an upper bound of the closest hit so far;
//   PT32_HIT     it certainly accepts (min_t clause included, max_t clause NOT), and the
//                hit distance t it computes lies in the closed float interval [lo, hi];
//   PT32_UNSURE  anything else: the caller runs the FP64 test.
//
// "Certainly" means for the values the REFERENCE computes, rounding of its FP64
// evaluation included.
//
// Triangle. Inputs: sv = float(o - p1) with the subtraction done in FP64 (one rounding: no
// cancellation error), d, e1, e2 rounded to float.
// With u = 2^-24, every float input within u relative of its FP64 value, FMA forms:
//
//   s1 = d x e2              |s1_i| <= 2 md me2,      error <= 7.1u md me2
//   s2 = sv x e1             |s2_i| <= 2 msv me1,     error <= 7.1u msv me1
//   D = s1.e1, U = s1.sv, V = s2.d, T = s2.e2
//                            error <= 45.3u x (md me1 me2, md me2 msv, md me1 msv, msv me1 me2)
//
// with md, me1, me2, msv the largest absolute components of d, e1, e2, sv. The bounds
// used are 64u = 2^-18 times those products (1.41 times the derived bound), and every
// decision keeps a further factor of two: that pays for the rounding of the bounds
// themselves and for the reference's own FP64 rounding (2^-29 of these bounds). All rays
// of the renderer have unit directions, so md is taken as 1. A floor of 2^-80 keeps the
// bounds meaningful when float products underflow (coordinates below 2^20 in magnitude:
// checked when a scene is uploaded); overflow gives NaN or inf, and every decision is a
// comparison that NaN fails, so such cases come out UNSURE.
//
// The hit distance: t32 = T / D evaluated with a 1-ulp reciprocal;
//   |t - t32| <= 0.75 (E_T + |t32| E_D) / |D| + 2^-22 |t32|        (given |D| >= 16 E_D);
// the interval uses 1.25 and 2^-21.
//
// Sphere (a = d.d, h = sv.d, c = sv.sv - r^2 with sv = float(o - centre), the reference's
// roots being (-h -+ sqrt(h^2 - a c)) / a up to its own rounding): a, h, c and the
// discriminant carry bounds of the same kind; see classify_sphere32.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define PT32_HD __host__ __device__ __forceinline__
#else
#define PT32_HD inline
#endif

namespace ptdev {

enum { PT32_MISS = 0, PT32_HIT = 1, PT32_UNSURE = 2 };

PT32_HD float pt32_max3abs(float a, float b, float c) { return fmaxf(fmaxf(fabsf(a), fabsf(b)), fabsf(c)); }
PT32_HD float pt32_rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rcpf(x);  // 1 ulp
#else
  return 1.0f / x;
#endif
}
PT32_HD float pt32_sqrt(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_sqrtf(x);  // 1 ulp
#else
  return sqrtf(x);
#endif
}

constexpr float PT32_C18 = 3.814697265625e-06f;          // 2^-18
constexpr float PT32_FLOOR = 8.271806125530277e-25f;     // 2^-80
constexpr float PT32_C46 = 1.4210854715202004e-14f;      // 2^-46
constexpr float PT32_MAX_COORD = 1048576.0f;             // 2^20: bound on scene coordinates and ray lengths

// The ray's origin is first moved along the ray to o' = o + t0 d (FP64), t0 being the distance at
// which the ray enters the primitive's bounding box, a float the walk has at hand: u and v do not
// depend on where along the ray the origin lies, t' = t - t0, and |o' - p1| is of the size of the
// primitive instead of the size of the scene, which is what the error of u, v and t scales with.
// sv = float(o' - p1) (FP64 subtraction), df = float(d) with |d| = 1, e1 / e2 = float edges.
// big >= (largest |coordinate| of any ray origin) + |t0|: bounds the FP64 rounding of o' and of the
// reference's own evaluation from the unmoved origin (both ~2^-52 big; charged as 2^-46 big).
// min_lo <= min_t <= min_hi (floats around the ray's FP64 min_t, min_t >= 0); best_hi: see above.
// dmin: for the primitive the ray's origin lies on, the caller knows |T| of the reference's test
// exactly (T depends on the origin only) and passes dmin >= 1.001 |T| / min_t: when |D| is
// certainly above it, |t| = |T / D| < min_t and the reference rejects. +inf for every other primitive.
PT32_HD int classify_tri32(const float sv[3], const float df[3], const float e1[3], const float e2[3], float min_lo,
                           float min_hi, float best_hi, float dmin, float t0, float big, float &lo, float &hi) {
  // s1 = d x e2, s2 = sv x e1
  const float s1x = fmaf(df[1], e2[2], -(df[2] * e2[1]));
  const float s1y = fmaf(df[2], e2[0], -(df[0] * e2[2]));
  const float s1z = fmaf(df[0], e2[1], -(df[1] * e2[0]));
  const float s2x = fmaf(sv[1], e1[2], -(sv[2] * e1[1]));
  const float s2y = fmaf(sv[2], e1[0], -(sv[0] * e1[2]));
  const float s2z = fmaf(sv[0], e1[1], -(sv[1] * e1[0]));
  const float D = fmaf(s1z, e1[2], fmaf(s1y, e1[1], s1x * e1[0]));
  const float U = fmaf(s1z, sv[2], fmaf(s1y, sv[1], s1x * sv[0]));
  const float V = fmaf(s2z, df[2], fmaf(s2y, df[1], s2x * df[0]));
  const float Tn = fmaf(s2z, e2[2], fmaf(s2y, e2[1], s2x * e2[0]));
  const float msv = pt32_max3abs(sv[0], sv[1], sv[2]);
  const float me1 = pt32_max3abs(e1[0], e1[1], e1[2]);
  const float me2 = pt32_max3abs(e2[0], e2[1], e2[2]);
  const float W = fmaf(PT32_C18, msv, PT32_C46 * big);
  const float m12 = me1 * me2;
  const float ED = fmaf(PT32_C18, m12, PT32_FLOOR);
  const float EU = fmaf(W, me2, PT32_FLOOR);
  const float EV = fmaf(W, me1, PT32_FLOOR);
  const float ET = fmaf(W, m12, PT32_FLOOR);
  // everything relative to the sign of D
  const float Ds = fabsf(D);
  const float Us = copysignf(1.0f, D) * U, Vs = copysignf(1.0f, D) * V, Ts = copysignf(1.0f, D) * Tn;
  if (!(Ds > 16.0f * ED) || !(big < PT32_MAX_COORD)) return PT32_UNSURE;
  if (Ds - 2.0f * ED > dmin) return PT32_MISS;
  const float EUV = EU + EV + ED;
  const float r = pt32_rcp(Ds);
  const float t32 = Ts * r;  // t - t0
  const float et = fmaf(1.25f * r, fmaf(fabsf(t32), ED, ET), fmaf(fabsf(t32), 4.76837158203125e-07f /* 2^-21 */, PT32_FLOOR));
  // back to the ray's own parameter; 2^-22 |t0| pays for the roundings of these three additions
  const float ts = t0 + t32, pad = fmaf(fabsf(t0), 2.384185791015625e-07f, et);
  lo = ts - pad;
  hi = ts + pad;
  const float W3 = Ds - Us - Vs;  // D - U - V: negative when u + v > 1
  if (Us < -2.0f * EU || Vs < -2.0f * EV || W3 < -2.0f * EUV || hi < min_lo || lo > best_hi) return PT32_MISS;
  if (Us > 2.0f * EU && Vs > 2.0f * EV && W3 > 2.0f * EUV && lo > min_hi) return PT32_HIT;
  return PT32_UNSURE;
}

// Sphere::test (src/scene/sphere.cpp:11-30): a = d.d, b = 2 (o - c).d, cc = |o - c|^2 - r^2,
// roots t1 <= t2 = -b/2a -+ sqrt(b^2 - 4 a cc)/2a; Sphere::intersect takes t1 if it lies in
// [min_t, max_t], else t2 if it does. sv = float(o' - centre) with the moved origin o' as above
// (the discriminant does not depend on the origin's place along the ray, the roots move by t0),
// r2 = float(r^2). With h = sv.d (so b = 2h), q = h^2 - a cc (a quarter of the discriminant), |d| = 1:
//   |a - 1| <= 2^-20,  E_h = 8u msv,  E_c = 8u (3 msv^2 + r2),  E_q = 2|h| E_h + E_h^2 + 1.01 E_c + 2^-19 |cc| + 4u (h^2 + |cc|)
// The bounds below use twice those, plus the FP64 roundings that scale with `big` (2^-46 big on
// lengths, 2^-45 big^2 on the discriminant). A root's interval comes from interval division by a
// in [1 - 2^-19, 1 + 2^-19] and the square root's bound E_q / sqrt(q) (given q >= 4 E_q).
PT32_HD int classify_sphere32(const float sv[3], const float df[3], float r2, float min_lo, float min_hi, float best_hi,
                              float t0, float big, float &lo, float &hi) {
  const float U8 = 9.5367431640625e-07f;  // 2^-20 = 16u: twice the 8u of the derivation
  const float wabs = PT32_C46 * big;
  const float msv = pt32_max3abs(sv[0], sv[1], sv[2]);
  const float h = fmaf(sv[2], df[2], fmaf(sv[1], df[1], sv[0] * df[0]));
  const float ss = fmaf(sv[2], sv[2], fmaf(sv[1], sv[1], sv[0] * sv[0]));
  const float cc = ss - r2;
  const float Eh = fmaf(U8, msv, wabs + PT32_FLOOR);
  const float Ec = fmaf(U8, fmaf(3.0f * msv, msv, r2), fmaf(8.0f * msv, wabs, PT32_FLOOR));
  const float q = fmaf(h, h, -cc);  // a = 1 to within 2^-20: its deviation is part of Eq
  const float Eq = fmaf(2.0f * fabsf(h) + Eh, Eh, fmaf(2.02f, Ec, fmaf(fabsf(cc), 3.814697265625e-06f /* 2^-18 */,
                                                                       fmaf(U8 * 0.5f, fmaf(h, h, fabsf(cc)), 2.0f * big * wabs + PT32_FLOOR))));
  if (!(big < PT32_MAX_COORD)) return PT32_UNSURE;
  if (q < -Eq) return PT32_MISS;  // discriminant certainly negative
  if (!(q > 4.0f * Eq)) return PT32_UNSURE;
  const float sq = pt32_sqrt(q);
  // |sqrt(q*) - sq| <= Eq / sq (q >= 4 Eq) + 2^-22 sq; roots (-h -+ sq) / a, a within 2^-19 of 1
  const float es = fmaf(Eq, pt32_rcp(sq), sq * 4.76837158203125e-07f) + Eh + fabsf(t0) * 2.384185791015625e-07f;
  const float t1 = -h - sq, t2 = -h + sq;
  const float e1 = fmaf(fabsf(t1), 3.814697265625e-06f, es), e2 = fmaf(fabsf(t2), 3.814697265625e-06f, es);
  const float lo1 = (t0 + t1) - e1, hi1 = (t0 + t1) + e1, lo2 = (t0 + t2) - e2, hi2 = (t0 + t2) + e2;
  if (lo1 > min_hi) {  // the first root is certainly at or beyond min_t: it is the sphere's answer
    lo = lo1; hi = hi1;
    return lo1 > best_hi ? PT32_MISS : PT32_HIT;
  }
  if (hi1 < min_lo) {  // the first root is certainly before min_t: the second decides
    if (hi2 < min_lo) return PT32_MISS;
    if (lo2 > min_hi) {
      lo = lo2; hi = hi2;
      return lo2 > best_hi ? PT32_MISS : PT32_HIT;
    }
  }
  return PT32_UNSURE;
}

}  // namespace ptdev
