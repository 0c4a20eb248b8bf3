// FP32 pre-test of a ray against a triangle, with rigorous error bounds.
//
// The reference decides every ray-triangle question with an FP64 Moeller-Trumbore
// test (src/scene/triangle.cpp:25-44). FP64 runs at half rate on the GPU's vector
// units and the traversal kernels are bound by instruction issue, so the walk asks
// a cheaper question first: evaluated in FP32, is the answer already certain?
//
//   TRI32_MISS    the reference's test certainly rejects the triangle (or its hit
//                 lies beyond `best_hi`, an upper bound of the closest hit so far);
//   TRI32_HIT     it certainly accepts, and the hit distance t it computes lies in
//                 the closed float interval [lo, hi], with min_t < lo;
//   TRI32_UNSURE  anything else: the caller runs the FP64 test.
//
// "Certainly" means for the values the REFERENCE computes, rounding of its FP64
// evaluation included. Derivation (u = 2^-24; every float input within u relative
// of its FP64 value; FMA forms, which round less often than the bound assumes):
//
//   sv = of - p1f            |sv - sv*| <= a,  a = 2u (mo + mp)          (cancellation:
//                            absolute in the magnitudes of o and p1, not of sv)
//   s1 = d x e2              |s1_i| <= 2 md me2,      error <= 8.2u md me2
//   s2 = sv x e1             |s2_i| <= 2 msv me1,     error <= 2a me1 + 8.2u msv me1
//   D = s1.e1                error <= 48.6u md me1 me2
//   U = s1.sv                error <= md me2 (42.6u msv + 6a)
//   V = s2.d,  T = s2.e2     error <= md me1 (48.6u msv + 6a), me1 me2 (48.6u msv + 6a)
//
// with mo, mp, md, me1, me2, msv the largest absolute components of o, p1, d, e1, e2,
// sv. With W = 2^-18 (msv + mo + mp) (2^-18 = 64u) the bounds used are
//   E_D = 2^-18 md me1 me2,  E_U = W md me2,  E_V = W md me1,  E_T = W me1 me2,
// each at least 1.3 times the derived bound, and every decision keeps a further
// factor of two: that pays for the rounding of the bounds themselves and for the
// reference's own FP64 rounding (2^-29 of these bounds). A floor of 2^-100 keeps the
// bounds meaningful when products underflow; overflow gives NaN or inf, and every
// decision below is a comparison that NaN fails, so such cases come out UNSURE.
//
// The hit distance: t32 = T / D evaluated with a 1-ulp reciprocal;
//   |t - t32| <= 1.25 (E_T + |t32| E_D) / |D| + 2^-21 |t32|      (given |D| >= 16 E_D).
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define PT32_HD __host__ __device__ __forceinline__
#else
#define PT32_HD inline
#endif

namespace ptdev {

// 48 bytes, three 16-byte loads. A sphere has me1 < 0 (and is always UNSURE).
struct Tri32 {
  float p1[3], e1[3], e2[3];
  float mp, me1, me2;  // largest |component| of p1, e1, e2 (of the float values)
};

enum { TRI32_MISS = 0, TRI32_HIT = 1, TRI32_UNSURE = 2 };

PT32_HD float pt32_max3abs(float a, float b, float c) { return fmaxf(fmaxf(fabsf(a), fabsf(b)), fabsf(c)); }
PT32_HD float pt32_rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rcpf(x);  // 1 ulp
#else
  return 1.0f / x;
#endif
}

// min_lo <= min_t <= min_hi (floats around the ray's FP64 min_t, min_t >= 0).
PT32_HD int classify_tri32(const float of[3], const float df[3], const Tri32 &T, float min_lo, float min_hi,
                           float best_hi, float &lo, float &hi) {
  const float C18 = 3.814697265625e-06f;  // 2^-18
  const float FLOOR = 7.888609052210118e-31f;  // 2^-100
  const float svx = of[0] - T.p1[0], svy = of[1] - T.p1[1], svz = of[2] - T.p1[2];
  // s1 = d x e2, s2 = sv x e1
  const float s1x = fmaf(df[1], T.e2[2], -(df[2] * T.e2[1]));
  const float s1y = fmaf(df[2], T.e2[0], -(df[0] * T.e2[2]));
  const float s1z = fmaf(df[0], T.e2[1], -(df[1] * T.e2[0]));
  const float s2x = fmaf(svy, T.e1[2], -(svz * T.e1[1]));
  const float s2y = fmaf(svz, T.e1[0], -(svx * T.e1[2]));
  const float s2z = fmaf(svx, T.e1[1], -(svy * T.e1[0]));
  const float D = fmaf(s1z, T.e1[2], fmaf(s1y, T.e1[1], s1x * T.e1[0]));
  const float U = fmaf(s1z, svz, fmaf(s1y, svy, s1x * svx));
  const float V = fmaf(s2z, df[2], fmaf(s2y, df[1], s2x * df[0]));
  const float Tn = fmaf(s2z, T.e2[2], fmaf(s2y, T.e2[1], s2x * T.e2[0]));
  const float md = pt32_max3abs(df[0], df[1], df[2]);
  const float mo = pt32_max3abs(of[0], of[1], of[2]);
  const float msv = pt32_max3abs(svx, svy, svz);
  const float W = C18 * (msv + mo + T.mp);
  const float m12 = T.me1 * T.me2;
  const float ED = fmaf(C18 * md, m12, FLOOR);
  const float EU = fmaf(W * md, T.me2, FLOOR);
  const float EV = fmaf(W * md, T.me1, FLOOR);
  const float ET = fmaf(W, m12, FLOOR);
  // everything relative to the sign of D
  const float Ds = fabsf(D);
  const float Us = copysignf(1.0f, D) * U, Vs = copysignf(1.0f, D) * V, Ts = copysignf(1.0f, D) * Tn;
  if (!(Ds > 16.0f * ED)) return TRI32_UNSURE;
  const float EUV = EU + EV + ED;
  const float r = pt32_rcp(Ds);
  const float t32 = Ts * r;
  const float et = fmaf(1.25f * r, fmaf(fabsf(t32), ED, ET), fmaf(fabsf(t32), 4.76837158203125e-07f /* 2^-21 */, FLOOR));
  lo = t32 - et;
  hi = t32 + et;
  const float W3 = Ds - Us - Vs;  // D - U - V: negative when u + v > 1
  if (Us < -2.0f * EU || Vs < -2.0f * EV || W3 < -2.0f * EUV || hi < min_lo || lo > best_hi) return TRI32_MISS;
  if (Us > 2.0f * EU && Vs > 2.0f * EV && W3 > 2.0f * EUV && lo > min_hi) return TRI32_HIT;
  return TRI32_UNSURE;
}

// Host: the float record of a triangle given p1 and the edges e1 = p2 - p1, e2 = p3 - p1 as
// the reference rounds them in FP64.
inline Tri32 make_tri32(const double p1[3], const double e1[3], const double e2[3]) {
  Tri32 t;
  for (int k = 0; k < 3; k++) { t.p1[k] = (float)p1[k]; t.e1[k] = (float)e1[k]; t.e2[k] = (float)e2[k]; }
  t.mp = pt32_max3abs(t.p1[0], t.p1[1], t.p1[2]);
  t.me1 = pt32_max3abs(t.e1[0], t.e1[1], t.e1[2]);
  t.me2 = pt32_max3abs(t.e2[0], t.e2[1], t.e2[2]);
  return t;
}
inline Tri32 make_sphere32(const double c[3]) {
  Tri32 t;
  for (int k = 0; k < 3; k++) { t.p1[k] = (float)c[k]; t.e1[k] = 0; t.e2[k] = 0; }
  t.mp = pt32_max3abs(t.p1[0], t.p1[1], t.p1[2]);
  t.me1 = -1.0f;
  t.me2 = 0;
  return t;
}

}  // namespace ptdev
