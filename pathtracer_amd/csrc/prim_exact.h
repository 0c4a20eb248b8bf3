// The reference's FP64 primitive test on the device, directed rounding of doubles to floats, and
// scalar-load access to wave-uniform tables.
#pragma once
#include "device_types.h"

namespace ptk {

// Read-only tables reached through wave-uniform addresses (the lights, the records of their end
// primitives) are read through pointers in the constant address space: the compiler then fetches
// them with scalar loads (one request per wave through the scalar cache, the values in scalar
// registers) instead of one vector load per lane. Nothing is ever written through such a pointer;
// the host fills these tables before any kernel runs.
#if defined(__HIP_DEVICE_COMPILE__)
#define PT_CONST_AS __attribute__((address_space(4)))
#else
#define PT_CONST_AS  // the host pass only parses the device code
#endif
template <class T>
__device__ __forceinline__ const PT_CONST_AS T *uniform_table(const T *p) {
  return (const PT_CONST_AS T *)p;
}
__device__ __forceinline__ int32_t low_word(double x) { return __double2loint(x); }

// the largest float <= x and the smallest float >= x (x not NaN). Stepping to the neighbouring
// float is an integer step on the bit pattern: away from zero adds one, towards zero subtracts
// one, and from +-0 the neighbour is the smallest subnormal of the wanted sign.
__device__ __forceinline__ float f_step(float f, bool up) {
  const uint32_t b = __float_as_uint(f);
  if ((b << 1) == 0u) return __uint_as_float(up ? 1u : 0x80000001u);
  return __uint_as_float(((b >> 31) != 0u) == up ? b - 1u : b + 1u);
}
__device__ __forceinline__ float f_down(double x) {
  const float f = (float)x;
  return ((double)f > x) ? f_step(f, false) : f;
}
__device__ __forceinline__ float f_up(double x) {
  const float f = (float)x;
  return ((double)f < x) ? f_step(f, true) : f;
}

// The reference's FP64 test of primitive p (Triangle::test, src/scene/triangle.cpp:25-44;
// Sphere::test / Sphere::intersect, src/scene/sphere.cpp:11-72) in its operation order,
// WITHOUT the max_t clause: false when the reference rejects the primitive for any other
// reason; else t (u, v for a triangle) are the values it computes. For a sphere t is the
// first root that is not below min_t: the reference takes the first root inside
// [min_t, max_t], and when the first root at or beyond min_t lies beyond max_t so does the
// second, so comparing this t with max_t decides as the reference does.
// P: the primitive's FP64 record (PRIM_STRIDE doubles), through either kind of pointer
template <class Rec>
__device__ __forceinline__ bool exact_test_rec(Rec P, const v3 &o, const v3 &d, double min_t, double &t, double &u, double &v) {
  const int32_t info = low_word(P[9]);
  t = 0; u = 0; v = 0;
  if (!(info & 1)) {
    const v3 p1 = V(P[0], P[1], P[2]), e1 = V(P[3], P[4], P[5]), e2 = V(P[6], P[7], P[8]);
    const v3 normal = vcross(e1, e2);
    if (vdot(normal, d) == 0) return false;
    const v3 sv = vsub(o, p1), s1 = vcross(d, e2), s2 = vcross(sv, e1);
    const double rse1 = 1 / (vdot(s1, e1));
    t = vdot(s2, e2) * rse1;
    u = vdot(s1, sv) * rse1;
    v = vdot(s2, d) * rse1;
    return !(t < min_t || u < 0 || v < 0 || u + v > 1);
  }
  const v3 c = V(P[0], P[1], P[2]);
  const double r2 = P[3];
  const v3 ominc = vsub(o, c);
  const double a = vdot(d, d), b = 2 * vdot(ominc, d), cc = vdot(ominc, ominc) - r2;
  const double discr = b * b - 4 * a * cc;
  if (discr < 0) return false;
  const double r2a = 1 / (2 * a);
  const double b2a = -b * r2a, sq = sqrt(discr) * r2a;
  const double t1 = b2a - sq, t2 = b2a + sq;
  if (t1 >= min_t) { t = t1; return true; }
  if (t2 >= min_t) { t = t2; return true; }
  return false;
}
__device__ __forceinline__ bool exact_test(const DevScene &S, int p, const v3 &o, const v3 &d, double min_t, double &t,
                                           double &u, double &v) {
  return exact_test_rec(S.prim + (size_t)p * PRIM_STRIDE, o, d, min_t, t, u, v);
}

}  // namespace ptk
