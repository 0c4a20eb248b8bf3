// Device-side FP64 helpers. Operation order follows the reference's CGL vector
// library (CGL/include/CGL/vector3D.h:96-262, CGL/src/matrix3x3.cpp:110-114) so
// that, compiled with -ffp-contract=off, every intermediate rounds exactly as it
// does in the reference's CPU renderer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ptdev {

#define PT_PI (3.14159265358979323)   // CGL/include/CGL/misc.h:11
#define PT_EPS_D (0.00000000001)      // CGL/include/CGL/misc.h:12
#define PT_RRT 0.7                    // src/pathtracer/pathtracer.cpp:158

struct v3 {
  double x, y, z;
};
__device__ __forceinline__ v3 V(double x, double y, double z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
__device__ __forceinline__ v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ v3 vscale(v3 a, double c) { return V(a.x * c, a.y * c, a.z * c); }
__device__ __forceinline__ v3 vdivs(v3 a, double c) { double rc = 1.0 / c; return V(rc * a.x, rc * a.y, rc * a.z); }
__device__ __forceinline__ double vdot(v3 u, v3 v) { return u.x * v.x + u.y * v.y + u.z * v.z; }
__device__ __forceinline__ v3 vcross(v3 u, v3 v) {
  return V(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x);
}
__device__ __forceinline__ double vnorm2(v3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
__device__ __forceinline__ double vnorm(v3 a) { return sqrt(vnorm2(a)); }
__device__ __forceinline__ v3 vunit(v3 a) { double r = 1. / vnorm(a); return vscale(a, r); }
struct m3 {
  v3 c[3];
};
__device__ __forceinline__ v3 mmul(const m3 &m, v3 v) {
  return vadd(vadd(vscale(m.c[0], v.x), vscale(m.c[1], v.y)), vscale(m.c[2], v.z));
}
__device__ __forceinline__ m3 mtrans(const m3 &a) {
  m3 b;
  b.c[0] = V(a.c[0].x, a.c[1].x, a.c[2].x);
  b.c[1] = V(a.c[0].y, a.c[1].y, a.c[2].y);
  b.c[2] = V(a.c[0].z, a.c[1].z, a.c[2].z);
  return b;
}
// std::min / std::max operand order (NaN and signed-zero behaviour included)
__device__ __forceinline__ double dmin(double a, double b) { return (b < a) ? b : a; }
__device__ __forceinline__ double dmax(double a, double b) { return (a < b) ? b : a; }

// Philox-4x32-10, counter (c0..c3), key (k0,k1)
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0;
    uint32_t n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// word / (2^32-1), clamped to [1e-7, 0.99999999] (src/util/random_util.h:15-22)
__device__ __forceinline__ double u01(uint32_t w) {
  const double rmax = 1.0 / 4294967295.0;
  double v = (double)w * rmax;
  return dmin(dmax(v, 0.0000001), 0.99999999);
}

}  // namespace ptdev
