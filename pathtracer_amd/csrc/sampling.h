// Shading frames, samplers and the lights' sample_L, in the reference's operation order. Device code only.
#pragma once
#include "device_types.h"

namespace ptk {

// make_coord_space, src/pathtracer/bsdf.cpp:21-41
__device__ __forceinline__ void make_coord_space(m3 &o2w, v3 n) {
  v3 z = n, h = n;
  if (fabs(h.x) <= fabs(h.y) && fabs(h.x) <= fabs(h.z)) h.x = 1.0;
  else if (fabs(h.y) <= fabs(h.x) && fabs(h.y) <= fabs(h.z)) h.y = 1.0;
  else h.z = 1.0;
  z = vscale(z, 1. / vnorm(z));
  v3 y = vcross(h, z);
  y = vscale(y, 1. / vnorm(y));
  v3 x = vcross(z, y);
  x = vscale(x, 1. / vnorm(x));
  o2w.c[0] = x; o2w.c[1] = y; o2w.c[2] = z;
}
// CosineWeightedHemisphereSampler3D::get_sample, src/pathtracer/sampler.cpp:64-73
__device__ __forceinline__ v3 cosine_sample(double Xi1, double Xi2, double &pdf) {
  double r = sqrt(Xi1);
  double theta = 2. * PT_PI * Xi2;
  pdf = sqrt(1 - Xi1) / PT_PI;
  return V(r * cos(theta), r * sin(theta), sqrt(1 - Xi1));
}
// UniformHemisphereSampler3D::get_sample, src/pathtracer/sampler.cpp:35-49 (float trig)
__device__ __forceinline__ v3 uniform_hemisphere_sample(double Xi1, double Xi2) {
  double theta = acos(Xi1);
  double phi = 2.0 * PT_PI * Xi2;
  double xs = sinf((float)theta) * cosf((float)phi);
  double ys = sinf((float)theta) * sinf((float)phi);
  double zs = cosf((float)theta);
  return V(xs, ys, zs);
}
__device__ __forceinline__ bool light_is_delta(int kind) {  // src/scene/light.h:21,36,52,67,85
  return kind == PTAMD_LIGHT_DIRECTIONAL || kind == PTAMD_LIGHT_POINT || kind == PTAMD_LIGHT_SPOT;
}
// SceneLight::sample_L for every light kind, src/scene/light.cpp:17-93
// kind, q: the light's kind and its 16 parameters (through either kind of pointer)
template <class Rec>
__device__ __forceinline__ v3 light_sample_rec(int kind, Rec q, v3 p, double sx, double sy, v3 &wi, double &dist, double &pdf) {
  const v3 radiance = V(q[0], q[1], q[2]);
  switch (kind) {
    case PTAMD_LIGHT_AREA: {
      const v3 position = V(q[3], q[4], q[5]), direction = V(q[6], q[7], q[8]);
      const v3 dim_x = V(q[9], q[10], q[11]), dim_y = V(q[12], q[13], q[14]);
      const double area = q[15];
      const double ax = sx - 0.5f, ay = sy - 0.5f;
      const v3 dd = vsub(vadd(vadd(position, vscale(dim_x, ax)), vscale(dim_y, ay)), p);
      const double cosTheta = vdot(dd, direction);
      const double sqDist = vnorm2(dd);
      const double dl = sqrt(sqDist);
      wi = vdivs(dd, dl);
      dist = dl;
      pdf = sqDist / (area * fabs(cosTheta));
      return cosTheta < 0 ? radiance : V(0, 0, 0);
    }
    case PTAMD_LIGHT_POINT: {
      const v3 dd = vsub(V(q[3], q[4], q[5]), p);
      wi = vunit(dd);
      dist = vnorm(dd);
      pdf = 1.0;
      return radiance;
    }
    case PTAMD_LIGHT_DIRECTIONAL:
      wi = V(q[6], q[7], q[8]);
      dist = INFINITY;
      pdf = 1.0;
      return radiance;
    case PTAMD_LIGHT_INF_HEMISPHERE: {
      const v3 dir = uniform_hemisphere_sample(sx, sy);
      m3 s2w;
      s2w.c[0] = V(1, 0, 0); s2w.c[1] = V(0, 0, -1); s2w.c[2] = V(0, 1, 0);
      wi = mmul(s2w, dir);
      dist = INFINITY;
      pdf = 1.0 / (2.0 * PT_PI);
      return radiance;
    }
    default:  // SpotLight::sample_L is a stub that returns no radiance
      wi = V(0, 0, 0);
      dist = 0;
      pdf = 0;
      return V(0, 0, 0);
  }
}

}  // namespace ptk
