// Per-pixel luminance and the adaptive-sampling test of raytrace_pixel, shared by the kernel that
// folds finished paths into the pixel sums and by the kernels that filter the active-pixel list.
#pragma once
#include "device_types.h"

namespace ptk {

// Vector3D::illum (CGL/include/CGL/vector3D.h:231-233): double arithmetic, float result
__device__ __forceinline__ float illum(v3 v) { return (float)(0.2126f * v.x + 0.7152f * v.y + 0.0722f * v.z); }

// The adaptive test of pathtracer.cpp:250-258 at i = n_done.
// A pixel of a round's list goes on to the next round if it used every sample
// traced for it so far (it did not stop at a test inside an earlier round) and the
// test at i = n_done says so.
__device__ __forceinline__ bool pixel_continues(const double *st, int32_t i, double tol);
// With n_done < 0 every pixel is judged at its own count, as the loop of
// raytrace_pixel would: done at ns_aa samples, tested at every positive multiple
// of the batch size.
__device__ __forceinline__ bool pixel_survives(const double *st, int32_t n_done, double tol, int32_t batch, int32_t ns_aa) {
  if (n_done >= 0) return st[5] >= (double)n_done && pixel_continues(st, n_done, tol);
  const int32_t n = (int32_t)st[5];
  if (n >= ns_aa) return false;
  return n == 0 || n % batch != 0 || pixel_continues(st, n, tol);
}
__device__ __forceinline__ bool pixel_continues(const double *st, int32_t i, double tol) {
  const double s1 = st[0], s2 = st[1];
  const double one_over_i = 1.0 / i, one_over_im1 = 1.0 / (i - 1);
  const double mean = s1 * one_over_i;
  const double var = one_over_im1 * (s2 - s1 * s1 * one_over_i);
  const double I = 1.96 * sqrt(var * one_over_i);
  return !(I <= tol * mean);
}

}  // namespace ptk
