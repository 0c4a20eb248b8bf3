// Device layer of the render hot path for MI355X (gfx950): a wavefront path
// tracer. One frame is rendered in rounds of `samples_per_batch` samples per
// still-active pixel; each round runs, per path depth,
//
//   extend   closest-hit BVH traversal of the ray queue      (BVHAccel::intersect)
//   shade    emission, Russian roulette, BSDF sample, NEE    (at_least_one_bounce_radiance,
//            sample generation; wave64 ballot/prefix appends  estimate_direct_lighting_*)
//   shadow   any-hit traversal of the shadow-ray queue
//   resolve  ordered sum of the unoccluded light samples
//
// and finally `accumulate` folds the finished paths into the per-pixel sums in
// sample order, exactly as PathTracer::raytrace_pixel does
// (src/pathtracer/pathtracer.cpp:234-272 of the reference).
//
// All arithmetic that decides a result is FP64 in the reference's operation
// order and this file is compiled with -ffp-contract=off. Queue sizes live on
// the device; the host never synchronises inside a round.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ptamd.h"
#include "device_math.h"
#include "error_state.h"

using namespace ptdev;

namespace {

// ------------------------------------------------------------------ device data
// Interior BVH node, one 128-byte L2 line: both children's boxes plus their
// codes. A code >= 0 is an interior node index; a negative code is a leaf,
// ~((start << 4) | (count - 1)); LEAF_EMPTY marks "no child".
struct alignas(128) DevNode {
  double lb[6];
  double rb[6];
  int32_t left, right;
  int32_t pad[6];
};
static_assert(sizeof(DevNode) == 128, "node must be one cache line");
constexpr int32_t LEAF_EMPTY = INT32_MIN;

// Primitive record, 96 bytes. Triangle: p1, e1 = p2-p1, e2 = p3-p1,
// normal = e1 x e2 (all rounded exactly as Triangle::test computes them on the
// fly, src/scene/triangle.cpp:27-28). Sphere: centre, r*r.
constexpr int PRIM_STRIDE = 12;

struct DevScene {
  const DevNode *nodes;
  const double *prim;       // [n_prims * PRIM_STRIDE]
  const int32_t *prim_info; // bit 0: sphere, bits 1..: material index
  const double *prim_nrm;   // [n_prims * 9]
  const int32_t *mat_kind;
  const double *mat_param;
  const int32_t *light_kind;
  const double *light_param;
  const double *jitter;
  int32_t n_lights, jitter_n, root_code, stack_cap;
};

struct FrameParams {
  int32_t w, h, max_ray_depth, accum, ns_area_light, hemisphere, nls, bpv;
  uint32_t seed_lo, seed_hi;
  double tan_h, tan_v, n_clip, f_clip;
  v3 cam_pos;
  m3 c2w;
};

struct Ctl {
  uint32_t n_cur, n_next, n_shadow, pad;
  unsigned long long st_rays, st_shadow, st_vertices, st_nodes, st_tests;
  long long filt_count;
};

// Ray queue entry arrays (SoA, capacity = max paths)
struct RayQ {
  int32_t *path;
  double *o[3], *d[3], *T[3];
};
struct HitBuf {
  int32_t *prim;
  double *t, *u, *v;
};
struct ShadowQ {
  int32_t *slot;     // entry * nls + q
  double *d[3];
  double *max_t;     // hemisphere mode: cos term instead
};
struct VertexBuf {
  double *hp[3];      // hit point per queue entry (shadow-ray origin)
  uint8_t *flag;      // bit 0: has light samples, bit 1: result goes to `pending`
  double *contrib[3]; // [cap * nls]
  uint8_t *vis;       // [cap * nls]
};
struct PathBuf {
  double *rad[3];
  double *pend[3];
};

// ------------------------------------------------------------------ intersection
// BBox::intersect (src/scene/bbox.cpp:10-47) against the live max_t.
// The reference's std::min / std::max are written here as fmin / fmax
// (v_min_f64 / v_max_f64, one instruction each instead of compare + two selects).
// They differ only when an operand is NaN, i.e. when a ray runs exactly parallel
// to a slab (a direction component of exactly 0) AND starts exactly on its plane;
// sampled directions never have an exactly zero component.
__device__ __forceinline__ bool slab(const double *bb, v3 o, v3 inv, double min_t, double max_t, double &tmin_out) {
  const double timinx = (bb[0] - o.x) * inv.x, timaxx = (bb[3] - o.x) * inv.x;
  double tmin = fmin(timinx, timaxx), tmax = fmax(timinx, timaxx);
  const double timiny = (bb[1] - o.y) * inv.y, timaxy = (bb[4] - o.y) * inv.y;
  tmin = fmax(tmin, fmin(timiny, timaxy));
  tmax = fmin(tmax, fmax(timiny, timaxy));
  const double timinz = (bb[2] - o.z) * inv.z, timaxz = (bb[5] - o.z) * inv.z;
  tmin = fmax(tmin, fmin(timinz, timaxz));
  tmax = fmin(tmax, fmax(timinz, timaxz));
  tmin_out = tmin;
  return !(tmin > tmax || tmin > max_t || tmax < min_t || tmax < 0);
}

// Per-lane traversal stack: the first STACK_LDS levels live in LDS (entry
// (level, lane) at lds[level * blockDim.x + tid], conflict-free for any mix of
// levels), deeper levels spill to a per-thread strip of global memory.
constexpr int STACK_LDS = 12;
struct TravStack {
  uint2 *lds;
  uint2 *glob;
  uint32_t gstride, gtid;
  __device__ __forceinline__ void put(int level, uint2 e) const {
    if (level < STACK_LDS) lds[level * blockDim.x + threadIdx.x] = e;
    else glob[(size_t)(level - STACK_LDS) * gstride + gtid] = e;
  }
  __device__ __forceinline__ uint2 get(int level) const {
    if (level < STACK_LDS) return lds[level * blockDim.x + threadIdx.x];
    return glob[(size_t)(level - STACK_LDS) * gstride + gtid];
  }
};

struct Hit {
  double t, u, v;
  int32_t prim;
};

// Closest hit (ANY = false) or any hit (ANY = true) of one ray.
//
// The reference walks the tree with an explicit stack in a fixed order and keeps
// the first of equally distant hits (src/scene/bvh.cpp:164-200). Primitives are
// stored here in exactly that visiting order, so "closest t, then lowest index"
// reproduces its answer for ANY traversal order; this one goes near child first
// and drops stacked subtrees that start beyond the current hit.
//
template <bool ANY, bool COUNT>
__device__ __forceinline__ bool traverse(const DevScene &S, v3 o, v3 d, double min_t, double max_t0, Hit &hit,
                                         const TravStack &stack, uint32_t &n_nodes, uint32_t &n_tests) {
  const v3 inv = V(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);  // src/pathtracer/ray.h:41
  double best_t = max_t0;
  hit.prim = -1;
  hit.t = max_t0;
  hit.u = hit.v = 0;
  int sp = 0;
  int32_t cur = S.root_code;
  for (;;) {
    if (cur >= 0) {
      const DevNode *nd = S.nodes + cur;
      if (COUNT) n_nodes += 2;
      double tl, tr;
      const bool hl = nd->left != LEAF_EMPTY && slab(nd->lb, o, inv, min_t, best_t, tl);
      const bool hr = nd->right != LEAF_EMPTY && slab(nd->rb, o, inv, min_t, best_t, tr);
      const int32_t cl = nd->left, cr = nd->right;
      if (hl && hr) {
        const bool left_first = tl <= tr;
        const int32_t far_c = left_first ? cr : cl;
        const double far_t = left_first ? tr : tl;
        // entry distance rounded DOWN to float: may only delay a cull
        float ft = (float)far_t;
        if ((double)ft > far_t) ft = nextafterf(ft, -INFINITY);
        if (!(far_t == far_t)) ft = -INFINITY;
        stack.put(sp, make_uint2((uint32_t)far_c, __float_as_uint(ft)));
        sp++;
        cur = left_first ? cl : cr;
        continue;
      } else if (hl) {
        cur = cl;
        continue;
      } else if (hr) {
        cur = cr;
        continue;
      }
    } else if (cur != LEAF_EMPTY) {
      const uint32_t code = ~(uint32_t)cur;
      const int start = (int)(code >> 4), cnt = (int)(code & 15u) + 1;
      for (int p = start; p < start + cnt; p++) {
        if (COUNT) n_tests++;
        const double *P = S.prim + (size_t)p * PRIM_STRIDE;
        const int32_t info = S.prim_info[p];
        double t, u = 0, v = 0;
        bool ok;
        if (!(info & 1)) {
          // Triangle::test, src/scene/triangle.cpp:25-44
          const v3 p1 = V(P[0], P[1], P[2]), e1 = V(P[3], P[4], P[5]), e2 = V(P[6], P[7], P[8]);
          const v3 normal = V(P[9], P[10], P[11]);
          if (vdot(normal, d) == 0) continue;
          const v3 s = vsub(o, p1), s1 = vcross(d, e2), s2 = vcross(s, e1);
          const double rse1 = 1 / (vdot(s1, e1));
          t = vdot(s2, e2) * rse1;
          u = vdot(s1, s) * rse1;
          v = vdot(s2, d) * rse1;
          ok = !(t < min_t || t > best_t || u < 0 || v < 0 || u + v > 1);
        } else {
          // Sphere::test / Sphere::intersect, src/scene/sphere.cpp:11-72
          const v3 c = V(P[0], P[1], P[2]);
          const double r2 = P[3];
          const v3 ominc = vsub(o, c);
          const double a = vdot(d, d), b = 2 * vdot(ominc, d), cc = vdot(ominc, ominc) - r2;
          const double discr = b * b - 4 * a * cc;
          if (discr < 0) continue;
          const double r2a = 1 / (2 * a);
          const double b2a = -b * r2a, sq = sqrt(discr) * r2a;
          const double t1 = b2a - sq, t2 = b2a + sq;
          if (t1 <= best_t && t1 >= min_t) { t = t1; ok = true; }
          else if (t2 <= best_t && t2 >= min_t) { t = t2; ok = true; }
          else { t = 0; ok = false; }
        }
        if (ok) {
          // ok means min_t <= t <= best_t. Every such hit shortens the ray
          // (r.max_t = t); it is RECORDED when strictly closer than anything
          // before (bvh.cpp:187), or equally close as the recorded hit but
          // earlier in the reference's visiting order (lower index here).
          if (ANY) {
            if (t < max_t0) { hit.prim = p; hit.t = t; return true; }
          } else if (t < best_t || (hit.prim >= 0 && p < hit.prim)) {
            hit.prim = p; hit.t = t; hit.u = u; hit.v = v;
          }
          best_t = t;
        }
      }
    }
    // pop, skipping subtrees that begin beyond the current hit
    for (;;) {
      if (sp == 0) return hit.prim >= 0;
      sp--;
      const uint2 e = stack.get(sp);
      if ((double)__uint_as_float(e.y) > best_t) continue;
      cur = (int32_t)e.x;
      break;
    }
  }
}

// ------------------------------------------------------------------ sampling
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
__device__ __forceinline__ v3 light_sample_L(const DevScene &S, int li, v3 p, double sx, double sy, v3 &wi,
                                             double &dist, double &pdf) {
  const double *q = S.light_param + 16 * li;
  const v3 radiance = V(q[0], q[1], q[2]);
  switch (S.light_kind[li]) {
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

// ------------------------------------------------------------------ kernels
__global__ void k_set_counts(Ctl *ctl, uint32_t n_cur) {
  ctl->n_cur = n_cur;
  ctl->n_next = 0;
  ctl->n_shadow = 0;
}

// end of one depth: fold the stage's counts into the statistics, make the next
// queue current
__global__ void k_advance(Ctl *ctl, int count_vertices) {
  ctl->st_rays += (unsigned long long)ctl->n_cur + ctl->n_shadow;
  ctl->st_shadow += ctl->n_shadow;
  (void)count_vertices;
  ctl->n_cur = ctl->n_next;
  ctl->n_next = 0;
  ctl->n_shadow = 0;
}

// Camera::generate_ray, src/pathtracer/camera.cpp:190-203
__device__ __forceinline__ void camera_ray(const FrameParams &fp, double x, double y, v3 &o, v3 &d) {
  const v3 sensor = V((x - 0.5) * 2 * fp.tan_h, (y - 0.5) * 2 * fp.tan_v, -1);
  const v3 dir = mmul(fp.c2w, sensor);
  o = fp.cam_pos;
  d = vunit(dir);
}

// One thread per path of the round: pixel = pixels[k / ns], sample = base + k % ns.
// raytrace_pixel's ray set-up, src/pathtracer/pathtracer.cpp:259-261, with
// JitteredSampler2D::get_sample (src/pathtracer/sampler.cpp:14-18).
__global__ void k_generate(FrameParams fp, DevScene S, const int32_t *__restrict__ pixels, uint32_t n_paths,
                           int32_t sample_base, int32_t ns, RayQ q, PathBuf pb) {
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n_paths; k += stride) {
    const int32_t pix = pixels[k / (uint32_t)ns];
    const int32_t s = sample_base + (int32_t)(k % (uint32_t)ns);
    const int32_t x = pix % fp.w, y = pix / fp.w;
    uint32_t wd[4];
    philox4x32_10((uint32_t)s, 0u, fp.seed_hi, 0u, fp.seed_lo, (uint32_t)pix, wd);
    const int n = S.jitter_n;
    const size_t ji = (size_t)floor(u01(wd[0]) * n);
    const size_t jj = (size_t)floor(u01(wd[1]) * n);
    const double *js = S.jitter + 2 * (ji * (size_t)n + jj);
    const double sxp = (double)x + js[0], syp = (double)y + js[1];
    v3 o, d;
    camera_ray(fp, sxp / (double)fp.w, syp / (double)fp.h, o, d);
    q.path[k] = (int32_t)k;
    q.o[0][k] = o.x; q.o[1][k] = o.y; q.o[2][k] = o.z;
    q.d[0][k] = d.x; q.d[1][k] = d.y; q.d[2][k] = d.z;
    q.T[0][k] = 1.0; q.T[1][k] = 1.0; q.T[2][k] = 1.0;
    pb.rad[0][k] = 0; pb.rad[1][k] = 0; pb.rad[2][k] = 0;
    if (pb.pend[0]) { pb.pend[0][k] = 0; pb.pend[1][k] = 0; pb.pend[2][k] = 0; }
  }
}

// camera rays through caller-given sensor positions (ptamd_trace_primary)
__global__ void k_generate_primary(FrameParams fp, const double *__restrict__ sx, const double *__restrict__ sy,
                                   uint32_t n, RayQ q) {
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) {
    v3 o, d;
    camera_ray(fp, sx[k], sy[k], o, d);
    q.path[k] = (int32_t)k;
    q.o[0][k] = o.x; q.o[1][k] = o.y; q.o[2][k] = o.z;
    q.d[0][k] = d.x; q.d[1][k] = d.y; q.d[2][k] = d.z;
  }
}

// Ray and vertex records live at FIXED positions (a bounce ray at its parent's
// list position, light sample q of vertex i at slot i*nls + q) and the queues are
// ordered index lists built by ballot / prefix-sum compaction: no atomics, and
// the order is the same on every run.
template <bool COUNT>
__global__ void __launch_bounds__(256)
    k_extend(DevScene S, RayQ q, const int32_t *__restrict__ list, HitBuf hb, Ctl *ctl, double min_t, double max_t,
             uint2 *gstack) {
  extern __shared__ uint2 lds_stack[];
  const TravStack stack{lds_stack, gstack, gridDim.x * blockDim.x, blockIdx.x * blockDim.x + threadIdx.x};
  const uint32_t n = ctl->n_cur;
  const uint32_t stride = gridDim.x * blockDim.x;
  uint32_t cn = 0, ct = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint32_t j = list ? (uint32_t)list[i] : i;
    const v3 o = V(q.o[0][j], q.o[1][j], q.o[2][j]);
    const v3 d = V(q.d[0][j], q.d[1][j], q.d[2][j]);
    Hit h;
    traverse<false, COUNT>(S, o, d, min_t, max_t, h, stack, cn, ct);
    hb.prim[i] = h.prim;
    hb.t[i] = h.t;
    hb.u[i] = h.u;
    hb.v[i] = h.v;
  }
  if (COUNT) {
    atomicAdd(&ctl->st_nodes, (unsigned long long)cn);
    atomicAdd(&ctl->st_tests, (unsigned long long)ct);
  }
}

// Shadow rays: light sampling needs only "is anything in the way"
// (pathtracer.cpp:127); hemisphere sampling needs the closest hit's emission
// (pathtracer.cpp:79-82). vis[slot]: 2 = ray pending, 1 = light sample counts, 0 = not.
template <bool HEMI, bool COUNT>
__global__ void __launch_bounds__(256)
    k_shadow(DevScene S, ShadowQ sq, VertexBuf vb, Ctl *ctl, int32_t nls, uint2 *gstack) {
  extern __shared__ uint2 lds_stack[];
  const TravStack stack{lds_stack, gstack, gridDim.x * blockDim.x, blockIdx.x * blockDim.x + threadIdx.x};
  const uint32_t n = ctl->n_shadow;
  const uint32_t stride = gridDim.x * blockDim.x;
  uint32_t cn = 0, ct = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int32_t slot = sq.slot[i];
    const uint32_t e = (uint32_t)slot / (uint32_t)nls;
    const v3 o = V(vb.hp[0][e], vb.hp[1][e], vb.hp[2][e]);
    const v3 d = V(sq.d[0][slot], sq.d[1][slot], sq.d[2][slot]);
    Hit h;
    if (!HEMI) {
      const bool occluded = traverse<true, COUNT>(S, o, d, PT_EPS_D, sq.max_t[slot], h, stack, cn, ct);
      vb.vis[slot] = occluded ? 0 : 1;
    } else {
      const bool found = traverse<false, COUNT>(S, o, d, PT_EPS_D, (double)INFINITY, h, stack, cn, ct);
      if (found) {
        const int32_t m = S.prim_info[h.prim] >> 1;
        v3 light_L = V(0, 0, 0);
        if (S.mat_kind[m] == PTAMD_MAT_EMISSION) light_L = V(S.mat_param[3 * m], S.mat_param[3 * m + 1], S.mat_param[3 * m + 2]);
        // f * light_L * dot(wi, w2o*n) * 2 * PI; f waits in the contrib slot, the cosine in max_t
        const v3 f = V(vb.contrib[0][slot], vb.contrib[1][slot], vb.contrib[2][slot]);
        const v3 term = vscale(vscale(vscale(vmul(f, light_L), sq.max_t[slot]), 2), PT_PI);
        vb.contrib[0][slot] = term.x; vb.contrib[1][slot] = term.y; vb.contrib[2][slot] = term.z;
      }
      vb.vis[slot] = found ? 1 : 0;
    }
  }
  if (COUNT) {
    atomicAdd(&ctl->st_nodes, (unsigned long long)cn);
    atomicAdd(&ctl->st_tests, (unsigned long long)ct);
  }
}

struct Compact {
  uint8_t *pflag;        // [cap] entry pushes a bounce ray
  uint32_t *cnt_next;    // [cap/256 + 1] bounce rays per 256-entry block, then their exclusive scan
  uint32_t *cnt_shadow;  // [cap/256 + 1] shadow rays per 256-entry block, then their exclusive scan
};

// One thread per entry of the current list at path depth `depth` (1 = the camera
// ray's hit). Restates at_least_one_bounce_radiance (pathtracer.cpp:160-201)
// iteratively: the path carries its throughput T, every shaded vertex adds
// T * direct light.
__global__ void __launch_bounds__(256)
    k_shade(FrameParams fp, DevScene S, RayQ q, const int32_t *__restrict__ list, HitBuf hb, RayQ qn, ShadowQ sq,
            VertexBuf vb, PathBuf pb, Compact cp, Ctl *ctl, int32_t depth, const int32_t *__restrict__ pixels,
            int32_t sample_base, int32_t ns) {
  __shared__ uint32_t wsum[2][4];
  const uint32_t n = ctl->n_cur;
  const uint32_t stride = gridDim.x * blockDim.x;
  const int nls = fp.nls;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (uint32_t base_i = blockIdx.x * blockDim.x; base_i < n; base_i += stride) {
    const uint32_t i = base_i + threadIdx.x;
    const bool valid = i < n;
    bool do_nee = false, push = false;
    v3 hit_p = V(0, 0, 0), f = V(0, 0, 0), T = V(0, 0, 0), next_d = V(0, 0, 0), next_T = V(0, 0, 0);
    v3 n_sh = V(0, 0, 1);
    m3 o2w, w2o;
    o2w.c[0] = o2w.c[1] = o2w.c[2] = V(0, 0, 0);
    w2o = o2w;
    int32_t path = 0;
    uint32_t rng_s = 0, rng_pix = 0, rng_base = 0, n_emit = 0;
    if (valid) {
      const uint32_t j = list ? (uint32_t)list[i] : i;
      const int32_t prim = hb.prim[i];
      path = q.path[j];
      if (prim < 0) {
        // bounce ray left the scene: without accumulation the vertex before it
        // keeps its direct light (pathtracer.cpp:196-198)
        if (!fp.accum && depth > 1) {
          pb.rad[0][path] += pb.pend[0][path];
          pb.rad[1][path] += pb.pend[1][path];
          pb.rad[2][path] += pb.pend[2][path];
        }
        vb.flag[i] = 0;
      } else {
        const int32_t info = S.prim_info[prim];
        const int32_t mat = info >> 1;
        const int32_t kind = S.mat_kind[mat];
        const v3 ro = V(q.o[0][j], q.o[1][j], q.o[2][j]);
        const v3 rd = V(q.d[0][j], q.d[1][j], q.d[2][j]);
        const double t = hb.t[i];
        if (!(info & 1)) {
          // (1 - u - v) * n1 + u * n2 + v * n3, src/scene/triangle.cpp:66
          const double u = hb.u[i], v = hb.v[i];
          const double *N = S.prim_nrm + (size_t)prim * 9;
          n_sh = vadd(vadd(vscale(V(N[0], N[1], N[2]), 1 - u - v), vscale(V(N[3], N[4], N[5]), u)),
                      vscale(V(N[6], N[7], N[8]), v));
        } else {
          // Sphere::normal of r.o + t * r.d, src/scene/sphere.cpp:69, sphere.h:73-75
          const double *P = S.prim + (size_t)prim * PRIM_STRIDE;
          const v3 pp = vadd(ro, vscale(rd, t));
          n_sh = vunit(vsub(pp, V(P[0], P[1], P[2])));
        }
        if (depth == 1) {
          T = V(1, 1, 1);
          // zero_bounce_radiance: emission counts only at the camera ray's hit
          // (pathtracer.cpp:221-226)
          if ((fp.accum || fp.max_ray_depth == 0) && kind == PTAMD_MAT_EMISSION) {
            pb.rad[0][path] += S.mat_param[3 * mat];
            pb.rad[1][path] += S.mat_param[3 * mat + 1];
            pb.rad[2][path] += S.mat_param[3 * mat + 2];
          }
        } else {
          T = V(q.T[0][j], q.T[1][j], q.T[2][j]);
          if (!fp.accum) { pb.pend[0][path] = 0; pb.pend[1][path] = 0; pb.pend[2][path] = 0; }
        }
        uint8_t flag = 0;
        if (fp.max_ray_depth > 0) {
          make_coord_space(o2w, n_sh);
          w2o = mtrans(o2w);
          hit_p = vadd(ro, vscale(rd, t));
          // BSDF::f: only the Lambertian one is non-zero (bsdf.cpp:52-58; the
          // emission, mirror, glass, refraction and microfacet ones return 0)
          if (kind == PTAMD_MAT_DIFFUSE)
            f = vdivs(V(S.mat_param[3 * mat], S.mat_param[3 * mat + 1], S.mat_param[3 * mat + 2]), PT_PI);
          // the path's random numbers: key (seed, pixel), counter (sample, block)
          const uint32_t k = (uint32_t)path;
          rng_pix = (uint32_t)pixels[k / (uint32_t)ns];
          rng_s = (uint32_t)(sample_base + (int32_t)(k % (uint32_t)ns));
          rng_base = 1u + (uint32_t)(depth - 1) * (uint32_t)fp.bpv;
          uint32_t wd[4];
          philox4x32_10(rng_s, rng_base, fp.seed_hi, 0u, fp.seed_lo, rng_pix, wd);
          const bool ignore_RRT = depth == 1 && fp.max_ray_depth > 1;
          const bool not_terminate =
              ignore_RRT || (depth < fp.max_ray_depth && (!fp.accum || (u01(wd[0]) < PT_RRT)));
          // a surface with f = 0 adds nothing and passes nothing on: its light
          // samples and its bounce are skipped (the reference traces them and
          // multiplies by zero)
          const bool lit = kind == PTAMD_MAT_DIFFUSE;
          do_nee = lit && nls > 0;
          if (do_nee) flag = 1 | ((!fp.accum && not_terminate) ? 2 : 0);
          if (not_terminate && lit) {
            double pdf;
            const v3 wi = cosine_sample(u01(wd[1]), u01(wd[2]), pdf);
            next_d = mmul(o2w, wi);
            // f * abs_cos_theta(wi) / pdf / RRT, applied to what the rest of the path returns
            next_T = vmul(T, vdivs(vdivs(vscale(f, fabs(wi.z)), pdf), PT_RRT));
            push = true;
          }
        }
        vb.flag[i] = flag;
        if (flag) { vb.hp[0][i] = hit_p.x; vb.hp[1][i] = hit_p.y; vb.hp[2][i] = hit_p.z; }
      }
      cp.pflag[i] = push ? 1 : 0;
    }
    // ---- light samples into their slots (estimate_direct_lighting_*, pathtracer.cpp:47-135)
    if (do_nee) {
      int qq = 0;
      if (!fp.hemisphere) {
        for (int li = 0; li < S.n_lights; li++) {
          const int nsamp = light_is_delta(S.light_kind[li]) ? 1 : fp.ns_area_light;
          for (int jj = 0; jj < nsamp; jj++, qq++) {
            uint32_t wd[4];
            philox4x32_10(rng_s, rng_base + 1u + (uint32_t)(qq >> 1), fp.seed_hi, 0u, fp.seed_lo, rng_pix, wd);
            const double sx = u01(wd[2 * (qq & 1)]), sy = u01(wd[2 * (qq & 1) + 1]);
            double pdf, dist;
            v3 wi;
            const v3 radiance = light_sample_L(S, li, hit_p, sx, sy, wi, dist, pdf);
            const v3 wi_o = mmul(w2o, wi);
            const size_t slot = (size_t)i * nls + qq;
            if (!(wi_o.z < 0 || (radiance.x == 0 && radiance.y == 0 && radiance.z == 0))) {
              const v3 term = vdivs(vscale(vmul(f, radiance), fabs(wi_o.z)), pdf);
              sq.d[0][slot] = wi.x; sq.d[1][slot] = wi.y; sq.d[2][slot] = wi.z;
              sq.max_t[slot] = dist;
              vb.contrib[0][slot] = term.x; vb.contrib[1][slot] = term.y; vb.contrib[2][slot] = term.z;
              vb.vis[slot] = 2;
              n_emit++;
            } else {
              vb.vis[slot] = 0;
            }
          }
        }
      } else {
        for (qq = 0; qq < nls; qq++) {
          uint32_t wd[4];
          philox4x32_10(rng_s, rng_base + 1u + (uint32_t)(qq >> 1), fp.seed_hi, 0u, fp.seed_lo, rng_pix, wd);
          const v3 wi = uniform_hemisphere_sample(u01(wd[2 * (qq & 1)]), u01(wd[2 * (qq & 1) + 1]));
          const v3 wd3 = mmul(o2w, wi);
          const double cs = vdot(wi, mmul(w2o, n_sh));
          const size_t slot = (size_t)i * nls + qq;
          sq.d[0][slot] = wd3.x; sq.d[1][slot] = wd3.y; sq.d[2][slot] = wd3.z;
          sq.max_t[slot] = cs;
          vb.contrib[0][slot] = f.x; vb.contrib[1][slot] = f.y; vb.contrib[2][slot] = f.z;
          vb.vis[slot] = 2;
          n_emit++;
        }
      }
    } else if (valid && nls > 0) {
      for (int qq = 0; qq < nls; qq++) vb.vis[(size_t)i * nls + qq] = 0;
    }
    // ---- bounce ray at this entry's position in the next ray arrays
    if (push) {
      qn.path[i] = path;
      qn.o[0][i] = hit_p.x; qn.o[1][i] = hit_p.y; qn.o[2][i] = hit_p.z;
      qn.d[0][i] = next_d.x; qn.d[1][i] = next_d.y; qn.d[2][i] = next_d.z;
      qn.T[0][i] = next_T.x; qn.T[1][i] = next_T.y; qn.T[2][i] = next_T.z;
    }
    // ---- per-block counts for the compaction: wave64 ballot / reduction, then LDS
    const unsigned long long pm = __ballot(push);
    uint32_t em = n_emit;
    for (int off = 32; off > 0; off >>= 1) em += __shfl_down(em, off);
    if (lane == 0) { wsum[0][wv] = (uint32_t)__popcll(pm); wsum[1][wv] = em; }
    __syncthreads();
    if (threadIdx.x == 0) {
      const uint32_t vbk = base_i >> 8;
      cp.cnt_next[vbk] = wsum[0][0] + wsum[0][1] + wsum[0][2] + wsum[0][3];
      cp.cnt_shadow[vbk] = wsum[1][0] + wsum[1][1] + wsum[1][2] + wsum[1][3];
    }
    __syncthreads();
  }
}

// Exclusive scan of the two per-block count arrays (one workgroup); the totals
// become the sizes of the next ray list and of the shadow-ray list.
__global__ void __launch_bounds__(1024) k_scan2(Compact cp, Ctl *ctl) {
  __shared__ uint32_t tmp[2][1024];
  __shared__ uint32_t carry[2];
  const uint32_t n_blocks = (ctl->n_cur + 255u) >> 8;
  if (threadIdx.x < 2) carry[threadIdx.x] = 0;
  __syncthreads();
  for (uint32_t base = 0; base < n_blocks; base += 1024) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t a = i < n_blocks ? cp.cnt_next[i] : 0, b = i < n_blocks ? cp.cnt_shadow[i] : 0;
    tmp[0][threadIdx.x] = a;
    tmp[1][threadIdx.x] = b;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
      const uint32_t x = threadIdx.x >= off ? tmp[0][threadIdx.x - off] : 0;
      const uint32_t y = threadIdx.x >= off ? tmp[1][threadIdx.x - off] : 0;
      __syncthreads();
      tmp[0][threadIdx.x] += x;
      tmp[1][threadIdx.x] += y;
      __syncthreads();
    }
    if (i < n_blocks) {
      cp.cnt_next[i] = carry[0] + tmp[0][threadIdx.x] - a;
      cp.cnt_shadow[i] = carry[1] + tmp[1][threadIdx.x] - b;
    }
    __syncthreads();
    if (threadIdx.x == 1023) { carry[0] += tmp[0][1023]; carry[1] += tmp[1][1023]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    ctl->n_next = carry[0];
    ctl->n_shadow = carry[1];
  }
}

// Writes the two ordered index lists from the flags and the scanned block offsets.
__global__ void __launch_bounds__(256)
    k_scatter(Compact cp, VertexBuf vb, int32_t *__restrict__ list_next, ShadowQ sq, Ctl *ctl, int32_t nls) {
  __shared__ uint32_t wcnt[4];
  const uint32_t n = ctl->n_cur;
  const uint32_t n_blocks = (n + 255u) >> 8;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned long long below = (1ull << lane) - 1ull;
  for (uint32_t vbk = blockIdx.x; vbk < n_blocks; vbk += gridDim.x) {
    const uint32_t i = (vbk << 8) + threadIdx.x;
    // bounce rays
    {
      const bool keep = i < n && cp.pflag[i];
      const unsigned long long m = __ballot(keep);
      if (lane == 0) wcnt[wv] = (uint32_t)__popcll(m);
      __syncthreads();
      uint32_t off = cp.cnt_next[vbk];
      for (int k = 0; k < wv; k++) off += wcnt[k];
      if (keep) list_next[off + (uint32_t)__popcll(m & below)] = (int32_t)i;
      __syncthreads();
    }
    // shadow rays, in slot order (vertex-major: the samples of one vertex stay adjacent)
    uint32_t run = cp.cnt_shadow[vbk];
    const size_t slot0 = (size_t)(vbk << 8) * nls, slot_end = (size_t)n * nls;
    for (int c = 0; c < nls; c++) {
      const size_t s = slot0 + (size_t)c * 256 + threadIdx.x;
      const bool keep = s < slot_end && s < slot0 + (size_t)256 * nls && vb.vis[s] == 2;
      const unsigned long long m = __ballot(keep);
      if (lane == 0) wcnt[wv] = (uint32_t)__popcll(m);
      __syncthreads();
      uint32_t off = run;
      for (int k = 0; k < wv; k++) off += wcnt[k];
      if (keep) sq.slot[off + (uint32_t)__popcll(m & below)] = (int32_t)s;
      run += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
      __syncthreads();
    }
  }
}

// L_out = sum of the unoccluded samples in sampling order, then / sample_count
// (pathtracer.cpp:116-134, 85); the path gains T * L_out.
__global__ void k_resolve(FrameParams fp, RayQ q, const int32_t *__restrict__ list, VertexBuf vb, PathBuf pb, Ctl *ctl,
                          int32_t depth) {
  const uint32_t n = ctl->n_cur;
  const uint32_t stride = gridDim.x * blockDim.x;
  const int nls = fp.nls;
  unsigned long long verts = 0;
  for (uint32_t base_i = blockIdx.x * blockDim.x; base_i < n; base_i += stride) {
    const uint32_t i = base_i + threadIdx.x;
    if (i < n) {
      const uint8_t flag = vb.flag[i];
      if (flag) {
        verts++;
        v3 L = V(0, 0, 0);
        const size_t b = (size_t)i * nls;
        for (int qq = 0; qq < nls; qq++) {
          if (vb.vis[b + qq] == 1) L = vadd(L, V(vb.contrib[0][b + qq], vb.contrib[1][b + qq], vb.contrib[2][b + qq]));
        }
        L = vscale(L, 1. / (double)nls);
        const uint32_t j = list ? (uint32_t)list[i] : i;
        const v3 T = depth == 1 ? V(1, 1, 1) : V(q.T[0][j], q.T[1][j], q.T[2][j]);
        const v3 val = vmul(T, L);
        const int32_t path = q.path[j];
        if (flag & 2) {
          pb.pend[0][path] = val.x; pb.pend[1][path] = val.y; pb.pend[2][path] = val.z;
        } else {
          pb.rad[0][path] += val.x; pb.rad[1][path] += val.y; pb.rad[2][path] += val.z;
        }
      }
    }
  }
  // one add per wave
  for (int off = 32; off > 0; off >>= 1) verts += __shfl_down(verts, off);
  if ((threadIdx.x & 63) == 0 && verts) atomicAdd(&ctl->st_vertices, verts);
}

// Vector3D::illum (CGL/include/CGL/vector3D.h:231-233): double arithmetic, float result
__device__ __forceinline__ float illum(v3 v) { return (float)(0.2126f * v.x + 0.7152f * v.y + 0.0722f * v.z); }

// Folds the round's finished paths into the per-pixel sums, in sample order
// (pathtracer.cpp:262-266). state: [pixel*6] = s1, s2, sum.rgb, n.
__global__ void k_accumulate(const int32_t *__restrict__ pixels, uint32_t n_pixels, int32_t ns, PathBuf pb,
                             double *__restrict__ state) {
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < n_pixels; p += stride) {
    double *st = state + (size_t)pixels[p] * 6;
    double s1 = st[0], s2 = st[1];
    v3 sum = V(st[2], st[3], st[4]);
    const size_t b = (size_t)p * (uint32_t)ns;
    for (int s = 0; s < ns; s++) {
      const v3 L = V(pb.rad[0][b + s], pb.rad[1][b + s], pb.rad[2][b + s]);
      const double xl = illum(L);
      s1 += xl;
      s2 += xl * xl;
      sum = vadd(sum, L);
    }
    st[0] = s1; st[1] = s2; st[2] = sum.x; st[3] = sum.y; st[4] = sum.z;
    st[5] += (double)ns;
  }
}

// The adaptive test of pathtracer.cpp:250-258 at i = n_done.
__device__ __forceinline__ bool pixel_continues(const double *st, int32_t i, double tol) {
  const double s1 = st[0], s2 = st[1];
  const double one_over_i = 1.0 / i, one_over_im1 = 1.0 / (i - 1);
  const double mean = s1 * one_over_i;
  const double var = one_over_im1 * (s2 - s1 * s1 * one_over_i);
  const double I = 1.96 * sqrt(var * one_over_i);
  return !(I <= tol * mean);
}

// Ordered stream compaction of the still-active pixels, three small kernels:
// per-block counts, scan of the counts, scatter.
__global__ void k_filter_count(const int32_t *__restrict__ pixels, uint32_t n, int32_t n_done, double tol,
                               const double *__restrict__ state, uint32_t *__restrict__ block_count) {
  __shared__ uint32_t wave_cnt[4];
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool keep = i < n && pixel_continues(state + (size_t)pixels[i] * 6, n_done, tol);
  const unsigned long long m = __ballot(keep);
  if (__lane_id() == 0) wave_cnt[threadIdx.x >> 6] = (uint32_t)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) block_count[blockIdx.x] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}
__global__ void k_filter_scan(uint32_t *block_count, uint32_t n_blocks, long long *total_out) {
  // one block; exclusive scan in place
  __shared__ uint32_t carry;
  __shared__ uint32_t tmp[1024];
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (uint32_t base = 0; base < n_blocks; base += 1024) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < n_blocks ? block_count[i] : 0;
    tmp[threadIdx.x] = v;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
      const uint32_t a = threadIdx.x >= off ? tmp[threadIdx.x - off] : 0;
      __syncthreads();
      tmp[threadIdx.x] += a;
      __syncthreads();
    }
    if (i < n_blocks) block_count[i] = carry + tmp[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += tmp[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) *total_out = (long long)carry;
}
__global__ void k_filter_scatter(const int32_t *__restrict__ pixels, uint32_t n, int32_t n_done, double tol,
                                 const double *__restrict__ state, const uint32_t *__restrict__ block_offset,
                                 int32_t *__restrict__ out) {
  __shared__ uint32_t wave_cnt[4];
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool keep = i < n && pixel_continues(state + (size_t)pixels[i] * 6, n_done, tol);
  const unsigned long long m = __ballot(keep);
  const int lane = __lane_id(), wv = threadIdx.x >> 6;
  if (lane == 0) wave_cnt[wv] = (uint32_t)__popcll(m);
  __syncthreads();
  uint32_t off = block_offset[blockIdx.x];
  for (int k = 0; k < wv; k++) off += wave_cnt[k];
  if (keep) out[off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = pixels[i];
}

// pixel_sum / n and the sample count, pathtracer.cpp:268-271
__global__ void k_frame_resolve(const double *__restrict__ state, uint32_t n_pix, double *__restrict__ hdr,
                                int32_t *__restrict__ count) {
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < n_pix; p += stride) {
    const double *st = state + (size_t)p * 6;
    const double nn = st[5];
    if (nn > 0) {
      const double r = 1. / nn;
      hdr[3 * (size_t)p] = st[2] * r; hdr[3 * (size_t)p + 1] = st[3] * r; hdr[3 * (size_t)p + 2] = st[4] * r;
    } else {
      hdr[3 * (size_t)p] = 0; hdr[3 * (size_t)p + 1] = 0; hdr[3 * (size_t)p + 2] = 0;
    }
    count[p] = (int32_t)nn;
  }
}

__global__ void k_iota(int32_t *out, uint32_t n, int32_t w, int32_t x0, int32_t y0, int32_t rw) {
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = (y0 + (int32_t)(i / (uint32_t)rw)) * w + x0 + (int32_t)(i % (uint32_t)rw);
}

}  // namespace

// ====================================================================== host side
#define HIPCHK(expr)                                                                       \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) {                                                                \
      ptamd::set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                 \
      return PTAMD_ERR_DEVICE;                                                             \
    }                                                                                      \
  } while (0)

struct ptamd_context {
  int device = 0;
  int num_cu = 256;
  hipStream_t own_stream = nullptr;
  // scene
  bool have_scene = false, have_camera = false, have_frame = false;
  DevScene S{};
  std::vector<void *> scene_allocs;
  std::vector<int32_t> dev2ref;  // device primitive index -> caller's primitive index
  std::vector<int32_t> host_light_kinds;
  int32_t n_prims = 0;
  ptamd_camera cam{};
  ptamd_render_params params{};
  FrameParams fp{};
  // work buffers
  int64_t max_paths = 4 << 20;
  int64_t cap_paths = 0;
  int cap_nls = 0;
  bool cap_pend = false;
  void *slab = nullptr;
  RayQ q[2]{};
  int32_t *list[2]{nullptr, nullptr};
  Compact cp{};
  HitBuf hb{};
  ShadowQ sq{};
  VertexBuf vb{};
  PathBuf pb{};
  Ctl *ctl = nullptr;
  uint32_t *block_count = nullptr;
  size_t block_count_cap = 0;
  uint2 *gstack = nullptr;  // overflow levels of the per-lane traversal stacks
  size_t gstack_bytes = 0;
  // profiling
  bool prof_events = false, prof_counters = false;
  struct Ev { int cls; hipEvent_t a, b; };
  std::vector<Ev> events;
  ptamd_stats stats{};
};

namespace {

void free_scene(ptamd_context *c) {
  for (void *p : c->scene_allocs) (void)hipFree(p);
  c->scene_allocs.clear();
  c->have_scene = false;
}

template <typename T>
int upload(ptamd_context *c, const std::vector<T> &h, const T **out) {
  void *d = nullptr;
  size_t bytes = std::max<size_t>(h.size() * sizeof(T), 16);
  HIPCHK(hipMalloc(&d, bytes));
  c->scene_allocs.push_back(d);
  if (!h.empty()) HIPCHK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  *out = (const T *)d;
  return PTAMD_OK;
}

int light_samples_per_vertex(const ptamd_context *c, const ptamd_render_params *p, const std::vector<int32_t> &kinds) {
  (void)c;
  if (p->hemisphere) return (int)kinds.size() * p->ns_area_light;
  int n = 0;
  for (int k : kinds) n += (k == PTAMD_LIGHT_DIRECTIONAL || k == PTAMD_LIGHT_POINT || k == PTAMD_LIGHT_SPOT) ? 1 : p->ns_area_light;
  return n;
}

struct Launcher {
  ptamd_context *c;
  hipStream_t s;
  void begin(int cls) {
    c->stats.launches[cls]++;
    if (c->prof_events) {
      ptamd_context::Ev e;
      e.cls = cls;
      (void)hipEventCreate(&e.a);
      (void)hipEventCreate(&e.b);
      (void)hipEventRecord(e.a, s);
      c->events.push_back(e);
    }
  }
  void end() {
    if (c->prof_events) (void)hipEventRecord(c->events.back().b, s);
  }
};

void collect_events(ptamd_context *c) {
  for (auto &e : c->events) {
    float ms = 0;
    if (hipEventSynchronize(e.b) == hipSuccess && hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) c->stats.kernel_ms[e.cls] += ms;
    (void)hipEventDestroy(e.a);
    (void)hipEventDestroy(e.b);
  }
  c->events.clear();
}

int ensure_buffers(ptamd_context *c, int64_t paths, int nls, bool pend) {
  if (c->slab && c->cap_paths >= paths && c->cap_nls >= nls && (c->cap_pend || !pend)) return PTAMD_OK;
  if (c->slab) {
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipFree(c->slab));
    c->slab = nullptr;
  }
  paths = std::max<int64_t>(paths, c->cap_paths);
  nls = std::max(nls, std::max(c->cap_nls, 1));
  pend = pend || c->cap_pend;
  const size_t P = (size_t)paths, PN = P * (size_t)nls;
  auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
  size_t total = 0;
  auto take = [&](size_t bytes) { size_t o = total; total += al(bytes); return o; };
  size_t off_q[2][10], off_hit[4], off_sq[5], off_vb[8], off_pb[6], off_list[2], off_cp[3];
  for (int k = 0; k < 2; k++) {
    off_q[k][0] = take(P * 4);
    for (int j = 1; j < 10; j++) off_q[k][j] = take(P * 8);
  }
  off_hit[0] = take(P * 4);
  for (int j = 1; j < 4; j++) off_hit[j] = take(P * 8);
  off_sq[0] = take(PN * 4);
  for (int j = 1; j < 5; j++) off_sq[j] = take(PN * 8);
  for (int j = 0; j < 3; j++) off_vb[j] = take(P * 8);
  off_vb[3] = take(P);
  for (int j = 4; j < 7; j++) off_vb[j] = take(PN * 8);
  off_vb[7] = take(PN);
  for (int j = 0; j < 6; j++) off_pb[j] = take(P * 8);
  for (int k = 0; k < 2; k++) off_list[k] = take(P * 4);
  off_cp[0] = take(P);
  off_cp[1] = take((P / 256 + 2) * 4);
  off_cp[2] = take((P / 256 + 2) * 4);
  size_t off_ctl = take(sizeof(Ctl));
  HIPCHK(hipMalloc(&c->slab, total));
  char *b = (char *)c->slab;
  for (int k = 0; k < 2; k++) {
    c->q[k].path = (int32_t *)(b + off_q[k][0]);
    for (int j = 0; j < 3; j++) {
      c->q[k].o[j] = (double *)(b + off_q[k][1 + j]);
      c->q[k].d[j] = (double *)(b + off_q[k][4 + j]);
      c->q[k].T[j] = (double *)(b + off_q[k][7 + j]);
    }
  }
  c->hb.prim = (int32_t *)(b + off_hit[0]);
  c->hb.t = (double *)(b + off_hit[1]);
  c->hb.u = (double *)(b + off_hit[2]);
  c->hb.v = (double *)(b + off_hit[3]);
  c->sq.slot = (int32_t *)(b + off_sq[0]);
  for (int j = 0; j < 3; j++) c->sq.d[j] = (double *)(b + off_sq[1 + j]);
  c->sq.max_t = (double *)(b + off_sq[4]);
  for (int j = 0; j < 3; j++) c->vb.hp[j] = (double *)(b + off_vb[j]);
  c->vb.flag = (uint8_t *)(b + off_vb[3]);
  for (int j = 0; j < 3; j++) c->vb.contrib[j] = (double *)(b + off_vb[4 + j]);
  c->vb.vis = (uint8_t *)(b + off_vb[7]);
  for (int j = 0; j < 3; j++) {
    c->pb.rad[j] = (double *)(b + off_pb[j]);
    c->pb.pend[j] = (double *)(b + off_pb[3 + j]);
  }
  for (int k = 0; k < 2; k++) c->list[k] = (int32_t *)(b + off_list[k]);
  c->cp.pflag = (uint8_t *)(b + off_cp[0]);
  c->cp.cnt_next = (uint32_t *)(b + off_cp[1]);
  c->cp.cnt_shadow = (uint32_t *)(b + off_cp[2]);
  c->ctl = (Ctl *)(b + off_ctl);
  HIPCHK(hipMemset(c->ctl, 0, sizeof(Ctl)));
  c->cap_paths = paths;
  c->cap_nls = nls;
  c->cap_pend = true;
  return PTAMD_OK;
}

int setup_frame_params(ptamd_context *c, const ptamd_render_params *p) {
  if (!c->have_scene || !c->have_camera) {
    ptamd::set_error("scene and camera must be set before rendering");
    return PTAMD_ERR_STATE;
  }
  if (!p || p->w <= 0 || p->h <= 0 || p->ns_aa < 0 || p->samples_per_batch <= 0 || p->ns_area_light < 0 ||
      p->max_ray_depth < 0 || (int64_t)p->w * p->h > (int64_t)1 << 30) {
    ptamd::set_error("invalid render parameters");
    return PTAMD_ERR_ARG;
  }
  FrameParams &fp = c->fp;
  fp.w = p->w; fp.h = p->h;
  fp.max_ray_depth = p->max_ray_depth;
  fp.accum = p->accumulate_bounces ? 1 : 0;
  fp.ns_area_light = p->ns_area_light;
  fp.hemisphere = p->hemisphere ? 1 : 0;
  fp.seed_lo = (uint32_t)p->seed;
  fp.seed_hi = (uint32_t)(p->seed >> 32);
  const ptamd_camera &cam = c->cam;
  // tan(radians(fov) * 0.5) with the host's libm, as the reference evaluates
  // it per ray (camera.cpp:197)
  fp.tan_h = tan(cam.hFov * (PT_PI / 180) * 0.5);
  fp.tan_v = tan(cam.vFov * (PT_PI / 180) * 0.5);
  fp.n_clip = cam.nClip;
  fp.f_clip = cam.fClip;
  fp.cam_pos.x = cam.pos[0]; fp.cam_pos.y = cam.pos[1]; fp.cam_pos.z = cam.pos[2];
  for (int k = 0; k < 3; k++) {
    fp.c2w.c[k].x = cam.c2w[3 * k]; fp.c2w.c[k].y = cam.c2w[3 * k + 1]; fp.c2w.c[k].z = cam.c2w[3 * k + 2];
  }
  return PTAMD_OK;
}

int grid_for(const ptamd_context *c, int per_cu) { return c->num_cu * per_cu; }

}  // namespace

extern "C" {

int ptamd_create(int32_t device_id, ptamd_context **out) {
  if (!out) { ptamd::set_error("null output pointer"); return PTAMD_ERR_ARG; }
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    ptamd::set_error("no HIP device available: the render hot path needs a GPU and has no CPU fallback");
    return PTAMD_ERR_DEVICE;
  }
  if (device_id < 0 || device_id >= n) { ptamd::set_error("device id out of range"); return PTAMD_ERR_ARG; }
  HIPCHK(hipSetDevice(device_id));
  ptamd_context *c = new ptamd_context();
  c->device = device_id;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) c->num_cu = prop.multiProcessorCount;
  if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) c->own_stream = nullptr;
  *out = c;
  return PTAMD_OK;
}

void ptamd_destroy(ptamd_context *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  collect_events(c);
  free_scene(c);
  if (c->slab) (void)hipFree(c->slab);
  if (c->block_count) (void)hipFree(c->block_count);
  if (c->gstack) (void)hipFree(c->gstack);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

int ptamd_set_max_paths(ptamd_context *c, int64_t max_paths) {
  if (!c) { ptamd::set_error("null context"); return PTAMD_ERR_ARG; }
  if (max_paths <= 0) max_paths = 4 << 20;
  if (max_paths > ((int64_t)1 << 27)) max_paths = (int64_t)1 << 27;
  c->max_paths = max_paths;
  return PTAMD_OK;
}

int ptamd_set_profiling(ptamd_context *c, int32_t events, int32_t counters) {
  if (!c) { ptamd::set_error("null context"); return PTAMD_ERR_ARG; }
  c->prof_events = events != 0;
  c->prof_counters = counters != 0;
  return PTAMD_OK;
}

int ptamd_upload_scene(ptamd_context *c, const ptamd_scene_desc *d) {
  if (!c || !d) { ptamd::set_error("null argument"); return PTAMD_ERR_ARG; }
  HIPCHK(hipSetDevice(c->device));
  if (d->n_prims <= 0 || d->n_nodes <= 0 || d->root < 0 || d->root >= d->n_nodes || d->n_materials <= 0 ||
      d->jitter_n <= 0 || !d->prim_type || !d->prim_pos || !d->prim_nrm || !d->prim_material || !d->node_bbox ||
      !d->node_left || !d->node_right || !d->node_start || !d->node_count || !d->mat_kind || !d->mat_param ||
      !d->jitter_table || (d->n_lights > 0 && (!d->light_kind || !d->light_param)) || d->n_lights < 0) {
    ptamd::set_error("scene description is incomplete");
    return PTAMD_ERR_ARG;
  }
  // ---- validate the tree and find the reference's visiting order: depth-first,
  // right child first, leaf primitives in ascending order (bvh.cpp:177-196)
  const int N = d->n_nodes, NP = d->n_prims;
  std::vector<int32_t> visit_order;  // caller's primitive indices in visiting order
  visit_order.reserve(NP);
  std::vector<int32_t> leaf_new_start(N, -1);
  std::vector<char> seen(N, 0);
  std::vector<char> prim_seen(NP, 0);
  int max_depth = 0;
  {
    std::vector<std::pair<int32_t, int>> st;
    st.push_back({d->root, 1});
    while (!st.empty()) {
      auto [n, depth] = st.back();
      st.pop_back();
      if (n < 0 || n >= N || seen[n]) { ptamd::set_error("BVH node graph is not a tree"); return PTAMD_ERR_SCENE; }
      seen[n] = 1;
      max_depth = std::max(max_depth, depth);
      if (d->node_count[n] > 0) {
        const int s = d->node_start[n], cnt = d->node_count[n];
        if (s < 0 || s + cnt > NP) { ptamd::set_error("BVH leaf range outside the primitive array"); return PTAMD_ERR_SCENE; }
        if (cnt > 16) { ptamd::set_error("BVH leaves of more than 16 primitives are not supported"); return PTAMD_ERR_SCENE; }
        leaf_new_start[n] = (int32_t)visit_order.size();
        for (int p = s; p < s + cnt; p++) {
          if (prim_seen[p]) { ptamd::set_error("primitive referenced by two leaves"); return PTAMD_ERR_SCENE; }
          prim_seen[p] = 1;
          visit_order.push_back(p);
        }
      } else {
        st.push_back({d->node_left[n], depth + 1});   // popped second
        st.push_back({d->node_right[n], depth + 1});  // popped first
      }
    }
  }
  if ((int)visit_order.size() != NP) { ptamd::set_error("BVH leaves do not cover every primitive"); return PTAMD_ERR_SCENE; }
  for (int p = 0; p < NP; p++) {
    if (d->prim_material[p] < 0 || d->prim_material[p] >= d->n_materials) { ptamd::set_error("primitive material index out of range"); return PTAMD_ERR_SCENE; }
    if (d->prim_type[p] != PTAMD_PRIM_TRIANGLE && d->prim_type[p] != PTAMD_PRIM_SPHERE) { ptamd::set_error("unknown primitive type"); return PTAMD_ERR_SCENE; }
  }
  // ---- interior nodes get dense indices; children are coded
  std::vector<int32_t> interior_index(N, -1);
  int n_interior = 0;
  for (int n = 0; n < N; n++)
    if (seen[n] && d->node_count[n] <= 0) interior_index[n] = n_interior++;
  auto code_of = [&](int n) -> int32_t {
    if (d->node_count[n] > 0) return ~(int32_t)(((uint32_t)leaf_new_start[n] << 4) | (uint32_t)(d->node_count[n] - 1));
    return interior_index[n];
  };
  std::vector<DevNode> nodes((size_t)std::max(n_interior, 1));
  memset(nodes.data(), 0, nodes.size() * sizeof(DevNode));
  int32_t root_code;
  if (d->node_count[d->root] > 0) {
    // a single-leaf tree: wrap it in one interior node whose other child is empty
    DevNode &nd = nodes[0];
    memcpy(nd.lb, d->node_bbox + 6 * (size_t)d->root, 48);
    nd.left = code_of(d->root);
    nd.right = LEAF_EMPTY;
    root_code = 0;
  } else {
    for (int n = 0; n < N; n++) {
      if (interior_index[n] < 0) continue;
      DevNode &nd = nodes[interior_index[n]];
      const int l = d->node_left[n], r = d->node_right[n];
      memcpy(nd.lb, d->node_bbox + 6 * (size_t)l, 48);
      memcpy(nd.rb, d->node_bbox + 6 * (size_t)r, 48);
      nd.left = code_of(l);
      nd.right = code_of(r);
    }
    root_code = interior_index[d->root];
  }
  // ---- primitive records in visiting order
  std::vector<double> prim((size_t)NP * PRIM_STRIDE, 0.0), nrm((size_t)NP * 9);
  std::vector<int32_t> info(NP);
  for (int i = 0; i < NP; i++) {
    const int p = visit_order[i];
    const double *P = d->prim_pos + 9 * (size_t)p;
    double *R = prim.data() + (size_t)i * PRIM_STRIDE;
    if (d->prim_type[p] == PTAMD_PRIM_TRIANGLE) {
      // e1 = p2 - p1, e2 = p3 - p1, normal = cross(e1, e2): one rounding each,
      // identical to evaluating them per ray; this file is built with
      // -ffp-contract=off so the host compiler cannot fuse them either
      volatile double e1x = P[3] - P[0], e1y = P[4] - P[1], e1z = P[5] - P[2];
      volatile double e2x = P[6] - P[0], e2y = P[7] - P[1], e2z = P[8] - P[2];
      volatile double a0 = e1y * e2z, a1 = e1z * e2y, b0 = e1z * e2x, b1 = e1x * e2z, c0 = e1x * e2y, c1 = e1y * e2x;
      R[0] = P[0]; R[1] = P[1]; R[2] = P[2];
      R[3] = e1x; R[4] = e1y; R[5] = e1z;
      R[6] = e2x; R[7] = e2y; R[8] = e2z;
      R[9] = a0 - a1; R[10] = b0 - b1; R[11] = c0 - c1;
      info[i] = d->prim_material[p] << 1;
    } else {
      R[0] = P[0]; R[1] = P[1]; R[2] = P[2]; R[3] = P[4];
      info[i] = (d->prim_material[p] << 1) | 1;
    }
    memcpy(nrm.data() + (size_t)i * 9, d->prim_nrm + 9 * (size_t)p, 72);
  }
  free_scene(c);
  c->dev2ref = visit_order;
  c->n_prims = NP;
  DevScene S{};
  int rc;
  if ((rc = upload(c, nodes, &S.nodes))) return rc;
  if ((rc = upload(c, prim, &S.prim))) return rc;
  if ((rc = upload(c, info, &S.prim_info))) return rc;
  if ((rc = upload(c, nrm, &S.prim_nrm))) return rc;
  std::vector<int32_t> mk(d->mat_kind, d->mat_kind + d->n_materials);
  std::vector<double> mp(d->mat_param, d->mat_param + 3 * (size_t)d->n_materials);
  std::vector<int32_t> lk(d->light_kind, d->light_kind + d->n_lights);
  std::vector<double> lp(d->light_param, d->light_param + 16 * (size_t)d->n_lights);
  std::vector<double> jt(d->jitter_table, d->jitter_table + 2 * (size_t)d->jitter_n * d->jitter_n);
  if ((rc = upload(c, mk, &S.mat_kind))) return rc;
  if ((rc = upload(c, mp, &S.mat_param))) return rc;
  if ((rc = upload(c, lk, &S.light_kind))) return rc;
  if ((rc = upload(c, lp, &S.light_param))) return rc;
  if ((rc = upload(c, jt, &S.jitter))) return rc;
  S.n_lights = d->n_lights;
  S.jitter_n = d->jitter_n;
  S.root_code = root_code;
  S.stack_cap = max_depth + 1;
  c->S = S;
  c->have_scene = true;
  c->host_light_kinds = lk;
  return PTAMD_OK;
}

int ptamd_set_camera(ptamd_context *c, const ptamd_camera *cam) {
  if (!c || !cam) { ptamd::set_error("null argument"); return PTAMD_ERR_ARG; }
  c->cam = *cam;
  c->have_camera = true;
  return PTAMD_OK;
}


int ptamd_frame_begin(ptamd_context *c, const ptamd_render_params *p) {
  if (!c) { ptamd::set_error("null context"); return PTAMD_ERR_ARG; }
  HIPCHK(hipSetDevice(c->device));
  int rc = setup_frame_params(c, p);
  if (rc) return rc;
  c->params = *p;
  c->fp.nls = light_samples_per_vertex(c, p, c->host_light_kinds);
  c->fp.bpv = 1 + (c->fp.nls + 1) / 2;
  c->have_frame = true;
  if (!c->events.empty()) {
    HIPCHK(hipDeviceSynchronize());
    collect_events(c);  // timings of earlier frames are not part of this one
  }
  memset(&c->stats, 0, sizeof c->stats);
  if (c->ctl) HIPCHK(hipMemset(c->ctl, 0, sizeof(Ctl)));
  return PTAMD_OK;
}

// Launch shape of the traversal kernels: 256-thread workgroups, STACK_LDS stack
// levels per lane in LDS, the rest (if the tree is deeper) in a global strip.
struct TravLaunch {
  int block, grid;
  size_t lds;
  uint2 *gstack;
};
static int traversal_launch(ptamd_context *c, TravLaunch *t) {
  t->block = 256;
  t->lds = (size_t)t->block * std::min(c->S.stack_cap, (int32_t)STACK_LDS) * sizeof(uint2);
  // resident workgroups per CU: bounded by LDS (160 KiB per CU) and by registers
  // (about 5 waves per SIMD); twice that many in the grid so that the tail of a
  // launch still has work to balance
  const int wg_per_cu = std::max(1, std::min<int>((int)((160 * 1024) / std::max<size_t>(t->lds, 1)), 5));
  t->grid = c->num_cu * wg_per_cu * 2;
  const size_t levels = (size_t)std::max(0, c->S.stack_cap - STACK_LDS);
  const size_t need = levels * (size_t)t->grid * t->block * sizeof(uint2);
  if (need > c->gstack_bytes) {
    if (c->gstack) { HIPCHK(hipDeviceSynchronize()); HIPCHK(hipFree(c->gstack)); c->gstack = nullptr; }
    HIPCHK(hipMalloc((void **)&c->gstack, need));
    c->gstack_bytes = need;
  }
  t->gstack = c->gstack;
  return PTAMD_OK;
}

int ptamd_trace_pixels(ptamd_context *c, const int32_t *d_pixels, int64_t n, int32_t sample_base, int32_t n_samples,
                       double *d_state, void *stream) {
  if (!c || !c->have_frame) { ptamd::set_error("ptamd_frame_begin must be called first"); return PTAMD_ERR_STATE; }
  if (n < 0 || n_samples <= 0 || sample_base < 0 || (n > 0 && (!d_pixels || !d_state))) {
    ptamd::set_error("invalid arguments to ptamd_trace_pixels");
    return PTAMD_ERR_ARG;
  }
  if (n == 0) return PTAMD_OK;
  HIPCHK(hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)stream;
  const FrameParams &fp = c->fp;
  const int nls = std::max(fp.nls, 1);
  int64_t px_per_chunk = std::max<int64_t>(1, c->max_paths / n_samples);
  px_per_chunk = std::min(px_per_chunk, n);
  int rc = ensure_buffers(c, px_per_chunk * n_samples, nls, !fp.accum);
  if (rc) return rc;
  TravLaunch tl;
  rc = traversal_launch(c, &tl);
  if (rc) return rc;
  const int tblock = tl.block, tgrid = tl.grid;
  const size_t lds = tl.lds;
  uint2 *gstack = tl.gstack;
  const int ggrid = grid_for(c, 8);
  PathBuf pb = c->pb;
  if (fp.accum) pb.pend[0] = pb.pend[1] = pb.pend[2] = nullptr;
  Launcher L{c, s};
  const int depths = std::max(1, fp.max_ray_depth);
  for (int64_t p0 = 0; p0 < n; p0 += px_per_chunk) {
    const int64_t np = std::min(px_per_chunk, n - p0);
    const uint32_t n_paths = (uint32_t)(np * n_samples);
    const int32_t *pix = d_pixels + p0;
    L.begin(PTAMD_K_OTHER);
    hipLaunchKernelGGL(k_set_counts, 1, 1, 0, s, c->ctl, n_paths);
    L.end();
    L.begin(PTAMD_K_GENERATE);
    hipLaunchKernelGGL(k_generate, ggrid, 256, 0, s, fp, c->S, pix, n_paths, sample_base, n_samples, c->q[0], pb);
    L.end();
    int cur = 0;
    for (int depth = 1; depth <= depths; depth++) {
      const double min_t = depth == 1 ? fp.n_clip : PT_EPS_D;
      const double max_t = depth == 1 ? fp.f_clip : (double)INFINITY;
      const int32_t *lst = depth == 1 ? nullptr : c->list[cur];
      L.begin(PTAMD_K_EXTEND);
      if (c->prof_counters) hipLaunchKernelGGL(k_extend<true>, tgrid, tblock, lds, s, c->S, c->q[cur], lst, c->hb, c->ctl, min_t, max_t, gstack);
      else hipLaunchKernelGGL(k_extend<false>, tgrid, tblock, lds, s, c->S, c->q[cur], lst, c->hb, c->ctl, min_t, max_t, gstack);
      L.end();
      L.begin(PTAMD_K_SHADE);
      hipLaunchKernelGGL(k_shade, ggrid, 256, 0, s, fp, c->S, c->q[cur], lst, c->hb, c->q[cur ^ 1], c->sq, c->vb, pb, c->cp,
                         c->ctl, depth, pix, sample_base, n_samples);
      L.end();
      L.begin(PTAMD_K_OTHER);
      hipLaunchKernelGGL(k_scan2, 1, 1024, 0, s, c->cp, c->ctl);
      hipLaunchKernelGGL(k_scatter, ggrid, 256, 0, s, c->cp, c->vb, c->list[cur ^ 1], c->sq, c->ctl, fp.max_ray_depth > 0 ? fp.nls : 0);
      L.end();
      if (fp.max_ray_depth > 0 && fp.nls > 0) {
        L.begin(PTAMD_K_SHADOW);
        if (fp.hemisphere) {
          if (c->prof_counters) hipLaunchKernelGGL((k_shadow<true, true>), tgrid, tblock, lds, s, c->S, c->sq, c->vb, c->ctl, nls, gstack);
          else hipLaunchKernelGGL((k_shadow<true, false>), tgrid, tblock, lds, s, c->S, c->sq, c->vb, c->ctl, nls, gstack);
        } else {
          if (c->prof_counters) hipLaunchKernelGGL((k_shadow<false, true>), tgrid, tblock, lds, s, c->S, c->sq, c->vb, c->ctl, nls, gstack);
          else hipLaunchKernelGGL((k_shadow<false, false>), tgrid, tblock, lds, s, c->S, c->sq, c->vb, c->ctl, nls, gstack);
        }
        L.end();
        L.begin(PTAMD_K_RESOLVE);
        hipLaunchKernelGGL(k_resolve, ggrid, 256, 0, s, fp, c->q[cur], lst, c->vb, pb, c->ctl, depth);
        L.end();
      }
      L.begin(PTAMD_K_OTHER);
      hipLaunchKernelGGL(k_advance, 1, 1, 0, s, c->ctl, 1);
      L.end();
      cur ^= 1;
    }
    L.begin(PTAMD_K_ACCUMULATE);
    hipLaunchKernelGGL(k_accumulate, ggrid, 256, 0, s, pix, (uint32_t)np, n_samples, pb, d_state);
    L.end();
  }
  HIPCHK(hipGetLastError());
  c->stats.samples += (uint64_t)n * (uint64_t)n_samples;
  return PTAMD_OK;
}

int ptamd_adaptive_filter(ptamd_context *c, const int32_t *d_in, int64_t n, int32_t n_done, const double *d_state,
                          int32_t *d_out, int64_t *d_n_out, void *stream) {
  if (!c || !c->have_frame) { ptamd::set_error("ptamd_frame_begin must be called first"); return PTAMD_ERR_STATE; }
  if (n < 0 || n_done < 2 || !d_n_out || (n > 0 && (!d_in || !d_out || !d_state))) {
    ptamd::set_error("invalid arguments to ptamd_adaptive_filter");
    return PTAMD_ERR_ARG;
  }
  HIPCHK(hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)stream;
  const uint32_t nb = (uint32_t)((n + 255) / 256);
  if (nb > c->block_count_cap) {
    if (c->block_count) { HIPCHK(hipDeviceSynchronize()); HIPCHK(hipFree(c->block_count)); }
    c->block_count_cap = std::max<size_t>(nb, 4096);
    HIPCHK(hipMalloc((void **)&c->block_count, c->block_count_cap * sizeof(uint32_t)));
  }
  Launcher L{c, s};
  L.begin(PTAMD_K_FILTER);
  if (nb) hipLaunchKernelGGL(k_filter_count, nb, 256, 0, s, d_in, (uint32_t)n, n_done, c->params.max_tolerance, d_state, c->block_count);
  hipLaunchKernelGGL(k_filter_scan, 1, 1024, 0, s, c->block_count, nb, (long long *)d_n_out);
  if (nb) hipLaunchKernelGGL(k_filter_scatter, nb, 256, 0, s, d_in, (uint32_t)n, n_done, c->params.max_tolerance, d_state, c->block_count, d_out);
  L.end();
  HIPCHK(hipGetLastError());
  return PTAMD_OK;
}

int ptamd_frame_resolve(ptamd_context *c, const double *d_state, double *d_hdr, int32_t *d_count, void *stream) {
  if (!c || !c->have_frame) { ptamd::set_error("ptamd_frame_begin must be called first"); return PTAMD_ERR_STATE; }
  if (!d_state || !d_hdr || !d_count) { ptamd::set_error("null buffer"); return PTAMD_ERR_ARG; }
  HIPCHK(hipSetDevice(c->device));
  Launcher L{c, (hipStream_t)stream};
  L.begin(PTAMD_K_OTHER);
  hipLaunchKernelGGL(k_frame_resolve, grid_for(c, 4), 256, 0, (hipStream_t)stream, d_state,
                     (uint32_t)(c->params.w * c->params.h), d_hdr, d_count);
  L.end();
  HIPCHK(hipGetLastError());
  return PTAMD_OK;
}

int ptamd_get_stats(ptamd_context *c, ptamd_stats *out) {
  if (!c || !out) { ptamd::set_error("null argument"); return PTAMD_ERR_ARG; }
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipDeviceSynchronize());
  collect_events(c);
  if (c->ctl) {
    Ctl h;
    HIPCHK(hipMemcpy(&h, c->ctl, sizeof h, hipMemcpyDeviceToHost));
    c->stats.rays = h.st_rays;
    c->stats.shadow_rays = h.st_shadow;
    c->stats.path_vertices = h.st_vertices;
    c->stats.nodes_visited = h.st_nodes;
    c->stats.prim_tests = h.st_tests;
  }
  *out = c->stats;
  return PTAMD_OK;
}

int ptamd_render(ptamd_context *c, const ptamd_render_params *p, double *hdr_out, int32_t *count_out) {
  if (!c || !p || !hdr_out || !count_out) { ptamd::set_error("null argument"); return PTAMD_ERR_ARG; }
  int rc = ptamd_frame_begin(c, p);
  if (rc) return rc;
  const int x0 = std::max(0, p->x0), y0 = std::max(0, p->y0), x1 = std::min(p->w, p->x1), y1 = std::min(p->h, p->y1);
  const int64_t npix = (int64_t)p->w * p->h;
  const int64_t nrect = (x1 > x0 && y1 > y0) ? (int64_t)(x1 - x0) * (y1 - y0) : 0;
  double *d_state = nullptr, *d_hdr = nullptr;
  int32_t *d_list[2] = {nullptr, nullptr}, *d_count = nullptr;
  long long *d_n = nullptr;
  auto cleanup = [&]() {
    (void)hipFree(d_state); (void)hipFree(d_hdr); (void)hipFree(d_list[0]); (void)hipFree(d_list[1]);
    (void)hipFree(d_count); (void)hipFree(d_n);
  };
#define HIPCHK_CLEAN(expr)                                                              \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess) {                                                             \
      ptamd::set_error(std::string(#expr) + ": " + hipGetErrorString(e_));              \
      cleanup();                                                                        \
      return PTAMD_ERR_DEVICE;                                                          \
    }                                                                                   \
  } while (0)
  HIPCHK_CLEAN(hipMalloc((void **)&d_state, npix * 6 * sizeof(double)));
  HIPCHK_CLEAN(hipMalloc((void **)&d_hdr, npix * 3 * sizeof(double)));
  HIPCHK_CLEAN(hipMalloc((void **)&d_count, npix * sizeof(int32_t)));
  HIPCHK_CLEAN(hipMalloc((void **)&d_list[0], std::max<int64_t>(nrect, 1) * sizeof(int32_t)));
  HIPCHK_CLEAN(hipMalloc((void **)&d_list[1], std::max<int64_t>(nrect, 1) * sizeof(int32_t)));
  HIPCHK_CLEAN(hipMalloc((void **)&d_n, sizeof(long long)));
  hipStream_t s = c->own_stream;
  HIPCHK_CLEAN(hipMemsetAsync(d_state, 0, npix * 6 * sizeof(double), s));
  if (nrect) hipLaunchKernelGGL(k_iota, grid_for(c, 4), 256, 0, s, d_list[0], (uint32_t)nrect, p->w, x0, y0, x1 - x0);
  HIPCHK_CLEAN(hipStreamSynchronize(s));
  auto t0 = std::chrono::steady_clock::now();
  int64_t n = nrect;
  int cur = 0, done = 0;
  while (done < p->ns_aa && n > 0) {
    if (done > 0) {
      rc = ptamd_adaptive_filter(c, d_list[cur], n, done, d_state, d_list[cur ^ 1], (int64_t *)d_n, s);
      if (rc) { cleanup(); return rc; }
      long long hn = 0;
      HIPCHK_CLEAN(hipMemcpyAsync(&hn, d_n, sizeof hn, hipMemcpyDeviceToHost, s));
      HIPCHK_CLEAN(hipStreamSynchronize(s));
      n = hn;
      cur ^= 1;
      if (n == 0) break;
    }
    const int ns = std::min(p->samples_per_batch, p->ns_aa - done);
    rc = ptamd_trace_pixels(c, d_list[cur], n, done, ns, d_state, s);
    if (rc) { cleanup(); return rc; }
    done += ns;
  }
  rc = ptamd_frame_resolve(c, d_state, d_hdr, d_count, s);
  if (rc) { cleanup(); return rc; }
  HIPCHK_CLEAN(hipStreamSynchronize(s));
  c->stats.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  HIPCHK_CLEAN(hipMemcpy(hdr_out, d_hdr, npix * 3 * sizeof(double), hipMemcpyDeviceToHost));
  HIPCHK_CLEAN(hipMemcpy(count_out, d_count, npix * sizeof(int32_t), hipMemcpyDeviceToHost));
  cleanup();
#undef HIPCHK_CLEAN
  return PTAMD_OK;
}

int ptamd_trace_primary(ptamd_context *c, int64_t n, const double *sx, const double *sy, int32_t *prim_out,
                        double *t_out) {
  if (!c || n < 0 || (n > 0 && (!sx || !sy || !prim_out || !t_out))) { ptamd::set_error("invalid arguments"); return PTAMD_ERR_ARG; }
  ptamd_render_params p{};
  p.w = 1; p.h = 1; p.x1 = 1; p.y1 = 1; p.ns_aa = 1; p.samples_per_batch = 1;
  int rc = setup_frame_params(c, &p);
  if (rc) return rc;
  HIPCHK(hipSetDevice(c->device));
  const int64_t chunk = std::min<int64_t>(std::max<int64_t>(n, 1), c->max_paths);
  rc = ensure_buffers(c, chunk, std::max(c->cap_nls, 1), false);
  if (rc) return rc;
  TravLaunch tl;
  rc = traversal_launch(c, &tl);
  if (rc) return rc;
  const int tblock = tl.block, tgrid = tl.grid;
  const size_t lds = tl.lds;
  hipStream_t s = c->own_stream;
  // the queue's throughput arrays are unused here and double as staging for sx, sy
  double *d_sx = c->q[0].T[0], *d_sy = c->q[0].T[1];
  std::vector<int32_t> prim((size_t)chunk);
  for (int64_t b = 0; b < n; b += chunk) {
    const int64_t m = std::min(chunk, n - b);
    HIPCHK(hipMemcpyAsync(d_sx, sx + b, m * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d_sy, sy + b, m * sizeof(double), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_set_counts, 1, 1, 0, s, c->ctl, (uint32_t)m);
    hipLaunchKernelGGL(k_generate_primary, grid_for(c, 4), 256, 0, s, c->fp, d_sx, d_sy, (uint32_t)m, c->q[0]);
    hipLaunchKernelGGL(k_extend<false>, tgrid, tblock, lds, s, c->S, c->q[0], (const int32_t *)nullptr, c->hb, c->ctl, c->fp.n_clip, c->fp.f_clip, tl.gstack);
    HIPCHK(hipMemcpyAsync(prim.data(), c->hb.prim, m * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(t_out + b, c->hb.t, m * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    for (int64_t i = 0; i < m; i++) {
      prim_out[b + i] = prim[i] >= 0 ? c->dev2ref[prim[i]] : -1;
      if (prim[i] < 0) t_out[b + i] = -1.0;
    }
  }
  return PTAMD_OK;
}

}  // extern "C"
