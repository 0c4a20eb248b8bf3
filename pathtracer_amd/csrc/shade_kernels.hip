// Kernels of a round other than the BVH walks: path generation, shading (k_shade), the light samples
// (k_light), the ordered compaction of the next ray list and the shadow-ray list (k_scan2,
// k_scatter), the sum of the unoccluded light samples (k_resolve) and the per-pixel accumulation
// with the adaptive test (k_accumulate). See ptamd_device.hip for how a round strings them together.
#include "kernels.h"
#include "pixel_test.h"
#include "prim_exact.h"
#include "sampling.h"

namespace ptk {

// Start of a chunk: pixels [p0, p0 + chunk) of the shard, ns paths each.
__global__ void k_set_counts(Ctl *ctl, PixView pv, uint32_t p0, uint32_t chunk, uint32_t ns) {
  const uint32_t have = pv.count();
  const uint32_t np = have > p0 ? (have - p0 < chunk ? have - p0 : chunk) : 0u;
  ctl->n_pix = np;
  ctl->n_cur = np * ns;
  ctl->n_next = 0;
  ctl->n_shadow = 0;
  ctl->grab[0] = ctl->grab[1] = ctl->grab[2] = 0;
}

// Packed path slots (PathMap): the planned sample counts of the chunk's pixels, summed per 256 pixels,
// scanned, and turned into every pixel's first slot; the total is the chunk's number of paths.
__global__ void k_plan_count(PixView pv, uint32_t p0, uint32_t chunk, RoundPlan rp, uint32_t *__restrict__ block_count) {
  __shared__ uint32_t part[4];
  const uint32_t have = pv.count();
  const uint32_t np = have > p0 ? (have - p0 < chunk ? have - p0 : chunk) : 0u;
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t n = 0;
  if (p < np) {
    int32_t first;
    n = (uint32_t)rp.planned(rp.state + (size_t)pv.at(p0 + p) * 6, first);
  }
  for (int off = 32; off > 0; off >>= 1) n += __shfl_down(n, off);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = n;
  __syncthreads();
  if (threadIdx.x == 0) block_count[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}
// one workgroup: exclusive scan of the block sums in place; starts the chunk as k_set_counts does
__global__ void __launch_bounds__(1024) k_plan_scan(uint32_t *block_count, uint32_t n_blocks, Ctl *ctl, PixView pv, uint32_t p0, uint32_t chunk) {
  __shared__ uint32_t wtot[16];
  __shared__ uint32_t carry;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (uint32_t base = 0; base < n_blocks; base += 1024) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < n_blocks ? block_count[i] : 0u;
    uint32_t x = v;
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t u = __shfl_up(x, off);
      if (lane >= off) x += u;
    }
    if (lane == 63) wtot[wv] = x;
    __syncthreads();
    uint32_t pre = carry;
    for (int k = 0; k < wv; k++) pre += wtot[k];
    if (i < n_blocks) block_count[i] = pre + x - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry = pre + x;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const uint32_t have = pv.count();
    ctl->n_pix = have > p0 ? (have - p0 < chunk ? have - p0 : chunk) : 0u;
    ctl->n_cur = carry;
    ctl->n_next = 0;
    ctl->n_shadow = 0;
    ctl->grab[0] = ctl->grab[1] = ctl->grab[2] = 0;
  }
}
__global__ void k_plan_offsets(PixView pv, uint32_t p0, Ctl *ctl, RoundPlan rp, const uint32_t *__restrict__ block_offset,
                               uint32_t *__restrict__ off, uint32_t *__restrict__ slot_pix) {
  __shared__ uint32_t wtot[4];
  const uint32_t np = ctl->n_pix;
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t n = 0;
  if (p < np) {
    int32_t first;
    n = (uint32_t)rp.planned(rp.state + (size_t)pv.at(p0 + p) * 6, first);
  }
  uint32_t x = n;
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t u = __shfl_up(x, o);
    if (lane >= o) x += u;
  }
  if (lane == 63) wtot[wv] = x;
  __syncthreads();
  uint32_t pre = block_offset[blockIdx.x];
  for (int k = 0; k < wv; k++) pre += wtot[k];
  if (p < np) {
    const uint32_t s0 = pre + x - n, s1 = pre + x;  // the pixel's slots
    off[p] = s0;
    for (uint32_t b = (s0 + 255u) >> 8; (b << 8) < s1; b++) slot_pix[b] = p;  // the 256-slot blocks that begin inside them
  }
  if (p + 1 == np) off[np] = pre + x;  // == ctl->n_cur
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
__global__ void k_generate(FrameParams fp, DevScene S, PixView pv, uint32_t p0, Ctl *ctl, RoundPlan rp, PathMap pm, RayQ q, PathBuf pb) {
  const uint32_t stride = gridDim.x * blockDim.x;
  const uint32_t n_paths = ctl->n_cur, np = ctl->n_pix;
  unsigned long long dead = 0;
  for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n_paths; k += stride) {
    uint32_t usk;
    const int32_t pix = pv.at(p0 + pm.owner(k, np, usk));
    int32_t first;
    const int32_t planned = rp.planned(rp.state + (size_t)pix * 6, first);
    const int32_t sk = (int32_t)usk;
    q.path[k] = (int32_t)k;
    pb.rad[0][k] = 0; pb.rad[1][k] = 0; pb.rad[2][k] = 0;
    if (pb.pend[0]) { pb.pend[0][k] = 0; pb.pend[1][k] = 0; pb.pend[2][k] = 0; }
    if (sk >= planned) {
      // a slot this round does not use (only where slots are not packed, see PathMap): a zero direction
      // tells k_extend to skip it
      q.d[0][k] = 0; q.d[1][k] = 0; q.d[2][k] = 0;
      dead++;
      continue;
    }
    const int32_t s = first + sk;
    pb.pix_smp[k] = make_int2(pix, s);
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
    // (every camera ray starts at fp.cam_pos: the kernels that read a camera ray take its origin from there,
    // and the 24 bytes per path are neither written nor read)
    q.d[0][k] = d.x; q.d[1][k] = d.y; q.d[2][k] = d.z;
  }
  block_add(&ctl->st_dead, dead);
}

// camera rays through caller-given sensor positions (ptamd_trace_primary)
__global__ void k_generate_primary(FrameParams fp, const double *__restrict__ sx, const double *__restrict__ sy,
                                   uint32_t n, RayQ q) {
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) {
    v3 o, d;
    camera_ray(fp, sx[k], sy[k], o, d);
    q.path[k] = (int32_t)k;
    q.d[0][k] = d.x; q.d[1][k] = d.y; q.d[2][k] = d.z;
  }
}

// hit distances of traced camera rays (ptamd_trace_primary): the walk names the primitive, the
// reference's test gives its t
__global__ void k_primary_t(DevScene S, RayQ q, const int32_t *__restrict__ prim, uint32_t n, double min_t,
                            double *__restrict__ t_out, v3 cam_o) {
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) {
    double t = -1.0, u, v;
    if (prim[k] >= 0)
      exact_test(S, prim[k], cam_o, V(q.d[0][k], q.d[1][k], q.d[2][k]), min_t, t, u, v);
    t_out[k] = t;
  }
}

// Specular mode (ptamd_render_params.specular): the ideal mirror / refraction /
// glass BSDFs that the reference declares and leaves empty
// (advanced_bsdf.cpp:21-27, 99-106, 124-134, 146-169); the model is stated in
// include/ptamd.h. A bounce ray that leaves such a Dirac-delta BSDF carries
// VIA_DELTA in its path word: an emitter it hits is seen directly.
constexpr int32_t VIA_DELTA = 1 << 30, PATH_MASK = VIA_DELTA - 1;
__device__ __forceinline__ bool is_delta_kind(int32_t k) {
  return k == PTAMD_MAT_MIRROR || k == PTAMD_MAT_REFRACTION || k == PTAMD_MAT_GLASS;
}
__device__ __forceinline__ bool refract_z(v3 wo, double ior, v3 &wi, double &eta) {
  eta = wo.z > 0 ? 1.0 / ior : ior;
  const double c2 = 1.0 - eta * eta * (1.0 - wo.z * wo.z);
  if (c2 < 0) return false;
  const double cz = sqrt(c2);
  wi = V(-eta * wo.x, -eta * wo.y, wo.z > 0 ? -cz : cz);
  return true;
}
// returns f; wi = 0 and f = 0 when the path ends (refraction at total internal reflection)
__device__ __forceinline__ v3 specular_sample_f(int32_t kind, const double *a3, const double *e4, v3 wo, double u, v3 &wi,
                                               double &pdf) {
  const v3 a = V(a3[0], a3[1], a3[2]);
  const double ior = e4[3];
  double eta;
  pdf = 1.0;
  if (kind == PTAMD_MAT_MIRROR) {
    wi = V(-wo.x, -wo.y, wo.z);
    return vdivs(a, fabs(wi.z));
  }
  if (kind == PTAMD_MAT_REFRACTION) {
    if (!refract_z(wo, ior, wi, eta)) { wi = V(0, 0, 0); return V(0, 0, 0); }
    return vdivs(vdivs(a, fabs(wi.z)), eta * eta);
  }
  const v3 refl = V(e4[0], e4[1], e4[2]);
  v3 wt;
  if (!refract_z(wo, ior, wt, eta)) {
    wi = V(-wo.x, -wo.y, wo.z);
    return vdivs(refl, fabs(wi.z));
  }
  double r0 = (1.0 - ior) / (1.0 + ior);
  r0 = r0 * r0;
  const double m = 1.0 - fabs(wo.z);
  const double R = r0 + (1.0 - r0) * (m * m * m * m * m);
  if (u < R) {
    wi = V(-wo.x, -wo.y, wo.z);
    pdf = R;
    return vdivs(vscale(refl, R), fabs(wi.z));
  }
  wi = wt;
  pdf = 1.0 - R;
  return vdivs(vdivs(vscale(a, 1.0 - R), fabs(wi.z)), eta * eta);
}

// Visibility bytes of consecutive slots, stored four at a time when a vertex's run
// of slots is word-aligned (nls a multiple of 4).
struct VisWriter {
  uint8_t *vis;
  bool words;
  uint32_t acc;
  __device__ __forceinline__ void put(size_t slot, uint32_t v) {
    if (!words) { vis[slot] = (uint8_t)v; return; }
    acc |= v << (8 * (slot & 3));
    if ((slot & 3) == 3) {
      *reinterpret_cast<uint32_t *>(vis + (slot & ~(size_t)3)) = acc;
      acc = 0;
    }
  }
};
// One thread per entry of the current list at path depth `depth` (1 = the camera
// ray's hit). Restates at_least_one_bounce_radiance (pathtracer.cpp:160-201)
// iteratively: the path carries its throughput T, every shaded vertex adds
// T * direct light.
// (82 VGPRs. With the light-sample loop inside, one lane running all samples of its vertex, it was
// 131: the samples now run one lane each in k_light.)
template <bool HEMI>
__global__ void __launch_bounds__(256)
    k_shade(FrameParams fp, DevScene S, RayQ q, const int32_t *__restrict__ list, HitBuf hb, RayQ qn, ShadowQ sq,
            VertexBuf vb, PathBuf pb, Compact cp, Ctl *ctl, int32_t depth) {
  __shared__ uint32_t wsum[2][4];
  const uint32_t n = ctl->n_cur;
  const int nls = fp.nls;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  {
    // one workgroup per 256 entries (the grid covers the launch's upper bound; see ptamd_trace_list):
    // without a loop around it the body keeps far fewer kernel arguments live in scalar registers
    const uint32_t base_i = blockIdx.x * blockDim.x;
    if (base_i >= n) return;
    const uint32_t i = base_i + threadIdx.x;
    const bool valid = i < n;
    bool do_nee = false, push = false;
    v3 hit_p = V(0, 0, 0), f = V(0, 0, 0), T = V(0, 0, 0), next_d = V(0, 0, 0), next_T = V(0, 0, 0);
    v3 n_sh = V(0, 0, 1);
    m3 o2w, w2o;
    o2w.c[0] = o2w.c[1] = o2w.c[2] = V(0, 0, 0);
    w2o = o2w;
    int32_t path = 0, via = 0, nee_mat = 0;
    uint32_t rng_s = 0, rng_pix = 0, rng_base = 0, n_emit = 0;
    if (valid) {
      const uint32_t j = list ? (uint32_t)list[i] : i;
      const int32_t prim = hb.prim[i];
      const int32_t path_word = q.path[j];
      path = path_word & PATH_MASK;
      if (prim < 0) {
        // bounce ray left the scene: without accumulation the vertex before it
        // keeps its direct light (pathtracer.cpp:196-198)
        if (prim == -1 && !fp.accum && depth > 1) {
          pb.rad[0][path] += pb.pend[0][path];
          pb.rad[1][path] += pb.pend[1][path];
          pb.rad[2][path] += pb.pend[2][path];
        }
        vb.flag[i] = 0;
      } else {
        // (bit 0: sphere, bits 1..: material; the copy in the primitive's FP64 record, which exact_test reads anyway)
        const int32_t info = low_word(S.prim[(size_t)prim * PRIM_STRIDE + 9]);
        const int32_t mat = info >> 1;
        nee_mat = mat;
        const int32_t kind = S.mat_kind[mat];
        const v3 ro = depth == 1 ? fp.cam_pos : V(q.o[0][j], q.o[1][j], q.o[2][j]);  // (camera rays: see k_generate)
        const v3 rd = V(q.d[0][j], q.d[1][j], q.d[2][j]);
        // the walk only names the primitive: its t, u, v as the reference computes them
        double t, u, v;
        exact_test(S, prim, ro, rd, depth == 1 ? fp.n_clip : PT_EPS_D, t, u, v);
        if (!(info & 1)) {
          // (1 - u - v) * n1 + u * n2 + v * n3, src/scene/triangle.cpp:66
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
        const bool delta = fp.specular && is_delta_kind(kind);
        if (fp.specular && (path_word & VIA_DELTA) && kind == PTAMD_MAT_EMISSION) {
          // seen through a chain of delta bounces: no light sample could have covered it
          const v3 E = V(S.mat_param[3 * mat], S.mat_param[3 * mat + 1], S.mat_param[3 * mat + 2]);
          const v3 val = vmul(T, E);
          pb.rad[0][path] += val.x; pb.rad[1][path] += val.y; pb.rad[2][path] += val.z;
        }
        if (fp.max_ray_depth > 0) {
          make_coord_space(o2w, n_sh);
          w2o = mtrans(o2w);
          hit_p = vadd(ro, vscale(rd, t));
          // BSDF::f: only the Lambertian one is non-zero (bsdf.cpp:52-58; the
          // emission, mirror, glass, refraction and microfacet ones return 0)
          if (kind == PTAMD_MAT_DIFFUSE)
            f = vdivs(V(S.mat_param[3 * mat], S.mat_param[3 * mat + 1], S.mat_param[3 * mat + 2]), PT_PI);
          // the path's random numbers: key (seed, pixel), counter (sample, block)
          const int2 ps = pb.pix_smp[path];
          rng_pix = (uint32_t)ps.x;
          rng_s = (uint32_t)ps.y;
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
          if (not_terminate && (lit || delta)) {
            double pdf;
            v3 wi;
            if (delta) {
              const v3 wo = mmul(w2o, vneg(rd));
              f = specular_sample_f(kind, S.mat_param + 3 * mat, S.mat_extra + 4 * mat, wo, u01(wd[1]), wi, pdf);
              push = !(f.x == 0 && f.y == 0 && f.z == 0);
              via = VIA_DELTA;
            } else {
              wi = cosine_sample(u01(wd[1]), u01(wd[2]), pdf);
              push = true;
            }
            next_d = mmul(o2w, wi);
            // f * abs_cos_theta(wi) / pdf / RRT, applied to what the rest of the path returns
            next_T = vmul(T, vdivs(vdivs(vscale(f, fabs(wi.z)), pdf), PT_RRT));
          }
        }
        vb.flag[i] = flag;
        // the hit point: origin of the vertex's shadow rays and of its bounce ray (vb.hp IS qn.o)
        if (flag || push) {
          vb.hp[0][i] = hit_p.x; vb.hp[1][i] = hit_p.y; vb.hp[2][i] = hit_p.z;
          // Every ray that leaves this vertex meets the vertex's own triangle at t = T / D with
          // T = ((o - p1) x e1) . e2, a number that depends on the origin only and is zero up to
          // FP64 rounding; the reference rejects it through its min_t clause unless the ray
          // grazes the triangle (|D| tiny). T is computed here as Triangle::test computes it
          // (src/scene/triangle.cpp:33-36), and the walks get the |D| above which |T / D| is
          // certainly below min_t (prim32.h, classify_tri32: dmin).
          int32_t self = (int32_t)PRIM_NONE;
          float dmin = INFINITY;
          if (!(info & 1)) {
            const double *P = S.prim + (size_t)prim * PRIM_STRIDE;
            const v3 sv = vsub(hit_p, V(P[0], P[1], P[2]));
            const double Tn = vdot(vcross(sv, V(P[3], P[4], P[5])), V(P[6], P[7], P[8]));
            const float tu = fmaxf(f_up(fabs(Tn)), 7.888609052210118e-31f /* 2^-100 */);
            dmin = (tu * 1.002f) / f_down(PT_EPS_D);
            self = prim;
          }
          vb.self_code[i] = self;
          vb.self_dmin[i] = dmin;
        }
      }
      cp.pflag[i] = push ? 1 : 0;
    }
    // ---- bounce ray at this entry's position in the next ray arrays (first: it frees its registers
    // for the light-sample loop)
    if (push) {
      qn.path[i] = path | via;
      qn.d[0][i] = next_d.x; qn.d[1][i] = next_d.y; qn.d[2][i] = next_d.z;
      qn.T[0][i] = next_T.x; qn.T[1][i] = next_T.y; qn.T[2][i] = next_T.z;
    }
    // ---- light samples (estimate_direct_lighting_*, pathtracer.cpp:47-135). Importance sampling of the
    // lights runs one lane per SAMPLE in the next launch, k_light: this kernel leaves it what it needs of
    // the vertex. Hemisphere sampling (-H) draws its samples here.
    if (!HEMI) {
      if (do_nee) {
        vb.nz[0][i] = o2w.c[2].x; vb.nz[1][i] = o2w.c[2].y; vb.nz[2][i] = o2w.c[2].z;
        vb.nee[i] = make_uint4(rng_s, rng_pix, (uint32_t)nee_mat, 0u);
      }
    } else {
      VisWriter vis{vb.vis, (nls & 3) == 0, 0u};
      if (do_nee) {
        for (int qq = 0; qq < nls; qq++) {
          uint32_t wd[4];
          philox4x32_10(rng_s, rng_base + 1u + (uint32_t)(qq >> 1), fp.seed_hi, 0u, fp.seed_lo, rng_pix, wd);
          const v3 wi = uniform_hemisphere_sample(u01(wd[2 * (qq & 1)]), u01(wd[2 * (qq & 1) + 1]));
          const v3 wd3 = mmul(o2w, wi);
          const double cs = vdot(wi, mmul(w2o, n_sh));
          const size_t slot = (size_t)i * nls + qq;
          sq.ray[slot] = SRay{{wd3.x, wd3.y, wd3.z}, cs};
          vb.contrib[slot] = Contrib{{f.x, f.y, f.z}};
          vis.put(slot, 2);
          n_emit++;
        }
      } else if (valid && nls > 0) {
        for (int qq = 0; qq < nls; qq++) vis.put((size_t)i * nls + qq, 0);
      }
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
      // (importance sampling: zero here, k_light's workgroups add their shadow rays)
      cp.cnt_shadow[vbk] = wsum[1][0] + wsum[1][1] + wsum[1][2] + wsum[1][3];
    }
    __syncthreads();
  }
}

// The light samples of the vertices k_shade has just shaded, one lane per SAMPLE: slot = entry * nls + q
// (estimate_direct_lighting_importance, pathtracer.cpp:93-135; the sample's light and its index within
// that light follow from q as the reference's two loops enumerate them). A vertex's samples sit in
// adjacent lanes, so what they share (hit point, frame axis, f, counter words) is one fetch per
// wave and the 32- and 24-byte records of a wave's 64 slots are contiguous. Per sample: a Philox
// block, sample_L, the cosine against the frame's z axis (w2o * wi needs only that row), the
// reference's test of the primitives at the light's far end, the shadow-ray and contribution
// records, the visibility byte; per workgroup the number of shadow rays, added to the count of its
// 256-entry block for the compaction (an integer sum: the same on every run).
// Split off k_shade, where one lane ran the whole light loop of its vertex at 131 registers.
// (one workgroup per 256 slots, `blk` its number)
__device__ __forceinline__ void light_block(uint32_t blk, const FrameParams &fp, const DevScene &S, const VertexBuf &vb, const ShadowQ &sq,
                                            const Compact &cp, Ctl *ctl, const FastDiv &by_nls, uint32_t rng_base) {
  __shared__ uint32_t wsum[4];
  const uint32_t n = ctl->n_cur;
  const uint32_t nls = (uint32_t)fp.nls;
  const uint32_t n_slots = n * nls;  // < 2^31 (ptamd_trace_list)
  const uint32_t slot0 = blk * 256u;
  if (slot0 >= n_slots) return;
  const uint32_t slot = slot0 + threadIdx.x;
  const bool in = slot < n_slots;
  // Everything a lane needs of its vertex is fetched at once, before the flag says whether it is
  // needed (lanes past the end read the last slot's vertex): the kernel's time is the chain of its
  // memory round trips, not its arithmetic. For the same reason all that depends on the LIGHT only
  // (kind, parameters, end primitives and their records) is read through wave-uniform addresses,
  // i.e. scalar loads: the loop below runs over the lights, each taking the lanes whose sample
  // belongs to it, instead of each lane looking its light up.
  const uint32_t i = by_nls.div(in ? slot : n_slots - 1u);
  const uint32_t q = (in ? slot : n_slots - 1u) - i * nls;
  const uint8_t flag = vb.flag[i];
  const uint4 rk = vb.nee[i];
  const v3 hit_p = V(vb.hp[0][i], vb.hp[1][i], vb.hp[2][i]);
  const v3 z = V(vb.nz[0][i], vb.nz[1][i], vb.nz[2][i]);
  const bool want = in && (flag & 1);
  bool emit = false;
  if (__ballot(want)) {
    double sx = 0, sy = 0;
    if (want) {
      uint32_t wd[4];
      philox4x32_10(rk.x, rng_base + 1u + (q >> 1), fp.seed_hi, 0u, fp.seed_lo, rk.y, wd);
      sx = u01(wd[2 * (q & 1)]);
      sy = u01(wd[2 * (q & 1) + 1]);
    }
    const auto *light_kind = uniform_table(S.light_kind);
    const auto *light_param = uniform_table(S.light_param);
    const auto *light_tag = uniform_table(S.light_tag);
    const auto *light_end = uniform_table(S.light_end);
    const auto *prim_rec = uniform_table(S.prim);
    uint32_t base = 0;
    for (int li = 0; li < S.n_lights; li++) {  // wave-uniform
      const int kind = light_kind[li];
      const uint32_t nsamp = light_is_delta(kind) ? 1u : (uint32_t)fp.ns_area_light;
      const bool mine = want && q >= base && q < base + nsamp;
      base += nsamp;
      if (!__ballot(mine)) continue;
      double pdf = 1, dist = 0, wi_o_z = 0;
      v3 wi = V(0, 0, 0), radiance = V(0, 0, 0);
      bool e = false;
      if (mine) {
        radiance = light_sample_rec(kind, light_param + 16 * li, hit_p, sx, sy, wi, dist, pdf);
        // (w2o * wi).z: the row of w2o that is the frame's z axis, summed as mmul sums it
        wi_o_z = z.x * wi.x + z.y * wi.y + z.z * wi.z;
        e = !(wi_o_z < 0 || (radiance.x == 0 && radiance.y == 0 && radiance.z == 0));
      }
      // The primitives at the far end of this light's shadow rays are tested here, by the reference's
      // test and its rule (bvh.cpp:187: a hit strictly before the ray's end occludes); a ray they
      // occlude is never traced, and the walk of the others skips them.
      // (Measured and rejected: the walk's FP32 pre-test in front of each of these tests, which
      // settles the triangle of the quad the sample point does not lie on: 9 % slower. A host-made
      // table of the sample point's barycentric coordinates in each end triangle, which lets a lane
      // pass over the triangle its ray certainly misses so that the wave runs the test once, each
      // lane on its own triangle: the per-lane record costs more than the second test saves, whether
      // loaded per lane (k_light 28.5 -> 30.9 ms) or picked out of the scalar-loaded records (34.0).
      // A grid-stride loop instead of one workgroup per 256 slots: 30.5 to 31.8 ms. Compiled for 6 or
      // 7 waves per SIMD instead of the 5 its 88 registers allow: 29.2 and 33.7 ms.)
      if (S.has_end && light_tag[li]) {
#pragma unroll 1
        for (int k = 0; k < 4; k++) {
          const int32_t ep = light_end[4 * li + k];  // wave-uniform, and so is its record
          if (ep < 0) break;
          if (e) {
            double te, ue, ve;
            if (exact_test_rec(prim_rec + (size_t)ep * PRIM_STRIDE, hit_p, wi, PT_EPS_D, te, ue, ve) && te < dist) e = false;
          }
        }
      }
      if (e) {
        // DiffuseBSDF::f, bsdf.cpp:52-58 (only Lambertian vertices take light samples)
        const double *rho = S.mat_param + 3 * (size_t)rk.z;
        const v3 f = vdivs(V(rho[0], rho[1], rho[2]), PT_PI);
        const v3 term = vdivs(vscale(vmul(f, radiance), fabs(wi_o_z)), pdf);
        sq.ray[slot] = SRay{{wi.x, wi.y, wi.z}, dist};
        vb.contrib[slot] = Contrib{{term.x, term.y, term.z}};
        emit = true;
      }
    }
  }
  if (in) vb.vis[slot] = emit ? 2 : 0;
  const unsigned long long m = __ballot(emit);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = (uint32_t)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t t = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    // 256 consecutive slots from a multiple of 256 lie inside one block of 256 entries x nls slots
    if (t) atomicAdd(&cp.cnt_shadow[blk / nls], t);
  }
}

__global__ void __launch_bounds__(256)
    k_light(FrameParams fp, DevScene S, VertexBuf vb, ShadowQ sq, Compact cp, Ctl *ctl, FastDiv by_nls, uint32_t rng_base) {
  light_block(blockIdx.x, fp, S, vb, sq, cp, ctl, by_nls, rng_base);
}

// Exclusive scan of the two per-block count arrays (one workgroup); the totals
// become the sizes of the next ray list and of the shadow-ray list. This is also where a depth ends for the
// counters: every depth has a control block of its own (a ring of four, ptamd_trace_list), and the one
// workgroup that knows this depth's totals sets up the next depth's block (its list length, the length of the
// list before it, fresh claim counters) and folds the totals into the statistics, which saves a launch per depth.
__global__ void __launch_bounds__(1024) k_scan2(Compact cp, Ctl *ctl, Ctl *next) {
  // per pass: IT counts per thread summed serially (read and written sixteen bytes at a time), a shuffle scan
  // inside each wave, the 16 wave totals through LDS (three barriers per 1024 x IT counts: the kernel is one
  // workgroup and its time is the number of passes, two for the 65536 blocks of a 16 Mi-path launch)
  __shared__ uint32_t wtot[2][16];
  __shared__ uint32_t carry[2];
  constexpr int IT = 32;
  const uint32_t n_blocks = (ctl->n_cur + 255u) >> 8;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (threadIdx.x < 2) carry[threadIdx.x] = 0;
  __syncthreads();
  for (uint32_t base = 0; base < n_blocks; base += 1024 * IT) {
    const uint32_t i0 = base + threadIdx.x * IT;
    const bool whole = i0 + IT <= n_blocks;  // (the arrays are 256-byte aligned, i0 a multiple of 32)
    uint32_t a[IT], b[IT], sa = 0, sb = 0;
    if (whole) {
#pragma unroll
      for (int k = 0; k < IT; k += 4) {
        const uint4 va = *reinterpret_cast<const uint4 *>(cp.cnt_next + i0 + k), vb4 = *reinterpret_cast<const uint4 *>(cp.cnt_shadow + i0 + k);
        a[k] = va.x; a[k + 1] = va.y; a[k + 2] = va.z; a[k + 3] = va.w;
        b[k] = vb4.x; b[k + 1] = vb4.y; b[k + 2] = vb4.z; b[k + 3] = vb4.w;
      }
    } else {
#pragma unroll
      for (int k = 0; k < IT; k++) {
        const bool in = i0 + k < n_blocks;
        a[k] = in ? cp.cnt_next[i0 + k] : 0u;
        b[k] = in ? cp.cnt_shadow[i0 + k] : 0u;
      }
    }
#pragma unroll
    for (int k = 0; k < IT; k++) { sa += a[k]; sb += b[k]; }
    uint32_t xa = sa, xb = sb;  // inclusive scan over the wave's threads
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t ua = __shfl_up(xa, off), ub = __shfl_up(xb, off);
      if (lane >= off) { xa += ua; xb += ub; }
    }
    if (lane == 63) { wtot[0][wv] = xa; wtot[1][wv] = xb; }
    __syncthreads();
    uint32_t pa = carry[0], pb = carry[1];
    for (int k = 0; k < wv; k++) { pa += wtot[0][k]; pb += wtot[1][k]; }
    uint32_t ea = pa + xa - sa, eb = pb + xb - sb;  // exclusive prefix of this thread's first count
    // the counts become their exclusive prefixes, in place
#pragma unroll
    for (int k = 0; k < IT; k++) {
      const uint32_t ta = a[k], tb = b[k];
      a[k] = ea; b[k] = eb;
      ea += ta; eb += tb;
    }
    if (whole) {
#pragma unroll
      for (int k = 0; k < IT; k += 4) {
        *reinterpret_cast<uint4 *>(cp.cnt_next + i0 + k) = make_uint4(a[k], a[k + 1], a[k + 2], a[k + 3]);
        *reinterpret_cast<uint4 *>(cp.cnt_shadow + i0 + k) = make_uint4(b[k], b[k + 1], b[k + 2], b[k + 3]);
      }
    } else {
#pragma unroll
      for (int k = 0; k < IT; k++)
        if (i0 + k < n_blocks) { cp.cnt_next[i0 + k] = a[k]; cp.cnt_shadow[i0 + k] = b[k]; }
    }
    __syncthreads();
    if (threadIdx.x == 1023) { carry[0] = pa + xa; carry[1] = pb + xb; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const uint32_t n_cur = ctl->n_cur;
    ctl->n_next = carry[0];
    ctl->n_shadow = carry[1];
    ctl->st_rays += (unsigned long long)n_cur + carry[1];
    ctl->st_shadow += carry[1];
    next->n_pix = ctl->n_pix;
    next->n_prev = n_cur;
    next->n_cur = carry[0];
    next->n_next = 0;
    next->n_shadow = 0;
    next->grab[0] = next->grab[1] = next->grab[2] = 0;
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
    if ((nls & 3) == 0) {
      // one thread per entry, its nls visibility bytes read as words: an exclusive scan of the entries' counts
      // over the workgroup, then every thread writes its own slots (one pass and two barriers instead of nls)
      const uint8_t *vrow = vb.vis + (size_t)i * nls;
      uint32_t mine = 0;
      if (i < n)
        for (int w = 0; w < nls; w += 4) mine += (uint32_t)__popc(*reinterpret_cast<const uint32_t *>(vrow + w) & 0x02020202u);
      uint32_t incl = mine;
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(incl, off);
        if (lane >= off) incl += up;
      }
      if (lane == 63) wcnt[wv] = incl;
      __syncthreads();
      uint32_t at = run + incl - mine;
      for (int k = 0; k < wv; k++) at += wcnt[k];
      if (mine) {
        for (int w = 0; w < nls; w += 4) {
          const uint32_t vw = *reinterpret_cast<const uint32_t *>(vrow + w);
#pragma unroll
          for (int b = 0; b < 4; b++)
            if ((vw >> (8 * b)) & 2u) sq.slot[at++] = (int32_t)((size_t)i * nls + w + b);
        }
      }
      __syncthreads();
      continue;
    }
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
// (A shadow ray whose walk left one primitive undecided, vis = 3, is settled here by the
// reference's FP64 test of that primitive: bvh.cpp:187, occluded when it is hit strictly before
// the ray's end.)
__device__ __forceinline__ bool pending_visible(const DevScene &S, const VertexBuf &vb, const SRay *rays, uint32_t i,
                                                size_t slot) {
  const SRay r = rays[slot];
  double t, u, v;
  const bool ok = exact_test(S, vb.pend_prim[slot], V(vb.hp[0][i], vb.hp[1][i], vb.hp[2][i]), V(r.d[0], r.d[1], r.d[2]),
                             PT_EPS_D, t, u, v);
  return !(ok && t < r.max_t);
}
// (one workgroup per 256 entries, as k_shade; `blk` its number, n the length of the list)
__device__ __forceinline__ void resolve_block(uint32_t blk, uint32_t n, const FrameParams &fp, const DevScene &S, const RayQ &q,
                                              const int32_t *__restrict__ list, const VertexBuf &vb, const SRay *__restrict__ rays,
                                              const PathBuf &pb, Ctl *ctl, int32_t depth) {
  const int nls = fp.nls;
  unsigned long long verts = 0;
  {
    const uint32_t base_i = blk * blockDim.x;
    if (base_i >= n) return;
    const uint32_t i = base_i + threadIdx.x;
    if (i < n) {
      // Loads that do not depend on each other are issued together: the kernel does
      // almost no arithmetic, its time is the chain of memory round trips.
      const uint8_t flag = vb.flag[i];
      const uint32_t j = list ? (uint32_t)list[i] : i;
      if (flag) {
        verts++;
        const int32_t path = q.path[j] & PATH_MASK;
        const v3 T = depth == 1 ? V(1, 1, 1) : V(q.T[0][j], q.T[1][j], q.T[2][j]);
        v3 L = V(0, 0, 0);
        const size_t b = (size_t)i * nls;
        if ((nls & 3) == 0) {
          for (int q4 = 0; q4 < nls; q4 += 4) {
            uint32_t vw = *reinterpret_cast<const uint32_t *>(vb.vis + b + q4);
            // undecided slots first, one after the other (a rolled loop: the FP64 test's registers once)
            if (vw & 0x02020202u) {
#pragma unroll 1
              for (int qq = 0; qq < 4; qq++) {
                if (((vw >> (8 * qq)) & 0xffu) == 3u) {
                  const uint32_t vs = pending_visible(S, vb, rays, i, b + q4 + qq) ? 1u : 0u;
                  vw = (vw & ~(0xffu << (8 * qq))) | (vs << (8 * qq));
                }
              }
            }
            if (!(vw & 0x01010101u)) continue;  // none of the four is visible (the bytes are 0, 1 or 2 by now)
            Contrib cc[4];
#pragma unroll
            for (int qq = 0; qq < 4; qq++) cc[qq] = vb.contrib[b + q4 + qq];  // all four once one is visible
#pragma unroll
            for (int qq = 0; qq < 4; qq++)
              if (((vw >> (8 * qq)) & 0xffu) == 1u) L = vadd(L, V(cc[qq].c[0], cc[qq].c[1], cc[qq].c[2]));
          }
        } else {
          for (int qq = 0; qq < nls; qq++) {
            const uint32_t vs = vb.vis[b + qq];
            if (vs == 1u || (vs == 3u && pending_visible(S, vb, rays, i, b + qq))) {
              const Contrib cc = vb.contrib[b + qq];
              L = vadd(L, V(cc.c[0], cc.c[1], cc.c[2]));
            }
          }
        }
        // Nothing visible (or only zeros): T * L = +-0 would be added, which changes no sum (they start at +0, and
        // x + -0 = x, +0 + +0 = +0), unless T is not finite (inf * 0 = NaN, as in the reference): the
        // read-modify-write of the path's radiance is left out.
        // (a quarter of the headline frame's vertices: k_resolve 16.1 -> 15.3 ms)
        const bool add = (flag & 2) || !(L.x == 0 && L.y == 0 && L.z == 0) ||
                         !(fabs(T.x) < INFINITY && fabs(T.y) < INFINITY && fabs(T.z) < INFINITY);
        if (add) {
          L = vscale(L, 1. / (double)nls);
          const v3 val = vmul(T, L);
          if (flag & 2) {
            pb.pend[0][path] = val.x; pb.pend[1][path] = val.y; pb.pend[2][path] = val.z;
          } else {
            pb.rad[0][path] += val.x; pb.rad[1][path] += val.y; pb.rad[2][path] += val.z;
          }
        }
      }
    }
  }
  block_add(&ctl->st_vertices, verts);
}
__global__ void k_resolve(FrameParams fp, DevScene S, RayQ q, const int32_t *__restrict__ list, VertexBuf vb,
                          const SRay *__restrict__ rays, PathBuf pb, Ctl *ctl, int32_t depth) {
  resolve_block(blockIdx.x, ctl->n_cur, fp, S, q, list, vb, rays, pb, ctl, depth);
}
// k_light of depth d + 1 and k_resolve of depth d in ONE launch: the first is bound by instruction issue,
// the second by its memory round trips, and they touch different records once the per-sample records, the
// vertex flags and the ray queues are kept per depth (ptamd_trace_list). Launched on two streams the two
// kernels ran one after the other; as workgroups of one grid they share the compute units. Every workgroup
// takes one role; the resolve workgroups are spread evenly among the light workgroups (of the first L + R
// workgroups, L light and R resolve, workgroup b resolves iff floor((b + 1) R / (L + R)) > floor(b R / (L + R))),
// so that both kinds are resident throughout the launch. `prev`: the list, the queue, the vertex and sample
// records of depth d, whose length k_advance has kept in Ctl::n_prev.
__global__ void __launch_bounds__(256)
    k_light_resolve(FrameParams fp, DevScene S, VertexBuf vb, ShadowQ sq, Compact cp, Ctl *ctl, FastDiv by_nls, uint32_t rng_base,
                    ResolveArgs prev) {
  const uint32_t n_prev = ctl->n_prev;
  const unsigned long long L = ((unsigned long long)ctl->n_cur * (uint32_t)fp.nls + 255u) >> 8, R = (n_prev + 255u) >> 8;
  const unsigned long long b = blockIdx.x, total = L + R;
  if (b >= total) return;
  const unsigned long long r_before = b * R / total;
  if ((b + 1) * R / total > r_before) resolve_block((uint32_t)r_before, n_prev, fp, S, prev.q, prev.list, prev.vb, prev.rays, prev.pb, ctl, prev.depth);
  else light_block((uint32_t)(b - r_before), fp, S, vb, sq, cp, ctl, by_nls, rng_base);
}
// Folds the round's finished paths into the per-pixel sums, in sample order:
// the loop of raytrace_pixel (pathtracer.cpp:249-267) over radiances that are
// already computed. A round may hold several batches of a pixel, traced ahead of
// their adaptive tests; the test runs here between the batches, and what was
// traced past a pixel's stop is dropped. state: [pixel*6] = s1, s2, sum.rgb, n.
// (One thread per pixel: its loads are not coalesced, but a wave per pixel with coalesced
// loads and the ordered sum passed lane to lane was measured 2.5 times slower; the serial
// additions, not the loads, are what a pixel costs.)
__global__ void k_accumulate(PixView pv, uint32_t p0, Ctl *ctl, RoundPlan rp, PathMap pm, PathBuf pb, double *__restrict__ state) {
  const uint32_t stride = gridDim.x * blockDim.x;
  const uint32_t n_pixels = ctl->n_pix;
  unsigned long long kept = 0;
  for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < n_pixels; p += stride) {
    double *st = state + (size_t)pv.at(p0 + p) * 6;
    int32_t first;
    const int32_t planned = rp.planned(st, first);
    double acc[2] = {st[0], st[1]};
    v3 sum = V(st[2], st[3], st[4]);
    const size_t b = pm.first_slot(p);
    // runs of samples between two tests: inside a run nothing depends on the sums, so its loads
    // are issued eight at a time; the additions stay in sample order
    int s = 0;
    int next_test = (first / rp.batch + 1) * rp.batch;  // the test comes before sample i = next_test
    while (s < planned) {
      if (s > 0 && first + s == next_test) {
        if (!pixel_continues(acc, next_test, rp.tol)) break;
        next_test += rp.batch;
      }
      const int run_end = next_test - first < planned ? next_test - first : planned;
      for (; s + 8 <= run_end; s += 8) {
        double Lx[8], Ly[8], Lz[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { Lx[k] = pb.rad[0][b + s + k]; Ly[k] = pb.rad[1][b + s + k]; Lz[k] = pb.rad[2][b + s + k]; }
#pragma unroll
        for (int k = 0; k < 8; k++) {
          const v3 L = V(Lx[k], Ly[k], Lz[k]);
          const double xl = illum(L);
          acc[0] += xl;
          acc[1] += xl * xl;
          sum = vadd(sum, L);
        }
      }
      for (; s < run_end; s++) {
        const v3 L = V(pb.rad[0][b + s], pb.rad[1][b + s], pb.rad[2][b + s]);
        const double xl = illum(L);
        acc[0] += xl;
        acc[1] += xl * xl;
        sum = vadd(sum, L);
      }
    }
    st[0] = acc[0]; st[1] = acc[1]; st[2] = sum.x; st[3] = sum.y; st[4] = sum.z;
    st[5] += (double)s;
    kept += (unsigned long long)s;
  }
  block_add(&ctl->st_samples, kept);
}

template __global__ void k_shade<false>(FrameParams, DevScene, RayQ, const int32_t *, HitBuf, RayQ, ShadowQ, VertexBuf, PathBuf, Compact,
                                        Ctl *, int32_t);
template __global__ void k_shade<true>(FrameParams, DevScene, RayQ, const int32_t *, HitBuf, RayQ, ShadowQ, VertexBuf, PathBuf, Compact,
                                       Ctl *, int32_t);

}  // namespace ptk
