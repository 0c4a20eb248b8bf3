// The traversal kernels: k_extend (closest hit of a ray list), k_shadow (first hit of the shadow-ray
// list) and k_trace2 (the bounce rays of the next depth and the shadow rays of this one in one launch
// of persistent waves). The walk itself is in walk.h.
#include "kernels.h"
#include "sampling.h"
#include "walk.h"

namespace ptk {

// Ray and vertex records live at FIXED positions (a bounce ray at its parent's
// list position, light sample q of vertex i at slot i*nls + q) and the queues are
// ordered index lists built by ballot / prefix-sum compaction: no atomics, and
// the order is the same on every run.
constexpr int32_t PRIM_UNUSED = -2;  // hit record of a path slot that holds no ray
struct ExtendPolicy {
  RayQ q;
  const int32_t *list;
  HitBuf hb;
  double min_t, max_t;
  const int32_t *self_code;  // VertexBuf::self_code / self_dmin: a bounce ray sits at its vertex's entry
  const float *dmin;
  v3 cam_o;  // where every camera ray starts (a list-less launch traces camera rays: k_generate does not store their origins)
  __device__ __forceinline__ double min_t_all() const { return min_t; }
  // false: the slot holds no ray (see k_generate); its hit record says so
  __device__ __forceinline__ bool load(uint32_t i, v3 &o, v3 &d, double &tmax, uint32_t &self) const {
    const uint32_t j = list ? (uint32_t)list[i] : i;
    self = list ? (uint32_t)self_code[j] : PRIM_NONE;  // camera rays start on nothing
    d = V(q.d[0][j], q.d[1][j], q.d[2][j]);
    if (!list && d.x == 0 && d.y == 0 && d.z == 0) {
      hb.prim[i] = PRIM_UNUSED;
      return false;
    }
    o = list ? V(q.o[0][j], q.o[1][j], q.o[2][j]) : cam_o;
    tmax = max_t;
    return true;
  }
  __device__ __forceinline__ double ray_end(uint32_t) const { return max_t; }
  __device__ __forceinline__ float self_dmin(uint32_t i) const { return dmin[list[i]]; }
  // (the hit's t, u, v are computed by the kernel that shades it)
  __device__ __forceinline__ void store(uint32_t i, const Hit &h, bool) const { hb.prim[i] = h.prim; }
};

template <bool COUNT>
__global__ void TRAV_BOUNDS_N(PTAMD_WAVES_EXTEND)
    k_extend(DevScene S, RayQ q, const int32_t *__restrict__ list, HitBuf hb, Ctl *ctl, double min_t, double max_t,
             uint2 *gstack, const int32_t *__restrict__ self_code, const float *__restrict__ self_dmin, v3 cam_o) {
  extern __shared__ uint2 lds_stack[];
  const TravStack stack{(lds_uint2 *)lds_stack, gstack, gridDim.x * blockDim.x, blockIdx.x * blockDim.x + threadIdx.x, PTAMD_STACK_EXTEND};
  uint32_t cn = 0, ct = 0, cx = 0;
  const ExtendPolicy pol{q, list, hb, min_t, max_t, self_code, self_dmin, cam_o};
  PTAMD_WAVELOG_BEGIN
  trace_rays<0, COUNT, PTAMD_REFILL_EXTEND>(S, pol, ctl->n_cur, &ctl->grab[0], stack, cn, ct, cx);
  PTAMD_WAVELOG_END(S)
  if (COUNT) {
    atomicAdd(&ctl->st_nodes, (unsigned long long)cn);
    atomicAdd(&ctl->st_tests, (unsigned long long)ct);
    atomicAdd(&ctl->st_exact, (unsigned long long)cx);
  }
}

// Shadow rays: light sampling needs only "is anything in the way"
// (pathtracer.cpp:127); hemisphere sampling needs the closest hit's emission
// (pathtracer.cpp:79-82). vis[slot]: 2 = ray pending, 1 = light sample counts, 0 = not.
template <bool HEMI>
struct ShadowPolicy {
  DevScene S;
  ShadowQ sq;
  VertexBuf vb;
  FastDiv by_nls;  // slot -> entry (exact for slot < 2^31, nls < 2^16: both enforced by ptamd_trace_list)
  uint32_t nls, ns_area_light;
  __device__ __forceinline__ double min_t_all() const { return PT_EPS_D; }
  // the tag of the light that sample q of a vertex goes to (k_light's loop over the lights, backwards;
  // once per ray, and only in scenes where some light has end primitives: linear in the number of lights)
  __device__ __forceinline__ uint32_t light_tag_of(uint32_t q) const {
    for (int li = 0; li < S.n_lights; li++) {
      const uint32_t ns = light_is_delta(S.light_kind[li]) ? 1u : ns_area_light;
      if (q < ns) return (uint32_t)S.light_tag[li];
      q -= ns;
    }
    return 0u;
  }
  __device__ __forceinline__ bool load(uint32_t i, v3 &o, v3 &d, double &tmax, uint32_t &self) const {
    const int32_t slot = sq.slot[i];
    const uint32_t e = by_nls.div((uint32_t)slot);
    self = (uint32_t)vb.self_code[e];
    if (!HEMI && S.has_end) self |= light_tag_of((uint32_t)slot - e * nls) << PRIM_BITS;
    o = V(vb.hp[0][e], vb.hp[1][e], vb.hp[2][e]);
    const SRay r = sq.ray[slot];
    d = V(r.d[0], r.d[1], r.d[2]);
    tmax = HEMI ? (double)INFINITY : r.max_t;
    return true;
  }
  __device__ __forceinline__ double ray_end(uint32_t i) const {
    return HEMI ? (double)INFINITY : sq.ray[sq.slot[i]].max_t;
  }
  __device__ __forceinline__ float self_dmin(uint32_t i) const { return vb.self_dmin[by_nls.div((uint32_t)sq.slot[i])]; }
  __device__ __forceinline__ void store(uint32_t i, const Hit &h, bool found) const {
    const int32_t slot = sq.slot[i];
    if (!HEMI) {
      // 3: one primitive is left to decide (k_resolve)
      vb.vis[slot] = found ? (h.certain ? 0 : 3) : 1;
      if (found && !h.certain) vb.pend_prim[slot] = h.prim;
    } else {
      if (found) {
        const int32_t m = S.prim_info[h.prim] >> 1;
        v3 light_L = V(0, 0, 0);
        if (S.mat_kind[m] == PTAMD_MAT_EMISSION) light_L = V(S.mat_param[3 * m], S.mat_param[3 * m + 1], S.mat_param[3 * m + 2]);
        // f * light_L * dot(wi, w2o*n) * 2 * PI; f waits in the contrib slot, the cosine in max_t
        const Contrib cf = vb.contrib[slot];
        const v3 f = V(cf.c[0], cf.c[1], cf.c[2]);
        const v3 term = vscale(vscale(vscale(vmul(f, light_L), sq.ray[slot].max_t), 2), PT_PI);
        vb.contrib[slot] = Contrib{{term.x, term.y, term.z}};
      }
      vb.vis[slot] = found ? 1 : 0;
    }
  }
};

template <bool HEMI, bool COUNT>
__global__ void TRAV_BOUNDS_N(PTAMD_WAVES_SHADOW)
    k_shadow(DevScene S, ShadowQ sq, VertexBuf vb, Ctl *ctl, FastDiv by_nls, uint2 nls_nsa, uint2 *gstack, uint32_t fuse_below) {
  // (the launch before, k_trace2, has traced these rays itself if the two lists were short: same test)
  if (ctl->n_next + ctl->n_shadow < fuse_below) return;
  extern __shared__ uint2 lds_stack[];
  const TravStack stack{(lds_uint2 *)lds_stack, gstack, gridDim.x * blockDim.x, blockIdx.x * blockDim.x + threadIdx.x, PTAMD_STACK_SHADOW};
  uint32_t cn = 0, ct = 0, cx = 0;
  const ShadowPolicy<HEMI> pol{S, sq, vb, by_nls, nls_nsa.x, nls_nsa.y};
  PTAMD_WAVELOG_BEGIN
  if (HEMI) trace_rays<0, COUNT, PTAMD_REFILL_SHADOW>(S, pol, ctl->n_shadow, &ctl->grab[1], stack, cn, ct, cx);
  else trace_rays<1, COUNT, PTAMD_REFILL_SHADOW>(S, pol, ctl->n_shadow, &ctl->grab[1], stack, cn, ct, cx);
  PTAMD_WAVELOG_END(S)
  if (COUNT) {
    atomicAdd(&ctl->st_nodes, (unsigned long long)cn);
    atomicAdd(&ctl->st_tests, (unsigned long long)ct);
    atomicAdd(&ctl->st_exact, (unsigned long long)cx);
  }
}

// The bounce rays of the next depth (list and count: the queue k_scatter has just
// written, Ctl::n_next) and the shadow rays of this depth, see trace_two_lists.
template <bool HEMI, bool COUNT>
__global__ void TRAV_BOUNDS_N(PTAMD_WAVES_EXTEND)
    k_trace2(DevScene S, RayQ q, const int32_t *__restrict__ list, HitBuf hb, ShadowQ sq, VertexBuf vb, Ctl *ctl,
             FastDiv by_nls, uint2 nls_nsa, uint2 *gstack, uint32_t fuse_below) {
  extern __shared__ uint2 lds_stack[];
  const TravStack stack{(lds_uint2 *)lds_stack, gstack, gridDim.x * blockDim.x, blockIdx.x * blockDim.x + threadIdx.x, PTAMD_STACK_EXTEND};
  uint32_t cn = 0, ct = 0, cx = 0;
  const ExtendPolicy pe{q, list, hb, PT_EPS_D, (double)INFINITY, vb.self_code, vb.self_dmin, V(0, 0, 0)};  // (always a list)
  for (int k = 0; k < 3; k++) vb.hp[k] = q.o[k];  // the same arrays (see ptamd_trace_list): one set of pointers to hold
  const ShadowPolicy<HEMI> ps{S, sq, vb, by_nls, nls_nsa.x, nls_nsa.y};
  // Short lists are traced together (one drain phase instead of two, see trace_two_lists). Long
  // shadow-ray lists are left to k_shadow, the next launch: first-hit walks carry less state and
  // run at 7 waves per SIMD there instead of 5 here, which is worth more than the drain phase.
  const uint32_t n_e = ctl->n_next;
  const uint32_t n_s = (n_e + ctl->n_shadow < fuse_below) ? ctl->n_shadow : 0u;
  if (blockIdx.x == 0 && threadIdx.x == 0) {  // the only writer of these two during the launch
    ctl->st_two_ext += n_e;
    ctl->st_two_shadow += n_s;
  }
  PTAMD_WAVELOG_BEGIN
  trace_two_lists<HEMI, COUNT, PTAMD_REFILL_EXTEND>(S, pe, n_e, &ctl->grab[2], ps, n_s, &ctl->grab[1], stack, cn, ct, cx);
  PTAMD_WAVELOG_END(S)
  if (COUNT) {
    atomicAdd(&ctl->st_nodes, (unsigned long long)cn);
    atomicAdd(&ctl->st_tests, (unsigned long long)ct);
    atomicAdd(&ctl->st_exact, (unsigned long long)cx);
  }
}

template __global__ void k_extend<false>(DevScene, RayQ, const int32_t *, HitBuf, Ctl *, double, double, uint2 *, const int32_t *, const float *, v3);
template __global__ void k_extend<true>(DevScene, RayQ, const int32_t *, HitBuf, Ctl *, double, double, uint2 *, const int32_t *, const float *, v3);
#define PTK_INST2(K, H, C) template __global__ void K<H, C>
PTK_INST2(k_shadow, false, false)(DevScene, ShadowQ, VertexBuf, Ctl *, FastDiv, uint2, uint2 *, uint32_t);
PTK_INST2(k_shadow, false, true)(DevScene, ShadowQ, VertexBuf, Ctl *, FastDiv, uint2, uint2 *, uint32_t);
PTK_INST2(k_shadow, true, false)(DevScene, ShadowQ, VertexBuf, Ctl *, FastDiv, uint2, uint2 *, uint32_t);
PTK_INST2(k_shadow, true, true)(DevScene, ShadowQ, VertexBuf, Ctl *, FastDiv, uint2, uint2 *, uint32_t);
PTK_INST2(k_trace2, false, false)(DevScene, RayQ, const int32_t *, HitBuf, ShadowQ, VertexBuf, Ctl *, FastDiv, uint2, uint2 *, uint32_t);
PTK_INST2(k_trace2, false, true)(DevScene, RayQ, const int32_t *, HitBuf, ShadowQ, VertexBuf, Ctl *, FastDiv, uint2, uint2 *, uint32_t);
PTK_INST2(k_trace2, true, false)(DevScene, RayQ, const int32_t *, HitBuf, ShadowQ, VertexBuf, Ctl *, FastDiv, uint2, uint2 *, uint32_t);
PTK_INST2(k_trace2, true, true)(DevScene, RayQ, const int32_t *, HitBuf, ShadowQ, VertexBuf, Ctl *, FastDiv, uint2, uint2 *, uint32_t);
#undef PTK_INST2

}  // namespace ptk
