// Data the kernels of the device layer work on, as the host hands it to them: the scene, the frame's
// parameters, the ray / vertex / path buffers of a round (structure-of-arrays views into one slab),
// the control block with the list lengths and statistics.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ptamd.h"
#include "dev_node.h"
#include "device_math.h"
#include "fastdiv.h"
#include "trav_config.h"

namespace ptk {
using namespace ptdev;

struct DevScene {
  const DevNode *nodes;
  const double *prim;       // [n_prims * PRIM_STRIDE]
  const uint4 *prim32;      // [n_prims * 3]: the FP32 pre-test's record, see ptamd_upload_scene
  const int32_t *prim_info; // bit 0: sphere, bits 1..: material index
  const double *prim_nrm;   // [n_prims * 9]
  const int32_t *mat_kind;
  const double *mat_param;
  const double *mat_extra;  // [n_materials*4] glass reflectance, ior (specular mode)
  const int32_t *light_kind;
  const double *light_param;
  // Primitives that lie where a light's shadow rays END (the emissive quad an area light sits on), see
  // find_light_end_prims: light_end[4 * li + k] = primitive or -1; light_tag[li] = the tag (1..7) their
  // child codes carry, 0 if the light has none; has_end = any light has some.
  const int32_t *light_end;
  const int32_t *light_tag;
  int32_t has_end;
  const double *jitter;
  int32_t n_lights, jitter_n, root_code, stack_cap;
  float coord_bound;  // no ray starts farther from the origin than this on any axis (prim32.h: big)
#ifdef PTAMD_WAVELOG
  // measurement build: every traversal wave writes when it started and ended (100 MHz ticks) at
  // wavelog[2 * (wavelog_base + wave)]; the host names a fresh range per launch (tools/wave_drain.py)
  unsigned long long *wavelog;
  uint32_t wavelog_base, wavelog_cap;
#endif
};
#ifdef PTAMD_WAVELOG
#define PTAMD_WAVELOG_BEGIN const unsigned long long wl_t0 = wall_clock64();
#define PTAMD_WAVELOG_END(S)                                                                                   \
  if ((threadIdx.x & 63) == 0) {                                                                              \
    const uint32_t wl_w = (S).wavelog_base + blockIdx.x * (blockDim.x / 64u) + (threadIdx.x >> 6);           \
    if (wl_w < (S).wavelog_cap) { (S).wavelog[2 * (size_t)wl_w] = wl_t0; (S).wavelog[2 * (size_t)wl_w + 1] = wall_clock64(); } \
  }
#else
#define PTAMD_WAVELOG_BEGIN
#define PTAMD_WAVELOG_END(S)
#endif

struct FrameParams {
  int32_t w, h, max_ray_depth, accum, ns_area_light, hemisphere, nls, bpv, specular, pad;
  uint32_t seed_lo, seed_hi;
  double tan_h, tan_v, n_clip, f_clip;
  v3 cam_pos;
  m3 c2w;
};

// One per path depth in flight: the kernels of depth d work on block d & 3 of a ring (CTL_RING), whose list lengths
// the previous depth's k_scan2 has set; the statistics are summed over the ring when they are read.
constexpr int CTL_RING = 4;
struct Ctl {
  uint32_t n_cur, n_next, n_shadow, n_pix;  // sizes of the ray lists; pixels of the running chunk
  uint32_t n_prev, pad_;  // the previous depth's n_cur (set by that depth's k_scan2): k_light_resolve settles that depth's light samples
  uint32_t grab[4];  // next unclaimed ray of a list: [0] k_extend, [1] shadow rays, [2] bounce rays in k_trace2 (persistent drivers)
  unsigned long long st_rays, st_shadow, st_vertices, st_nodes, st_tests, st_samples, st_dead;
  unsigned long long st_exact;  // primitive tests that needed the FP64 path (counted with st_tests)
  unsigned long long st_two_ext, st_two_shadow;  // rays of either kind traced by k_trace2 launches
};

// A shard of a device-resident pixel list: entries offset, offset + stride, ...
// below the list's length, which is read on the device when d_n is given (the
// host then only knows an upper bound).
struct PixView {
  const int32_t *pixels;
  const long long *d_n;
  long long n;
  uint32_t offset, stride;
  __device__ __forceinline__ long long total() const { return d_n ? *d_n : n; }
  __device__ __forceinline__ uint32_t count() const {
    const long long t = total();
    return t > (long long)offset ? (uint32_t)((t - offset + stride - 1) / stride) : 0u;
  }
  __device__ __forceinline__ int32_t at(uint32_t k) const { return pixels[offset + (size_t)k * stride]; }
};

// Ray queue entry arrays (SoA, capacity = max paths)
struct RayQ {
  int32_t *path;
  double *o[3], *d[3], *T[3];
};
struct HitBuf {
  int32_t *prim;
  double *t;  // only ptamd_trace_primary asks for distances
};
// Per light sample (slot = entry * nls + q) the records are arrays of structures
// (shadow ray 32 bytes, contribution 24): the nls samples of one vertex are written by
// adjacent lanes of k_light as one contiguous run, and a shadow ray reads its record with
// two 16-byte loads.
struct alignas(32) SRay {
  double d[3];
  double max_t;  // hemisphere mode: cos term instead
};
struct Contrib {  // 24 bytes: the kernels that touch it are bound by the bytes they move
  double c[3];
};
struct ShadowQ {
  int32_t *slot;  // entry * nls + q
  SRay *ray;      // [cap * nls]
};
struct VertexBuf {
  double *hp[3];     // hit point per queue entry (shadow-ray origin): the origin arrays of the next ray queue
  uint8_t *flag;     // bit 0: has light samples, bit 1: result goes to `pending`
  Contrib *contrib;  // [cap * nls]
  uint8_t *vis;      // [cap * nls]
  // The primitive a vertex lies on, as a child code, when every ray leaving the vertex may skip it
  // (see k_shade), else CHILD_EMPTY; and the |D| above which the reference's test of that primitive
  // certainly fails its min_t clause.
  int32_t *self_code;  // [cap]
  float *self_dmin;    // [cap]
  int32_t *pend_prim;  // [cap * nls] primitive a shadow ray could not decide in FP32 (vis = 3), for k_resolve
  // What k_light needs of a vertex that takes light samples (written by k_shade): the z axis of its
  // shading frame (make_coord_space: n / |n|), and one 16-byte word: the Philox counter / key words
  // (sample index, pixel) of its path and its material (f is recomputed from it: three multiplications).
  double *nz[3];  // [cap]
  uint4 *nee;     // [cap] {sample, pixel, material, -}
};
struct PathBuf {
  double *rad[3];
  double *pend[3];
  // the pixel and the sample index of a path slot (written by k_generate): the Philox key and counter words of its
  // path, side by side (one 8-byte gather per vertex instead of two of 4)
  int2 *pix_smp;
};
// What k_resolve needs of a depth, for the launch that settles it beside the next depth's light samples (k_light_resolve)
struct ResolveArgs {
  RayQ q;
  const int32_t *list;
  VertexBuf vb;
  const SRay *rays;
  PathBuf pb;
  int32_t depth;
};
// Where a chunk's pixels keep their paths. With every listed pixel tracing the same number of samples, pixel
// p of the chunk owns slots [p ns, (p + 1) ns). When every pixel traces as many batches as ITS statistics
// ask for (RoundPlan::planned), the slots are packed: pixel p owns [off[p], off[p + 1]), off being the
// exclusive prefix sum of the planned counts (k_plan_*), so that a launch holds no unused slot.
struct PathMap {
  const uint32_t *off;       // [pixels of the chunk + 1], or null: ns slots per pixel
  const uint32_t *slot_pix;  // packed: the pixel (index within the chunk) that owns slot 256 b, for every b
  uint32_t ns;
  __device__ __forceinline__ uint32_t first_slot(uint32_t p) const { return off ? off[p] : p * ns; }
  // the pixel (index within the chunk) that owns slot k < off[np], and k's place among that pixel's slots
  __device__ __forceinline__ uint32_t owner(uint32_t k, uint32_t np, uint32_t &sk) const {
    if (!off) { sk = k % ns; return k / ns; }
    // the owner of the 256-slot block's first slot, then at most 256 pixels further on
    uint32_t lo = slot_pix[k >> 8], hi = lo + 257u < np ? lo + 257u : np;  // off[lo] <= k < off[hi]
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (off[mid] <= k) lo = mid; else hi = mid;
    }
    sk = k - off[lo];
    return lo;
  }
};

// What a round traces for one pixel. With sample_base >= 0 the caller drives the
// cadence: samples [sample_base, sample_base + ns) for every listed pixel. With
// sample_base < 0 every pixel continues from its own count (state[5]) and gets
// as many batches, up to ns / batch, as its statistics say it will still need:
// the adaptive test passes once 1.96 sqrt(var / n) <= tol mean, i.e. from
// n = (1.96 / tol)^2 var / mean^2 on. A pixel predicted to stop soon is not
// traced far ahead of its tests; a wrong guess only costs it another round. The
// prediction never changes which samples a pixel ends up with (samples are
// keyed by their index, tests run at every multiple of `batch`), only when they
// are traced.
struct RoundPlan {
  const double *state;
  double tol;
  int32_t sample_base, ns, batch, ns_aa;
  int32_t predict;  // 0: every pixel gets all ns samples (a launch too small for dropped samples to matter)
  double margin;    // samples planned per sample predicted (see planned)
  __device__ __forceinline__ int32_t first_of(const double *st) const {
    return sample_base >= 0 ? sample_base : (int32_t)st[5];
  }
  // number of samples to trace for the pixel with state row st; first = index of the first
  __device__ __forceinline__ int32_t planned(const double *st, int32_t &first) const {
    if (sample_base >= 0) { first = sample_base; return ns; }
    const int32_t n = (int32_t)st[5];
    first = n;
    int32_t want = ns;
    if (predict && n >= 2) {
      const double s1 = st[0], s2 = st[1];
      const double mean = s1 / n, var = (s2 - s1 * s1 / n) / (n - 1);
      const double c = 1.96 / tol;
      // margin: erring on the long side costs a few dropped samples, erring on the short side a round
      const double need = margin * (c * c * var / (mean * mean) - n);
      if (need == need) {  // not NaN
        double kb = ceil(need / batch);
        kb = kb < 1 ? 1 : (kb > (double)(ns / batch) ? (double)(ns / batch) : kb);
        want = (int32_t)kb * batch;
      }
    }
    const int32_t left = ns_aa - n;
    return want < left ? want : (left > 0 ? left : 0);
  }
};

struct Compact {
  uint8_t *pflag;        // [cap] entry pushes a bounce ray
  uint32_t *cnt_next;    // [cap/256 + 1] bounce rays per 256-entry block, then their exclusive scan
  uint32_t *cnt_shadow;  // [cap/256 + 1] shadow rays per 256-entry block, then their exclusive scan
};

// Adds the workgroup's sum of v to *dst with ONE atomic (every thread of the workgroup must call
// it): adds to one address are served one after the other, ~20 ns each, and a launch does not end
// before its last one, so an add per wave of a 2048-workgroup launch is a 0.16 ms tail.
__device__ __forceinline__ void block_add(unsigned long long *dst, unsigned long long v) {
  __shared__ unsigned long long part[16];
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int k = 0; k < (int)((blockDim.x + 63) >> 6); k++) t += part[k];
    if (t) atomicAdd(dst, t);
  }
}

}  // namespace ptk
