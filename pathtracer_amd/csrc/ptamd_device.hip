// Device layer of the render hot path for MI355X (gfx950): a wavefront path
// tracer. One frame is rendered in rounds of one or more batches of
// `samples_per_batch` samples per still-active pixel; each round runs, per path
// depth,
//
//   extend   closest-hit BVH traversal of the ray list       (BVHAccel::intersect)
//   shade    emission, Russian roulette, BSDF sample, light   (at_least_one_bounce_radiance,
//            samples into fixed slots; per-block counts        estimate_direct_lighting_*)
//   scan2, scatter   ordered compaction: next ray list, shadow-ray list
//   shadow   any-hit traversal of the shadow-ray list
//   resolve  ordered sum of the unoccluded light samples
//
// The shadow rays of depth d and the bounce rays of depth d + 1 are both ready once
// depth d is shaded and are traced by ONE launch (k_trace2); k_extend only traces
// the camera rays and k_shadow the last depth's shadow rays.
//
// and finally `accumulate` folds the finished paths into the per-pixel sums in
// sample order, applying the adaptive test between batches, exactly as
// PathTracer::raytrace_pixel does (src/pathtracer/pathtracer.cpp:234-272 of the
// reference).
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
#include "fastdiv.h"
#include "prim32.h"

using namespace ptdev;

namespace {

// ------------------------------------------------------------------ device data
// 4-wide BVH node in 64 bytes (half an L2 line): the four child boxes are 8-bit
// offsets on a per-node power-of-two grid, plus four child codes. The tree is the
// reference's binary SAH tree collapsed (a node's children are opened, largest box
// first, while the result fits four slots), with every primitive as its own child
// (so each primitive is culled by its own box before its own test). A code >= 0 is
// a node index, a negative code other than CHILD_EMPTY (an unused slot) is a
// primitive: ~code = primitive | light tag << 24 (PRIM_BITS below).
//
// Why so small: the traversal kernels are bound by the number of 16-byte
// per-lane loads the vector L1 can serve for divergent addresses (TCP and TD
// ~90 % busy in rocprofv3), so a node visit costs 4 loads instead of the 8 of a
// plain FP32 node or the 16 of two FP64 boxes.
//
// Plane q of axis k is org[k] + q * 2^(e_k), a float by construction (org is a
// multiple of 2^e_k and |org| / 2^e_k < 2^22), and the host rounds q outward, so
// the decoded box CONTAINS the reference's FP64 box. Together with the
// conservative float slab test below, a child is never culled when the
// reference's FP64 test (src/scene/bbox.cpp:10-47) would have accepted it; which
// hit wins is decided by the primitive tests only.
//
// The lower planes of the four children on one axis share a word (byte k: child k),
// and so do the upper planes: which of the two a ray meets first depends only on the
// sign of its direction, so the walk picks the near and the far WORD once per axis
// instead of ordering two distances per child and axis.
struct alignas(64) DevNode {
  float org[3];
  uint32_t exps;    // bytes 0..2: e_k as signed 8-bit integers; byte 3: the light tag its tagged primitive children carry (0: none, or not all the same)
  uint32_t qlo[3];  // per axis: lower planes of children 0..3
  uint32_t qhi[3];  // per axis: upper planes
  int32_t child[4];
  int32_t pad[2];
};
static_assert(sizeof(DevNode) == 64, "node must be half a cache line");
constexpr int32_t CHILD_EMPTY = INT32_MIN;
// A primitive's child code is ~(primitive | tag << 24): primitives number below 2^24 (checked at upload),
// tag != 0 marks an end primitive of light tag - 1 (DevScene::light_end). A ray carries a word of the
// same shape (Walk::skip): the primitive it starts on, or PRIM_NONE, and the tag of the light it is a
// shadow ray to, or 0.
constexpr uint32_t PRIM_BITS = 24, PRIM_NONE = (1u << PRIM_BITS) - 1u;
constexpr uint32_t MAX_LIGHT_TAG = 7;

// FP64 primitive record, 80 bytes (5 loads), read by exact_test. Triangle: p1,
// e1 = p2-p1, e2 = p3-p1 rounded exactly as Triangle::test computes them on the fly
// (src/scene/triangle.cpp:27); sphere: centre, r*r. The 10th slot holds
// prim_info (bit 0: sphere, bits 1..: material) in its low 32 bits.
constexpr int PRIM_STRIDE = 10;
// FP32 pre-test record, 48 bytes (3 loads), the one the walk normally reads: p1 (or the
// sphere's centre) as three doubles, so that o - p1 is formed in FP64 without cancellation
// error, then e1, e2 as floats. A sphere has NaN in e1.x and float(r*r) in e1.y.

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

struct Ctl {
  uint32_t n_cur, n_next, n_shadow, n_pix;  // sizes of the ray lists; pixels of the running chunk
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
// one lane of k_shade as one contiguous run, and a shadow ray reads its record with
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
      // margin (1.5): erring on the long side costs a few dropped samples, erring on the short side a round
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

// ------------------------------------------------------------------ intersection
// The FP32 box test. For a node with grid origin org and step s = 2^e, the
// distance along the ray to plane q of one axis is
//     t*(q) = (org + q s - o) / d
// (o, d: the ray's FP64 origin and direction). The kernel evaluates it as one FMA,
//     t(q) = fma(q, A, B),   A = s * inv,   B = (org - of) * inv,
// with of = float(o) and inv = v_rcp_f32(float(d)). Error budget: inv is off by
// < 2^-22 relative (conversion 2^-24, reciprocal 1 ulp); A is exact (a power-of-two
// scaling); B carries two more roundings; the FMA one; of differs from o by at
// most 2^-24 |of|. Hence
//     |t(q) - t*(q)| < (255 |A| + |B|) 2^-21 + |of inv| 2^-23.9 .
// The test uses E = (255 |A| + |B|) 2^-20 + |of inv| 2^-22 and moves every plane
// distance by E in the direction that makes the box bigger for this ray: the
// plane met first (lo when A >= 0) by -E, the other by +E; both are folded into B
// once per node, so a child costs six FMAs. The reference's four rejection tests
// (src/scene/bbox.cpp:38) follow, with min_t rounded down and the current max_t
// rounded up. The margin is 2^30 times the FP64 rounding of the reference's own
// evaluation, so a child is rejected here only if the reference rejects it.
//
// A direction component of exactly +-0 gives inv = +-inf. Then either every
// plane distance of that axis is NaN, which fminf/fmaxf drop (the axis does not
// constrain, as when the reference's origin is inside the slab), or the near
// distance is +inf / the far one -inf, which happens only for an origin outside
// the node's slab, where the reference rejects as well.
// Assumes scene coordinates and distances far from the ends of the float range
// (|x| in [2^-60, 2^60]).
struct RayF {
  float inv[3];
};
#ifndef PTAMD_PK_FMA
#define PTAMD_PK_FMA 0
#endif
typedef float float2v __attribute__((ext_vector_type(2)));
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
__device__ __forceinline__ RayF make_rayf(v3 d) {
  RayF r;
  r.inv[0] = __builtin_amdgcn_rcpf((float)d.x);
  r.inv[1] = __builtin_amdgcn_rcpf((float)d.y);
  r.inv[2] = __builtin_amdgcn_rcpf((float)d.z);
  return r;
}

// Per-lane traversal stack: the first STACK_LDS levels live in LDS (entry
// (level, lane) at lds[level * 256 + tid], conflict-free for any mix of levels),
// deeper levels spill to a per-thread strip of global memory. LDS level
// STACK_LDS is a scratch slot: a push that turns out not to be needed is
// redirected there instead of being branched around.
// (levels: a TravStack member, chosen per kernel from the LDS its occupancy leaves)
#ifndef PTAMD_STACK_EXTEND
#define PTAMD_STACK_EXTEND 14  // 5 workgroups per CU x 30 KiB
#endif
#ifndef PTAMD_STACK_SHADOW
#define PTAMD_STACK_SHADOW 10  // 7 workgroups per CU x 22 KiB
#endif
// Build-time tuning of the traversal kernels (defaults are the measured best): workgroups
// launched per resident workgroup slot by the grid-stride driver (REFILL == 0).
#ifndef PTAMD_GRID_MULT
#define PTAMD_GRID_MULT 8
#endif
#ifndef PTAMD_IN_ORDER
#define PTAMD_IN_ORDER 1  // first-hit walks take a node's children in the node's order (see node_step)
#endif
#ifndef PTAMD_NODE_STEPS
#define PTAMD_NODE_STEPS 2
#endif
#ifndef PTAMD_NODE_TAGS
#define PTAMD_NODE_TAGS 1  // end primitives are rejected in the box step of the node that holds them (see node_step)
#endif
// idle lanes at which a persistent wave refills (0: grid-stride driver), see trace_rays
#ifndef PTAMD_REFILL_EXTEND
#define PTAMD_REFILL_EXTEND 24
#endif
#ifndef PTAMD_REFILL_SHADOW
#define PTAMD_REFILL_SHADOW 40
#endif
#ifndef PTAMD_REFILL_TWO_S
#define PTAMD_REFILL_TWO_S 40  // the shadow-ray list of k_trace2
#endif
#ifndef PTAMD_WAVES_EXTEND
#define PTAMD_WAVES_EXTEND 5  // the closest-hit walk with lane refill needs ~96 registers; spilling costs more than a wave
#endif
#ifndef PTAMD_WAVES_SHADOW
#define PTAMD_WAVES_SHADOW 7  // first-hit walks: 72 registers, 4 of them spilled; 8 waves lose
#endif
#define TRAV_BOUNDS_N(n) __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(n, n)))
constexpr int TRAV_BLOCK = 256;
// (the LDS pointer carries its address space: as a generic pointer it made the gfx950
// backend of ROCm 7.2 emit an invalid null check in the two-list kernel)
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(3))) uint2 lds_uint2;
#else
typedef uint2 lds_uint2;  // the host pass only parses the device code
#endif
struct TravStack {
  lds_uint2 *lds;
  uint2 *glob;
  uint32_t gstride, gtid;
  int STACK_LDS;  // levels held in LDS
  // (the global strip is accessed with non-temporal loads and stores so that the optimiser cannot
  // merge the two branches into one access through a generic pointer)
  __device__ __forceinline__ void put(int level, uint2 e) const {
    if (level < STACK_LDS) {
      lds[level * TRAV_BLOCK + threadIdx.x] = e;
    } else {
      uint32_t *g = reinterpret_cast<uint32_t *>(glob + ((size_t)(level - STACK_LDS) * gstride + gtid));
      __builtin_nontemporal_store(e.x, g);
      __builtin_nontemporal_store(e.y, g + 1);
    }
  }
  __device__ __forceinline__ uint2 get(int level) const {
    if (level < STACK_LDS) return lds[level * TRAV_BLOCK + threadIdx.x];
    const uint32_t *g = reinterpret_cast<const uint32_t *>(glob + ((size_t)(level - STACK_LDS) * gstride + gtid));
    return make_uint2(__builtin_nontemporal_load(g), __builtin_nontemporal_load(g + 1));
  }
  // LDS-only forms; level STACK_LDS is the scratch slot
  __device__ __forceinline__ void put_lds(int level, uint2 e) const { lds[level * TRAV_BLOCK + threadIdx.x] = e; }
  __device__ __forceinline__ uint2 get_lds(int level) const { return lds[level * TRAV_BLOCK + threadIdx.x]; }
};

struct Hit {
  int32_t prim;  // -1: none
  bool certain;  // first-hit walks: false = `prim` is a primitive the FP32 test could not decide (and no other occludes)
};

__device__ __forceinline__ void cswap(float &ta, int32_t &ca, float &tb, int32_t &cb) {
  if (tb < ta) {
    const float t = ta; ta = tb; tb = t;
    const int32_t c = ca; ca = cb; cb = c;
  }
}

// The reference's FP64 test of primitive p (Triangle::test, src/scene/triangle.cpp:25-44;
// Sphere::test / Sphere::intersect, src/scene/sphere.cpp:11-72) in its operation order,
// WITHOUT the max_t clause: false when the reference rejects the primitive for any other
// reason; else t (u, v for a triangle) are the values it computes. For a sphere t is the
// first root that is not below min_t: the reference takes the first root inside
// [min_t, max_t], and when the first root at or beyond min_t lies beyond max_t so does the
// second, so comparing this t with max_t decides as the reference does.
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

// State of one ray's walk.
//
// The reference walks its binary tree with an explicit stack in a fixed order
// and keeps the first of equally distant hits (src/scene/bvh.cpp:164-200):
// its answer is the primitive with the smallest t below the ray's max_t, and
// among equal t the one it visits first. Primitives are stored here in exactly
// that visiting order, so "closest t, then lowest index" reproduces its answer
// for ANY traversal order and ANY tree over the same primitives; this walk takes
// the nearest child of a 4-wide node first and drops stacked subtrees that start
// beyond the current hit.
//
// Primitive tests run in FP32 with error bounds (prim32.h) and the walk keeps the
// candidate hit as (primitive, [lo_f, best_f]): an interval that contains the
// distance the reference computes for it. A primitive certainly missed costs nothing
// more; a certain hit whose interval lies entirely below the candidate's replaces it;
// only when the FP32 answer is not certain, or two intervals overlap, does the walk
// run the reference's FP64 test (exact_test) on the primitives in question and compare
// the exact distances (and indices). Without a hit the interval is the ray's own max_t
// rounded down and up. The exact t, u, v of the final hit are computed once, by the
// kernel that shades it (k_shade), with every lane busy.
struct Walk {
  v3 o, d;
  RayF rf;
  float lo_f, best_f;
  int32_t hit_prim;
  int sp;
  int32_t cur;
  float cur_t;        // where the ray enters cur's box (the box test's conservative near distance)
  uint32_t skip;  // primitive the ray starts on (VertexBuf::self_code) | tag of the light it ends on << 24
  // (a first-hit walk that has found a certain occluder says so with lo_f = NaN)
  __device__ __forceinline__ Hit result() const { return Hit{hit_prim, lo_f != lo_f}; }

  // (min_t is the same for every ray of a launch: the drivers hold it, and its float
  // neighbours, in scalar registers)
  __device__ __forceinline__ void begin(const DevScene &S, v3 o_, v3 d_, double tmax, uint32_t skip_) {
    o = o_; d = d_;
    skip = skip_;
    rf = make_rayf(d);
    lo_f = f_down(tmax);
    best_f = f_up(tmax);
    hit_prim = -1;
    sp = 0;
    cur = S.root_code;
    cur_t = 0.f;
  }

  // pop, skipping subtrees that begin beyond the current hit
  __device__ __forceinline__ void pop(const TravStack &stack) {
    cur = CHILD_EMPTY;
    if (sp <= stack.STACK_LDS) {
      while (sp > 0) {
        sp--;
        const uint2 e = stack.get_lds(sp);
        if (__uint_as_float(e.y) > best_f) continue;
        cur = (int32_t)e.x;
        cur_t = __uint_as_float(e.y);
        break;
      }
    } else {
      while (sp > 0) {
        sp--;
        const uint2 e = stack.get(sp);
        if (__uint_as_float(e.y) > best_f) continue;
        cur = (int32_t)e.x;
        cur_t = __uint_as_float(e.y);
        break;
      }
    }
  }

  // cur >= 0: test the node's four child boxes, descend into the nearest
  // in_order (the same for every lane of the wave): the accepted children are taken in the node's own
  // order instead of nearest first. A first-hit walk of a ray that nothing occludes, the common
  // shadow ray, visits every box the ray meets whatever the order, so ordering its children
  // (a quarter of a box step's instructions) buys nothing.
  template <bool COUNT>
  __device__ __forceinline__ void node_step(const DevScene &S, const TravStack &stack, uint32_t &n_nodes, float m0,
                                            bool in_order = false) {
    const float FINF = INFINITY;
    const uint4 *np = reinterpret_cast<const uint4 *>(S.nodes + cur);
    const uint4 w0 = np[0], w1 = np[1], w2 = np[2];
    const uint2 w3 = *reinterpret_cast<const uint2 *>(np + 3);
    if (COUNT) n_nodes += 4;
    const float org[3] = {__uint_as_float(w0.x), __uint_as_float(w0.y), __uint_as_float(w0.z)};
    const uint32_t qlo[3] = {w1.x, w1.y, w1.z}, qhi[3] = {w1.w, w2.x, w2.y};
    // float(o) and the slack |of inv| 2^-22 are recomputed here rather than carried by the walk:
    // six registers less per lane are worth more than the nine instructions
    const float of[3] = {(float)o.x, (float)o.y, (float)o.z};
    float A[3], Bn[3], Bf[3];
    uint32_t qn[3], qf[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const int e = (int)(w0.w << (24 - 8 * a)) >> 24;  // signed byte a
      A[a] = __builtin_ldexpf(rf.inv[a], e);
      const float B = (org[a] - of[a]) * rf.inv[a];
      const float slack = fabsf(of[a] * rf.inv[a]) * 2.38418579e-07f;  // 2^-22
      const float E = fmaf(fmaf(fabsf(A[a]), 255.f, fabsf(B)), 9.53674316e-07f /* 2^-20 */, slack);
      // the plane met first is the lower one when the ray runs up the axis, else the upper one
      const bool down = rf.inv[a] < 0.f;
      qn[a] = down ? qhi[a] : qlo[a];
      qf[a] = down ? qlo[a] : qhi[a];
      Bn[a] = B - E;
      Bf[a] = B + E;
    }
    int32_t c[4] = {(int32_t)w2.z, (int32_t)w2.w, (int32_t)w3.x, (int32_t)w3.y};
    float t[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      float tn[3], tf[3];
#pragma unroll
      for (int a = 0; a < 3; a++) {
#if PTAMD_PK_FMA
        // the two planes of an axis in one packed FMA (v_pk_fma_f32, the scale broadcast to both halves):
        // measured 1 % slower than the two scalar FMAs (12 x 5.1 cycles against 24 x 3.4 should win; the
        // register pairs it pins cost more), kept as a build option
        const float2v qq = {(float)((qn[a] >> (8 * k)) & 0xffu), (float)((qf[a] >> (8 * k)) & 0xffu)};
        const float2v aa = {A[a], A[a]}, bb = {Bn[a], Bf[a]};
        const float2v tt = __builtin_elementwise_fma(qq, aa, bb);
        tn[a] = tt.x;
        tf[a] = tt.y;
#else
        tn[a] = fmaf((float)((qn[a] >> (8 * k)) & 0xffu), A[a], Bn[a]);
        tf[a] = fmaf((float)((qf[a] >> (8 * k)) & 0xffu), A[a], Bf[a]);
#endif
      }
      // bbox.cpp:38 with NaNs (a direction component of zero; fmaxf / fminf drop them) accepting:
      // reject iff max(tn, min_t, 0) > min(tf, max_t)
      const float lo2 = fmaxf(fmaxf(fmaxf(tn[0], tn[1]), tn[2]), m0);
      const float hi2 = fminf(fminf(fminf(tf[0], tf[1]), tf[2]), best_f);
      t[k] = (c[k] != CHILD_EMPTY && !(lo2 > hi2)) ? lo2 : FINF;
    }
#if PTAMD_NODE_TAGS
    // A node some of whose children are end primitives of the light this shadow ray goes to (k_shade has
    // tested them): they are passed over here, before they cost the lane a trip each. The node's header
    // names the tag, so the other nodes pay one comparison.
    {
      const uint32_t ntag = w0.w >> 24;
      if (ntag != 0u && ntag == (skip >> PRIM_BITS & MAX_LIGHT_TAG)) {
#pragma unroll
        for (int k = 0; k < 4; k++)
          if ((~(uint32_t)c[k] >> PRIM_BITS) == ntag) t[k] = FINF;
      }
    }
#endif
    if (in_order) {
      const int a0 = t[0] < FINF, a1 = t[1] < FINF, a2 = t[2] < FINF, a3 = t[3] < FINF;
      const int n01 = a0 + a1, n012 = n01 + a2, nacc = n012 + a3;
      if (nacc == 0) { pop(stack); return; }
      // the first accepted child is next; each later one goes on the stack, above those before it
      const bool p1 = a1 && a0, p2 = a2 && n01 > 0, p3 = a3 && n012 > 0;
      if (sp + 3 <= stack.STACK_LDS) {
        stack.put_lds(p1 ? sp : stack.STACK_LDS, make_uint2((uint32_t)c[1], __float_as_uint(t[1])));
        stack.put_lds(p2 ? sp + n01 - 1 : stack.STACK_LDS, make_uint2((uint32_t)c[2], __float_as_uint(t[2])));
        stack.put_lds(p3 ? sp + n012 - 1 : stack.STACK_LDS, make_uint2((uint32_t)c[3], __float_as_uint(t[3])));
      } else {
        if (p1) stack.put(sp, make_uint2((uint32_t)c[1], __float_as_uint(t[1])));
        if (p2) stack.put(sp + n01 - 1, make_uint2((uint32_t)c[2], __float_as_uint(t[2])));
        if (p3) stack.put(sp + n012 - 1, make_uint2((uint32_t)c[3], __float_as_uint(t[3])));
      }
      sp += nacc - 1;
      cur = a0 ? c[0] : (a1 ? c[1] : (a2 ? c[2] : c[3]));
      cur_t = a0 ? t[0] : (a1 ? t[1] : (a2 ? t[2] : t[3]));
      return;
    }
    // (Measured and rejected: finding only the nearest child, 3 or 4 compare-and-swaps instead of the 5 of a
    // full sort, and pushing the others in slot order: the frame 0.1 to 0.9 % slower, k_extend 17 % slower.)
    cswap(t[0], c[0], t[1], c[1]);
    cswap(t[2], c[2], t[3], c[3]);
    cswap(t[0], c[0], t[2], c[2]);
    cswap(t[1], c[1], t[3], c[3]);
    cswap(t[1], c[1], t[2], c[2]);
    if (t[0] < FINF) {
      // the accepted children are a prefix of the sorted four; the farthest goes
      // deepest. nv = accepted among 1..3; c[k] lands on level sp + nv - k.
      const int nv = (t[1] < FINF) + (t[2] < FINF) + (t[3] < FINF);
      if (sp + 3 <= stack.STACK_LDS) {
#pragma unroll
        for (int k = 1; k < 4; k++)
          stack.put_lds(k <= nv ? sp + nv - k : stack.STACK_LDS, make_uint2((uint32_t)c[k], __float_as_uint(t[k])));
      } else {
#pragma unroll
        for (int k = 1; k < 4; k++)
          if (k <= nv) stack.put(sp + nv - k, make_uint2((uint32_t)c[k], __float_as_uint(t[k])));
      }
      sp += nv;
      cur = c[0];
      cur_t = t[0];
    } else {
      pop(stack);
    }
  }

  // cur is a primitive: FP32 test, FP64 only when that is not certain (see above); returns true
  // when the walk is over.
  // ANY: 0 closest hit, 1 first hit, 2 per ray (`first_hit`)
  // ray_end(): the ray's own FP64 max_t (asked for only on the FP64 path)
  // self_dmin(): VertexBuf::self_dmin of the vertex the ray leaves (asked for at the primitive it starts on)
  // min_lo <= min_t <= min_hi: its float neighbours
  //
  // Two questions come up on almost every ray and cannot be settled in FP32: the primitive the ray
  // starts on (t = 0 up to FP64 rounding, against min_t = 1e-11) and, for a shadow ray, the light's own
  // primitive at its far end (t = max_t up to rounding). The first is settled by the kernel that
  // made the ray (self_code / self_dmin). The second is left open: a first-hit walk remembers ONE
  // undecided primitive and goes on looking for a certain occluder; if none turns up, the kernel
  // that consumes the ray (k_resolve) runs the reference's test on that primitive, with every
  // lane busy. Only a second undecided primitive is tested in FP64 on the spot.
  template <int ANY, bool COUNT, class End, class Dmin>
  __device__ __forceinline__ bool prim_step(const DevScene &S, const TravStack &stack, uint32_t &n_tests, uint32_t &n_exact,
                                            double min_t, float min_lo, float min_hi, const End &ray_end,
                                            const Dmin &self_dmin, bool first_hit = false) {
    const uint32_t code = ~(uint32_t)cur;
    const int p = (int)(code & PRIM_NONE);
    // an end primitive of the light this shadow ray goes to: k_shade has tested it (and found no hit)
    if ((code >> PRIM_BITS) != 0u && (code >> PRIM_BITS) == (skip >> PRIM_BITS)) {
      pop(stack);
      return cur == CHILD_EMPTY;
    }
    if (COUNT) n_tests++;
    const uint4 *R = S.prim32 + (size_t)p * 3;
    const uint4 r0 = R[0], r1 = R[1], r2 = R[2];
    float dmin = INFINITY;
    if ((uint32_t)p == (skip & PRIM_NONE)) dmin = self_dmin();
    const double p1x = __hiloint2double((int)r0.y, (int)r0.x), p1y = __hiloint2double((int)r0.w, (int)r0.z),
                 p1z = __hiloint2double((int)r1.y, (int)r1.x);
    // the origin moved to where the ray enters the primitive's box (prim32.h)
    const double t0 = (double)cur_t;
    const float sv[3] = {(float)(fma(t0, d.x, o.x) - p1x), (float)(fma(t0, d.y, o.y) - p1y), (float)(fma(t0, d.z, o.z) - p1z)};
    const float big = S.coord_bound + fabsf(cur_t);
    const float df[3] = {(float)d.x, (float)d.y, (float)d.z};
    const float e1[3] = {__uint_as_float(r1.z), __uint_as_float(r1.w), __uint_as_float(r2.x)};
    const float e2[3] = {__uint_as_float(r2.y), __uint_as_float(r2.z), __uint_as_float(r2.w)};
    float lo = 0.f, hi = 0.f;
    int cls;
    if (e1[0] == e1[0]) cls = classify_tri32(sv, df, e1, e2, min_lo, min_hi, best_f, dmin, cur_t, big, lo, hi);
    else cls = classify_sphere32(sv, df, e1[1], min_lo, min_hi, best_f, cur_t, big, lo, hi);  // NaN marks a sphere; e1[1] = r^2
    const bool first = ANY == 1 || (ANY == 2 && first_hit);
    bool unsure = cls == PT32_UNSURE;
    if (cls == PT32_HIT) {
      if (hi < lo_f) {  // certainly closer than the candidate, or than the ray's own end
        hit_prim = p;
        if (first) {
          lo_f = NAN;
        } else {
          lo_f = lo;
          best_f = hi;
        }
      } else {
        unsure = true;
      }
    }
    if (unsure) {
      if (first && hit_prim < 0) {
        hit_prim = p;  // left to the consumer of the ray
      } else {
        if (COUNT) n_exact++;
        double t, u, v;
        if (exact_test(S, p, o, d, min_t, t, u, v)) {
          if (first) {
            if (t < ray_end()) { hit_prim = p; lo_f = NAN; }
          } else if (hit_prim < 0) {
            // (bvh.cpp:187: a hit counts when strictly closer than the ray's end)
            if (t < ray_end()) { hit_prim = p; lo_f = f_down(t); best_f = f_up(t); }
          } else {
            double tc, uc, vc;
            exact_test(S, hit_prim, o, d, min_t, tc, uc, vc);
            if (t < tc || (t == tc && p < hit_prim)) hit_prim = p;
            else t = tc;
            lo_f = f_down(t);
            best_f = f_up(t);
          }
        }
      }
    }
    if (first && lo_f != lo_f) return true;
    pop(stack);
    return cur == CHILD_EMPTY;
  }

  // One trip of an active lane; true when its walk is over: PTAMD_NODE_STEPS box steps, then a
  // primitive test for the lanes at a primitive (which includes lanes a box step has just brought
  // there). Measured and rejected, with the FP32 primitive tests in place: a primitive test after
  // every box step (4 to 9 % slower); putting off the second box step or the primitive test when
  // fewer than 16 to 32 lanes want it (a step costs the wave the same however few lanes run it):
  // within 1 % either way on CBbunny, 1 to 7 % slower on the 100k-triangle scene.
  template <int ANY, bool COUNT, class End, class Dmin>
  __device__ __forceinline__ bool trip(const DevScene &S, const TravStack &stack, uint32_t &n_nodes, uint32_t &n_tests,
                                       uint32_t &n_exact, float m0, double min_t, float min_lo, float min_hi,
                                       const End &ray_end, const Dmin &self_dmin, bool first_hit = false) {
    // (wave-uniform: every lane that is here traces a first-hit ray)
    const bool in_order = PTAMD_IN_ORDER && (ANY == 1 || (ANY == 2 && __ballot(!first_hit) == 0ull));
#pragma unroll
    for (int k = 0; k < PTAMD_NODE_STEPS; k++)
      if (cur >= 0) node_step<COUNT>(S, stack, n_nodes, m0, in_order);
    if (cur == CHILD_EMPTY) return true;
    if (cur < 0) return prim_step<ANY, COUNT>(S, stack, n_tests, n_exact, min_t, min_lo, min_hi, ray_end, self_dmin, first_hit);
    return false;
  }
};

// Traces a list of rays. Policy: load(i, o, d, max_t) reads ray i, min_t_all() is the min_t of all of them;
// store(i, hit, found) writes its result. One trip of the walk: PTAMD_NODE_STEPS
// box steps for the lanes at a node, then a primitive test for the lanes at a
// primitive (which includes lanes a box step just brought there). Running the two
// kinds of step one after the other, instead of one kind per trip, keeps more
// lanes busy per instruction.
//
// Two drivers:
//  REFILL == 0: one ray per lane per grid-stride step; the wave waits for its
//    slowest ray (which takes about three times the average number of trips).
//  REFILL > 0: persistent waves. A wave claims rays in chunks of GRAB from a
//    counter shared by the launch, and once REFILL of its lanes are idle they are
//    given the next rays. Which lane traces a ray does not affect where its
//    result is stored. Measured on MI355X: -18 % for shadow rays and -19 % for
//    bounce rays PROVIDED the kernel does not spill: the bookkeeping costs about
//    ten registers, and with spills in the loop the same driver was 30 to 90 %
//    slower than the grid-stride one. Hence closest-hit walks run at 5 waves per
//    SIMD (no spills) and first-hit walks, which carry less state, at 7.
//
// Also measured and rejected: claims that shrink towards the end of the list (guided
// self-scheduling, so that no wave holds unstarted rays while others idle): 7 % slower, a
// claim stalls the whole wave for an atomic's round trip and the end of a launch is bound by
// its longest walks, not by imbalance; sorting each workgroup's 256 rays by direction
// octant (no gain); one box step or one primitive test per trip (if / else),
// or box steps until every lane has a primitive (while-while): 5 % and 25 %
// slower; cutting the list into one contiguous band per XCD so that each XCD's
// L2 holds only its band's share of the scene: 12 % slower (the bands differ in
// cost and the XCDs with easy bands idle).
#ifndef PTAMD_GRAB
#define PTAMD_GRAB 256
#endif
// How a persistent wave gets rays from a list of n. Its FIRST chunk is fixed: wave w owns
// [w g, (w + 1) g). Only what lies beyond waves x g is handed out through the shared counter.
// Thousands of waves adding to one address are served one after the other (~20 ns each): with
// every wave's first claim going through the counter a launch of 64 rays took 105 us (now 52),
// and every launch began with such a queue.
struct Claims {
  uint32_t n, g, dyn0;  // list length, chunk size, first ray handed out through the counter
  uint32_t *counter;
  // own_chunks = false: everything goes through the counter (a list the waves reach one by one, when
  // they run out of another: a wave held up by long walks would sit on its own chunk)
  __device__ __forceinline__ void init(uint32_t n_, uint32_t *counter_, uint32_t waves, bool own_chunks) {
    n = n_;
    counter = counter_;
    const uint32_t per_wave = n / waves;
    // PTAMD_GRAB, less when the list is too short to give every wave that many
    g = per_wave >= PTAMD_GRAB ? PTAMD_GRAB : (per_wave < 64u ? 64u : (per_wave & ~63u));
    dyn0 = own_chunks ? waves * g : 0u;
  }
  // the wave's own first chunk; false if the list ends before it
  __device__ __forceinline__ bool first(uint32_t wave, uint32_t &next, uint32_t &end) const {
    next = wave * g;
    end = next + g < n ? next + g : n;
    return next < n;
  }
  // A further chunk; false when the list is handed out. Call with all lanes active; the result is
  // wave-uniform and lives in scalar registers. (Looking at the counter with a plain load before
  // adding to it, to spare the adds that come back empty at the end of a list, was measured: the
  // loads queue up behind the adds on the same line, +6 % frame time.)
  __device__ __forceinline__ bool more(int lane, uint32_t &next, uint32_t &end) const {
    if (dyn0 >= n) return false;  // the waves' own chunks cover the whole list
    uint32_t base = 0;
    if (lane == 0) base = dyn0 + atomicAdd(counter, g);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    if (base >= n) return false;
    next = base;
    end = base + g < n ? base + g : n;
    return true;
  }
};

template <int ANY, bool COUNT, int REFILL, class Policy>
__device__ __forceinline__ void trace_rays(const DevScene &S, const Policy &pol, uint32_t n, uint32_t *grab,
                                           const TravStack &stack, uint32_t &n_nodes, uint32_t &n_tests, uint32_t &n_exact) {
  Walk w;
  const double min_t = pol.min_t_all();
  const float min_lo = f_down(min_t), min_hi = f_up(min_t);
  const float m0 = fmaxf(min_lo, 0.f);
  if (REFILL == 0) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
      v3 o, d;
      double tmax;
      uint32_t self;
      if (!pol.load(i, o, d, tmax, self)) continue;
      w.begin(S, o, d, tmax, self);
      while (!w.template trip<ANY, COUNT>(S, stack, n_nodes, n_tests, n_exact, m0, min_t, min_lo, min_hi,
                                          [&]() { return pol.ray_end(i); }, [&]() { return pol.self_dmin(i); })) {
      }
      pol.store(i, w.result(), w.hit_prim >= 0);
    }
    return;
  }
  const int lane = threadIdx.x & 63;
  const unsigned long long below = (1ull << lane) - 1ull;
  Claims cl;
  cl.init(n, grab, gridDim.x * (TRAV_BLOCK / 64u), true);
  uint32_t next = 0, end = 0, idx = 0;
  bool exhausted = false, active = false;
  bool unstored = false;  // the lane's finished walk has yet to write its result (see trace_two_lists)
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (TRAV_BLOCK / 64u) + (threadIdx.x >> 6)));
  if (!cl.first(wave, next, end)) { exhausted = true; next = end = 0; }
  w.cur = CHILD_EMPTY;
  w.sp = 0;
  for (;;) {
    const unsigned long long idle = __ballot(!active);
    const int n_idle = __popcll(idle);
    if (n_idle >= REFILL && !exhausted) {
      if (next >= end && !cl.more(lane, next, end)) { exhausted = true; next = end = 0; }
      if (unstored) {
        pol.store(idx, w.result(), w.hit_prim >= 0);
        unstored = false;
      }
      if (next < end) {
        // idle lanes take the next rays in lane order
        const uint32_t mine = next + (uint32_t)__popcll(idle & below);
        if (!active && mine < end) {
          idx = mine;
          v3 o, d;
          double tmax;
          uint32_t self;
          if (pol.load(idx, o, d, tmax, self)) {
            w.begin(S, o, d, tmax, self);
            active = true;
          }
        }
        next = next + (uint32_t)n_idle < end ? next + (uint32_t)n_idle : end;
      }
    }
    if (!__ballot(active)) {
      if (exhausted) break;  // every wave gets here: the list is finite and claimed monotonically
      continue;
    }
    if (active) {
      const bool done = w.template trip<ANY, COUNT>(S, stack, n_nodes, n_tests, n_exact, m0, min_t, min_lo, min_hi,
                                                    [&]() { return pol.ray_end(idx); }, [&]() { return pol.self_dmin(idx); });
      if (done) {
        active = false;
        unstored = true;
      }
    }
  }
  if (unstored) pol.store(idx, w.result(), w.hit_prim >= 0);
}

// Two ray lists in one launch of persistent waves: list E (closest-hit rays, policy
// pe) is claimed first, then list S (policy ps; first-hit rays unless CLOSEST_S).
// Every traversal launch ends with a drain phase in which each wave finishes the
// rays it still holds while the rest of the GPU idles: 0.1 to 0.2 ms for closest-hit
// rays on a 100k-triangle scene, however short the list (measured: a wave whose list
// has run out still holds rays on about 18 lanes and needs some 22 more trips, each a
// chain of dependent fetches and instructions that takes 3.6 us whether or not other
// waves share the SIMD). The bounce rays of depth d + 1 and the shadow rays of depth d
// are both ready once depth d has been shaded, so they are traced together: the shadow
// rays fill the lanes the last bounce-ray walks leave idle, and a depth costs one such
// phase instead of two.
template <bool CLOSEST_S, bool COUNT, int REFILL, class PolE, class PolS>
__device__ __forceinline__ void trace_two_lists(const DevScene &S, const PolE &pe, uint32_t n_e, uint32_t *grab_e,
                                                const PolS &ps, uint32_t n_s, uint32_t *grab_s, const TravStack &stack,
                                                uint32_t &n_nodes, uint32_t &n_tests, uint32_t &n_exact) {
  Walk w;
  const double min_t = pe.min_t_all();  // == ps.min_t_all(): both kinds of ray leave a surface
  const float min_lo = f_down(min_t), min_hi = f_up(min_t);
  const float m0 = fmaxf(min_lo, 0.f);
  const uint32_t waves = gridDim.x * (TRAV_BLOCK / 64u);
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (TRAV_BLOCK / 64u) + (threadIdx.x >> 6)));
  const int lane = threadIdx.x & 63;
  const unsigned long long below = (1ull << lane) - 1ull;
  Claims cl;  // of the list being claimed
  cl.init(n_e, grab_e, waves, true);
  uint32_t next = 0, end = 0, idx = 0;
  int phase = 0;  // list being claimed: 0 = E, 1 = S, 2 = both are handed out (wave-uniform)
  bool own_chunk = true;  // the wave has yet to take its own first chunk of the list of `phase`
  bool active = false, from_s = false;
  bool unstored = false;  // the lane's finished walk has yet to write its result
  w.cur = CHILD_EMPTY;
  w.sp = 0;
  for (;;) {
    const unsigned long long idle = __ballot(!active);
    const int n_idle = __popcll(idle);
    // (short first-hit walks pay relatively more for a refill pass than long closest-hit walks:
    // their list is refilled in bigger groups)
    if (n_idle >= (phase == 0 ? REFILL : PTAMD_REFILL_TWO_S) && phase < 2) {
      if (next >= end) {
        const bool got = own_chunk ? cl.first(wave, next, end) : cl.more(lane, next, end);
        own_chunk = false;
        if (!got) {  // this list is handed out: on to the next one
          phase++;
          cl.init(n_s, grab_s, waves, false);
          next = end = 0;
          continue;
        }
      }
      // Results are written here, by all the lanes that have finished since the last refill at once,
      // not by the few lanes that finish in any one trip.
      if (unstored) {
        if (from_s) ps.store(idx, w.result(), w.hit_prim >= 0);
        else pe.store(idx, w.result(), w.hit_prim >= 0);
        unstored = false;
      }
      const uint32_t mine = next + (uint32_t)__popcll(idle & below);
      if (!active && mine < end) {
        idx = mine;
        v3 o, d;
        double tmax;
        uint32_t self;
        const bool have = phase == 0 ? pe.load(idx, o, d, tmax, self) : ps.load(idx, o, d, tmax, self);
        if (have) {
          w.begin(S, o, d, tmax, self);
          active = true;
          from_s = phase == 1;
        }
      }
      next = next + (uint32_t)n_idle < end ? next + (uint32_t)n_idle : end;
    }
    if (!__ballot(active)) {
      if (phase >= 2) break;  // every wave gets here: both lists are finite and claimed monotonically
      continue;
    }
    if (active) {
      const bool done = w.template trip<2, COUNT>(S, stack, n_nodes, n_tests, n_exact, m0, min_t, min_lo, min_hi,
                                                  [&]() { return from_s ? ps.ray_end(idx) : pe.ray_end(idx); },
                                                  [&]() { return from_s ? ps.self_dmin(idx) : pe.self_dmin(idx); },
                                                  !CLOSEST_S && from_s);
      if (done) {
        active = false;
        unstored = true;
      }
    }
  }
  if (unstored) {
    if (from_s) ps.store(idx, w.result(), w.hit_prim >= 0);
    else pe.store(idx, w.result(), w.hit_prim >= 0);
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

// ------------------------------------------------------------------ kernels
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

// end of one depth: fold the stage's counts into the statistics, make the next
// queue current
__global__ void k_advance(Ctl *ctl) {
  ctl->st_rays += (unsigned long long)ctl->n_cur + ctl->n_shadow;
  ctl->st_shadow += ctl->n_shadow;
  ctl->n_cur = ctl->n_next;
  ctl->n_next = 0;
  ctl->n_shadow = 0;
  ctl->grab[0] = ctl->grab[1] = ctl->grab[2] = 0;
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
__global__ void k_generate(FrameParams fp, DevScene S, PixView pv, uint32_t p0, Ctl *ctl, RoundPlan rp, RayQ q, PathBuf pb) {
  const uint32_t stride = gridDim.x * blockDim.x;
  const uint32_t n_paths = ctl->n_cur;
  const int32_t ns = rp.ns;
  unsigned long long dead = 0;
  for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n_paths; k += stride) {
    const int32_t pix = pv.at(p0 + k / (uint32_t)ns);
    int32_t first;
    const int32_t planned = rp.planned(rp.state + (size_t)pix * 6, first);
    const int32_t sk = (int32_t)(k % (uint32_t)ns);
    q.path[k] = (int32_t)k;
    pb.rad[0][k] = 0; pb.rad[1][k] = 0; pb.rad[2][k] = 0;
    if (pb.pend[0]) { pb.pend[0][k] = 0; pb.pend[1][k] = 0; pb.pend[2][k] = 0; }
    if (sk >= planned) {
      // a slot this round does not use: a zero direction tells k_extend to skip it
      q.d[0][k] = 0; q.d[1][k] = 0; q.d[2][k] = 0;
      dead++;
      continue;
    }
    const int32_t s = first + sk;
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
    q.o[0][k] = o.x; q.o[1][k] = o.y; q.o[2][k] = o.z;
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
    q.o[0][k] = o.x; q.o[1][k] = o.y; q.o[2][k] = o.z;
    q.d[0][k] = d.x; q.d[1][k] = d.y; q.d[2][k] = d.z;
  }
}

// hit distances of traced camera rays (ptamd_trace_primary): the walk names the primitive, the
// reference's test gives its t
__global__ void k_primary_t(DevScene S, RayQ q, const int32_t *__restrict__ prim, uint32_t n, double min_t,
                            double *__restrict__ t_out) {
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) {
    double t = -1.0, u, v;
    if (prim[k] >= 0)
      exact_test(S, prim[k], V(q.o[0][k], q.o[1][k], q.o[2][k]), V(q.d[0][k], q.d[1][k], q.d[2][k]), min_t, t, u, v);
    t_out[k] = t;
  }
}

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
    o = V(q.o[0][j], q.o[1][j], q.o[2][j]);
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
             uint2 *gstack, const int32_t *__restrict__ self_code, const float *__restrict__ self_dmin) {
  extern __shared__ uint2 lds_stack[];
  const TravStack stack{(lds_uint2 *)lds_stack, gstack, gridDim.x * blockDim.x, blockIdx.x * blockDim.x + threadIdx.x, PTAMD_STACK_EXTEND};
  uint32_t cn = 0, ct = 0, cx = 0;
  const ExtendPolicy pol{q, list, hb, min_t, max_t, self_code, self_dmin};
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
  // the tag of the light that sample q of a vertex goes to (k_shade's loop over the lights, backwards;
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
  const ExtendPolicy pe{q, list, hb, PT_EPS_D, (double)INFINITY, vb.self_code, vb.self_dmin};
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

struct Compact {
  uint8_t *pflag;        // [cap] entry pushes a bounce ray
  uint32_t *cnt_next;    // [cap/256 + 1] bounce rays per 256-entry block, then their exclusive scan
  uint32_t *cnt_shadow;  // [cap/256 + 1] shadow rays per 256-entry block, then their exclusive scan
};

// One thread per entry of the current list at path depth `depth` (1 = the camera
// ray's hit). Restates at_least_one_bounce_radiance (pathtracer.cpp:160-201)
// iteratively: the path carries its throughput T, every shaded vertex adds
// T * direct light.
// (131 VGPRs, 3 waves per SIMD. Forcing 4 waves (127 registers, 2 spilled) gains 0.5 % at
// most: the kernel is bound by its FP64 arithmetic and the bytes it moves, not by latency.)
template <bool HEMI>
__global__ void __launch_bounds__(256)
    k_shade(FrameParams fp, DevScene S, RayQ q, const int32_t *__restrict__ list, HitBuf hb, RayQ qn, ShadowQ sq,
            VertexBuf vb, PathBuf pb, Compact cp, Ctl *ctl, int32_t depth, PixView pv, uint32_t p0, RoundPlan rp) {
  __shared__ uint32_t wsum[2][4];
  const uint32_t n = ctl->n_cur;
  const int nls = fp.nls;
  const int ns = rp.ns;
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
        const int32_t info = S.prim_info[prim];
        const int32_t mat = info >> 1;
        nee_mat = mat;
        const int32_t kind = S.mat_kind[mat];
        const v3 ro = V(q.o[0][j], q.o[1][j], q.o[2][j]);
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
          const uint32_t k = (uint32_t)path;
          rng_pix = (uint32_t)pv.at(p0 + k / (uint32_t)ns);
          rng_s = (uint32_t)(rp.first_of(rp.state + (size_t)rng_pix * 6) + (int32_t)(k % (uint32_t)ns));
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
#ifndef PTAMD_LIGHT_LOOP
#define PTAMD_LIGHT_LOOP 0  // workgroups per CU of a grid-stride k_light (0: one workgroup per 256 slots of the launch's upper bound)
#endif
#ifndef PTAMD_LIGHT_WAVES
#define PTAMD_LIGHT_WAVES 0  // waves per SIMD k_light is compiled for (0: the compiler's choice)
#endif
#if PTAMD_LIGHT_WAVES
#define LIGHT_BOUNDS __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PTAMD_LIGHT_WAVES, PTAMD_LIGHT_WAVES)))
#else
#define LIGHT_BOUNDS __launch_bounds__(256)
#endif
__global__ void LIGHT_BOUNDS
    k_light(FrameParams fp, DevScene S, VertexBuf vb, ShadowQ sq, Compact cp, Ctl *ctl, FastDiv by_nls, uint32_t rng_base) {
  __shared__ uint32_t wsum[4];
  const uint32_t n = ctl->n_cur;
  const uint32_t nls = (uint32_t)fp.nls;
  const uint32_t n_slots = n * nls;  // < 2^31 (ptamd_trace_list)
#if PTAMD_LIGHT_LOOP
  for (uint32_t blk = blockIdx.x; blk * 256u < n_slots; blk += gridDim.x) {
#else
  const uint32_t blk = blockIdx.x;
  {
#endif
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
      // settles the triangle of the quad the sample point does not lie on: 9 % slower.)
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
#if PTAMD_LIGHT_LOOP
  __syncthreads();
#endif
  }
}

// Exclusive scan of the two per-block count arrays (one workgroup); the totals
// become the sizes of the next ray list and of the shadow-ray list.
__global__ void __launch_bounds__(1024) k_scan2(Compact cp, Ctl *ctl) {
  // per pass: 8 counts per thread summed serially, a shuffle scan inside each wave, the 16 wave
  // totals through LDS (three barriers per 8192 counts)
  __shared__ uint32_t wtot[2][16];
  __shared__ uint32_t carry[2];
  constexpr int IT = 8;
  const uint32_t n_blocks = (ctl->n_cur + 255u) >> 8;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (threadIdx.x < 2) carry[threadIdx.x] = 0;
  __syncthreads();
  for (uint32_t base = 0; base < n_blocks; base += 1024 * IT) {
    const uint32_t i0 = base + threadIdx.x * IT;
    uint32_t a[IT], b[IT], sa = 0, sb = 0;
#pragma unroll
    for (int k = 0; k < IT; k++) {
      const bool in = i0 + k < n_blocks;
      a[k] = in ? cp.cnt_next[i0 + k] : 0u;
      b[k] = in ? cp.cnt_shadow[i0 + k] : 0u;
      sa += a[k];
      sb += b[k];
    }
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
#pragma unroll
    for (int k = 0; k < IT; k++) {
      if (i0 + k < n_blocks) {
        cp.cnt_next[i0 + k] = ea;
        cp.cnt_shadow[i0 + k] = eb;
      }
      ea += a[k];
      eb += b[k];
    }
    __syncthreads();
    if (threadIdx.x == 1023) { carry[0] = pa + xa; carry[1] = pb + xb; }
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
__global__ void k_resolve(FrameParams fp, DevScene S, RayQ q, const int32_t *__restrict__ list, VertexBuf vb,
                          const SRay *__restrict__ rays, PathBuf pb, Ctl *ctl, int32_t depth) {
  const uint32_t n = ctl->n_cur;
  const int nls = fp.nls;
  unsigned long long verts = 0;
  {  // one workgroup per 256 entries, as k_shade
    const uint32_t base_i = blockIdx.x * blockDim.x;
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
            Contrib cc[4];
#pragma unroll
            for (int qq = 0; qq < 4; qq++) cc[qq] = vb.contrib[b + q4 + qq];  // whether visible or not
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
  block_add(&ctl->st_vertices, verts);
}

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

// Folds the round's finished paths into the per-pixel sums, in sample order:
// the loop of raytrace_pixel (pathtracer.cpp:249-267) over radiances that are
// already computed. A round may hold several batches of a pixel, traced ahead of
// their adaptive tests; the test runs here between the batches, and what was
// traced past a pixel's stop is dropped. state: [pixel*6] = s1, s2, sum.rgb, n.
// (One thread per pixel: its loads are not coalesced, but a wave per pixel with coalesced
// loads and the ordered sum passed lane to lane was measured 2.5 times slower; the serial
// additions, not the loads, are what a pixel costs.)
__global__ void k_accumulate(PixView pv, uint32_t p0, Ctl *ctl, RoundPlan rp, PathBuf pb, double *__restrict__ state) {
  const uint32_t stride = gridDim.x * blockDim.x;
  const uint32_t n_pixels = ctl->n_pix;
  unsigned long long kept = 0;
  for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < n_pixels; p += stride) {
    double *st = state + (size_t)pv.at(p0 + p) * 6;
    int32_t first;
    const int32_t planned = rp.planned(st, first);
    double acc[2] = {st[0], st[1]};
    v3 sum = V(st[2], st[3], st[4]);
    const size_t b = (size_t)p * (uint32_t)rp.ns;
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

// Ordered stream compaction of the still-active pixels, three small kernels:
// per-block counts, scan of the counts, scatter.
__global__ void k_filter_count(PixView pv, int32_t n_done, double tol, int32_t batch, int32_t ns_aa,
                               const double *__restrict__ state, uint32_t *__restrict__ block_count) {
  __shared__ uint32_t wave_cnt[4];
  const uint32_t n = pv.count();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool keep = i < n && pixel_survives(state + (size_t)pv.at(i) * 6, n_done, tol, batch, ns_aa);
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
__global__ void k_filter_scatter(PixView pv, int32_t n_done, double tol, int32_t batch, int32_t ns_aa,
                                 const double *__restrict__ state, const uint32_t *__restrict__ block_offset,
                                 int32_t *__restrict__ out) {
  __shared__ uint32_t wave_cnt[4];
  const uint32_t n = pv.count();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool keep = i < n && pixel_survives(state + (size_t)pv.at(i) * 6, n_done, tol, batch, ns_aa);
  const unsigned long long m = __ballot(keep);
  const int lane = __lane_id(), wv = threadIdx.x >> 6;
  if (lane == 0) wave_cnt[wv] = (uint32_t)__popcll(m);
  __syncthreads();
  uint32_t off = block_offset[blockIdx.x];
  for (int k = 0; k < wv; k++) off += wave_cnt[k];
  if (keep) out[off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = pv.at(i);
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

// State rows of a shard's pixels, gathered into a dense block (what a rank sends
// after a round) and scattered back from the blocks of all ranks: entry i of the
// whole list was traced by rank i % world and is row i / world of its block.
__global__ void k_pack_rows(PixView pv, const double *__restrict__ state, double *__restrict__ rows) {
  const uint32_t n = pv.count();
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n * 6u; t += stride) {
    const uint32_t k = t / 6u, f = t % 6u;
    rows[t] = state[(size_t)pv.at(k) * 6 + f];
  }
}
__global__ void k_unpack_rows(PixView pv, uint32_t world, uint32_t skip_rank, const double *__restrict__ rows, uint32_t m,
                              double *__restrict__ state) {
  const uint32_t n = pv.count();
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n * 6u; t += stride) {
    const uint32_t i = t / 6u, f = t % 6u;
    const uint32_t r = i % world, k = i / world;
    if (r != skip_rank) state[(size_t)pv.at(i) * 6 + f] = rows[((size_t)r * m + k) * 6 + f];
  }
}

// Block counts of ptamd_filter_list_static: the one entry point that allocates nothing. Each GPU
// the library runs on holds its own copy of this array, at its own address, until the process
// ends; host code reaches it only through the pointer hipGetSymbolAddress returned on that GPU.
constexpr int64_t FILTER_STATIC_MAX_PIXELS = PTAMD_FILTER_STATIC_MAX_PIXELS;
constexpr size_t FILTER_STATIC_BLOCKS = (size_t)(FILTER_STATIC_MAX_PIXELS / 256);
static __device__ __align__(256) uint32_t scratch[FILTER_STATIC_BLOCKS];

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

// Frame-level buffers of ptamd_render / ptamd_render_list (see ensure_frame_scratch)
struct FrameScratch {
  int32_t *d_list[3] = {nullptr, nullptr, nullptr};  // two survivor lists in turn; [2]: ptamd_render's first list
  int64_t cap_list = 0;
  double *d_state = nullptr, *d_hdr = nullptr;
  int32_t *d_count = nullptr;
  int64_t cap_pix = 0;
  int64_t *d_n = nullptr, *h_n = nullptr;  // two list lengths on the device, their pinned host copies
  hipEvent_t ev[2] = {nullptr, nullptr};
};

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
  bool clear_ctl = false;  // the next ptamd_trace_list zeroes the device-side counters first
  double plan_margin = 1.5;  // RoundPlan::margin (PTAMD_PLAN_MARGIN in the environment: experiments)
  bool trace_two = true;  // bounce rays of depth d + 1 and shadow rays of depth d in one launch (PTAMD_TRACE_TWO=0: apart)
  uint32_t fuse_below = 16u << 20;  // ... when the two lists together are shorter than this (PTAMD_FUSE_BELOW)
  int64_t max_paths = 16 << 20;  // paths per launch: large launches amortise the drain of the slowest waves
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
  FrameScratch fs;
  double scene_max_coord = 0;
  uint32_t *static_block_count = nullptr;  // this GPU's copy of `scratch` (ptamd_filter_list_static)
  uint2 *gstack = nullptr;  // overflow levels of the per-lane traversal stacks
  size_t gstack_bytes = 0;
  // profiling
  bool prof_events = false, prof_counters = false;
  uint32_t event_mask = 0xffffffffu;  // kernel classes whose launches are bracketed by events
  struct Ev { int cls; hipEvent_t a, b; };
  std::vector<Ev> events;
  std::vector<hipEvent_t> event_pool;  // recycled: creating an event per launch costs more than the launch
  hipEvent_t take_event() {
    hipEvent_t e = nullptr;
    if (!event_pool.empty()) { e = event_pool.back(); event_pool.pop_back(); }
    else (void)hipEventCreate(&e);
    return e;
  }
  ptamd_stats stats{};
};

namespace {

void free_frame_scratch(ptamd_context *c);
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
  bool timed = false;
  void begin(int cls) {
    c->stats.launches[cls]++;
    timed = c->prof_events && ((c->event_mask >> cls) & 1u);
    if (timed) {
      ptamd_context::Ev e;
      e.cls = cls;
      e.a = c->take_event();
      e.b = c->take_event();
      (void)hipEventRecord(e.a, s);
      c->events.push_back(e);
    }
  }
  void end() {
    if (timed) (void)hipEventRecord(c->events.back().b, s);
  }
};

#ifdef PTAMD_WAVELOG
struct WaveLogEntry { int kind; uint32_t base, waves; };
static std::vector<WaveLogEntry> g_wavelog_launches;
static uint32_t g_wavelog_used = 0;
// names the range of the launch about to be queued (kind: 0 k_extend, 1 k_trace2, 2 k_shadow)
static void wavelog_next(ptamd_context *c, int kind, int grid) {
  const uint32_t waves = (uint32_t)grid * (TRAV_BLOCK / 64u);
  if (!c->S.wavelog) {
    c->S.wavelog_cap = 24u << 20;
    if (hipMalloc((void **)&c->S.wavelog, (size_t)c->S.wavelog_cap * 16) != hipSuccess) { c->S.wavelog = nullptr; c->S.wavelog_cap = 0; }
    else (void)hipMemset(c->S.wavelog, 0, (size_t)c->S.wavelog_cap * 16);
  }
  c->S.wavelog_base = g_wavelog_used;
  if (g_wavelog_used + waves <= c->S.wavelog_cap) {
    g_wavelog_launches.push_back({kind, g_wavelog_used, waves});
    g_wavelog_used += waves;
  } else {
    c->S.wavelog_base = c->S.wavelog_cap;  // full: nothing more is written
  }
}
// at ptamd_destroy: "kind base waves" per launch, then start / end ticks per wave, to $PTAMD_WAVELOG
static void wavelog_dump(ptamd_context *c) {
  const char *path = getenv("PTAMD_WAVELOG");
  if (!path || !c->S.wavelog) return;
  std::vector<unsigned long long> h((size_t)g_wavelog_used * 2);
  if (hipMemcpy(h.data(), c->S.wavelog, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return;
  FILE *f = fopen(path, "wb");
  if (!f) return;
  const uint32_t n = (uint32_t)g_wavelog_launches.size();
  fwrite(&n, 4, 1, f);
  for (const auto &e : g_wavelog_launches) { const uint32_t r[3] = {(uint32_t)e.kind, e.base, e.waves}; fwrite(r, 4, 3, f); }
  fwrite(h.data(), 8, h.size(), f);
  fclose(f);
}
#define PTAMD_WAVELOG_NEXT(c, kind, grid) wavelog_next(c, kind, grid)
#else
#define PTAMD_WAVELOG_NEXT(c, kind, grid)
#endif
void collect_events(ptamd_context *c) {
  for (auto &e : c->events) {
    float ms = 0;
    if (hipEventSynchronize(e.b) == hipSuccess && hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) c->stats.kernel_ms[e.cls] += ms;
    c->event_pool.push_back(e.a);
    c->event_pool.push_back(e.b);
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
  size_t off_q[2][10], off_hit[4], off_sq[5], off_vb[8], off_pb[6], off_list[2], off_cp[3], off_nee[4];
  for (int k = 0; k < 2; k++) {
    off_q[k][0] = take(P * 4);
    for (int j = 1; j < 10; j++) off_q[k][j] = take(P * 8);
  }
  off_hit[0] = take(P * 4);
  off_hit[1] = take(P * 8);
  off_hit[2] = take(P * 4);
  off_hit[3] = take(P * 4);
  off_sq[0] = take(PN * 4);
  off_sq[1] = take(PN * sizeof(SRay));
  off_vb[3] = take(P);
  off_vb[4] = take(PN * sizeof(Contrib));
  off_vb[7] = take(PN + 4);
  off_vb[5] = take(PN * 4);
  for (int j = 0; j < 3; j++) off_nee[j] = take(P * 8);
  off_nee[3] = take(P * 16);
  for (int j = 0; j < 6; j++) off_pb[j] = take(P * 8);
  for (int k = 0; k < 2; k++) off_list[k] = take(P * 4);
  off_cp[0] = take(P);
  off_cp[1] = take((P / 256 + 2) * 4);
  off_cp[2] = take((P / 256 + 2) * 4);
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
  c->vb.self_code = (int32_t *)(b + off_hit[2]);
  c->vb.self_dmin = (float *)(b + off_hit[3]);
  c->vb.pend_prim = (int32_t *)(b + off_vb[5]);
  c->sq.slot = (int32_t *)(b + off_sq[0]);
  c->sq.ray = (SRay *)(b + off_sq[1]);
  for (int j = 0; j < 3; j++) c->vb.hp[j] = nullptr;  // set per depth: the next ray queue's origin arrays
  c->vb.flag = (uint8_t *)(b + off_vb[3]);
  c->vb.contrib = (Contrib *)(b + off_vb[4]);
  c->vb.vis = (uint8_t *)(b + off_vb[7]);
  for (int j = 0; j < 3; j++) c->vb.nz[j] = (double *)(b + off_nee[j]);
  c->vb.nee = (uint4 *)(b + off_nee[3]);
  for (int j = 0; j < 3; j++) {
    c->pb.rad[j] = (double *)(b + off_pb[j]);
    c->pb.pend[j] = (double *)(b + off_pb[3 + j]);
  }
  for (int k = 0; k < 2; k++) c->list[k] = (int32_t *)(b + off_list[k]);
  c->cp.pflag = (uint8_t *)(b + off_cp[0]);
  c->cp.cnt_next = (uint32_t *)(b + off_cp[1]);
  c->cp.cnt_shadow = (uint32_t *)(b + off_cp[2]);
  c->cap_paths = paths;
  c->cap_nls = nls;
  c->cap_pend = true;
  return PTAMD_OK;
}

int setup_frame_params(const ptamd_context *c, const ptamd_render_params *p, FrameParams &fp) {
  if (!c->have_scene || !c->have_camera) {
    ptamd::set_error("scene and camera must be set before rendering");
    return PTAMD_ERR_STATE;
  }
  if (!p || p->w <= 0 || p->h <= 0 || p->ns_aa < 0 || p->samples_per_batch <= 0 || p->ns_area_light < 0 ||
      p->max_ray_depth < 0 || (int64_t)p->w * p->h > (int64_t)1 << 30) {
    ptamd::set_error("invalid render parameters");
    return PTAMD_ERR_ARG;
  }
  fp.w = p->w; fp.h = p->h;
  fp.max_ray_depth = p->max_ray_depth;
  fp.accum = p->accumulate_bounces ? 1 : 0;
  fp.ns_area_light = p->ns_area_light;
  fp.hemisphere = p->hemisphere ? 1 : 0;
  fp.specular = p->specular ? 1 : 0;
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
  if (const char *e = getenv("PTAMD_TRACE_TWO")) c->trace_two = atoi(e) != 0;
  if (const char *e = getenv("PTAMD_FUSE_BELOW")) c->fuse_below = (uint32_t)strtoul(e, nullptr, 10);
  if (const char *e = getenv("PTAMD_PLAN_MARGIN")) { const double m = atof(e); if (m >= 1.0 && m <= 8.0) c->plan_margin = m; }
  // The counters live apart from the work buffers: those are reallocated when a later round needs
  // more room, and a frame's statistics must survive that.
  if (hipMalloc((void **)&c->ctl, sizeof(Ctl)) != hipSuccess || hipMemset(c->ctl, 0, sizeof(Ctl)) != hipSuccess) {
    ptamd::set_error("cannot allocate the device-side counters");
    ptamd_destroy(c);
    return PTAMD_ERR_DEVICE;
  }
  // Address of this device's copy of the static scratch array (the device is current: hipSetDevice
  // above). The array's name is only a host-side handle; kernels and memsets get this pointer.
  {
    void *sp = nullptr;
    const hipError_t e = hipGetSymbolAddress(&sp, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !sp) {
      ptamd::set_error(std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e));
      ptamd_destroy(c);
      return PTAMD_ERR_DEVICE;
    }
    c->static_block_count = (uint32_t *)sp;
  }
  *out = c;
  return PTAMD_OK;
}

void ptamd_destroy(ptamd_context *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  collect_events(c);
  for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
#ifdef PTAMD_WAVELOG
  wavelog_dump(c);
  if (c->S.wavelog) (void)hipFree(c->S.wavelog);
#endif
  free_scene(c);
  free_frame_scratch(c);
  if (c->slab) (void)hipFree(c->slab);
  if (c->ctl) (void)hipFree(c->ctl);
  if (c->block_count) (void)hipFree(c->block_count);
  if (c->gstack) (void)hipFree(c->gstack);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

int ptamd_set_max_paths(ptamd_context *c, int64_t max_paths) {
  if (!c) { ptamd::set_error("null context"); return PTAMD_ERR_ARG; }
  if (max_paths <= 0) max_paths = 16 << 20;
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

int ptamd_set_event_classes(ptamd_context *c, uint32_t mask) {
  if (!c) { ptamd::set_error("null context"); return PTAMD_ERR_ARG; }
  c->event_mask = mask;
  return PTAMD_OK;
}

// Fills the quantised boxes of a 4-wide node: per axis a grid step 2^e and an
// origin on that grid such that every child plane lies within 255 steps and every
// decoded plane origin + q * 2^e is exactly representable in float. lo planes are
// rounded down, hi planes up (checked in double, where all of this is exact).
static bool quantize_node(DevNode &n, const double boxes[][6], int nk) {
  n.exps = 0;
  for (int a = 0; a < 3; a++) {
    double lo = INFINITY, hi = -INFINITY;
    for (int k = 0; k < nk; k++) { lo = std::min(lo, boxes[k][a]); hi = std::max(hi, boxes[k][3 + a]); }
    if (!(std::isfinite(lo) && std::isfinite(hi)) || hi < lo) return false;
    int e;
    {
      const double ext = std::max(hi - lo, 1e-30);
      (void)std::frexp(ext / 255.0, &e);  // ext / 255 < 2^e
      e = std::max(e, -100);
    }
    for (;; e++) {
      if (e > 100) return false;
      const double s = std::ldexp(1.0, e);
      const double k0 = std::floor(lo / s);
      if (std::fabs(k0) > 4194304.0) continue;  // origin + 255 steps must stay below 2^24 steps
      const double org = k0 * s;
      if (std::ceil((hi - org) / s) > 255.0) continue;
      bool ok = true;
      uint8_t q[4][2];
      for (int k = 0; k < nk && ok; k++) {
        double ql = std::floor((boxes[k][a] - org) / s), qh = std::ceil((boxes[k][3 + a] - org) / s);
        while (ql > 0 && org + ql * s > boxes[k][a]) ql -= 1;
        while (org + qh * s < boxes[k][3 + a]) qh += 1;
        if (ql < 0 || qh > 255 || org + ql * s > boxes[k][a]) { ok = false; break; }
        // the device evaluates fmaf(q, s, org) in float: must be exact
        if ((double)(float)(org + ql * s) != org + ql * s || (double)(float)(org + qh * s) != org + qh * s) { ok = false; break; }
        q[k][0] = (uint8_t)ql; q[k][1] = (uint8_t)qh;
      }
      if (!ok || (double)(float)org != org) continue;
      n.org[a] = (float)org;
      n.exps |= (uint32_t)(uint8_t)(int8_t)e << (8 * a);
      n.qlo[a] = n.qhi[a] = 0;
      for (int k = 0; k < nk; k++) {
        n.qlo[a] |= (uint32_t)q[k][0] << (8 * k);
        n.qhi[a] |= (uint32_t)q[k][1] << (8 * k);
      }
      break;
    }
  }
  return true;
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
        if (cnt > 4) { ptamd::set_error("BVH leaves of more than 4 primitives are not supported (the reference builds with max_leaf_size = 4)"); return PTAMD_ERR_SCENE; }
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
  if (NP >= (int)PRIM_NONE) { ptamd::set_error("more than 2^24 - 2 primitives are not supported"); return PTAMD_ERR_SCENE; }
  // ---- collapse the binary tree into 4-wide nodes whose children are nodes or
  // single primitives. new_index[p] = position of the caller's primitive p in
  // visiting order (the device's primitive index).
  std::vector<int32_t> new_index(NP);
  for (int i = 0; i < NP; i++) new_index[visit_order[i]] = i;
  // ---- The primitives at the far end of a light's shadow rays. The CB scenes' light is a
  // rectangle that coincides with an emissive quad of the geometry (to within the 5e-8 the
  // quad's float coordinates are off), and the reference's shadow rays end ON the light
  // (max_t = distToLight, pathtracer.cpp:125): whether the quad occludes such a ray is decided
  // by the last bits of an FP64 test, which no float pre-test can settle, for every shadow ray
  // of the frame. So up to four triangles that lie in a light's plane and overlap its rectangle
  // are listed per light (the first seven area or point lights that have any): k_shade runs the reference's test on them
  // when it makes a shadow ray, and the ray's walk skips the primitives that carry its light's
  // tag. Which primitives are listed only decides where they are tested, not the result: a
  // shadow ray is occluded iff some primitive's test hits strictly before its end, in any order.
  std::vector<int32_t> light_end(4 * (size_t)std::max(d->n_lights, 1), -1), light_tag(std::max(d->n_lights, 1), 0);
  std::vector<uint8_t> prim_tag(NP, 0);
  int has_end = 0;
  {
    uint32_t next_tag = 1;
    for (int li = 0; li < d->n_lights && next_tag <= MAX_LIGHT_TAG; li++) {
      // (a point light's rays end on the light as well, pathtracer.cpp:125 with light.cpp:50-57, and
      // CBempty.dae puts one in the middle of its emissive quad: a rectangle of no extent, any plane)
      const bool area = d->light_kind[li] == PTAMD_LIGHT_AREA;
      if (!area && d->light_kind[li] != PTAMD_LIGHT_POINT) continue;
      const double *L = d->light_param + 16 * (size_t)li;
      const double *c0 = L + 3, *nrm = L + 6, *dx = L + 9, *dy = L + 12;
      double lo[3], hi[3], scale = 0;
      for (int k = 0; k < 3; k++) {
        const double ex = area ? 0.5 * (std::fabs(dx[k]) + std::fabs(dy[k])) : 0.0;
        lo[k] = c0[k] - ex; hi[k] = c0[k] + ex;
        scale = std::max(scale, std::max(std::fabs(lo[k]), std::fabs(hi[k])));
      }
      const double eps = 1e-5 * (1.0 + scale);
      int found = 0;
      for (int i = 0; i < NP && found < 4; i++) {  // in the device's primitive order
        const int p = visit_order[i];
        if (d->prim_type[p] != PTAMD_PRIM_TRIANGLE || prim_tag[p]) continue;
        const double *P = d->prim_pos + 9 * (size_t)p;
        bool in_plane = true, overlaps = true;
        if (area) {
          for (int v = 0; v < 3 && in_plane; v++) {
            const double h = (P[3 * v] - c0[0]) * nrm[0] + (P[3 * v + 1] - c0[1]) * nrm[1] + (P[3 * v + 2] - c0[2]) * nrm[2];
            in_plane = std::fabs(h) <= eps;
          }
        } else {
          // the point's distance to the triangle's plane
          const double e1[3] = {P[3] - P[0], P[4] - P[1], P[5] - P[2]}, e2[3] = {P[6] - P[0], P[7] - P[1], P[8] - P[2]};
          const double nx = e1[1] * e2[2] - e1[2] * e2[1], ny = e1[2] * e2[0] - e1[0] * e2[2], nz = e1[0] * e2[1] - e1[1] * e2[0];
          const double len = std::sqrt(nx * nx + ny * ny + nz * nz);
          in_plane = len > 0 && std::fabs((c0[0] - P[0]) * nx + (c0[1] - P[1]) * ny + (c0[2] - P[2]) * nz) <= eps * len;
        }
        if (!in_plane) continue;
        for (int k = 0; k < 3; k++) {
          const double tlo = std::min(P[k], std::min(P[3 + k], P[6 + k])), thi = std::max(P[k], std::max(P[3 + k], P[6 + k]));
          if (thi < lo[k] - eps || tlo > hi[k] + eps) overlaps = false;
        }
        if (!overlaps) continue;
        light_end[4 * (size_t)li + found++] = i;
        prim_tag[p] = (uint8_t)next_tag;
      }
      if (found) {
        light_tag[li] = (int32_t)next_tag++;
        has_end = 1;
      }
    }
  }
  struct Item { int32_t node; int32_t prim; };  // node >= 0: a node of the caller's tree; else the caller's primitive
  auto prim_box = [&](int p, double bb[6]) {
    const double *P = d->prim_pos + 9 * (size_t)p;
    if (d->prim_type[p] == PTAMD_PRIM_TRIANGLE) {
      for (int k = 0; k < 3; k++) {
        bb[k] = std::min(P[k], std::min(P[3 + k], P[6 + k]));
        bb[3 + k] = std::max(P[k], std::max(P[3 + k], P[6 + k]));
      }
    } else {  // o -/+ Vector3D(r, r, r), src/scene/sphere.h:32-34
      for (int k = 0; k < 3; k++) { bb[k] = P[k] - P[3]; bb[3 + k] = P[k] + P[3]; }
    }
  };
  auto item_box = [&](const Item &it, double bb[6]) {
    if (it.node >= 0) memcpy(bb, d->node_bbox + 6 * (size_t)it.node, 48);
    else prim_box(it.prim, bb);
  };
  auto area = [](const double bb[6]) {
    const double ex = bb[3] - bb[0], ey = bb[4] - bb[1], ez = bb[5] - bb[2];
    return ex * ey + ey * ez + ez * ex;
  };
  auto expand = [&](const Item &it, std::vector<Item> &out) {
    const int n = it.node;
    if (d->node_count[n] > 0) {
      for (int p = d->node_start[n]; p < d->node_start[n] + d->node_count[n]; p++) out.push_back({-1, p});
    } else {
      out.push_back({d->node_left[n], -1});
      out.push_back({d->node_right[n], -1});
    }
  };
  auto expanded_size = [&](const Item &it) { return d->node_count[it.node] > 0 ? d->node_count[it.node] : 2; };
  std::vector<DevNode> nodes;
  nodes.reserve((size_t)NP / 2 + 4);
  int depth4 = 0;
  struct Frame { int32_t node4; int slot; Item item; int depth; };
  std::vector<Frame> work;
  // a single-primitive leaf is just that primitive
  auto normalize = [&](Item it) {
    if (it.node >= 0 && d->node_count[it.node] == 1) return Item{-1, d->node_start[it.node]};
    return it;
  };
  int32_t root_code;
  {
    DevNode rootn;
    memset(&rootn, 0, sizeof rootn);
    for (int k = 0; k < 4; k++) rootn.child[k] = CHILD_EMPTY;
    nodes.push_back(rootn);
    root_code = 0;
    work.push_back({0, -1, normalize(Item{d->root, -1}), 1});
  }
  // `work` entries with slot == -1 fill node `node4` with the children of `item`
  while (!work.empty()) {
    Frame f = work.back();
    work.pop_back();
    depth4 = std::max(depth4, f.depth);
    std::vector<Item> kids;
    if (f.item.node < 0) {
      kids.push_back(f.item);  // the whole scene is one primitive
    } else {
      expand(f.item, kids);
      for (auto &k : kids) k = normalize(k);
      // open the largest openable child while the result still fits four slots
      for (;;) {
        int best = -1;
        double best_area = -1;
        for (size_t k = 0; k < kids.size(); k++) {
          if (kids[k].node < 0) continue;
          if ((int)kids.size() - 1 + expanded_size(kids[k]) > 4) continue;
          double bb[6];
          item_box(kids[k], bb);
          const double a = area(bb);
          if (a > best_area) { best_area = a; best = (int)k; }
        }
        if (best < 0) break;
        const Item it = kids[best];
        kids.erase(kids.begin() + best);
        std::vector<Item> sub;
        expand(it, sub);
        for (auto &k : sub) kids.push_back(normalize(k));
      }
    }
    double boxes[4][6];
    for (size_t k = 0; k < kids.size(); k++) item_box(kids[k], boxes[k]);
    if (!quantize_node(nodes[f.node4], boxes, (int)kids.size())) {
      ptamd::set_error("BVH node box cannot be represented (non-finite or out of float range)");
      return PTAMD_ERR_SCENE;
    }
    {
      // the light tag of the node's tagged primitive children, if they agree on one
      uint32_t ntag = 0;
      bool agree = true;
      for (size_t k = 0; k < kids.size(); k++)
        if (kids[k].node < 0 && prim_tag[kids[k].prim]) {
          if (ntag && ntag != prim_tag[kids[k].prim]) agree = false;
          ntag = prim_tag[kids[k].prim];
        }
      nodes[f.node4].exps = (nodes[f.node4].exps & 0x00ffffffu) | ((agree ? ntag : 0u) << 24);
    }
    for (size_t k = 0; k < kids.size(); k++) {
      if (kids[k].node < 0) {
        nodes[f.node4].child[k] = (int32_t)~((uint32_t)new_index[kids[k].prim] | ((uint32_t)prim_tag[kids[k].prim] << PRIM_BITS));
      } else {
        DevNode nn;
        memset(&nn, 0, sizeof nn);
        for (int j = 0; j < 4; j++) nn.child[j] = CHILD_EMPTY;
        const int32_t id = (int32_t)nodes.size();
        nodes.push_back(nn);
        nodes[f.node4].child[k] = id;
        work.push_back({id, -1, kids[k], f.depth + 1});
      }
    }
  }
  // deepest stack: three siblings pushed per level
  max_depth = 3 * depth4;
  // ---- primitive records in visiting order
  std::vector<double> prim((size_t)NP * PRIM_STRIDE, 0.0), nrm((size_t)NP * 9);
  std::vector<int32_t> info(NP);
  std::vector<uint4> prim32((size_t)NP * 3);
  bool coords_ok = true;
  double max_coord = 0;
  for (int i = 0; i < NP; i++) {
    const int p = visit_order[i];
    const double *P = d->prim_pos + 9 * (size_t)p;
    double *R = prim.data() + (size_t)i * PRIM_STRIDE;
    if (d->prim_type[p] == PTAMD_PRIM_TRIANGLE) {
      // e1 = p2 - p1, e2 = p3 - p1: one rounding each, identical to evaluating
      // them per ray
      R[0] = P[0]; R[1] = P[1]; R[2] = P[2];
      R[3] = P[3] - P[0]; R[4] = P[4] - P[1]; R[5] = P[5] - P[2];
      R[6] = P[6] - P[0]; R[7] = P[7] - P[1]; R[8] = P[8] - P[2];
      info[i] = d->prim_material[p] << 1;
    } else {
      R[0] = P[0]; R[1] = P[1]; R[2] = P[2]; R[3] = P[4];
      info[i] = (d->prim_material[p] << 1) | 1;
    }
    memcpy(R + 9, &info[i], 4);
    memcpy(nrm.data() + (size_t)i * 9, d->prim_nrm + 9 * (size_t)p, 72);
    // the FP32 pre-test's record
    unsigned char *Q = reinterpret_cast<unsigned char *>(prim32.data() + (size_t)i * 3);
    memcpy(Q, R, 24);
    float ef[6];
    if (d->prim_type[p] == PTAMD_PRIM_TRIANGLE) {
      for (int k = 0; k < 6; k++) ef[k] = (float)R[3 + k];
    } else {
      ef[0] = NAN;
      ef[1] = (float)R[3];
      for (int k = 2; k < 6; k++) ef[k] = 0.f;
    }
    memcpy(Q + 24, ef, 24);
    // prim32.h assumes coordinates (and so edge lengths) well inside the float range
    if (d->prim_type[p] == PTAMD_PRIM_TRIANGLE) {
      for (int k = 0; k < 9; k++) max_coord = std::max(max_coord, std::fabs(P[k]));
    } else {
      for (int k = 0; k < 3; k++) max_coord = std::max(max_coord, std::fabs(P[k]) + std::fabs(P[3]));
    }
  }
  coords_ok = max_coord < 1048576.0;  // also false for NaN
  if (!coords_ok) {
    ptamd::set_error("scene coordinates of 2^20 or more are not supported");
    return PTAMD_ERR_SCENE;
  }
  free_scene(c);
  c->dev2ref = visit_order;
  c->n_prims = NP;
  DevScene S{};
  int rc;
  if ((rc = upload(c, nodes, &S.nodes))) return rc;
  if ((rc = upload(c, prim, &S.prim))) return rc;
  if ((rc = upload(c, prim32, &S.prim32))) return rc;
  if ((rc = upload(c, info, &S.prim_info))) return rc;
  if ((rc = upload(c, nrm, &S.prim_nrm))) return rc;
  std::vector<int32_t> mk(d->mat_kind, d->mat_kind + d->n_materials);
  std::vector<double> mp(d->mat_param, d->mat_param + 3 * (size_t)d->n_materials);
  std::vector<int32_t> lk(d->light_kind, d->light_kind + d->n_lights);
  std::vector<double> lp(d->light_param, d->light_param + 16 * (size_t)d->n_lights);
  std::vector<double> jt(d->jitter_table, d->jitter_table + 2 * (size_t)d->jitter_n * d->jitter_n);
  std::vector<double> me(4 * (size_t)d->n_materials, 0.0);
  if (d->mat_extra) me.assign(d->mat_extra, d->mat_extra + 4 * (size_t)d->n_materials);
  if ((rc = upload(c, mk, &S.mat_kind))) return rc;
  if ((rc = upload(c, mp, &S.mat_param))) return rc;
  if ((rc = upload(c, me, &S.mat_extra))) return rc;
  if ((rc = upload(c, lk, &S.light_kind))) return rc;
  if ((rc = upload(c, light_end, &S.light_end))) return rc;
  if ((rc = upload(c, light_tag, &S.light_tag))) return rc;
  S.has_end = has_end;
  if ((rc = upload(c, lp, &S.light_param))) return rc;
  if ((rc = upload(c, jt, &S.jitter))) return rc;
  S.n_lights = d->n_lights;
  S.jitter_n = d->jitter_n;
  S.root_code = root_code;
  c->scene_max_coord = max_coord;
  {
    double far = max_coord;  // camera rays start at the camera, every other ray on a primitive
    if (c->have_camera) for (int k = 0; k < 3; k++) far = std::max(far, std::fabs(c->cam.pos[k]));
    S.coord_bound = (float)(1.01 * far + 1e-30);
  }
  S.stack_cap = max_depth + 1;
  c->S = S;
  c->have_scene = true;
  c->host_light_kinds = lk;
  return PTAMD_OK;
}

int ptamd_set_camera(ptamd_context *c, const ptamd_camera *cam) {
  if (!c || !cam) { ptamd::set_error("null argument"); return PTAMD_ERR_ARG; }
  c->cam = *cam;
  // camera rays start at the camera, every other ray on a primitive
  const double far = std::max(c->scene_max_coord, std::max(std::fabs(cam->pos[0]), std::max(std::fabs(cam->pos[1]), std::fabs(cam->pos[2]))));
  c->S.coord_bound = (float)(1.01 * far + 1e-30);
  c->have_camera = true;
  return PTAMD_OK;
}


int ptamd_frame_begin(ptamd_context *c, const ptamd_render_params *p) {
  if (!c) { ptamd::set_error("null context"); return PTAMD_ERR_ARG; }
  HIPCHK(hipSetDevice(c->device));
  int rc = setup_frame_params(c, p, c->fp);
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
  // The device-side counters are cleared by the frame's first ptamd_trace_list, on ITS stream: a
  // memset issued here, on the null stream, would not be ordered against an earlier frame still
  // running on a non-blocking stream.
  c->clear_ctl = true;
  return PTAMD_OK;
}

// Launch shape of the traversal kernels: 256-thread workgroups, PTAMD_STACK_* stack
// levels per lane in LDS, the rest (if the tree is deeper) in a global strip.
struct TravLaunch {
  int block;
  size_t lds_extend, lds_shadow;
  int grid_extend, grid_shadow;  // grid-stride driver: a few workgroups per resident slot; persistent: exactly the resident ones
  uint2 *gstack;
};
static int traversal_launch(ptamd_context *c, TravLaunch *t) {
  t->block = TRAV_BLOCK;
  t->lds_extend = (size_t)t->block * (PTAMD_STACK_EXTEND + 1) * sizeof(uint2);  // + the scratch level
  t->lds_shadow = (size_t)t->block * (PTAMD_STACK_SHADOW + 1) * sizeof(uint2);
  // resident workgroups per CU: bounded by LDS (160 KiB per CU) and by registers
  const int lds_e = (int)((160 * 1024) / t->lds_extend), lds_s = (int)((160 * 1024) / t->lds_shadow);
  t->grid_extend = c->num_cu * std::max(1, std::min(lds_e, PTAMD_WAVES_EXTEND)) * (PTAMD_REFILL_EXTEND ? 1 : PTAMD_GRID_MULT);
  t->grid_shadow = c->num_cu * std::max(1, std::min(lds_s, PTAMD_WAVES_SHADOW)) * (PTAMD_REFILL_SHADOW ? 1 : PTAMD_GRID_MULT);
  const size_t levels = (size_t)std::max(0, c->S.stack_cap - std::min(PTAMD_STACK_EXTEND, PTAMD_STACK_SHADOW));
  const size_t need = levels * (size_t)std::max(t->grid_extend, t->grid_shadow) * t->block * sizeof(uint2);
  if (need > c->gstack_bytes) {
    if (c->gstack) { HIPCHK(hipDeviceSynchronize()); HIPCHK(hipFree(c->gstack)); c->gstack = nullptr; }
    HIPCHK(hipMalloc((void **)&c->gstack, need));
    c->gstack_bytes = need;
  }
  t->gstack = c->gstack;
  return PTAMD_OK;
}

namespace {
int check_list(const ptamd_pixel_list *l, const char *who) {
  if (!l || l->n_max < 0 || l->stride < 1 || l->offset < 0 || (!l->d_n && (l->n < 0 || l->n > l->n_max)) ||
      (l->n_max > 0 && !l->d_pixels) || l->n_max > (int64_t)0x7fffffff) {
    ptamd::set_error(std::string("invalid pixel list passed to ") + who);
    return PTAMD_ERR_ARG;
  }
  return PTAMD_OK;
}
PixView view_of(const ptamd_pixel_list *l) {
  return PixView{l->d_pixels, (const long long *)l->d_n, (long long)l->n, (uint32_t)l->offset, (uint32_t)l->stride};
}
// entries of the shard, at most
int64_t shard_max(const ptamd_pixel_list *l) {
  return l->n_max > l->offset ? (l->n_max - l->offset + l->stride - 1) / l->stride : 0;
}
}  // namespace

int ptamd_trace_list(ptamd_context *c, const ptamd_pixel_list *list, int32_t sample_base, int32_t n_samples,
                     double *d_state, void *stream) {
  if (!c || !c->have_frame) { ptamd::set_error("ptamd_frame_begin must be called first"); return PTAMD_ERR_STATE; }
  int rc = check_list(list, "ptamd_trace_list");
  if (rc) return rc;
  const int64_t n = shard_max(list);
  if (n_samples <= 0 || sample_base < -1 || (n > 0 && !d_state) ||
      (sample_base < 0 && n_samples % c->params.samples_per_batch != 0)) {
    ptamd::set_error("invalid arguments to ptamd_trace_list");
    return PTAMD_ERR_ARG;
  }
  if (n == 0) return PTAMD_OK;
  HIPCHK(hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)stream;
  const FrameParams &fp = c->fp;
  const int nls = std::max(fp.nls, 1);
  if (nls >= (1 << 16)) {
    ptamd::set_error("more than 65535 light samples per vertex are not supported");
    return PTAMD_ERR_ARG;
  }
  const FastDiv by_nls = make_fastdiv((uint32_t)nls);
  const uint2 nls_nsa = make_uint2((uint32_t)nls, (uint32_t)std::max(fp.ns_area_light, 0));  // light samples per vertex, per area light
  // paths per launch: max_paths, but never more than a 24 GB slab of work buffers
  // (about 290 bytes per path + 70 per light-sample slot), and never so many that a light-sample
  // slot index (entry * nls + q, an int32) would pass 2^31
  const int64_t by_memory = std::min(((int64_t)24 << 30) / (290 + 70 * (int64_t)nls), (((int64_t)1 << 31) - 1) / nls);
  int64_t px_per_chunk = std::max<int64_t>(1, std::min(c->max_paths, by_memory) / n_samples);
  px_per_chunk = std::min(px_per_chunk, n);
  rc = ensure_buffers(c, px_per_chunk * n_samples, nls, !fp.accum);
  if (rc) return rc;
  TravLaunch tl;
  rc = traversal_launch(c, &tl);
  if (rc) return rc;
  if (c->clear_ctl) {
    HIPCHK(hipMemsetAsync(c->ctl, 0, sizeof(Ctl), s));
    c->clear_ctl = false;
  }
  const int tblock = tl.block;
  const int egrid = tl.grid_extend, sgrid = tl.grid_shadow;
  const size_t elds = tl.lds_extend, slds = tl.lds_shadow;
  uint2 *gstack = tl.gstack;
  const int ggrid = grid_for(c, 8);
  // k_shade runs one workgroup per 256 list entries; no list of this call is longer than a chunk's paths
  const int shade_grid = (int)((px_per_chunk * n_samples + 255) / 256);
  const int light_grid = (int)(((int64_t)shade_grid * 256 * nls + 255) / 256);  // k_light: one workgroup per 256 light-sample slots
  PathBuf pb = c->pb;
  if (fp.accum) pb.pend[0] = pb.pend[1] = pb.pend[2] = nullptr;
  const PixView pv = view_of(list);
  // per-pixel prediction only where a launch is big enough for dropped samples to cost time
  const RoundPlan rp{d_state, c->params.max_tolerance, sample_base, n_samples, c->params.samples_per_batch, c->params.ns_aa,
                     n * n_samples >= (1 << 20) ? 1 : 0, c->plan_margin};
  Launcher L{c, s};
  const int depths = std::max(1, fp.max_ray_depth);
  for (int64_t p0 = 0; p0 < n; p0 += px_per_chunk) {
    const uint32_t up0 = (uint32_t)p0;
    L.begin(PTAMD_K_GENERATE);
    hipLaunchKernelGGL(k_set_counts, 1, 1, 0, s, c->ctl, pv, up0, (uint32_t)px_per_chunk, (uint32_t)n_samples);
    hipLaunchKernelGGL(k_generate, ggrid, 256, 0, s, fp, c->S, pv, up0, c->ctl, rp, c->q[0], pb);
    L.end();
    int cur = 0;
    bool traced_ahead = false;  // this depth's bounce rays went with the previous depth's shadow rays (k_trace2)
    for (int depth = 1; depth <= depths; depth++) {
      const double min_t = depth == 1 ? fp.n_clip : PT_EPS_D;
      const double max_t = depth == 1 ? fp.f_clip : (double)INFINITY;
      const int32_t *lst = depth == 1 ? nullptr : c->list[cur];
      if (!traced_ahead) {
        L.begin(PTAMD_K_EXTEND);
        PTAMD_WAVELOG_NEXT(c, 0, egrid);
        if (c->prof_counters) hipLaunchKernelGGL((k_extend<true>), egrid, tblock, elds, s, c->S, c->q[cur], lst, c->hb, c->ctl, min_t, max_t, gstack, (const int32_t *)c->vb.self_code, (const float *)c->vb.self_dmin);
        else hipLaunchKernelGGL((k_extend<false>), egrid, tblock, elds, s, c->S, c->q[cur], lst, c->hb, c->ctl, min_t, max_t, gstack, (const int32_t *)c->vb.self_code, (const float *)c->vb.self_dmin);
        L.end();
      }
      traced_ahead = false;
      VertexBuf vb = c->vb;
      for (int k = 0; k < 3; k++) vb.hp[k] = c->q[cur ^ 1].o[k];  // one copy of the hit point serves both uses
      L.begin(PTAMD_K_SHADE);
      if (fp.hemisphere)
        hipLaunchKernelGGL(k_shade<true>, shade_grid, 256, 0, s, fp, c->S, c->q[cur], lst, c->hb, c->q[cur ^ 1], c->sq, vb, pb,
                           c->cp, c->ctl, depth, pv, up0, rp);
      else
        hipLaunchKernelGGL(k_shade<false>, shade_grid, 256, 0, s, fp, c->S, c->q[cur], lst, c->hb, c->q[cur ^ 1], c->sq, vb, pb,
                           c->cp, c->ctl, depth, pv, up0, rp);
      L.end();
      if (!fp.hemisphere && fp.max_ray_depth > 0 && fp.nls > 0) {
        L.begin(PTAMD_K_LIGHT);
        hipLaunchKernelGGL(k_light, PTAMD_LIGHT_LOOP ? std::min(light_grid, grid_for(c, PTAMD_LIGHT_LOOP)) : light_grid, 256, 0, s, fp, c->S, vb, c->sq, c->cp, c->ctl, by_nls,
                           1u + (uint32_t)(depth - 1) * (uint32_t)fp.bpv);
        L.end();
      }
      L.begin(PTAMD_K_OTHER);
      hipLaunchKernelGGL(k_scan2, 1, 1024, 0, s, c->cp, c->ctl);
      hipLaunchKernelGGL(k_scatter, ggrid, 256, 0, s, c->cp, vb, c->list[cur ^ 1], c->sq, c->ctl, fp.max_ray_depth > 0 ? fp.nls : 0);
      L.end();
      if (fp.max_ray_depth > 0 && fp.nls > 0) {
        uint32_t shadow_skip = 0;  // k_shadow skips lists that k_trace2 has traced: those shorter than this
        if (depth < depths && c->trace_two) {
          // the next depth's bounce rays (queue cur ^ 1, just written) together with this depth's shadow rays;
          // the hit records of this depth have been consumed by k_shade
          L.begin(PTAMD_K_TRACE2);
          PTAMD_WAVELOG_NEXT(c, 1, egrid);
          const bool hemi = fp.hemisphere != 0, cnt = c->prof_counters;
#define PTAMD_LAUNCH_TRACE2(H, C) hipLaunchKernelGGL((k_trace2<H, C>), egrid, tblock, elds, s, c->S, c->q[cur ^ 1], (const int32_t *)c->list[cur ^ 1], c->hb, c->sq, vb, c->ctl, by_nls, nls_nsa, gstack, c->fuse_below)
          if (hemi) { if (cnt) PTAMD_LAUNCH_TRACE2(true, true); else PTAMD_LAUNCH_TRACE2(true, false); }
          else { if (cnt) PTAMD_LAUNCH_TRACE2(false, true); else PTAMD_LAUNCH_TRACE2(false, false); }
#undef PTAMD_LAUNCH_TRACE2
          L.end();
          traced_ahead = true;
          shadow_skip = c->fuse_below;
        }
        if (shadow_skip != 0xffffffffu) {  // (all lists fused: not even the empty launch)
          L.begin(PTAMD_K_SHADOW);
          PTAMD_WAVELOG_NEXT(c, 2, sgrid);
          if (fp.hemisphere) {
            if (c->prof_counters) hipLaunchKernelGGL((k_shadow<true, true>), sgrid, tblock, slds, s, c->S, c->sq, vb, c->ctl, by_nls, nls_nsa, gstack, shadow_skip);
            else hipLaunchKernelGGL((k_shadow<true, false>), sgrid, tblock, slds, s, c->S, c->sq, vb, c->ctl, by_nls, nls_nsa, gstack, shadow_skip);
          } else {
            if (c->prof_counters) hipLaunchKernelGGL((k_shadow<false, true>), sgrid, tblock, slds, s, c->S, c->sq, vb, c->ctl, by_nls, nls_nsa, gstack, shadow_skip);
            else hipLaunchKernelGGL((k_shadow<false, false>), sgrid, tblock, slds, s, c->S, c->sq, vb, c->ctl, by_nls, nls_nsa, gstack, shadow_skip);
          }
          L.end();
        }
        L.begin(PTAMD_K_RESOLVE);
        hipLaunchKernelGGL(k_resolve, shade_grid, 256, 0, s, fp, c->S, c->q[cur], lst, vb, (const SRay *)c->sq.ray, pb, c->ctl, depth);
        hipLaunchKernelGGL(k_advance, 1, 1, 0, s, c->ctl);
        L.end();
      } else {
        L.begin(PTAMD_K_OTHER);
        hipLaunchKernelGGL(k_advance, 1, 1, 0, s, c->ctl);
        L.end();
      }
      cur ^= 1;
    }
    L.begin(PTAMD_K_ACCUMULATE);
    hipLaunchKernelGGL(k_accumulate, ggrid, 256, 0, s, pv, up0, c->ctl, rp, pb, d_state);
    L.end();
  }
  HIPCHK(hipGetLastError());
  return PTAMD_OK;
}

int ptamd_trace_pixels(ptamd_context *c, const int32_t *d_pixels, int64_t n, int32_t sample_base, int32_t n_samples,
                       double *d_state, void *stream) {
  if (n < 0 || (n > 0 && !d_pixels)) { ptamd::set_error("invalid arguments to ptamd_trace_pixels"); return PTAMD_ERR_ARG; }
  const ptamd_pixel_list l{d_pixels, nullptr, n, n, 0, 1};
  return ptamd_trace_list(c, &l, sample_base, n_samples, d_state, stream);
}

int ptamd_filter_list(ptamd_context *c, const ptamd_pixel_list *in, int32_t n_done, const double *d_state,
                      int32_t *d_out, int64_t *d_n_out, void *stream) {
  if (!c || !c->have_frame) { ptamd::set_error("ptamd_frame_begin must be called first"); return PTAMD_ERR_STATE; }
  int rc = check_list(in, "ptamd_filter_list");
  if (rc) return rc;
  const int64_t n = shard_max(in);
  if ((n_done < 1 && n_done != -1) || !d_n_out || (n > 0 && (!d_out || !d_state))) {
    ptamd::set_error("invalid arguments to ptamd_filter_list");
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
  const PixView pv = view_of(in);
  Launcher L{c, s};
  L.begin(PTAMD_K_FILTER);
  if (nb) hipLaunchKernelGGL(k_filter_count, nb, 256, 0, s, pv, n_done, c->params.max_tolerance, c->params.samples_per_batch,
                             c->params.ns_aa, d_state, c->block_count);
  hipLaunchKernelGGL(k_filter_scan, 1, 1024, 0, s, c->block_count, nb, (long long *)d_n_out);
  if (nb) hipLaunchKernelGGL(k_filter_scatter, nb, 256, 0, s, pv, n_done, c->params.max_tolerance, c->params.samples_per_batch,
                             c->params.ns_aa, d_state, c->block_count, d_out);
  L.end();
  HIPCHK(hipGetLastError());
  return PTAMD_OK;
}

// ptamd_filter_list with the per-block counts in the static device array instead of a buffer
// the library allocates: no hipMalloc / hipFree / event creation on any path, so it may be
// called where allocation is not allowed (under stream capture, in a serving loop). The
// kernels are the same three; they get the array through the pointer taken at ptamd_create.
// Nothing in the array has to start at zero: k_filter_count writes every count that
// k_filter_scan and k_filter_scatter read. Calls that overlap on one GPU share the array and race.
int ptamd_filter_list_static(ptamd_context *c, const ptamd_pixel_list *in, int32_t n_done, const double *d_state,
                             int32_t *d_out, int64_t *d_n_out, void *stream) {
  if (!c || !c->have_frame) { ptamd::set_error("ptamd_frame_begin must be called first"); return PTAMD_ERR_STATE; }
  int rc = check_list(in, "ptamd_filter_list_static");
  if (rc) return rc;
  const int64_t n = shard_max(in);
  if ((n_done < 1 && n_done != -1) || !d_n_out || (n > 0 && (!d_out || !d_state))) {
    ptamd::set_error("invalid arguments to ptamd_filter_list_static");
    return PTAMD_ERR_ARG;
  }
  if (n > FILTER_STATIC_MAX_PIXELS) {
    ptamd::set_error("ptamd_filter_list_static: the list is longer than PTAMD_FILTER_STATIC_MAX_PIXELS; "
                     "use ptamd_filter_list");
    return PTAMD_ERR_ARG;
  }
  if (!c->static_block_count) { ptamd::set_error("static scratch array is not set up"); return PTAMD_ERR_STATE; }
  HIPCHK(hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)stream;
  const uint32_t nb = (uint32_t)((n + 255) / 256);
  uint32_t *counts = c->static_block_count;
  const PixView pv = view_of(in);
  c->stats.launches[PTAMD_K_FILTER]++;
  if (nb) hipLaunchKernelGGL(k_filter_count, nb, 256, 0, s, pv, n_done, c->params.max_tolerance, c->params.samples_per_batch,
                             c->params.ns_aa, d_state, counts);
  hipLaunchKernelGGL(k_filter_scan, 1, 1024, 0, s, counts, nb, (long long *)d_n_out);
  if (nb) hipLaunchKernelGGL(k_filter_scatter, nb, 256, 0, s, pv, n_done, c->params.max_tolerance, c->params.samples_per_batch,
                             c->params.ns_aa, d_state, counts, d_out);
  HIPCHK(hipGetLastError());
  return PTAMD_OK;
}

int ptamd_adaptive_filter(ptamd_context *c, const int32_t *d_in, int64_t n, int32_t n_done, const double *d_state,
                          int32_t *d_out, int64_t *d_n_out, void *stream) {
  if (n < 0 || (n > 0 && !d_in)) { ptamd::set_error("invalid arguments to ptamd_adaptive_filter"); return PTAMD_ERR_ARG; }
  const ptamd_pixel_list l{d_in, nullptr, n, n, 0, 1};
  return ptamd_filter_list(c, &l, n_done, d_state, d_out, d_n_out, stream);
}

int ptamd_pack_rows(ptamd_context *c, const ptamd_pixel_list *list, const double *d_state, double *d_rows, void *stream) {
  if (!c) { ptamd::set_error("null context"); return PTAMD_ERR_ARG; }
  int rc = check_list(list, "ptamd_pack_rows");
  if (rc) return rc;
  if (shard_max(list) == 0) return PTAMD_OK;
  if (!d_state || !d_rows) { ptamd::set_error("null buffer"); return PTAMD_ERR_ARG; }
  HIPCHK(hipSetDevice(c->device));
  Launcher L{c, (hipStream_t)stream};
  L.begin(PTAMD_K_OTHER);
  hipLaunchKernelGGL(k_pack_rows, grid_for(c, 4), 256, 0, (hipStream_t)stream, view_of(list), d_state, d_rows);
  L.end();
  HIPCHK(hipGetLastError());
  return PTAMD_OK;
}

int ptamd_unpack_rows(ptamd_context *c, const ptamd_pixel_list *list, int32_t world, int32_t skip_rank,
                      const double *d_rows, int64_t rows_per_rank, double *d_state, void *stream) {
  if (!c) { ptamd::set_error("null context"); return PTAMD_ERR_ARG; }
  int rc = check_list(list, "ptamd_unpack_rows");
  if (rc) return rc;
  if (world < 1 || list->offset != 0 || list->stride != 1 || rows_per_rank < (list->n_max + world - 1) / world) {
    ptamd::set_error("invalid arguments to ptamd_unpack_rows");
    return PTAMD_ERR_ARG;
  }
  if (list->n_max == 0) return PTAMD_OK;
  if (!d_state || !d_rows) { ptamd::set_error("null buffer"); return PTAMD_ERR_ARG; }
  HIPCHK(hipSetDevice(c->device));
  Launcher L{c, (hipStream_t)stream};
  L.begin(PTAMD_K_OTHER);
  hipLaunchKernelGGL(k_unpack_rows, grid_for(c, 4), 256, 0, (hipStream_t)stream, view_of(list), (uint32_t)world,
                     skip_rank < 0 ? 0xffffffffu : (uint32_t)skip_rank, d_rows, (uint32_t)rows_per_rank, d_state);
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
    c->stats.rays = h.st_rays - h.st_dead;  // path slots a round left unused are not rays
    c->stats.shadow_rays = h.st_shadow;
    c->stats.path_vertices = h.st_vertices;
    c->stats.nodes_visited = h.st_nodes;
    c->stats.prim_tests = h.st_tests;
    c->stats.exact_tests = h.st_exact;
    c->stats.samples = h.st_samples;
    c->stats.two_ext_rays = h.st_two_ext;
    c->stats.two_shadow_rays = h.st_two_shadow;
  }
  *out = c->stats;
  return PTAMD_OK;
}

namespace {
// Buffers of ptamd_render / ptamd_render_list, kept by the context between frames (grown on demand).
int ensure_frame_scratch(ptamd_context *c, int64_t npix, int64_t nlist, bool outputs) {
  FrameScratch &f = c->fs;
  if (!f.d_n) {
    HIPCHK(hipMalloc((void **)&f.d_n, 2 * sizeof(int64_t)));
    HIPCHK(hipHostMalloc((void **)&f.h_n, 2 * sizeof(int64_t), hipHostMallocDefault));
    HIPCHK(hipEventCreateWithFlags(&f.ev[0], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&f.ev[1], hipEventDisableTiming));
  }
  if (nlist > f.cap_list) {
    HIPCHK(hipDeviceSynchronize());
    for (int k = 0; k < 3; k++) { if (f.d_list[k]) HIPCHK(hipFree(f.d_list[k])); f.d_list[k] = nullptr; }
    for (int k = 0; k < 3; k++) HIPCHK(hipMalloc((void **)&f.d_list[k], std::max<int64_t>(nlist, 1) * sizeof(int32_t)));
    f.cap_list = nlist;
  }
  if (outputs && npix > f.cap_pix) {
    HIPCHK(hipDeviceSynchronize());
    if (f.d_state) HIPCHK(hipFree(f.d_state));
    if (f.d_hdr) HIPCHK(hipFree(f.d_hdr));
    if (f.d_count) HIPCHK(hipFree(f.d_count));
    f.d_state = f.d_hdr = nullptr; f.d_count = nullptr;
    HIPCHK(hipMalloc((void **)&f.d_state, npix * 6 * sizeof(double)));
    HIPCHK(hipMalloc((void **)&f.d_hdr, npix * 3 * sizeof(double)));
    HIPCHK(hipMalloc((void **)&f.d_count, npix * sizeof(int32_t)));
    f.cap_pix = npix;
  }
  return PTAMD_OK;
}
void free_frame_scratch(ptamd_context *c) {
  FrameScratch &f = c->fs;
  for (int k = 0; k < 3; k++) if (f.d_list[k]) (void)hipFree(f.d_list[k]);
  if (f.d_state) (void)hipFree(f.d_state);
  if (f.d_hdr) (void)hipFree(f.d_hdr);
  if (f.d_count) (void)hipFree(f.d_count);
  if (f.d_n) (void)hipFree(f.d_n);
  if (f.h_n) (void)hipHostFree(f.h_n);
  for (hipEvent_t e : f.ev) if (e) (void)hipEventDestroy(e);
  f = FrameScratch{};
}
}  // namespace

int ptamd_render_list(ptamd_context *c, const ptamd_render_params *p, const int32_t *d_pixels, int64_t n, double *d_state,
                      void *stream) {
  if (!c || !p || n < 0 || (n > 0 && (!d_pixels || !d_state))) { ptamd::set_error("invalid arguments to ptamd_render_list"); return PTAMD_ERR_ARG; }
  int rc = ptamd_frame_begin(c, p);
  if (rc) return rc;
  if (n == 0) return PTAMD_OK;
  rc = ensure_frame_scratch(c, 0, n, false);
  if (rc) return rc;
  FrameScratch &f = c->fs;
  hipStream_t s = (hipStream_t)stream;
  {
    const int64_t init[2] = {n, n};
    HIPCHK(hipMemcpyAsync(f.d_n, init, sizeof init, hipMemcpyHostToDevice, s));
  }
  // Round r traces the survivors of the adaptive test after round r - 1. Their
  // number stays on the device; the host only needs an upper bound to size the
  // launches, and uses the count of one round earlier, read back asynchronously,
  // so the GPU never waits for the host between rounds.
  const int32_t *lists[2] = {d_pixels, nullptr};  // current list: the caller's, then the context's two in turn
  int32_t *own[2] = {f.d_list[0], f.d_list[1]};
  int64_t n_known = n;
  int cur = 0, done = 0;
  for (int r = 0; done < p->ns_aa && n_known > 0; r++) {
    ptamd_pixel_list l{lists[cur], f.d_n + cur, 0, n_known, 0, 1};
    if (r > 0) {
      int32_t *out = own[r & 1];
      rc = ptamd_filter_list(c, &l, done, d_state, out, f.d_n + (cur ^ 1), s);
      if (rc) return rc;
      cur ^= 1;
      lists[cur] = out;
      HIPCHK(hipMemcpyAsync(f.h_n + (r & 1), f.d_n + cur, sizeof(int64_t), hipMemcpyDeviceToHost, s));
      HIPCHK(hipEventRecord(f.ev[r & 1], s));
      if (r > 1) {
        HIPCHK(hipEventSynchronize(f.ev[(r - 1) & 1]));
        n_known = f.h_n[(r - 1) & 1];
        if (n_known == 0) break;
      }
      l = ptamd_pixel_list{lists[cur], f.d_n + cur, 0, n_known, 0, 1};
    }
    // a round too small to fill the GPU traces several batches per pixel ahead of
    // their adaptive tests (k_accumulate applies them and drops what was not needed)
    const int64_t paths = n_known * p->samples_per_batch;
    // (most pixels stop at their first test, so the first round traces ahead only when really small)
    const int64_t target = r == 0 ? (4 << 20) : (16 << 20);
    int64_t k = std::max<int64_t>(1, std::min<int64_t>(target / std::max<int64_t>(paths, 1), 32));
    const int ns = (int)std::min<int64_t>(k * p->samples_per_batch, p->ns_aa - done);
    rc = ptamd_trace_list(c, &l, done, ns, d_state, s);
    if (rc) return rc;
    done += ns;
  }
  return PTAMD_OK;
}

int ptamd_render(ptamd_context *c, const ptamd_render_params *p, double *hdr_out, int32_t *count_out) {
  if (!c || !p || !hdr_out || !count_out) { ptamd::set_error("null argument"); return PTAMD_ERR_ARG; }
  HIPCHK(hipSetDevice(c->device));
  const int x0 = std::max(0, p->x0), y0 = std::max(0, p->y0), x1 = std::min(p->w, p->x1), y1 = std::min(p->h, p->y1);
  const int64_t npix = (int64_t)p->w * p->h;
  const int64_t nrect = (x1 > x0 && y1 > y0) ? (int64_t)(x1 - x0) * (y1 - y0) : 0;
  if (npix <= 0) { ptamd::set_error("empty frame"); return PTAMD_ERR_ARG; }
  int rc = ensure_frame_scratch(c, npix, nrect, true);
  if (rc) return rc;
  FrameScratch &f = c->fs;
  hipStream_t s = c->own_stream;
  HIPCHK(hipMemsetAsync(f.d_state, 0, npix * 6 * sizeof(double), s));
  if (nrect) hipLaunchKernelGGL(k_iota, grid_for(c, 4), 256, 0, s, f.d_list[2], (uint32_t)nrect, p->w, x0, y0, x1 - x0);
  HIPCHK(hipStreamSynchronize(s));
  auto t0 = std::chrono::steady_clock::now();
  rc = ptamd_render_list(c, p, f.d_list[2], nrect, f.d_state, s);
  if (rc) return rc;
  rc = ptamd_frame_resolve(c, f.d_state, f.d_hdr, f.d_count, s);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(s));
  c->stats.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  HIPCHK(hipMemcpy(hdr_out, f.d_hdr, npix * 3 * sizeof(double), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(count_out, f.d_count, npix * sizeof(int32_t), hipMemcpyDeviceToHost));
  return PTAMD_OK;
}

int ptamd_trace_primary(ptamd_context *c, int64_t n, const double *sx, const double *sy, int32_t *prim_out,
                        double *t_out) {
  if (!c || n < 0 || (n > 0 && (!sx || !sy || !prim_out || !t_out))) { ptamd::set_error("invalid arguments"); return PTAMD_ERR_ARG; }
  ptamd_render_params p{};
  p.w = 1; p.h = 1; p.x1 = 1; p.y1 = 1; p.ns_aa = 1; p.samples_per_batch = 1;
  // (a frame that is open keeps its own parameters: these are local)
  FrameParams fp{};
  int rc = setup_frame_params(c, &p, fp);
  if (rc) return rc;
  HIPCHK(hipSetDevice(c->device));
  const int64_t chunk = std::min<int64_t>(std::max<int64_t>(n, 1), c->max_paths);
  rc = ensure_buffers(c, chunk, std::max(c->cap_nls, 1), false);
  if (rc) return rc;
  TravLaunch tl;
  rc = traversal_launch(c, &tl);
  if (rc) return rc;
  const int tblock = tl.block, tgrid = tl.grid_extend;
  const size_t lds = tl.lds_extend;
  hipStream_t s = c->own_stream;
  // the queue's throughput arrays are unused here and double as staging for sx, sy
  double *d_sx = c->q[0].T[0], *d_sy = c->q[0].T[1];
  std::vector<int32_t> prim((size_t)chunk);
  for (int64_t b = 0; b < n; b += chunk) {
    const int64_t m = std::min(chunk, n - b);
    HIPCHK(hipMemcpyAsync(d_sx, sx + b, m * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d_sy, sy + b, m * sizeof(double), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_set_counts, 1, 1, 0, s, c->ctl, PixView{nullptr, nullptr, (long long)m, 0u, 1u}, 0u, (uint32_t)m, 1u);
    hipLaunchKernelGGL(k_generate_primary, grid_for(c, 4), 256, 0, s, fp, d_sx, d_sy, (uint32_t)m, c->q[0]);
    hipLaunchKernelGGL((k_extend<false>), tgrid, tblock, lds, s, c->S, c->q[0], (const int32_t *)nullptr, c->hb, c->ctl, fp.n_clip, fp.f_clip, tl.gstack, (const int32_t *)nullptr, (const float *)nullptr);
    hipLaunchKernelGGL(k_primary_t, grid_for(c, 4), 256, 0, s, c->S, c->q[0], (const int32_t *)c->hb.prim, (uint32_t)m, fp.n_clip, c->hb.t);
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
