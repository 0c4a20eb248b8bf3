// The BVH walk of the traversal kernels: the conservative FP32 box test on quantised 4-wide nodes,
// the per-lane stack, the state of one ray's walk (node step, primitive step, trip), and the two
// drivers that feed rays to persistent waves. Device code only.
#pragma once
#include "device_types.h"
#include "prim32.h"
#include "prim_exact.h"

namespace ptk {

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
        // measured 1 % slower than the two scalar FMAs (a packed FMA issues every 4.3 cycles, a scalar one every
        // 2.3: no flops to gain, and the register pairs it pins cost), kept as a build option
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
    // A node some of whose children are end primitives of the light this shadow ray goes to (k_light has
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
    // an end primitive of the light this shadow ray goes to: k_light has tested it (and found no hit)
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

}  // namespace ptk
