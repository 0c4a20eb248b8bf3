// The 4-wide BVH node and the primitive codes of the device layer: plain data, shared by the
// host-side tree builder (tree4.cpp, no GPU headers needed) and the traversal kernels.
#pragma once
#include <stdint.h>

namespace ptk {

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
// plain FP32 node or the 16 of two FP64 boxes. (Measured again with the FP32 primitive
// tests in place: a 128-byte node with the 24 planes as floats, seven loads per lane, the
// near and far planes picked by address, saves the 24 half-rate byte-to-float conversions
// of a box step and still loses: first-hit walks 26 % slower, closest-hit walks 9 %.)
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

}  // namespace ptk
