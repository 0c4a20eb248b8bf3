// Host-side tree and primitive records of the device layer, see tree4.cpp.
#pragma once
#include <stdint.h>

#include <vector>

#include "../../include/ptamd.h"
#include "dev_node.h"

namespace ptamd {

struct Tree4 {
  struct alignas(16) Word4 { uint32_t x, y, z, w; };
  std::vector<ptk::DevNode> nodes;   // 4-wide nodes; node 0 is the root
  int32_t root_code = 0;
  int stack_cap = 0;                 // entries a lane's traversal stack can need
  std::vector<int32_t> visit_order;  // device primitive index -> the caller's primitive index
  std::vector<double> prim, nrm;     // FP64 records [n * PRIM_STRIDE] and vertex normals [n * 9], device order
  std::vector<int32_t> info;         // bit 0: sphere, bits 1..: material
  std::vector<Word4> prim32;         // [n * 3] the FP32 pre-test's records
  std::vector<int32_t> light_end;    // [4 * max(n_lights, 1)] primitives at the far end of a light's shadow rays, or -1
  std::vector<int32_t> light_tag;    // [max(n_lights, 1)] the tag their child codes carry, 0: none
  int has_end = 0;
  double max_coord = 0;              // largest coordinate magnitude of the scene
};

// Validates the caller's binary BVH (d must be complete: ptamd_upload_scene checks the pointers) and
// builds the above. PTAMD_OK, or an error code with the message left in ptamd::set_error.
int build_tree4(const ptamd_scene_desc *d, Tree4 &out);

}  // namespace ptamd
