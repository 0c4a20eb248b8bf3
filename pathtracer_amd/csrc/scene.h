// Host scene pipeline: COLLADA scene -> world-space static scene -> SAH BVH ->
// flat arrays for the device. Every step reproduces the reference's ordering
// and arithmetic so that primitive order, node order and all coordinates are
// bit-identical to what the reference's CPU renderer traverses.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "collada_loader.h"
#include "mathd.h"

namespace ptamd {

enum PrimType { PRIM_TRIANGLE = 0, PRIM_SPHERE = 1 };
// light kinds as the device sees them
enum LightKind { LK_DIRECTIONAL = 0, LK_INF_HEMISPHERE = 1, LK_POINT = 2, LK_SPOT = 3, LK_AREA = 4 };

struct Primitive {
  int type = PRIM_TRIANGLE;
  V3 p1, p2, p3;  // triangle vertices; sphere: p1 = centre, p2.x = r, p2.y = r*r
  V3 n1, n2, n3;  // triangle vertex normals
  V3 bb_min, bb_max;
  int material = 0;
};

struct Light {
  int kind = LK_AREA;
  V3 radiance, position, direction, dim_x, dim_y;
  double area = 0;
};

struct BvhNode {
  V3 bb_min, bb_max;
  int32_t l = -1, r = -1;       // children (interior)
  int32_t start = 0, count = 0;  // primitive range (leaf: count > 0)
};

struct CameraState {
  // src/pathtracer/camera.h:100-125
  double hFov = 0, vFov = 0, ar = 0, nClip = 0, fClip = 0;
  V3 pos, targetPos;
  double phi = 0, theta = 0, r = 0, minR = 0, maxR = 0;
  M3 c2w;
  size_t screenW = 0, screenH = 0;
  double screenDist = 0;

  void configure(const CameraInfo &info, size_t screenW, size_t screenH);
  void place(const V3 &targetPos, double phi, double theta, double r, double minR, double maxR);
  void set_screen_size(size_t screenW, size_t screenH);
  void compute_position();
  bool load_settings(const std::string &filename);
  bool dump_settings(const std::string &filename) const;
};

struct StaticScene {
  std::vector<Primitive> prims;  // BVH order after build_bvh(); scene order before
  std::vector<Material> materials;
  std::vector<Light> lights;
  std::vector<BvhNode> nodes;  // post-order, root = nodes.size()-1
  int32_t root = -1;
  V3 bbox_min, bbox_max;  // scene bounding box over objects (camera placement)
  bool bbox_empty = true;
  CameraState camera;
  double build_seconds = 0;
};

// Builds the world-space static scene from parsed COLLADA nodes, places the
// camera, and builds the BVH. Returns 0 on success.
int build_static_scene(const SceneInfo &info, StaticScene *out, std::string *err);

// SAH BVH over scene.prims (reorders prims in place, fills nodes/root).
void build_bvh(StaticScene *scene, size_t max_leaf_size = 4);

// The fixed pre-jittered 16x16 sub-pixel table: 512 draws of a default-seeded
// mt19937 through the reference's random_uniform().
void make_jitter_table(std::vector<double> *xy /* 2*n*n */, size_t n = 16);

}  // namespace ptamd
