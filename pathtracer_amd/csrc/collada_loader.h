// COLLADA 1.4 (.dae) scene loader for the subset the reference reads:
// node transforms, up-axis fix-up, cameras, lights, polylist meshes, and the
// CGL <extra> profile for area lights, spheres and BSDFs.
// Semantics follow src/scene/collada/collada.cpp of the reference (cited per
// function in the .cpp); errors are returned, never exit()ed.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "mathd.h"

namespace ptamd {

enum MaterialKind { MAT_DIFFUSE = 0, MAT_EMISSION = 1, MAT_MIRROR = 2, MAT_MICROFACET = 3, MAT_REFRACTION = 4, MAT_GLASS = 5 };

struct Material {
  int kind = MAT_DIFFUSE;
  V3 a = V3(0.5, 0.5, 0.5);  // reflectance | radiance | eta | transmittance
  V3 b;                      // k (microfacet) | reflectance (glass)
  double alpha = 0;          // microfacet alpha | roughness
  double ior = 0;
};

enum LightType { LIGHT_NONE = 0, LIGHT_AMBIENT, LIGHT_DIRECTIONAL, LIGHT_AREA, LIGHT_POINT, LIGHT_SPOT };

struct CameraInfo {
  V3 view_dir, up_dir;
  float hFov = 0, vFov = 0, nClip = 0, fClip = 0;
};

struct LightInfo {
  int light_type = LIGHT_NONE;
  V3 spectrum = V3(1, 1, 1);
  V3 position = V3(0, 0, 0);
  V3 direction = V3(0, 0, -1);
  V3 up = V3(0, 1, 0);
};

struct SphereInfo {
  float radius = 0;
  bool has_material = false;
  Material material;
};

struct PolymeshInfo {
  std::vector<V3> vertices;
  std::vector<std::vector<size_t>> polygons;  // vertex indices per polygon
  bool has_material = false;
  Material material;
};

enum InstanceType { INST_CAMERA = 0, INST_LIGHT, INST_SPHERE, INST_POLYMESH };

struct Node {
  std::string id, name;
  int type = -1;
  M4 transform = M4::identity();
  std::shared_ptr<CameraInfo> camera;
  std::shared_ptr<LightInfo> light;
  std::shared_ptr<SphereInfo> sphere;
  std::shared_ptr<PolymeshInfo> polymesh;
};

struct SceneInfo {
  std::vector<Node> nodes;
};

// Returns 0 on success; on failure returns non-zero and fills err.
int collada_load(const char *filename, SceneInfo *out, std::string *err);

}  // namespace ptamd
