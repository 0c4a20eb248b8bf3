#include "scene.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <fstream>
#include <limits>
#include <map>
#include <random>
#include <set>
#include <stdexcept>

namespace ptamd {
namespace {

constexpr double INF = std::numeric_limits<double>::infinity();

// std::min / std::max argument order matters for NaN and signed zeros
// (src/scene/bbox.h:69-94 uses std::min(min.x, p.x)).
inline void bb_expand(V3 &mn, V3 &mx, const V3 &p) {
  mn.x = std::min(mn.x, p.x); mn.y = std::min(mn.y, p.y); mn.z = std::min(mn.z, p.z);
  mx.x = std::max(mx.x, p.x); mx.y = std::max(mx.y, p.y); mx.z = std::max(mx.z, p.z);
}
inline void bb_expand(V3 &mn, V3 &mx, const V3 &omn, const V3 &omx) {
  mn.x = std::min(mn.x, omn.x); mn.y = std::min(mn.y, omn.y); mn.z = std::min(mn.z, omn.z);
  mx.x = std::max(mx.x, omx.x); mx.y = std::max(mx.y, omx.y); mx.z = std::max(mx.z, omx.z);
}
inline double bb_surface_area(const V3 &mn, const V3 &mx) {  // bbox.h:104-119
  if (mn.x > mx.x || mn.y > mx.y || mn.z > mx.z) return 0.0;
  V3 e = mx - mn;
  return 2 * (e.x * e.z + e.x * e.y + e.y * e.z);
}

// ---------------------------------------------------------------------------
// Half-edge connectivity, index based. Element creation order, the halfedge
// each vertex/face ends up pointing to, and the walks used for vertex normals
// follow src/util/halfEdgeMesh.cpp:29-404 and src/util/halfEdgeMesh.h:492-547.
struct HalfedgeMesh {
  struct HE { int next = -1, twin = -1, vertex = -1, face = -1; };
  std::vector<HE> he;
  std::vector<int> v_he;        // vertex -> halfedge
  std::vector<V3> v_pos, v_nrm;
  std::vector<int> f_he;        // face -> halfedge (faces first, then boundary loops)
  std::vector<char> f_boundary;
  size_t n_faces = 0;

  void build(const std::vector<std::vector<size_t>> &polygons, const std::vector<V3> &positions) {
    std::map<size_t, int> indexToVertex;
    std::vector<size_t> degree_seen;
    for (const auto &p : polygons) {
      if (p.size() < 3) throw std::runtime_error("each polygon must have at least three vertices");
      std::set<size_t> distinct;
      for (size_t i : p) {
        distinct.insert(i);
        auto it = indexToVertex.find(i);
        if (it == indexToVertex.end()) {
          int v = (int)v_he.size();
          v_he.push_back(-1);
          indexToVertex[i] = v;
          degree_seen.push_back(1);
        } else {
          degree_seen[it->second]++;
        }
      }
      if (distinct.size() < p.size()) throw std::runtime_error("a polygon does not have distinct vertices");
    }
    n_faces = polygons.size();
    f_he.assign(n_faces, -1);
    f_boundary.assign(n_faces, 0);

    std::map<std::pair<size_t, size_t>, int> pairToHalfedge;
    for (size_t f = 0; f < n_faces; f++) {
      const auto &p = polygons[f];
      size_t degree = p.size();
      std::vector<int> faceHalfedges;
      for (size_t i = 0; i < degree; i++) {
        size_t a = p[i], b = p[(i + 1) % degree];
        std::pair<size_t, size_t> ab(a, b);
        if (pairToHalfedge.count(ab)) throw std::runtime_error("mesh is nonmanifold or inconsistently oriented");
        int hab = (int)he.size();
        he.push_back(HE());
        pairToHalfedge[ab] = hab;
        he[hab].face = (int)f;
        f_he[f] = hab;
        he[hab].vertex = indexToVertex[a];
        v_he[he[hab].vertex] = hab;
        faceHalfedges.push_back(hab);
        auto iba = pairToHalfedge.find(std::make_pair(b, a));
        if (iba != pairToHalfedge.end()) {
          he[hab].twin = iba->second;
          he[iba->second].twin = hab;
        } else {
          he[hab].twin = -1;
        }
      }
      for (size_t i = 0; i < degree; i++) he[faceHalfedges[i]].next = faceHalfedges[(i + 1) % degree];
    }

    // boundary vertices: advance to a twinless halfedge (:236-248)
    for (size_t v = 0; v < v_he.size(); v++) {
      int h = v_he[v];
      do {
        if (he[h].twin < 0) {
          v_he[v] = h;
          break;
        }
        h = he[he[h].twin].next;
      } while (h != v_he[v]);
    }

    // boundary loops; the loop also runs over halfedges appended meanwhile (:251-325)
    for (size_t h = 0; h < he.size(); h++) {
      if (he[h].twin >= 0) continue;
      int b = (int)f_he.size();
      f_he.push_back(-1);
      f_boundary.push_back(1);
      std::vector<int> boundaryHalfedges;
      int i = (int)h;
      do {
        int t = (int)he.size();
        he.push_back(HE());
        boundaryHalfedges.push_back(t);
        he[i].twin = t;
        he[t].twin = i;
        he[t].face = b;
        he[t].vertex = he[he[i].next].vertex;
        i = he[i].next;
        while (i != (int)h && he[i].twin >= 0) {
          i = he[i].twin;
          i = he[i].next;
        }
      } while (i != (int)h);
      size_t degree = boundaryHalfedges.size();
      for (size_t p = 0; p < degree; p++) {
        size_t q = (p - 1 + degree) % degree;
        he[boundaryHalfedges[p]].next = boundaryHalfedges[q];
      }
      // Face::halfedge() of a boundary loop is never assigned by the reference
      // and is not read on this path
    }

    for (size_t v = 0; v < v_he.size(); v++) v_he[v] = he[he[v_he[v]].twin].next;  // (:331-333)

    for (size_t v = 0; v < v_he.size(); v++) {  // manifold check (:336-364)
      size_t count = 0;
      int h = v_he[v];
      do {
        if (!f_boundary[he[h].face]) count++;
        h = he[he[h].twin].next;
      } while (h != v_he[v]);
      if (count != degree_seen[v]) throw std::runtime_error("at least one of the vertices is nonmanifold");
    }

    if (positions.size() != v_he.size())
      throw std::runtime_error("number of vertex positions is different from the number of distinct vertices");
    v_pos.assign(v_he.size(), V3());
    size_t i = 0;
    for (auto &e : indexToVertex) v_pos[e.second] = positions[i++];  // key order (:380-397)

    v_nrm.assign(v_he.size(), V3());
    for (size_t v = 0; v < v_he.size(); v++) v_nrm[v] = compute_normal((int)v);
  }

  bool vertex_is_boundary(int v) const {  // halfEdgeMesh.h:532-547
    int h = v_he[v];
    do {
      if (f_boundary[he[h].face]) return true;
      h = he[he[h].twin].next;
    } while (h != v_he[v]);
    return false;
  }

  V3 compute_normal(int v) const {  // halfEdgeMesh.h:492-515
    V3 normal(0., 0., 0.);
    V3 pi = v_pos[v];
    int h = v_he[v];
    if (vertex_is_boundary(v)) {
      do {
        V3 pj = v_pos[he[he[h].next].vertex];
        V3 pk = v_pos[he[he[he[h].next].next].vertex];
        V3 c = cross(pj - pi, pk - pi);
        normal = normal + c;
        h = he[he[h].next].twin;
      } while (h != v_he[v]);
    } else {
      do {
        V3 pj = v_pos[he[he[h].next].vertex];
        V3 pk = v_pos[he[he[he[h].next].next].vertex];
        V3 c = cross(pj - pi, pk - pi);
        normal = normal + c;
        h = he[he[h].twin].next;
      } while (h != v_he[v]);
    }
    normalize(normal);
    return normal;
  }
};

Light make_light(const LightInfo &li, const M4 &T) {
  Light l;
  switch (li.light_type) {
    case LIGHT_AMBIENT:  // gl_scene/ambient_light.h:14-24 -> InfiniteHemisphereLight
      l.kind = LK_INF_HEMISPHERE;
      l.radiance = li.spectrum;
      break;
    case LIGHT_DIRECTIONAL: {  // gl_scene/directional_light.h:14-27, scene/light.cpp:11-15
      l.kind = LK_DIRECTIONAL;
      l.radiance = li.spectrum;
      V3 dir = -to3D(T * V4(li.direction, 1));
      normalize(dir);
      l.direction = -unit(dir);  // dirToLight
      break;
    }
    case LIGHT_AREA: {  // gl_scene/area_light.h:14-33, scene/light.cpp:75-79
      l.kind = LK_AREA;
      l.radiance = li.spectrum;
      l.position = to3D(T * V4(li.position, 1));
      l.direction = to3D(T * V4(li.direction, 1)) - l.position;
      normalize(l.direction);
      V3 dim_y = li.up;
      V3 dim_x = cross(li.up, li.direction);
      l.dim_x = to3D(T * V4(dim_x, 1)) - l.position;
      l.dim_y = to3D(T * V4(dim_y, 1)) - l.position;
      l.area = norm(l.dim_x) * norm(l.dim_y);
      break;
    }
    case LIGHT_POINT:  // gl_scene/point_light.h:17-29
      l.kind = LK_POINT;
      l.radiance = li.spectrum;
      l.position = to3D(T * V4(li.position, 1));
      break;
    case LIGHT_SPOT:  // gl_scene/spot_light.h:14-28; sample_L is a stub returning 0
      l.kind = LK_SPOT;
      l.radiance = li.spectrum;
      l.position = to3D(T * V4(li.position, 1));
      l.direction = to3D(T * V4(li.direction, 1)) - l.position;
      normalize(l.direction);
      break;
    default:
      throw std::runtime_error("light without a supported type");
  }
  return l;
}

}  // namespace

// --------------------------------------------------------------------------- camera
// Camera set-up. The numbers below have to come out bit for bit as the reference's
// (src/pathtracer/camera.cpp:29-146) computes them, so every expression keeps the reference's
// operations and their order; the decomposition is this file's own.
namespace {
// tan of half a field of view given in degrees: tan(radians(fov) / 2)
double half_tan(double fov_deg) { return tan(radians(fov_deg) / 2); }
// the field of view (degrees) that spans `extent` pixels at `dist` pixels: 2 degrees(atan(extent / (2 dist)))
double fov_spanning(size_t extent, double dist) { return 2 * degrees(atan(((double)extent) / (2 * dist))); }
}  // namespace

// camera.cpp:29-47: take the COLLADA camera's fields of view, widen whichever is too narrow for the
// window's aspect ratio, and fix the distance of the image plane in pixels from the vertical one.
void CameraState::configure(const CameraInfo &info, size_t sw, size_t sh) {
  screenW = sw;
  screenH = sh;
  nClip = info.nClip;
  fClip = info.fClip;
  hFov = info.hFov;
  vFov = info.vFov;
  ar = static_cast<double>(screenW) / screenH;
  const double described = half_tan(hFov) / half_tan(vFov);  // the aspect ratio the two angles describe
  if (described < ar) hFov = 2 * degrees(atan(half_tan(vFov) * ar));
  else if (described > ar) vFov = 2 * degrees(atan(half_tan(hFov) / ar));
  screenDist = ((double)screenH) / (2.0 * half_tan(vFov));
}

// camera.cpp:53-65: spherical coordinates around a target; r is clamped, and an angle whose sine
// vanishes is nudged off the pole before it is stored.
void CameraState::place(const V3 &target, double phi_in, double theta_in, double r_in, double minR_in,
                        double maxR_in) {
  targetPos = target;
  minR = minR_in;
  maxR = maxR_in;
  r = std::min(std::max(r_in, minR_in), maxR_in);
  phi = (sin(phi_in) == 0) ? (phi_in + EPS_F) : phi_in;
  theta = theta_in;
  compute_position();
}

// camera.cpp:83-89: a new window keeps the image-plane distance, so both angles follow the size
void CameraState::set_screen_size(size_t sw, size_t sh) {
  screenW = sw;
  screenH = sh;
  ar = 1.0 * screenW / screenH;
  hFov = fov_spanning(screenW, screenDist);
  vFov = fov_spanning(screenH, screenDist);
}

// camera.cpp:123-146: position and camera-to-world frame from (phi, theta, r). Columns of c2w:
// screen x = up x back, screen y = back x screen x, and the unit vector from the target to the camera.
void CameraState::compute_position() {
  double s_phi = sin(phi);
  if (s_phi == 0) {
    phi += EPS_F;
    s_phi = sin(phi);
  }
  const V3 back(r * s_phi * sin(theta), r * cos(phi), r * s_phi * cos(theta));
  pos = targetPos + back;
  const V3 up(0, s_phi > 0 ? 1 : -1, 0);
  c2w.col[0] = cross(up, back);
  normalize(c2w.col[0]);
  c2w.col[1] = cross(back, c2w.col[0]);
  normalize(c2w.col[1]);
  c2w.col[2] = unit(back);
}

// The camera settings file (written by Camera::dump_settings, read by Camera::load_settings,
// src/pathtracer/camera.cpp:151-185) is 26 numbers separated by white space:
//   hFov vFov ar nClip fClip | pos.xyz targetPos.xyz | phi theta r minR maxR |
//   c2w row by row | screenW screenH screenDist | focalDistance lensRadius
// Both directions go through one table of where each number lives, so that they cannot drift
// apart. The thin-lens pair at the end is read and dropped (the reference's thin-lens camera is a
// stub that raytrace_pixel never calls), and written as the reference's defaults.
namespace {
struct SettingsSlot {
  double *d;     // a double field, or
  size_t *z;     // a size_t field (the screen size), or neither: a value that is not kept
  double spare;  // what is written for a value that is not kept
  char after;    // the separator the reference writes after it: ' ' or '\n'
};
std::vector<SettingsSlot> settings_layout(CameraState &c) {
  std::vector<SettingsSlot> t;
  auto num = [&](double &x, char after = ' ') { t.push_back({&x, nullptr, 0.0, after}); };
  num(c.hFov); num(c.vFov); num(c.ar); num(c.nClip); num(c.fClip, '\n');
  for (int k = 0; k < 3; k++) num(c.pos[k]);
  for (int k = 0; k < 3; k++) num(c.targetPos[k]);
  t.back().after = 'N';  // the reference ends this line with " \n" (a space after every element)
  num(c.phi); num(c.theta); num(c.r); num(c.minR); num(c.maxR, '\n');
  for (int row = 0; row < 3; row++)
    for (int col = 0; col < 3; col++) num(c.c2w.col[col][row]);
  t.back().after = 'N';
  t.push_back({nullptr, &c.screenW, 0.0, ' '});
  t.push_back({nullptr, &c.screenH, 0.0, ' '});
  num(c.screenDist, '\n');
  t.push_back({nullptr, nullptr, 4.7, ' '});   // focalDistance
  t.push_back({nullptr, nullptr, 0.0, '\n'});  // lensRadius
  return t;
}
}  // namespace

bool CameraState::load_settings(const std::string &filename) {
  std::ifstream file(filename);
  if (!file.is_open()) return false;
  CameraState next = *this;  // a file that ends early leaves the camera as it was
  for (const SettingsSlot &slot : settings_layout(next)) {
    if (slot.z) {
      file >> *slot.z;
    } else {
      double v = 0;
      file >> v;
      if (slot.d) *slot.d = v;
    }
    if (file.fail()) return false;
  }
  *this = next;
  return true;
}

bool CameraState::dump_settings(const std::string &filename) const {
  std::ofstream file(filename);
  if (!file.is_open()) return false;
  CameraState copy = *this;
  for (const SettingsSlot &slot : settings_layout(copy)) {
    if (slot.z) file << *slot.z;
    else file << (slot.d ? *slot.d : slot.spare);
    if (slot.after == 'N') file << " \n";
    else file << slot.after;
  }
  return (bool)file.flush();
}

// --------------------------------------------------------------------------- scene
int build_static_scene(const SceneInfo &info, StaticScene *out, std::string *err) {
  try {
    StaticScene &S = *out;
    S = StaticScene();
    // Application::init (src/application/application.cpp:98-104): dummy camera at 800x600
    size_t screenW = 800, screenH = 600;
    {
      CameraInfo ci;
      ci.hFov = 50;
      ci.vFov = 35;
      ci.nClip = 0.01;
      ci.fClip = 100;
      S.camera.configure(ci, screenW, screenH);
    }
    // Application::load (src/application/application.cpp:229-305)
    V3 c_pos, c_dir;
    V3 sb_min(INF, INF, INF), sb_max(-INF, -INF, -INF);
    for (const Node &node : info.nodes) {
      const M4 &T = node.transform;
      switch (node.type) {
        case INST_CAMERA: {
          c_pos = to3D(T * V4(c_pos, 1));
          // w = 1: the node's translation leaks into the view direction (:250)
          c_dir = unit(to3D(T * V4(node.camera->view_dir, 1)));
          S.camera.configure(*node.camera, screenW, screenH);
          break;
        }
        case INST_LIGHT:
          S.lights.push_back(make_light(*node.light, T));
          break;
        case INST_SPHERE: {  // application.cpp:348-353, gl_scene/sphere.cpp:12-20,48-54
          V3 position = projectTo3D(T * V4(0, 0, 0, 1));
          double scale = norm(to3D(T * V4(1, 0, 0, 0)));
          double r = node.sphere->radius * scale;
          int mat = (int)S.materials.size();
          S.materials.push_back(node.sphere->has_material ? node.sphere->material : Material());
          Primitive p;
          p.type = PRIM_SPHERE;
          p.p1 = position;
          p.p2 = V3(r, r * r, 0);
          p.material = mat;
          // scene/sphere.h:32-34
          p.bb_min = position - V3(r, r, r);
          p.bb_max = position + V3(r, r, r);
          S.prims.push_back(p);
          // gl_scene/sphere.cpp:44-46 (scene bbox for camera placement)
          bb_expand(sb_min, sb_max, V3(position.x - r, position.y - r, position.z - r),
                    V3(position.x + r, position.y + r, position.z + r));
          break;
        }
        case INST_POLYMESH: {  // gl_scene/mesh.cpp:22-43, scene/object.cpp:16-58, scene/triangle.cpp:9-21
          const PolymeshInfo &pm = *node.polymesh;
          std::vector<V3> vertices = pm.vertices;
          for (auto &v : vertices) v = projectTo3D(T * V4(v, 1));
          HalfedgeMesh mesh;
          mesh.build(pm.polygons, vertices);
          int mat = (int)S.materials.size();
          S.materials.push_back(pm.has_material ? pm.material : Material());
          for (size_t f = 0; f < mesh.n_faces; f++) {
            int h = mesh.f_he[f];
            int a = mesh.he[h].vertex;
            int b = mesh.he[mesh.he[h].next].vertex;
            int c = mesh.he[mesh.he[mesh.he[h].next].next].vertex;
            Primitive p;
            p.type = PRIM_TRIANGLE;
            p.p1 = mesh.v_pos[a]; p.p2 = mesh.v_pos[b]; p.p3 = mesh.v_pos[c];
            p.n1 = mesh.v_nrm[a]; p.n2 = mesh.v_nrm[b]; p.n3 = mesh.v_nrm[c];
            p.bb_min = p.p1; p.bb_max = p.p1;
            bb_expand(p.bb_min, p.bb_max, p.p2);
            bb_expand(p.bb_min, p.bb_max, p.p3);
            p.material = mat;
            S.prims.push_back(p);
          }
          // gl_scene/mesh.cpp:204-210: bbox over vertices in list order
          V3 mb_min(INF, INF, INF), mb_max(-INF, -INF, -INF);
          for (const V3 &v : mesh.v_pos) bb_expand(mb_min, mb_max, v);
          bb_expand(sb_min, sb_max, mb_min, mb_max);
          break;
        }
      }
    }
    S.bbox_min = sb_min;
    S.bbox_max = sb_max;
    S.bbox_empty = sb_min.x > sb_max.x || sb_min.y > sb_max.y || sb_min.z > sb_max.z;
    if (!S.bbox_empty) {
      V3 target = (sb_min + sb_max) / 2;  // BBox::centroid (bbox.h:96-98)
      V3 extent = sb_max - sb_min;
      double canonical_view_distance = norm(extent) / 2 * 1.5;
      double view_distance = canonical_view_distance * 2;
      double min_view_distance = canonical_view_distance / 10.0;
      double max_view_distance = canonical_view_distance * 20.0;
      S.camera.place(target, acos(c_dir.y), atan2(c_dir.x, c_dir.z), view_distance, min_view_distance,
                     max_view_distance);
    }
    auto t0 = std::chrono::steady_clock::now();
    build_bvh(&S, 4);
    S.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  } catch (const std::exception &e) {
    if (err) *err = std::string("scene error: ") + e.what();
    return -3;
  }
  return 0;
}

// --------------------------------------------------------------------------- BVH
namespace {

struct BvhBuilder {
  const std::vector<Primitive> &prims;
  std::vector<int> order;  // permutation being sorted, as the reference sorts its pointer vector
  std::vector<V3> centroid;
  std::vector<BvhNode> nodes;
  size_t max_leaf;

  BvhBuilder(const std::vector<Primitive> &p, size_t ml) : prims(p), max_leaf(ml) {
    order.resize(p.size());
    centroid.resize(p.size());
    for (size_t i = 0; i < p.size(); i++) {
      order[i] = (int)i;
      centroid[i] = (p[i].bb_min + p[i].bb_max) / 2;  // BBox::centroid
    }
  }

  void sort_axis(size_t s, size_t e, int axis) {
    const std::vector<V3> &c = centroid;
    // the same std::sort on the same sequence with the same strict-weak
    // comparator gives the same permutation as the reference's pointer sort
    if (axis == 0) std::sort(order.begin() + s, order.begin() + e, [&c](int a, int b) { return c[a].x < c[b].x; });
    else if (axis == 1) std::sort(order.begin() + s, order.begin() + e, [&c](int a, int b) { return c[a].y < c[b].y; });
    else std::sort(order.begin() + s, order.begin() + e, [&c](int a, int b) { return c[a].z < c[b].z; });
  }

  int construct(size_t s, size_t e) {  // src/scene/bvh.cpp:47-135
    V3 mn(INF, INF, INF), mx(-INF, -INF, -INF);
    for (size_t p = s; p < e; p++) bb_expand(mn, mx, prims[order[p]].bb_min, prims[order[p]].bb_max);
    BvhNode node;
    node.bb_min = mn;
    node.bb_max = mx;
    size_t n = e - s;
    if (n <= max_leaf) {
      node.start = (int32_t)s;
      node.count = (int32_t)n;
      node.l = node.r = -1;
    } else {
      size_t best_axis = 0, best_index = 0;
      double best_cost = INF;
      std::vector<V3> lmn(n + 1), lmx(n + 1), rmn(n + 1), rmx(n + 1);
      for (int axis = 0; axis < 3; axis++) {
        sort_axis(s, e, axis);
        V3 smn(INF, INF, INF), smx(-INF, -INF, -INF), emn(INF, INF, INF), emx(-INF, -INF, -INF);
        for (size_t p = 0; p < n; p++) {
          bb_expand(smn, smx, prims[order[s + p]].bb_min, prims[order[s + p]].bb_max);
          lmn[p] = smn;
          lmx[p] = smx;
        }
        for (size_t p = n; p-- > 0;) {
          bb_expand(emn, emx, prims[order[s + p]].bb_min, prims[order[s + p]].bb_max);
          rmn[p] = emn;
          rmx[p] = emx;
        }
        for (size_t p = 1; p < n; p++) {
          double cost = bb_surface_area(lmn[p - 1], lmx[p - 1]) * (double)(std::ptrdiff_t)p +
                        bb_surface_area(rmn[p], rmx[p]) * (double)(std::ptrdiff_t)(n - p);
          if (cost < best_cost) {
            best_cost = cost;
            best_axis = axis;
            best_index = p;
          }
        }
      }
      sort_axis(s, e, (int)best_axis);
      size_t mid = s + best_index;
      int l = construct(s, mid);
      int r = construct(mid, e);
      node.l = l;
      node.r = r;
      node.start = 0;
      node.count = 0;
    }
    nodes.push_back(node);
    return (int)nodes.size() - 1;
  }
};

}  // namespace

void build_bvh(StaticScene *scene, size_t max_leaf_size) {
  BvhBuilder b(scene->prims, max_leaf_size);
  scene->root = b.construct(0, scene->prims.size());
  std::vector<Primitive> sorted(scene->prims.size());
  for (size_t i = 0; i < sorted.size(); i++) sorted[i] = scene->prims[b.order[i]];
  scene->prims.swap(sorted);
  scene->nodes.swap(b.nodes);
}

void make_jitter_table(std::vector<double> *xy, size_t n) {
  // src/util/random_util.h:10-22 (default-seeded mt19937, scaled by 1/(2^32-1),
  // clamped) and src/pathtracer/sampler.h:78-89 (table construction order)
  std::mt19937 eng;
  const double rmax = 1.0 / (double)(eng.max() - eng.min());
  auto random_uniform = [&]() {
    double v = double(eng() - eng.min()) * rmax;
    return std::min(std::max(v, 0.0000001), 0.99999999);
  };
  xy->clear();
  for (size_t i = 0; i < n; i++)
    for (size_t j = 0; j < n; j++) {
      // the two draws are arguments of one constructor call; g++ evaluates
      // them right to left, so y is drawn before x
      double ry = random_uniform();
      double rx = random_uniform();
      xy->push_back((i + rx) / n);
      xy->push_back((j + ry) / n);
    }
}

}  // namespace ptamd
