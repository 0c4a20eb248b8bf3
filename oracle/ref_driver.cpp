// TEST INFRASTRUCTURE — headless driver that links the *unmodified reference
// sources* (compiled where they lie under /root/reference by oracle/Makefile,
// target `ref`) and dumps what the parity tests need: flattened scene + BVH
// arrays, camera, primary-ray hits, known-answer vectors, and full renders.
//
// It is never linked into, imported by or executed from the product path.
// Only tests/, tools/make_golden.py and bench.py's cpu_baseline leg run it.
//
// The reference's Application class cannot be compiled here (it needs GLU and
// the OSD text renderer), so the ~40 lines of scene/camera set-up it performs
// are restated below, each block citing the reference lines it follows.
//
// Built with -fno-access-control so that private members of the reference's
// classes (bvh->nodes, camera fields, ...) can be read for the dumps.

#include <getopt.h>
#include <unistd.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "CGL/CGL.h"
#include "pathtracer/bsdf.h"
#include "pathtracer/camera.h"
#include "pathtracer/pathtracer.h"
#include "pathtracer/raytraced_renderer.h"
#include "scene/bvh.h"
#include "scene/collada/collada.h"
#include "scene/gl_scene/ambient_light.h"
#include "scene/gl_scene/area_light.h"
#include "scene/gl_scene/directional_light.h"
#include "scene/gl_scene/mesh.h"
#include "scene/gl_scene/point_light.h"
#include "scene/gl_scene/scene.h"
#include "scene/gl_scene/sphere.h"
#include "scene/gl_scene/spot_light.h"
#include "scene/light.h"
#include "scene/sphere.h"
#include "scene/triangle.h"
#include "util/image.h"

#include "ptar.h"

using namespace CGL;
using namespace CGL::SceneObjects;
using Collada::CameraInfo;
using Collada::LightInfo;
using Collada::PolymeshInfo;
using Collada::SceneInfo;
using Collada::SphereInfo;

namespace {

struct RefApp {
  // Restates the state Application keeps for the headless path
  // (src/application/application.h:120-170).
  Camera camera;
  Camera canonicalCamera;
  GLScene::Scene *scene = nullptr;
  size_t screenW = 0, screenH = 0;
  double canonical_view_distance = 0;
  RaytracedRenderer *renderer = nullptr;
};

// src/application/application.cpp:98-104 (Application::init, dummy camera)
void app_init(RefApp &app) {
  app.screenW = 800;
  app.screenH = 600;
  CameraInfo cameraInfo;
  cameraInfo.hFov = 50;
  cameraInfo.vFov = 35;
  cameraInfo.nClip = 0.01;
  cameraInfo.fClip = 100;
  app.camera.configure(cameraInfo, app.screenW, app.screenH);
}

// src/application/application.cpp:318-339 (Application::init_light)
GLScene::SceneLight *app_init_light(LightInfo &light, const Matrix4x4 &transform) {
  switch (light.light_type) {
    case Collada::LightType::NONE: break;
    case Collada::LightType::AMBIENT: return new GLScene::AmbientLight(light);
    case Collada::LightType::DIRECTIONAL: return new GLScene::DirectionalLight(light, transform);
    case Collada::LightType::AREA: return new GLScene::AreaLight(light, transform);
    case Collada::LightType::POINT: return new GLScene::PointLight(light, transform);
    case Collada::LightType::SPOT: return new GLScene::SpotLight(light, transform);
    default: break;
  }
  return nullptr;
}

// src/application/application.cpp:229-305 (Application::load) together with
// init_camera (:307-313), init_sphere (:348-353), init_polymesh (:355-358).
void app_load(RefApp &app, SceneInfo *sceneInfo) {
  std::vector<Collada::Node> &nodes = sceneInfo->nodes;
  std::vector<GLScene::SceneLight *> lights;
  std::vector<GLScene::SceneObject *> objects;

  CameraInfo *c;
  Vector3D c_pos = Vector3D();
  Vector3D c_dir = Vector3D();

  int len = nodes.size();
  for (int i = 0; i < len; i++) {
    Collada::Node &node = nodes[i];
    Collada::Instance *instance = node.instance;
    const Matrix4x4 &transform = node.transform;
    switch (instance->type) {
      case Collada::Instance::CAMERA:
        c = static_cast<CameraInfo *>(instance);
        c_pos = (transform * Vector4D(c_pos, 1)).to3D();
        c_dir = (transform * Vector4D(c->view_dir, 1)).to3D().unit();
        app.camera.configure(*c, app.screenW, app.screenH);
        app.canonicalCamera.configure(*c, app.screenW, app.screenH);
        break;
      case Collada::Instance::LIGHT:
        lights.push_back(app_init_light(static_cast<LightInfo &>(*instance), transform));
        break;
      case Collada::Instance::SPHERE: {
        SphereInfo &sphere = static_cast<SphereInfo &>(*instance);
        const Vector3D &position = (transform * Vector4D(0, 0, 0, 1)).projectTo3D();
        double scale = (transform * Vector4D(1, 0, 0, 0)).to3D().norm();
        objects.push_back(new GLScene::Sphere(sphere, position, scale));
        break;
      }
      case Collada::Instance::POLYMESH:
        objects.push_back(new GLScene::Mesh(static_cast<PolymeshInfo &>(*instance), transform));
        break;
      case Collada::Instance::MATERIAL: break;
    }
  }

  app.scene = new GLScene::Scene(objects, lights);

  const BBox &bbox = app.scene->get_bbox();
  if (!bbox.empty()) {
    Vector3D target = bbox.centroid();
    app.canonical_view_distance = bbox.extent.norm() / 2 * 1.5;
    double view_distance = app.canonical_view_distance * 2;
    double min_view_distance = app.canonical_view_distance / 10.0;
    double max_view_distance = app.canonical_view_distance * 20.0;
    app.canonicalCamera.place(target, acos(c_dir.y), atan2(c_dir.x, c_dir.z),
                              view_distance, min_view_distance, max_view_distance);
    app.camera.place(target, acos(c_dir.y), atan2(c_dir.x, c_dir.z),
                     view_distance, min_view_distance, max_view_distance);
  }
}

// src/application/application.cpp:189-201 (Application::resize; mode is
// EDIT_MODE on the headless path, so the renderer is not touched here)
void app_resize(RefApp &app, size_t w, size_t h) {
  app.screenW = w;
  app.screenH = h;
  app.camera.set_screen_size(w, h);
}

// src/application/application.cpp:634-640 (Application::set_up_pathtracer)
void app_set_up_pathtracer(RefApp &app) {
  app.renderer->set_camera(&app.camera);
  app.renderer->set_scene(app.scene->get_static_scene());
  app.renderer->set_frame_size(app.screenW, app.screenH);
}

struct Args {
  size_t ns_aa = 1, max_ray_depth = 1, ns_area_light = 1, threads = 1;
  bool accum = true, hemisphere = false;
  size_t samples_per_batch = 32;
  float max_tolerance = 0.05f;
  size_t w = 0, h = 0;
  long cx = -1, cy = 0, cdx = 0, cdy = 0;  // -p cell
  std::string scene;
  std::string cam_in, cam_out;  // -c: Camera::load_settings before anything else; -D: Camera::dump_settings
  std::string png;  // -P: RaytracedRenderer::save_image writes the frame here (and its _rate.png)
};

RefApp *make_app(const Args &a) {
  SceneInfo *sceneInfo = new SceneInfo();
  if (Collada::ColladaParser::load(a.scene.c_str(), sceneInfo) < 0) {
    fprintf(stderr, "ref_driver: cannot load %s\n", a.scene.c_str());
    exit(2);
  }
  RefApp *app = new RefApp();
  // src/application/application.cpp:21-39 (Application ctor -> renderer)
  app->renderer = new RaytracedRenderer(a.ns_aa, a.max_ray_depth, a.accum, a.ns_area_light, 1, 1, 1,
                                        a.threads, a.samples_per_batch, a.max_tolerance, NULL,
                                        a.hemisphere, "ref", 0.0, 4.7);
  // src/application/main.cpp:215-221
  app_init(*app);
  app_load(*app, sceneInfo);
  if (a.w && a.h) app_resize(*app, a.w, a.h);
  app_set_up_pathtracer(*app);
  return app;
}

void put_v3(std::vector<double> &v, const Vector3D &p) {
  v.push_back(p.x);
  v.push_back(p.y);
  v.push_back(p.z);
}

int bsdf_kind(BSDF *b, double params[3]) {
  params[0] = params[1] = params[2] = 0;
  if (auto *d = dynamic_cast<DiffuseBSDF *>(b)) {
    params[0] = d->reflectance.x; params[1] = d->reflectance.y; params[2] = d->reflectance.z;
    return 0;
  }
  if (auto *e = dynamic_cast<EmissionBSDF *>(b)) {
    params[0] = e->radiance.x; params[1] = e->radiance.y; params[2] = e->radiance.z;
    return 1;
  }
  if (auto *m = dynamic_cast<MirrorBSDF *>(b)) {
    params[0] = m->reflectance.x; params[1] = m->reflectance.y; params[2] = m->reflectance.z;
    return 2;
  }
  if (dynamic_cast<MicrofacetBSDF *>(b)) return 3;
  if (auto *r = dynamic_cast<RefractionBSDF *>(b)) {
    params[0] = r->transmittance.x; params[1] = r->transmittance.y; params[2] = r->transmittance.z;
    return 4;
  }
  if (auto *g = dynamic_cast<GlassBSDF *>(b)) {
    params[0] = g->transmittance.x; params[1] = g->transmittance.y; params[2] = g->transmittance.z;
    return 5;
  }
  return -1;
}

void dump_camera(FILE *f, const Camera &cam) {
  std::vector<double> c;
  put_v3(c, cam.pos);
  for (int col = 0; col < 3; col++) put_v3(c, cam.c2w[col]);
  c.push_back(cam.hFov);
  c.push_back(cam.vFov);
  c.push_back(cam.nClip);
  c.push_back(cam.fClip);
  c.push_back((double)cam.screenW);
  c.push_back((double)cam.screenH);
  c.push_back(cam.screenDist);
  c.push_back(cam.ar);
  put_v3(c, cam.targetPos);
  c.push_back(cam.phi);
  c.push_back(cam.theta);
  c.push_back(cam.r);
  ptar_put(f, "camera", PTAR_F64, c.data(), c.size());
}

void dump_scene(RefApp *app, FILE *f) {
  RaytracedRenderer *R = app->renderer;
  BVHAccel *bvh = R->bvh;
  Scene *scene = R->scene;

  // scene-order primitive list, rebuilt exactly as build_accel does
  // (src/pathtracer/raytraced_renderer.cpp:351-375) to recover scene indices
  // the BVH's own copies are distinct objects, so match by geometry order:
  // BVHAccel copied the pointer vector (src/scene/bvh.cpp:17), hence pointer
  // identity with what get_primitives() returned during set_scene is lost for
  // us; we therefore index by (object, face) through a map built from a fresh
  // get_primitives() call compared by value.
  size_t n = bvh->primitives.size();
  std::vector<int32_t> ptype(n), pmat(n);
  std::vector<double> pos(n * 9, 0.0), nrm(n * 9, 0.0);
  std::map<BSDF *, int> matid;
  std::vector<BSDF *> mats;
  for (size_t i = 0; i < n; i++) {
    Primitive *p = bvh->primitives[i];
    BSDF *b = p->get_bsdf();
    if (!matid.count(b)) {
      matid[b] = (int)mats.size();
      mats.push_back(b);
    }
    pmat[i] = matid[b];
    if (Triangle *t = dynamic_cast<Triangle *>(p)) {
      ptype[i] = 0;
      const Vector3D *pv[3] = {&t->p1, &t->p2, &t->p3};
      const Vector3D *nv[3] = {&t->n1, &t->n2, &t->n3};
      for (int k = 0; k < 3; k++) {
        pos[i * 9 + k * 3 + 0] = pv[k]->x; pos[i * 9 + k * 3 + 1] = pv[k]->y; pos[i * 9 + k * 3 + 2] = pv[k]->z;
        nrm[i * 9 + k * 3 + 0] = nv[k]->x; nrm[i * 9 + k * 3 + 1] = nv[k]->y; nrm[i * 9 + k * 3 + 2] = nv[k]->z;
      }
    } else if (Sphere *s = dynamic_cast<Sphere *>(p)) {
      ptype[i] = 1;
      pos[i * 9 + 0] = s->o.x; pos[i * 9 + 1] = s->o.y; pos[i * 9 + 2] = s->o.z;
      pos[i * 9 + 3] = s->r; pos[i * 9 + 4] = s->r2;
    } else {
      ptype[i] = -1;
    }
  }
  ptar_put(f, "prim_type", PTAR_I32, ptype.data(), n);
  ptar_put(f, "prim_material", PTAR_I32, pmat.data(), n);
  ptar_put(f, "prim_pos", PTAR_F64, pos.data(), pos.size());
  ptar_put(f, "prim_nrm", PTAR_F64, nrm.data(), nrm.size());

  // materials are numbered in order of first appearance in BVH order
  std::vector<int32_t> mkind(mats.size());
  std::vector<double> mpar(mats.size() * 3);
  for (size_t i = 0; i < mats.size(); i++) mkind[i] = bsdf_kind(mats[i], &mpar[i * 3]);
  ptar_put(f, "mat_kind", PTAR_I32, mkind.data(), mkind.size());
  ptar_put(f, "mat_param", PTAR_F64, mpar.data(), mpar.size());

  size_t N = bvh->nodes.size();
  std::vector<double> nb(N * 6);
  std::vector<int32_t> nl(N), nr(N), ns(N), nc(N);
  for (size_t i = 0; i < N; i++) {
    const BVHNode &nd = bvh->nodes[i];
    nb[i * 6 + 0] = nd.bb.min.x; nb[i * 6 + 1] = nd.bb.min.y; nb[i * 6 + 2] = nd.bb.min.z;
    nb[i * 6 + 3] = nd.bb.max.x; nb[i * 6 + 4] = nd.bb.max.y; nb[i * 6 + 5] = nd.bb.max.z;
    if (nd.leaf) {
      nl[i] = -1; nr[i] = -1;
      ns[i] = (int32_t)(nd.start - bvh->primitives.begin());
      nc[i] = (int32_t)(nd.end - nd.start);
    } else {
      nl[i] = (int32_t)nd.l; nr[i] = (int32_t)nd.r; ns[i] = 0; nc[i] = 0;
    }
  }
  int32_t root = (int32_t)bvh->root;
  ptar_put(f, "node_bbox", PTAR_F64, nb.data(), nb.size());
  ptar_put(f, "node_left", PTAR_I32, nl.data(), N);
  ptar_put(f, "node_right", PTAR_I32, nr.data(), N);
  ptar_put(f, "node_start", PTAR_I32, ns.data(), N);
  ptar_put(f, "node_count", PTAR_I32, nc.data(), N);
  ptar_put(f, "root", PTAR_I32, &root, 1);

  // lights: kind 0 directional, 1 infinite hemisphere, 2 point, 3 spot (stub),
  // 4 area; 16 doubles each: radiance, position, direction, dim_x, dim_y, area
  size_t L = scene->lights.size();
  std::vector<int32_t> lk(L);
  std::vector<double> lp(L * 16, 0.0);
  for (size_t i = 0; i < L; i++) {
    SceneLight *l = scene->lights[i];
    double *q = &lp[i * 16];
    if (auto *a = dynamic_cast<AreaLight *>(l)) {
      lk[i] = 4;
      const Vector3D *v[5] = {&a->radiance, &a->position, &a->direction, &a->dim_x, &a->dim_y};
      for (int k = 0; k < 5; k++) { q[k * 3] = v[k]->x; q[k * 3 + 1] = v[k]->y; q[k * 3 + 2] = v[k]->z; }
      q[15] = a->area;
    } else if (auto *p = dynamic_cast<PointLight *>(l)) {
      lk[i] = 2;
      q[0] = p->radiance.x; q[1] = p->radiance.y; q[2] = p->radiance.z;
      q[3] = p->position.x; q[4] = p->position.y; q[5] = p->position.z;
    } else if (auto *d = dynamic_cast<DirectionalLight *>(l)) {
      lk[i] = 0;
      q[0] = d->radiance.x; q[1] = d->radiance.y; q[2] = d->radiance.z;
      q[6] = d->dirToLight.x; q[7] = d->dirToLight.y; q[8] = d->dirToLight.z;
    } else if (auto *h = dynamic_cast<InfiniteHemisphereLight *>(l)) {
      lk[i] = 1;
      q[0] = h->radiance.x; q[1] = h->radiance.y; q[2] = h->radiance.z;
    } else {
      lk[i] = 3;
    }
  }
  ptar_put(f, "light_kind", PTAR_I32, lk.data(), L);
  ptar_put(f, "light_param", PTAR_F64, lp.data(), lp.size());

  dump_camera(f, *R->camera);

  // the fixed pre-jittered 16x16 table (src/pathtracer/sampler.h:78-89)
  JitteredSampler2D *js = dynamic_cast<JitteredSampler2D *>(R->pt->gridSampler);
  std::vector<double> jt;
  for (const Vector2D &s : js->samples) { jt.push_back(s.x); jt.push_back(s.y); }
  ptar_put(f, "jitter_table", PTAR_F64, jt.data(), jt.size());
}

// Primary rays through fixed sub-pixel offsets; returns the BVH-order index of
// the hit primitive (or -1) and t. Follows raytrace_pixel's ray set-up
// (src/pathtracer/pathtracer.cpp:259-261) and the closest-hit call (:217).
void dump_primary(RefApp *app, FILE *f) {
  RaytracedRenderer *R = app->renderer;
  BVHAccel *bvh = R->bvh;
  size_t w = R->frame_w, h = R->frame_h;
  std::map<const Primitive *, int> index;
  for (size_t i = 0; i < bvh->primitives.size(); i++) index[bvh->primitives[i]] = (int)i;
  static const double offs[4][2] = {{0.5, 0.5}, {0.25, 0.75}, {0.8125, 0.1875}, {0.03125, 0.96875}};
  for (int o = 0; o < 4; o++) {
    std::vector<int32_t> ids(w * h);
    std::vector<double> ts(w * h);
    for (size_t y = 0; y < h; y++)
      for (size_t x = 0; x < w; x++) {
        Vector2D sample = Vector2D(x, y) + Vector2D(offs[o][0], offs[o][1]);
        Ray r = R->camera->generate_ray(sample.x / w, sample.y / h);
        r.depth = 1;
        Intersection isect;
        bool hit = bvh->intersect(r, &isect);
        ids[x + y * w] = hit ? index[isect.primitive] : -1;
        ts[x + y * w] = hit ? isect.t : -1.0;
      }
    char name[64];
    snprintf(name, sizeof name, "primary_id_%d", o);
    ptar_put(f, name, PTAR_I32, ids.data(), ids.size());
    snprintf(name, sizeof name, "primary_t_%d", o);
    ptar_put(f, name, PTAR_F64, ts.data(), ts.size());
  }
  ptar_put(f, "primary_offsets", PTAR_F64, &offs[0][0], 8);
}

// Known-answer vectors of the pure hot-path functions on scripted inputs.
void dump_kat(FILE *f) {
  // deterministic scripted inputs from a small LCG (not the reference RNG)
  uint64_t s = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return (double)(s >> 11) / 9007199254740992.0;
  };
  const int N = 256;
  // BBox::intersect (src/scene/bbox.cpp:10-47)
  {
    std::vector<double> in, out;
    for (int i = 0; i < N; i++) {
      Vector3D a(rnd() * 2 - 1, rnd() * 2 - 1, rnd() * 2 - 1), b(rnd() * 2 - 1, rnd() * 2 - 1, rnd() * 2 - 1);
      BBox bb(a);
      bb.expand(b);
      Vector3D o(rnd() * 4 - 2, rnd() * 4 - 2, rnd() * 4 - 2);
      Vector3D d(rnd() * 2 - 1, rnd() * 2 - 1, rnd() * 2 - 1);
      if (i % 16 == 3) d.x = 0;               // axis-parallel rays: inf inv_d
      if (i % 16 == 7) { d.y = 0; o.y = bb.min.y; }  // 0 * inf = NaN slab
      d.normalize();
      Ray r(o, d);
      r.min_t = (i % 5 == 0) ? 1e-11 : 0.0;
      r.max_t = (i % 3 == 0) ? rnd() * 3 : INF_D;
      double t0 = -7, t1 = -7;
      bool hit = bb.intersect(r, t0, t1);
      put_v3(in, bb.min); put_v3(in, bb.max); put_v3(in, o); put_v3(in, d);
      in.push_back(r.min_t); in.push_back(r.max_t);
      out.push_back(hit ? 1.0 : 0.0); out.push_back(t0); out.push_back(t1);
    }
    ptar_put(f, "kat_bbox_in", PTAR_F64, in.data(), in.size());
    ptar_put(f, "kat_bbox_out", PTAR_F64, out.data(), out.size());
  }
  // Triangle::intersect (src/scene/triangle.cpp:25-70). A Triangle needs a
  // Mesh; build one by hand through the public arrays.
  {
    std::vector<double> in, out;
    for (int i = 0; i < N; i++) {
      Triangle t;
      t.p1 = Vector3D(rnd() * 2 - 1, rnd() * 2 - 1, rnd() * 2 - 1);
      t.p2 = Vector3D(rnd() * 2 - 1, rnd() * 2 - 1, rnd() * 2 - 1);
      t.p3 = Vector3D(rnd() * 2 - 1, rnd() * 2 - 1, rnd() * 2 - 1);
      t.n1 = Vector3D(rnd(), rnd(), rnd());
      t.n2 = Vector3D(rnd(), rnd(), rnd());
      t.n3 = Vector3D(rnd(), rnd(), rnd());
      t.bsdf = nullptr;
      Vector3D o(rnd() * 4 - 2, rnd() * 4 - 2, rnd() * 4 - 2);
      // aim at a point near the triangle so that about half the rays hit
      double a = rnd() * 1.4 - 0.2, b = rnd() * 1.4 - 0.2;
      Vector3D target = t.p1 + a * (t.p2 - t.p1) + b * (t.p3 - t.p1);
      Vector3D d = (target - o).unit();
      Ray r(o, d);
      r.min_t = 1e-11;
      r.max_t = (i % 4 == 0) ? rnd() * 3 : INF_D;
      double max_t_in = r.max_t;
      Intersection is;
      is.t = -1; is.n = Vector3D(0, 0, 0);
      bool hit = t.intersect(r, &is);
      put_v3(in, t.p1); put_v3(in, t.p2); put_v3(in, t.p3);
      put_v3(in, t.n1); put_v3(in, t.n2); put_v3(in, t.n3);
      put_v3(in, o); put_v3(in, d);
      in.push_back(1e-11); in.push_back(max_t_in);
      out.push_back(hit ? 1.0 : 0.0); out.push_back(hit ? is.t : 0.0);
      out.push_back(hit ? is.n.x : 0.0); out.push_back(hit ? is.n.y : 0.0); out.push_back(hit ? is.n.z : 0.0);
      out.push_back(r.max_t);
    }
    ptar_put(f, "kat_tri_in", PTAR_F64, in.data(), in.size());
    ptar_put(f, "kat_tri_out", PTAR_F64, out.data(), out.size());
  }
  // Sphere::intersect (src/scene/sphere.cpp:11-72)
  {
    std::vector<double> in, out;
    for (int i = 0; i < N; i++) {
      Vector3D c(rnd() * 2 - 1, rnd() * 2 - 1, rnd() * 2 - 1);
      double rad = 0.1 + rnd();
      Sphere sp(nullptr, c, rad);
      Vector3D o(rnd() * 4 - 2, rnd() * 4 - 2, rnd() * 4 - 2);
      if (i % 8 == 1) o = c + Vector3D(rnd() - .5, rnd() - .5, rnd() - .5) * rad;  // inside
      Vector3D target = c + Vector3D(rnd() * 2 - 1, rnd() * 2 - 1, rnd() * 2 - 1) * rad * 1.2;
      Vector3D d = (target - o).unit();
      Ray r(o, d);
      r.min_t = 1e-11;
      r.max_t = (i % 4 == 0) ? rnd() * 3 : INF_D;
      double max_t_in = r.max_t;
      Intersection is;
      is.t = -1;
      // bsdf lookup dereferences the owning object; avoid it by calling the
      // test + branch logic through intersect only when object is set. The
      // reference's get_bsdf() reads object->get_bsdf(), so give it one.
      SphereObject so(c, rad, nullptr);
      sp.object = &so;
      bool hit = sp.intersect(r, &is);
      put_v3(in, c); in.push_back(rad); put_v3(in, o); put_v3(in, d);
      in.push_back(1e-11); in.push_back(max_t_in);
      out.push_back(hit ? 1.0 : 0.0); out.push_back(hit ? is.t : 0.0);
      out.push_back(hit ? is.n.x : 0.0); out.push_back(hit ? is.n.y : 0.0); out.push_back(hit ? is.n.z : 0.0);
      out.push_back(r.max_t);
    }
    ptar_put(f, "kat_sphere_in", PTAR_F64, in.data(), in.size());
    ptar_put(f, "kat_sphere_out", PTAR_F64, out.data(), out.size());
  }
  // make_coord_space (src/pathtracer/bsdf.cpp:21-41)
  {
    std::vector<double> in, out;
    for (int i = 0; i < N; i++) {
      Vector3D n(rnd() * 2 - 1, rnd() * 2 - 1, rnd() * 2 - 1);
      if (i % 8 == 0) n = Vector3D(0, 1, 0);
      if (i % 8 == 1) n = Vector3D(1, 0, 0);
      if (i % 8 == 2) n = Vector3D(0, 0, -1);
      if (i % 8 == 3) n = n * 0.37;  // unnormalised shading normal
      Matrix3x3 m;
      make_coord_space(m, n);
      put_v3(in, n);
      for (int c = 0; c < 3; c++) put_v3(out, m[c]);
    }
    ptar_put(f, "kat_frame_in", PTAR_F64, in.data(), in.size());
    ptar_put(f, "kat_frame_out", PTAR_F64, out.data(), out.size());
  }
  // HDRImageBuffer::toColor (src/util/image.h:183-198)
  {
    HDRImageBuffer hb(N, 1);
    std::vector<double> in;
    for (int i = 0; i < N; i++) {
      Vector3D v(rnd() * 1.5, rnd() * 0.2, rnd() * 3 - 1);
      if (i == 5) v = Vector3D(0, 0, 0);
      if (i == 6) v = Vector3D(NAN, 1e300, -0.0);
      hb.update_pixel(v, i, 0);
      put_v3(in, v);
    }
    ImageBuffer ib(N, 1);
    hb.toColor(ib, 0, 0, N, 1);
    std::vector<int32_t> out(N);
    for (int i = 0; i < N; i++) out[i] = (int32_t)ib.data[i];
    ptar_put(f, "kat_tonemap_in", PTAR_F64, in.data(), in.size());
    ptar_put(f, "kat_tonemap_out", PTAR_I32, out.data(), out.size());
  }
}

void dump_render(RefApp *app, FILE *f, const Args &a) {
  RaytracedRenderer *R = app->renderer;
  auto t0 = std::chrono::steady_clock::now();
  {
    // RaytracedRenderer::render_to_file minus the PNG write
    // (src/pathtracer/raytraced_renderer.cpp:331-336)
    std::unique_lock<std::mutex> lk(R->m_done);
    if (a.cx >= 0) {
      // the cell set-up of render_to_file / raytrace_cell
      // (src/pathtracer/raytraced_renderer.cpp:339-343, 612-626)
      R->render_cell = true;
      R->cell_tl = Vector2D(a.cx, a.cy);
      R->cell_br = Vector2D(a.cx + a.cdx, a.cy + a.cdy);
    }
    R->start_raytracing();
    R->cv_done.wait(lk, [R] { return R->state == RaytracedRenderer::DONE; });
    lk.unlock();
  }
  auto t1 = std::chrono::steady_clock::now();
  double secs = std::chrono::duration<double>(t1 - t0).count();
  size_t w = R->frame_w, h = R->frame_h;
  std::vector<double> hdr(w * h * 3);
  for (size_t i = 0; i < w * h; i++) {
    const Vector3D &v = R->pt->sampleBuffer.data[i];
    hdr[i * 3] = v.x; hdr[i * 3 + 1] = v.y; hdr[i * 3 + 2] = v.z;
  }
  std::vector<int32_t> fb(w * h);
  for (size_t i = 0; i < w * h; i++) fb[i] = (int32_t)R->frameBuffer.data[i];
  long long total = 0;
  for (int c : R->pt->sampleCountBuffer) total += c;
  double stats[6] = {secs, (double)R->bvh->total_rays, (double)R->bvh->total_isects, (double)total,
                     (double)a.threads, (double)(w * h)};
  ptar_put(f, "hdr", PTAR_F64, hdr.data(), hdr.size());
  ptar_put(f, "count", PTAR_I32, R->pt->sampleCountBuffer.data(), w * h);
  ptar_put(f, "rgba8", PTAR_I32, fb.data(), fb.size());
  ptar_put(f, "stats", PTAR_F64, stats, 6);
  fprintf(stderr, "ref_driver: %lld samples in %.3f s = %.4g samples/s (%d threads)\n", total, secs,
          total / secs, (int)a.threads);
  // the reference's own output stage: the PNG and its "_rate.png" (raytraced_renderer.cpp:680-751)
  if (!a.png.empty()) R->save_image(a.png);
  // the worker threads are joined by stop() in the reference; leave them
  R->stop();
}

void usage() {
  fprintf(stderr,
          "usage: ref_driver <dump|primary|render|kat> [-s spp -l light -m depth -o accum -t threads\n"
          "                  -a batch tol -H -r W H -p x y dx dy -P frame.png -c cam_in.txt -D cam_out.txt] -f out.ptar [scene.dae]\n");
}

}  // namespace

int main(int argc, char **argv) {
  if (argc < 2) { usage(); return 1; }
  std::string mode = argv[1];
  Args a;
  std::string out;
  int opt;
  optind = 2;
  // same option letters and two-word handling as src/application/main.cpp:126-187
  while ((opt = getopt(argc, argv, "s:l:t:m:o:f:r:a:Hp:P:c:D:")) != -1) {
    switch (opt) {
      case 'f': out = optarg; break;
      case 'r': a.w = atoi(argv[optind - 1]); a.h = atoi(argv[optind]); optind++; break;
      case 's': a.ns_aa = atoi(optarg); break;
      case 'l': a.ns_area_light = atoi(optarg); break;
      case 't': a.threads = atoi(optarg); break;
      case 'm': a.max_ray_depth = atoi(optarg); break;
      case 'o': a.accum = atoi(optarg) > 0; break;
      case 'a': a.samples_per_batch = atoi(argv[optind - 1]); a.max_tolerance = atof(argv[optind]); optind++; break;
      case 'H': a.hemisphere = true; break;
      case 'P': a.png = optarg; break;
      case 'c': a.cam_in = optarg; break;
      case 'D': a.cam_out = optarg; break;
      case 'p':
        a.cx = atoi(argv[optind - 1]); a.cy = atoi(argv[optind]);
        a.cdx = atoi(argv[optind + 1]); a.cdy = atoi(argv[optind + 2]);
        optind += 3;
        break;
      default: usage(); return 1;
    }
  }
  if (out.empty()) { usage(); return 1; }
  FILE *f = ptar_open_write(out.c_str());
  if (!f) { fprintf(stderr, "cannot write %s\n", out.c_str()); return 2; }
  if (mode == "kat") {
    dump_kat(f);
    fclose(f);
    return 0;
  }
  if (optind >= argc) { usage(); return 1; }
  a.scene = argv[optind];
  RefApp *app = make_app(a);
  // main.cpp:220-224: the settings file is loaded after the resize
  if (!a.cam_in.empty()) app->camera.load_settings(a.cam_in);
  if (!a.cam_out.empty()) app->camera.dump_settings(a.cam_out);
  if (mode == "dump") {
    dump_scene(app, f);
  } else if (mode == "primary") {
    dump_scene(app, f);
    dump_primary(app, f);
  } else if (mode == "render") {
    dump_camera(f, *app->renderer->camera);
    dump_render(app, f, a);
  } else {
    usage();
    return 1;
  }
  fclose(f);
  fflush(stdout);
  _exit(0);  // skip the reference's static destructors (primitives are never freed there)
}
