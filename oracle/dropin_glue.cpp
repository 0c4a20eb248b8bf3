// TEST INFRASTRUCTURE — the drop-in, compiled: the reference's own host code (its COLLADA
// loader, scene objects, BVHAccel build, Camera, RaytracedRenderer) with ONE member function
// replaced, RaytracedRenderer::start_raytracing (src/pathtracer/raytraced_renderer.cpp:268-329),
// which here hands the frame to libptamd.so through the C ABI instead of filling the tile queue
// and starting worker threads. It is the binding a maintainer of the reference would add,
// built by `make -C oracle dropin`: the
// reference's translation units are compiled unmodified where they lie; the object of
// raytraced_renderer.cpp gets its start_raytracing symbol weakened (objcopy) so that the
// definition below takes its place at link time.
//
// Built with -fno-access-control (the function below is a member of the reference's class, but
// the flattening also reads private members of BVHAccel, Camera and the sampler).
#include <cstdio>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

#include "CGL/CGL.h"
#include "pathtracer/bsdf.h"
#include "pathtracer/camera.h"
#include "pathtracer/pathtracer.h"
#include "pathtracer/raytraced_renderer.h"
#include "pathtracer/sampler.h"
#include "scene/bvh.h"
#include "scene/light.h"
#include "scene/scene.h"
#include "scene/sphere.h"
#include "scene/triangle.h"

#include "../include/ptamd.h"

using namespace CGL;
using namespace CGL::SceneObjects;

namespace {

ptamd_context *g_dev = nullptr;
const BVHAccel *g_uploaded = nullptr;  // the tree the device holds

void append_material(BSDF *b, std::vector<int32_t> &kind, std::vector<double> &param) {
  Vector3D v(0, 0, 0);
  int k = PTAMD_MAT_MICROFACET;
  if (auto *d = dynamic_cast<DiffuseBSDF *>(b)) { k = PTAMD_MAT_DIFFUSE; v = d->reflectance; }
  else if (auto *e = dynamic_cast<EmissionBSDF *>(b)) { k = PTAMD_MAT_EMISSION; v = e->radiance; }
  else if (auto *m = dynamic_cast<MirrorBSDF *>(b)) { k = PTAMD_MAT_MIRROR; v = m->reflectance; }
  else if (auto *r = dynamic_cast<RefractionBSDF *>(b)) { k = PTAMD_MAT_REFRACTION; v = r->transmittance; }
  else if (auto *g = dynamic_cast<GlassBSDF *>(b)) { k = PTAMD_MAT_GLASS; v = g->transmittance; }
  kind.push_back(k);
  param.push_back(v.x); param.push_back(v.y); param.push_back(v.z);
}

void append_light(SceneLight *l, std::vector<int32_t> &kind, std::vector<double> &param) {
  double q[16] = {0};
  int k = PTAMD_LIGHT_SPOT;
  auto put = [&](int at, const Vector3D &v) { q[at] = v.x; q[at + 1] = v.y; q[at + 2] = v.z; };
  if (auto *a = dynamic_cast<AreaLight *>(l)) {
    k = PTAMD_LIGHT_AREA;
    put(0, a->radiance); put(3, a->position); put(6, a->direction); put(9, a->dim_x); put(12, a->dim_y);
    q[15] = a->area;
  } else if (auto *p = dynamic_cast<PointLight *>(l)) {
    k = PTAMD_LIGHT_POINT;
    put(0, p->radiance); put(3, p->position);
  } else if (auto *d = dynamic_cast<DirectionalLight *>(l)) {
    k = PTAMD_LIGHT_DIRECTIONAL;
    put(0, d->radiance); put(6, d->dirToLight);
  } else if (auto *h = dynamic_cast<InfiniteHemisphereLight *>(l)) {
    k = PTAMD_LIGHT_INF_HEMISPHERE;
    put(0, h->radiance);
  }
  kind.push_back(k);
  param.insert(param.end(), q, q + 16);
}

// what build_accel has just built (raytraced_renderer.cpp:351-375), flattened for the device
int upload_to_gpu(RaytracedRenderer *R) {
  BVHAccel *bvh = R->bvh;
  std::vector<int32_t> type, mat, left, right, start, count, mkind, lkind;
  std::vector<double> pos, nrm, bbox, mparam, lparam, jitter;
  std::map<BSDF *, int> mat_index;
  for (Primitive *p : bvh->primitives) {  // BVH order, not scene order
    if (auto *t = dynamic_cast<Triangle *>(p)) {
      type.push_back(PTAMD_PRIM_TRIANGLE);
      for (const Vector3D *v : {&t->p1, &t->p2, &t->p3}) { pos.push_back(v->x); pos.push_back(v->y); pos.push_back(v->z); }
      for (const Vector3D *v : {&t->n1, &t->n2, &t->n3}) { nrm.push_back(v->x); nrm.push_back(v->y); nrm.push_back(v->z); }
    } else if (auto *s = dynamic_cast<Sphere *>(p)) {
      type.push_back(PTAMD_PRIM_SPHERE);
      const double sp[9] = {s->o.x, s->o.y, s->o.z, s->r, s->r2, 0, 0, 0, 0};
      pos.insert(pos.end(), sp, sp + 9);
      nrm.insert(nrm.end(), 9, 0.0);
    } else {
      fprintf(stderr, "dropin: unknown primitive class\n");
      return 1;
    }
    BSDF *b = p->get_bsdf();
    if (!mat_index.count(b)) { mat_index[b] = (int)mkind.size(); append_material(b, mkind, mparam); }
    mat.push_back(mat_index[b]);
  }
  for (const BVHNode &n : bvh->nodes) {  // post-order array, root = bvh->root
    const double bb[6] = {n.bb.min.x, n.bb.min.y, n.bb.min.z, n.bb.max.x, n.bb.max.y, n.bb.max.z};
    bbox.insert(bbox.end(), bb, bb + 6);
    left.push_back(n.leaf ? -1 : (int)n.l);
    right.push_back(n.leaf ? -1 : (int)n.r);
    start.push_back(n.leaf ? (int)(n.start - bvh->primitives.begin()) : 0);
    count.push_back(n.leaf ? (int)(n.end - n.start) : 0);
  }
  for (SceneLight *l : R->scene->lights) append_light(l, lkind, lparam);
  JitteredSampler2D *js = dynamic_cast<JitteredSampler2D *>(R->pt->gridSampler);
  for (const Vector2D &s : js->samples) { jitter.push_back(s.x); jitter.push_back(s.y); }

  ptamd_scene_desc d = {};
  d.n_prims = (int32_t)type.size(); d.n_nodes = (int32_t)left.size(); d.n_materials = (int32_t)mkind.size();
  d.n_lights = (int32_t)lkind.size(); d.root = (int32_t)bvh->root; d.jitter_n = 16;
  d.prim_type = type.data(); d.prim_pos = pos.data(); d.prim_nrm = nrm.data(); d.prim_material = mat.data();
  d.node_bbox = bbox.data(); d.node_left = left.data(); d.node_right = right.data();
  d.node_start = start.data(); d.node_count = count.data();
  d.mat_kind = mkind.data(); d.mat_param = mparam.data();
  d.light_kind = lkind.data(); d.light_param = lparam.data(); d.jitter_table = jitter.data();
  if (!g_dev && ptamd_create(0, &g_dev)) { fprintf(stderr, "dropin: %s\n", ptamd_last_error()); return 1; }
  if (ptamd_upload_scene(g_dev, &d)) { fprintf(stderr, "dropin: %s\n", ptamd_last_error()); return 1; }
  g_uploaded = bvh;
  return 0;
}

}  // namespace

namespace CGL {

void RaytracedRenderer::start_raytracing() {
  if (state != READY) return;
  if (g_uploaded != bvh && upload_to_gpu(this)) return;
  state = RENDERING;
  pt->clear();
  pt->set_frame_size(frameBuffer.w, frameBuffer.h);
  pt->bvh = bvh;
  pt->camera = camera;
  pt->scene = scene;
  ptamd_camera cam = {};  // what Camera::generate_ray reads (camera.cpp:190-203)
  cam.pos[0] = camera->pos.x; cam.pos[1] = camera->pos.y; cam.pos[2] = camera->pos.z;
  for (int c = 0; c < 3; c++) { cam.c2w[3 * c] = camera->c2w[c].x; cam.c2w[3 * c + 1] = camera->c2w[c].y; cam.c2w[3 * c + 2] = camera->c2w[c].z; }
  cam.hFov = camera->hFov; cam.vFov = camera->vFov; cam.nClip = camera->nClip; cam.fClip = camera->fClip;
  ptamd_set_camera(g_dev, &cam);

  ptamd_render_params p = {};
  p.w = (int32_t)frameBuffer.w; p.h = (int32_t)frameBuffer.h;
  p.x0 = render_cell ? (int32_t)cell_tl.x : 0; p.y0 = render_cell ? (int32_t)cell_tl.y : 0;
  p.x1 = render_cell ? (int32_t)cell_br.x : p.w; p.y1 = render_cell ? (int32_t)cell_br.y : p.h;
  if (!render_cell) frameBuffer.clear();
  p.ns_aa = (int32_t)pt->ns_aa; p.max_ray_depth = (int32_t)pt->max_ray_depth; p.accumulate_bounces = pt->isAccumBounces;
  p.ns_area_light = (int32_t)pt->ns_area_light; p.samples_per_batch = (int32_t)pt->samplesPerBatch;
  p.max_tolerance = pt->maxTolerance; p.hemisphere = pt->direct_hemisphere_sample; p.seed = 0;
  std::vector<double> hdr((size_t)3 * p.w * p.h);
  std::vector<int32_t> cnt((size_t)p.w * p.h);
  fprintf(stdout, "[PathTracer] Rendering... "); fflush(stdout);
  if (ptamd_render(g_dev, &p, hdr.data(), cnt.data())) {
    fprintf(stderr, "dropin: %s\n", ptamd_last_error());
    state = READY;
    return;
  }
  for (size_t i = 0; i < (size_t)p.w * p.h; i++) {  // same indexing: x + y*w, y = 0 at the bottom
    pt->sampleBuffer.data[i] = Vector3D(hdr[3 * i], hdr[3 * i + 1], hdr[3 * i + 2]);
    pt->sampleCountBuffer[i] = cnt[i];
  }
  pt->write_to_framebuffer(frameBuffer, p.x0, p.y0, p.x1, p.y1);
  ptamd_stats st;
  ptamd_get_stats(g_dev, &st);
  bvh->total_rays = st.rays;
  fprintf(stdout, "100%%! (%.4fs)\n", st.seconds);
  // stop() joins and deletes numWorkerThreads threads when the state is DONE (raytraced_renderer.cpp:
  // 230-237): give it threads to join
  for (int i = 0; i < numWorkerThreads; i++) workerThreads[i] = new std::thread([] {});
  // render_to_file calls this function with m_done held and then waits for state == DONE
  // (raytraced_renderer.cpp:331-336): the frame is finished before it gets to wait, so the state is
  // simply set (taking m_done here, as worker_thread does at :674-676, would deadlock)
  state = DONE;
  cv_done.notify_one();
}

}  // namespace CGL
