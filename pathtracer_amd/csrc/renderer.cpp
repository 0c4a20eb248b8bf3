#include "renderer.h"

#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <ctime>
#include <sstream>

extern "C" int ptamd_write_png_top_down(const char *path, const uint32_t *rgba, int32_t w, int32_t h);

namespace ptamd {

void HDRImageBuffer::toColor(ImageBuffer &target, size_t x0, size_t y0, size_t x1, size_t y1) const {
  // row-wise so that ptamd_tonemap_rgba8 (src/util/image.h:183-198) does the arithmetic
  for (size_t y = y0; y < y1 && y < h; ++y) {
    if (x1 <= x0) break;
    ptamd_tonemap_rgba8(&data[(x0 + y * w) * 3], (int32_t)(std::min(x1, w) - x0), 1, &target.data[x0 + y * w]);
  }
}

PathTracer::PathTracer() {}
PathTracer::~PathTracer() {}

void PathTracer::set_frame_size(size_t width, size_t height) {  // pathtracer.cpp:27-30
  sampleBuffer.resize(width, height);
  sampleCountBuffer.assign(width * height, 0);
}

void PathTracer::clear() {  // pathtracer.cpp:32-40
  bvh = nullptr;
  scene = nullptr;
  camera = nullptr;
  sampleBuffer.resize(0, 0);
  sampleCountBuffer.clear();
}

void PathTracer::write_to_framebuffer(ImageBuffer &framebuffer, size_t x0, size_t y0, size_t x1, size_t y1) {
  sampleBuffer.toColor(framebuffer, x0, y0, x1, y1);  // pathtracer.cpp:42-45
}

ptamd_render_params PathTracer::make_params(size_t x0, size_t y0, size_t x1, size_t y1) const {
  ptamd_render_params p{};
  p.w = (int32_t)sampleBuffer.w;
  p.h = (int32_t)sampleBuffer.h;
  p.x0 = (int32_t)x0; p.y0 = (int32_t)y0; p.x1 = (int32_t)x1; p.y1 = (int32_t)y1;
  p.ns_aa = (int32_t)ns_aa;
  p.max_ray_depth = (int32_t)max_ray_depth;
  p.accumulate_bounces = isAccumBounces ? 1 : 0;
  p.ns_area_light = (int32_t)ns_area_light;
  p.samples_per_batch = (int32_t)samplesPerBatch;
  p.hemisphere = direct_hemisphere_sample ? 1 : 0;
  p.max_tolerance = maxTolerance;
  p.seed = seed;
  p.specular = specular ? 1 : 0;
  return p;
}

// libptamd_multi.so is loaded only when several GPUs are asked for: it links RCCL, which a
// single-GPU run has no use for.
struct MultiGpu {
  void *lib = nullptr;
  ptamd_multi *h = nullptr;
  int (*create)(const int32_t *, int32_t, ptamd_multi **) = nullptr;
  void (*destroy)(ptamd_multi *) = nullptr;
  int (*upload_scene)(ptamd_multi *, const ptamd_scene_desc *) = nullptr;
  int (*set_camera)(ptamd_multi *, const ptamd_camera *) = nullptr;
  int (*render)(ptamd_multi *, const ptamd_render_params *, double *, int32_t *) = nullptr;
  int (*get_stats)(ptamd_multi *, int32_t, ptamd_stats *) = nullptr;
  int (*set_profiling)(ptamd_multi *, int32_t, int32_t) = nullptr;
  std::vector<int32_t> devices;
  ~MultiGpu() {
    if (h && destroy) destroy(h);
    if (lib) dlclose(lib);
  }
  bool load(std::string &error) {
    Dl_info info;
    std::string dir = ".";
    if (dladdr((void *)&ptamd_create, &info) && info.dli_fname) {
      dir = info.dli_fname;
      const size_t slash = dir.find_last_of('/');
      dir = slash == std::string::npos ? "." : dir.substr(0, slash);
    }
    const std::string path = dir + "/libptamd_multi.so";
    lib = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!lib) { error = std::string("cannot load ") + path + ": " + dlerror(); return false; }
    create = (decltype(create))dlsym(lib, "ptamd_multi_create");
    destroy = (decltype(destroy))dlsym(lib, "ptamd_multi_destroy");
    upload_scene = (decltype(upload_scene))dlsym(lib, "ptamd_multi_upload_scene");
    set_camera = (decltype(set_camera))dlsym(lib, "ptamd_multi_set_camera");
    render = (decltype(render))dlsym(lib, "ptamd_multi_render");
    get_stats = (decltype(get_stats))dlsym(lib, "ptamd_multi_get_stats");
    set_profiling = (decltype(set_profiling))dlsym(lib, "ptamd_multi_set_profiling");
    if (!create || !destroy || !upload_scene || !set_camera || !render || !get_stats || !set_profiling) {
      error = path + " lacks an entry point";
      return false;
    }
    return true;
  }
};

int PathTracer::raytrace_rect(size_t x0, size_t y0, size_t x1, size_t y1) {
  if (!bvh && !multi) return PTAMD_ERR_STATE;
  const size_t w = sampleBuffer.w, h = sampleBuffer.h;
  ptamd_render_params p = make_params(x0, y0, x1, y1);
  std::vector<double> hdr(w * h * 3);
  std::vector<int32_t> cnt(w * h);
  int rc = multi ? multi->render(multi->h, &p, hdr.data(), cnt.data()) : ptamd_render(bvh, &p, hdr.data(), cnt.data());
  if (rc) return rc;
  for (size_t y = y0; y < y1 && y < h; y++)
    for (size_t x = x0; x < x1 && x < w; x++) {
      const size_t i = x + y * w;
      sampleBuffer.data[3 * i] = hdr[3 * i];
      sampleBuffer.data[3 * i + 1] = hdr[3 * i + 1];
      sampleBuffer.data[3 * i + 2] = hdr[3 * i + 2];
      sampleCountBuffer[i] = cnt[i];
    }
  return PTAMD_OK;
}

int PathTracer::raytrace_pixel(size_t x, size_t y) { return raytrace_rect(x, y, x + 1, y + 1); }

// ---------------------------------------------------------------------------
RaytracedRenderer::RaytracedRenderer(size_t ns_aa, size_t max_ray_depth, bool isAccumBounces, size_t ns_area_light,
                                     size_t ns_diff, size_t ns_glsy, size_t ns_refr, size_t num_threads,
                                     size_t samples_per_batch, float max_tolerance, void *envmap,
                                     bool direct_hemisphere_sample, std::string filename_, double lensRadius_,
                                     double focalDistance_, int device_) {
  // raytraced_renderer.cpp:40-86
  (void)envmap;  // the reference's environment light is an unfinished stub that returns black
  state = INIT;
  pt = new PathTracer();
  pt->ns_aa = ns_aa;
  pt->max_ray_depth = max_ray_depth;
  pt->isAccumBounces = isAccumBounces;
  pt->ns_area_light = ns_area_light;
  pt->ns_diff = ns_diff;
  pt->ns_glsy = ns_diff;  // sic: the reference copies ns_diff here (raytraced_renderer.cpp:61)
  (void)ns_glsy;
  pt->ns_refr = ns_refr;
  pt->samplesPerBatch = samples_per_batch;
  pt->maxTolerance = max_tolerance;
  pt->direct_hemisphere_sample = direct_hemisphere_sample;
  lensRadius = lensRadius_;
  focalDistance = focalDistance_;
  filename = filename_;
  imageTileSize = 32;
  numWorkerThreads = num_threads;
  device = device_;
}

bool RaytracedRenderer::set_devices(const std::vector<int> &device_ids) {
  if (state != INIT || bvh || multi || device_ids.empty()) return false;
  if (device_ids.size() == 1) { device = device_ids[0]; return true; }
  MultiGpu *m = new MultiGpu;
  if (!m->load(error)) {
    fprintf(stderr, "[PathTracer] %s\n", error.c_str());
    delete m;
    return false;
  }
  m->devices.assign(device_ids.begin(), device_ids.end());
  multi = m;
  return true;
}

RaytracedRenderer::~RaytracedRenderer() {
  delete multi;
  if (bvh) ptamd_destroy(bvh);
  if (scene) ptamd_scene_free(scene);
  delete pt;
}

void RaytracedRenderer::set_scene(ptamd_scene *new_scene) {  // raytraced_renderer.cpp:106-128
  if (state != INIT) return;
  if (scene != nullptr) {
    ptamd_scene_free(scene);
    if (bvh) ptamd_destroy(bvh);
    bvh = nullptr;
  }
  scene = new_scene;
  // build_accel (raytraced_renderer.cpp:351-375): the tree was built with the
  // scene; here it is flattened and copied to the device
  fprintf(stdout, "[PathTracer] Building BVH... Done! (%.4f sec)\n", ptamd_scene_build_seconds(scene));
  ptamd_scene_desc desc;
  int rc = ptamd_scene_get_desc(scene, &desc);
  if (multi) {
    if (multi->h) { multi->destroy(multi->h); multi->h = nullptr; }
    if (!rc) rc = multi->create(multi->devices.data(), (int32_t)multi->devices.size(), &multi->h);
    if (!rc) rc = multi->upload_scene(multi->h, &desc);
  } else {
    if (!rc) rc = ptamd_create(device, &bvh);
    if (!rc) rc = ptamd_upload_scene(bvh, &desc);
  }
  if (rc) {
    error = ptamd_last_error();
    fprintf(stderr, "[PathTracer] %s\n", error.c_str());
    if (bvh) ptamd_destroy(bvh);
    bvh = nullptr;
    return;
  }
  if (multi) fprintf(stdout, "[PathTracer] Uploaded %d primitives, %d BVH nodes to %d GPUs\n", desc.n_prims, desc.n_nodes, (int)multi->devices.size());
  else fprintf(stdout, "[PathTracer] Uploaded %d primitives, %d BVH nodes to GPU %d\n", desc.n_prims, desc.n_nodes, device);
  if (has_valid_configuration()) state = READY;
}

void RaytracedRenderer::set_camera(const ptamd_camera *cam) {  // raytraced_renderer.cpp:136-149
  if (state != INIT) return;
  camera = cam;
  if (has_valid_configuration()) state = READY;
}

void RaytracedRenderer::set_frame_size(size_t width, size_t height) {  // raytraced_renderer.cpp:159-177
  if (state != INIT && state != READY) stop();
  frame_w = width;
  frame_h = height;
  frameBuffer.resize(width, height);
  cell_x0 = 0; cell_y0 = 0; cell_x1 = width; cell_y1 = height;
  render_cell = false;
  pt->set_frame_size(width, height);
  if (has_valid_configuration()) state = READY;
}

void RaytracedRenderer::stop() {  // raytraced_renderer.cpp:215-236
  switch (state) {
    case INIT:
    case READY: break;
    case VISUALIZE:
    case RENDERING:
    case DONE: state = READY; break;
  }
}

void RaytracedRenderer::clear() {  // raytraced_renderer.cpp:241-254
  if (state != READY) return;
  if (bvh) ptamd_destroy(bvh);
  bvh = nullptr;
  if (scene) ptamd_scene_free(scene);
  scene = nullptr;
  camera = nullptr;
  frameBuffer.resize(0, 0);
  state = INIT;
  render_cell = false;
  pt->clear();
}

void RaytracedRenderer::start_raytracing() {  // raytraced_renderer.cpp:268-329
  if (state != READY) return;
  if (!bvh && !(multi && multi->h)) { error = "no device context"; return; }
  state = RENDERING;
  const size_t width = frameBuffer.w, height = frameBuffer.h;
  pt->clear();
  pt->set_frame_size(width, height);
  pt->bvh = bvh;
  pt->multi = multi;
  pt->camera = camera;
  pt->scene = scene;
  int rc = multi ? multi->set_camera(multi->h, camera) : ptamd_set_camera(bvh, camera);
  if (!rc) rc = multi ? multi->set_profiling(multi->h, 0, profile_counters ? 1 : 0) : ptamd_set_profiling(bvh, 0, profile_counters ? 1 : 0);
  size_t x0 = 0, y0 = 0, x1 = width, y1 = height;
  if (!render_cell) {
    frameBuffer.clear();
  } else {
    x0 = cell_x0; y0 = cell_y0; x1 = std::min(cell_x1, width); y1 = std::min(cell_y1, height);
  }
  fprintf(stdout, "[PathTracer] Rendering... ");
  fflush(stdout);
  // the worker pool's tile loop (raytraced_renderer.cpp:585-610, 644-658)
  // becomes one device call over the whole rectangle
  if (!rc) rc = pt->raytrace_rect(x0, y0, x1, y1);
  if (rc) {
    error = ptamd_last_error();
    fprintf(stdout, "\n[PathTracer] Rendering failed: %s\n", error.c_str());
    state = READY;
    return;
  }
  pt->write_to_framebuffer(frameBuffer, x0, y0, x1, y1);
  if (multi) {
    // the frame's wall time and the sums of the GPUs' counters
    ptamd_stats part{};
    stats = ptamd_stats{};
    for (size_t r = 0; r < multi->devices.size(); r++) {
      multi->get_stats(multi->h, (int32_t)r, &part);
      stats.seconds = part.seconds;
      stats.samples += part.samples; stats.rays += part.rays; stats.shadow_rays += part.shadow_rays;
      stats.path_vertices += part.path_vertices; stats.prim_tests += part.prim_tests; stats.nodes_visited += part.nodes_visited;
      stats.exact_tests += part.exact_tests; stats.two_ext_rays += part.two_ext_rays; stats.two_shadow_rays += part.two_shadow_rays;
      stats.unused_slots += part.unused_slots;
      for (int k = 0; k < PTAMD_K_CLASSES; k++) { stats.launches[k] += part.launches[k]; stats.kernel_ms[k] += part.kernel_ms[k]; }
    }
  } else {
    ptamd_get_stats(bvh, &stats);
  }
  // raytraced_renderer.cpp:668-672
  fprintf(stdout, "100%%! (%.4fs)\n", stats.seconds);
  fprintf(stdout, "[PathTracer] BVH traced %llu rays.\n", (unsigned long long)stats.rays);
  fprintf(stdout, "[PathTracer] Average speed %.4f million rays per second.\n", (double)stats.rays / stats.seconds * 1e-6);
  if (stats.prim_tests)
    fprintf(stdout, "[PathTracer] Averaged %f intersection tests per ray.\n", (double)stats.prim_tests / (double)stats.rays);
  fprintf(stdout, "[PathTracer] %llu path samples, %.4f million samples per second.\n", (unsigned long long)stats.samples,
          (double)stats.samples / stats.seconds * 1e-6);
  state = DONE;
}

void RaytracedRenderer::raytrace_cell(ImageBuffer &buffer) {  // raytraced_renderer.cpp:612-638
  const size_t w = cell_x1 - cell_x0, h = cell_y1 - cell_y0;
  buffer.resize(w, h);
  stop();
  render_cell = true;
  start_raytracing();
  if (state != DONE) return;
  for (size_t y = cell_y0; y < cell_y1 && y < frame_h; y++)
    for (size_t x = cell_x0; x < cell_x1 && x < frame_w; x++)
      buffer.data[w * (y - cell_y0) + (x - cell_x0)] = frameBuffer.data[x + y * frame_w];
}

void RaytracedRenderer::render_to_file(std::string fname, size_t x, size_t y, size_t dx, size_t dy) {
  // raytraced_renderer.cpp:331-348
  if (x == (size_t)-1) {
    start_raytracing();
    save_image(fname);
    if (state == DONE) fprintf(stdout, "[PathTracer] Job completed.\n");
  } else {
    render_cell = true;
    cell_x0 = x; cell_y0 = y; cell_x1 = x + dx; cell_y1 = y + dy;
    ImageBuffer buffer;
    raytrace_cell(buffer);
    save_image(fname, &buffer);
    if (state == DONE) fprintf(stdout, "[PathTracer] Cell job completed.\n");
  }
}

void RaytracedRenderer::save_image(std::string fname, ImageBuffer *buffer) {  // raytraced_renderer.cpp:680-718
  if (state != DONE) return;
  if (!buffer) buffer = &frameBuffer;
  if (fname == "") {
    time_t t = time(nullptr);
    tm *lt = localtime(&t);
    std::stringstream ss;
    ss << this->filename << "_screenshot_" << lt->tm_mon + 1 << "-" << lt->tm_mday << "_" << lt->tm_hour << "-"
       << lt->tm_min << "-" << lt->tm_sec << ".png";
    fname = ss.str();
  }
  fprintf(stderr, "[PathTracer] Saving to file: %s... ", fname.c_str());
  int rc = ptamd_write_png(fname.c_str(), buffer->data.data(), (int32_t)buffer->w, (int32_t)buffer->h);
  if (rc) {
    error = ptamd_last_error();
    fprintf(stderr, "failed: %s\n", error.c_str());
    return;
  }
  fprintf(stderr, "Done!\n");
  save_sampling_rate_image(fname);
}

void RaytracedRenderer::save_sampling_rate_image(std::string fname) {  // raytraced_renderer.cpp:720-751
  const size_t w = frameBuffer.w, h = frameBuffer.h;
  if (!w || !h || fname.size() < 4 || pt->ns_aa == 0) return;
  std::vector<uint32_t> out(w * h);
  if (ptamd_sampling_rate_rgba8(pt->sampleCountBuffer.data(), (int32_t)w, (int32_t)h, (int32_t)pt->ns_aa, out.data())) return;
  ptamd_write_png_top_down((fname.substr(0, fname.size() - 4) + "_rate.png").c_str(), out.data(), (int32_t)w, (int32_t)h);
}

}  // namespace ptamd
