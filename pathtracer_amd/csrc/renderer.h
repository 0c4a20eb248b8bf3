// Host-side mirror of the reference's render controller API
// (RaytracedRenderer: src/pathtracer/raytraced_renderer.h:60-250, implementing
// OfflineRenderer: src/application/renderer.h:14-99; PathTracer:
// src/pathtracer/pathtracer.h:23-105). Same class, method and field names and
// the same state machine; start_raytracing() hands the frame to the GPU through
// the C ABI instead of spawning worker threads.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/ptamd.h"

namespace ptamd {
struct MultiGpu;  // libptamd_multi.so's entry points and one ptamd_multi handle (renderer.cpp)

// src/util/image.h:20-100 (RGBA8, R in the low byte, row 0 = bottom)
struct ImageBuffer {
  size_t w = 0, h = 0;
  std::vector<uint32_t> data;
  void resize(size_t w_, size_t h_) { w = w_; h = h_; data.assign(w * h, 0); }
  void clear() { std::fill(data.begin(), data.end(), 0u); }
  bool is_empty() const { return w == 0 && h == 0; }
};

// src/util/image.h:105-200 (3 doubles per pixel)
struct HDRImageBuffer {
  size_t w = 0, h = 0;
  std::vector<double> data;
  void resize(size_t w_, size_t h_) { w = w_; h = h_; data.assign(w * h * 3, 0.0); }
  void clear() { std::fill(data.begin(), data.end(), 0.0); }
  // toColor over [x0,x1) x [y0,y1)
  void toColor(ImageBuffer &target, size_t x0, size_t y0, size_t x1, size_t y1) const;
};

class PathTracer {
 public:
  PathTracer();
  ~PathTracer();
  void set_frame_size(size_t width, size_t height);
  void write_to_framebuffer(ImageBuffer &framebuffer, size_t x0, size_t y0, size_t x1, size_t y1);
  void clear();
  // Traces one pixel on the GPU and stores it in sampleBuffer/sampleCountBuffer.
  // Returns a ptamd status.
  int raytrace_pixel(size_t x, size_t y);
  // Traces the half-open rectangle; the bulk form of raytrace_pixel.
  int raytrace_rect(size_t x0, size_t y0, size_t x1, size_t y1);

  // Integrator sampling settings (pathtracer.h:73-83)
  size_t max_ray_depth = 1;
  size_t isAccumBounces = 1;
  size_t ns_aa = 1;
  size_t ns_area_light = 1;
  size_t ns_diff = 1, ns_glsy = 1, ns_refr = 1;
  size_t samplesPerBatch = 32;
  double maxTolerance = 0.05f;
  bool direct_hemisphere_sample = false;
  uint64_t seed = 0;  // key of the counter-based RNG (the reference's RNG is unseeded)
  bool specular = false;  // ideal mirror / refraction / glass instead of the reference's black stubs

  // Components (pathtracer.h:87-97): the device context stands in for bvh
  ptamd_context *bvh = nullptr;
  MultiGpu *multi = nullptr;  // several GPUs: the renderer's (see RaytracedRenderer::set_devices)
  ptamd_scene *scene = nullptr;
  const ptamd_camera *camera = nullptr;
  HDRImageBuffer sampleBuffer;
  std::vector<int> sampleCountBuffer;

  ptamd_render_params make_params(size_t x0, size_t y0, size_t x1, size_t y1) const;
};

class RaytracedRenderer {
 public:
  // Render on several GPUs of this node (the counterpart of num_threads workers): call before
  // set_scene. Loads libptamd_multi.so (and with it RCCL) from the directory of libptamd.so.
  bool set_devices(const std::vector<int> &device_ids);

  enum State { INIT, READY, VISUALIZE, RENDERING, DONE };

  RaytracedRenderer(size_t ns_aa = 1, size_t max_ray_depth = 4, bool is_accumulate_bounces = false,
                    size_t ns_area_light = 1, size_t ns_diff = 1, size_t ns_glsy = 1, size_t ns_refr = 1,
                    size_t num_threads = 1, size_t samples_per_batch = 32, float max_tolerance = 0.05f,
                    void *envmap = nullptr, bool direct_hemisphere_sample = false, std::string filename = "",
                    double lensRadius = 0.25, double focalDistance = 4.7, int device = 0);
  ~RaytracedRenderer();

  void set_scene(ptamd_scene *scene);           // takes ownership
  void set_camera(const ptamd_camera *camera);  // borrowed
  void set_frame_size(size_t width, size_t height);
  void stop();
  void clear();
  void start_raytracing();
  void render_to_file(std::string filename, size_t x, size_t y, size_t dx, size_t dy);
  void save_image(std::string filename = "", ImageBuffer *buffer = nullptr);
  void save_sampling_rate_image(std::string filename);
  bool is_done() const { return state == DONE; }
  State get_state() const { return state; }
  const std::string &last_error() const { return error; }
  const ptamd_stats &get_stats() const { return stats; }

  PathTracer *pt;
  ptamd_context *bvh = nullptr;  // the device context holds the flattened BVH
  MultiGpu *multi = nullptr;     // set_devices with more than one GPU: the GPUs' contexts instead of bvh
  ptamd_scene *scene = nullptr;
  const ptamd_camera *camera = nullptr;
  ImageBuffer frameBuffer;
  size_t frame_w = 0, frame_h = 0;
  size_t imageTileSize = 32;
  size_t numWorkerThreads = 1;  // kept for API compatibility; the GPU replaces the pool
  bool render_cell = false;
  size_t cell_x0 = 0, cell_y0 = 0, cell_x1 = 0, cell_y1 = 0;
  double lensRadius, focalDistance;
  std::string filename;
  bool profile_counters = false;

 private:
  bool has_valid_configuration() const { return scene && camera; }
  void raytrace_cell(ImageBuffer &buffer);
  State state = INIT;
  int device = 0;
  std::string error;
  ptamd_stats stats{};
};

}  // namespace ptamd
