// Device layer of the render hot path for MI355X (gfx950): a wavefront path
// tracer. This file is its host side: the C ABI of include/ptamd.h, the buffers, and the
// round driver (ptamd_trace_list) that strings the kernels of trace_kernels.hip,
// shade_kernels.hip and frame_kernels.hip together. One frame is rendered in rounds of one
// or more batches of `samples_per_batch` samples per still-active pixel; each round runs,
// per path depth,
//
//   extend   closest-hit BVH traversal of the ray list       (BVHAccel::intersect)
//   shade    emission, Russian roulette, BSDF sample, bounce  (at_least_one_bounce_radiance)
//            ray; per-block counts
//   light    the vertices' light samples into fixed slots,    (estimate_direct_lighting_*)
//            one lane per sample
//   scan2, scatter   ordered compaction: next ray list, shadow-ray list
//   shadow   any-hit traversal of the shadow-ray list
//   resolve  ordered sum of the unoccluded light samples
//
// The shadow rays of depth d and the bounce rays of depth d + 1 are both ready once
// depth d is shaded and are traced by ONE launch (k_trace2); k_extend only traces
// the camera rays and k_shadow the last depth's shadow rays and lists too long to share
// a launch. Where the shading of a depth leaves the paths' sums alone, `resolve` of depth
// d runs as workgroups of the launch that draws depth d + 1's light samples
// (k_light_resolve): one is bound by memory round trips, the other by instruction issue.
//
// and finally `accumulate` folds the finished paths into the per-pixel sums in
// sample order, applying the adaptive test between batches, exactly as
// PathTracer::raytrace_pixel does (src/pathtracer/pathtracer.cpp:234-272 of the
// reference).
//
// All arithmetic that decides a result is FP64 in the reference's operation
// order and the device layer is compiled with -ffp-contract=off. Queue sizes live on
// the device; the host never synchronises inside a round.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ptamd.h"
#include "device_types.h"
#include "error_state.h"
#include "kernels.h"
#include "tree4.h"

using namespace ptdev;
using namespace ptk;

// ====================================================================== host side
#define HIPCHK(expr)                                                                       \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) {                                                                \
      ptamd::set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                 \
      return PTAMD_ERR_DEVICE;                                                             \
    }                                                                                      \
  } while (0)

// Frame-level buffers of ptamd_render / ptamd_render_list (see ensure_frame_scratch)
struct FrameScratch {
  int32_t *d_list[3] = {nullptr, nullptr, nullptr};  // two survivor lists in turn; [2]: ptamd_render's first list
  int64_t cap_list = 0;
  double *d_state = nullptr, *d_hdr = nullptr;
  int32_t *d_count = nullptr;
  int64_t cap_pix = 0;
  int64_t *d_n = nullptr, *h_n = nullptr;  // two list lengths on the device, their pinned host copies
  hipEvent_t ev[2] = {nullptr, nullptr};
};

struct ptamd_context {
  int device = 0;
  int num_cu = 256;
  hipStream_t own_stream = nullptr;
  // scene
  bool have_scene = false, have_camera = false, have_frame = false;
  DevScene S{};
  std::vector<void *> scene_allocs;
  std::vector<int32_t> dev2ref;  // device primitive index -> caller's primitive index
  std::vector<int32_t> host_light_kinds;
  int32_t n_prims = 0;
  ptamd_camera cam{};
  ptamd_render_params params{};
  FrameParams fp{};
  // work buffers
  bool clear_ctl = false;  // the next ptamd_trace_list zeroes the device-side counters first
  // RoundPlan::margin (PTAMD_PLAN_MARGIN in the environment: experiments). 1.1 to 1.35 time alike, 1.0 costs a
  // round more per frame, 1.5 traces 1 - 2 % more that is then dropped, 2 and 3 still more.
  double plan_margin = 1.25;
  int64_t predict_above = 1 << 20;  // rounds of fewer paths give every pixel all its samples (PTAMD_PREDICT_ABOVE)
  bool trace_two = true;  // bounce rays of depth d + 1 and shadow rays of depth d in one launch (PTAMD_TRACE_TWO=0: apart)
  uint32_t fuse_below = 16u << 20;  // ... when the two lists together are shorter than this (PTAMD_FUSE_BELOW)
  int64_t max_paths = 16 << 20;  // paths per launch: large launches amortise the drain of the slowest waves
  int64_t cap_paths = 0;
  int cap_nls = 0;
  bool cap_pend = false;
  bool cap_dup = false;  // the per-depth second copies exist (see fuse_resolve)
  void *slab = nullptr;
  RayQ q[3]{};  // two in turn; the third only where a depth's queue must outlive the next depth's shading (fuse_resolve)
  // k_resolve of depth d beside k_light of depth d + 1 (k_light_resolve): the records the two would share are then
  // kept per depth parity, [0] being the copies of vb / sq (PTAMD_FUSE_RESOLVE=0: one after the other)
  bool fuse_resolve = true;
  uint8_t *flag_p[2]{nullptr, nullptr};
  Contrib *contrib_p[2]{nullptr, nullptr};
  uint8_t *vis_p[2]{nullptr, nullptr};
  int32_t *pend_prim_p[2]{nullptr, nullptr};
  SRay *ray_p[2]{nullptr, nullptr};
  int32_t *list[2]{nullptr, nullptr};
  Compact cp{};
  HitBuf hb{};
  ShadowQ sq{};
  VertexBuf vb{};
  PathBuf pb{};
  uint32_t *pix_off = nullptr;     // PathMap::off of the running chunk [cap_paths + 1]
  uint32_t *plan_blocks = nullptr; // per-256-pixel sums of the planned counts [cap_paths / 256 + 2]
  uint32_t *slot_pix = nullptr;    // PathMap::slot_pix [cap_paths / 256 + 2]
  Ctl *ctl = nullptr;
  uint32_t *block_count = nullptr;
  size_t block_count_cap = 0;
  FrameScratch fs;
  double scene_max_coord = 0;
  uint32_t *static_block_count = nullptr;  // this GPU's copy of `scratch` (ptamd_filter_list_static)
  uint2 *gstack = nullptr;  // overflow levels of the per-lane traversal stacks
  size_t gstack_bytes = 0;
  // profiling
  bool prof_events = false, prof_counters = false;
  uint32_t event_mask = 0xffffffffu;  // kernel classes whose launches are bracketed by events
  struct Ev { int cls; hipEvent_t a, b; };
  std::vector<Ev> events;
  std::vector<hipEvent_t> event_pool;  // recycled: creating an event per launch costs more than the launch
  hipEvent_t take_event() {
    hipEvent_t e = nullptr;
    if (!event_pool.empty()) { e = event_pool.back(); event_pool.pop_back(); }
    else (void)hipEventCreate(&e);
    return e;
  }
  ptamd_stats stats{};
};

namespace {

void free_frame_scratch(ptamd_context *c);
void free_scene(ptamd_context *c) {
  for (void *p : c->scene_allocs) (void)hipFree(p);
  c->scene_allocs.clear();
  c->have_scene = false;
}

template <typename T>
int upload(ptamd_context *c, const std::vector<T> &h, const T **out) {
  void *d = nullptr;
  size_t bytes = std::max<size_t>(h.size() * sizeof(T), 16);
  HIPCHK(hipMalloc(&d, bytes));
  c->scene_allocs.push_back(d);
  if (!h.empty()) HIPCHK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  *out = (const T *)d;
  return PTAMD_OK;
}

int light_samples_per_vertex(const ptamd_context *c, const ptamd_render_params *p, const std::vector<int32_t> &kinds) {
  (void)c;
  if (p->hemisphere) return (int)kinds.size() * p->ns_area_light;
  int n = 0;
  for (int k : kinds) n += (k == PTAMD_LIGHT_DIRECTIONAL || k == PTAMD_LIGHT_POINT || k == PTAMD_LIGHT_SPOT) ? 1 : p->ns_area_light;
  return n;
}

struct Launcher {
  ptamd_context *c;
  hipStream_t s;
  bool timed = false;
  void begin(int cls) {
    c->stats.launches[cls]++;
    timed = c->prof_events && ((c->event_mask >> cls) & 1u);
    if (timed) {
      ptamd_context::Ev e;
      e.cls = cls;
      e.a = c->take_event();
      e.b = c->take_event();
      (void)hipEventRecord(e.a, s);
      c->events.push_back(e);
    }
  }
  void end() {
    if (timed) (void)hipEventRecord(c->events.back().b, s);
  }
};

#ifdef PTAMD_WAVELOG
struct WaveLogEntry { int kind; uint32_t base, waves; };
static std::vector<WaveLogEntry> g_wavelog_launches;
static uint32_t g_wavelog_used = 0;
// names the range of the launch about to be queued (kind: 0 k_extend, 1 k_trace2, 2 k_shadow)
static void wavelog_next(ptamd_context *c, int kind, int grid) {
  const uint32_t waves = (uint32_t)grid * (TRAV_BLOCK / 64u);
  if (!c->S.wavelog) {
    c->S.wavelog_cap = 24u << 20;
    if (hipMalloc((void **)&c->S.wavelog, (size_t)c->S.wavelog_cap * 16) != hipSuccess) { c->S.wavelog = nullptr; c->S.wavelog_cap = 0; }
    else (void)hipMemset(c->S.wavelog, 0, (size_t)c->S.wavelog_cap * 16);
  }
  c->S.wavelog_base = g_wavelog_used;
  if (g_wavelog_used + waves <= c->S.wavelog_cap) {
    g_wavelog_launches.push_back({kind, g_wavelog_used, waves});
    g_wavelog_used += waves;
  } else {
    c->S.wavelog_base = c->S.wavelog_cap;  // full: nothing more is written
  }
}
// at ptamd_destroy: "kind base waves" per launch, then start / end ticks per wave, to $PTAMD_WAVELOG
static void wavelog_dump(ptamd_context *c) {
  const char *path = getenv("PTAMD_WAVELOG");
  if (!path || !c->S.wavelog) return;
  std::vector<unsigned long long> h((size_t)g_wavelog_used * 2);
  if (hipMemcpy(h.data(), c->S.wavelog, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return;
  FILE *f = fopen(path, "wb");
  if (!f) return;
  const uint32_t n = (uint32_t)g_wavelog_launches.size();
  fwrite(&n, 4, 1, f);
  for (const auto &e : g_wavelog_launches) { const uint32_t r[3] = {(uint32_t)e.kind, e.base, e.waves}; fwrite(r, 4, 3, f); }
  fwrite(h.data(), 8, h.size(), f);
  fclose(f);
}
#define PTAMD_WAVELOG_NEXT(c, kind, grid) wavelog_next(c, kind, grid)
#else
#define PTAMD_WAVELOG_NEXT(c, kind, grid)
#endif
void collect_events(ptamd_context *c) {
  for (auto &e : c->events) {
    float ms = 0;
    if (hipEventSynchronize(e.b) == hipSuccess && hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) c->stats.kernel_ms[e.cls] += ms;
    c->event_pool.push_back(e.a);
    c->event_pool.push_back(e.b);
  }
  c->events.clear();
}

int ensure_buffers(ptamd_context *c, int64_t paths, int nls, bool pend, bool dup) {
  if (c->slab && c->cap_paths >= paths && c->cap_nls >= nls && (c->cap_pend || !pend) && (c->cap_dup || !dup)) return PTAMD_OK;
  if (c->slab) {
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipFree(c->slab));
    c->slab = nullptr;
  }
  paths = std::max<int64_t>(paths, c->cap_paths);
  nls = std::max(nls, std::max(c->cap_nls, 1));
  pend = pend || c->cap_pend;
  dup = dup || c->cap_dup;
  const size_t P = (size_t)paths, PN = P * (size_t)nls;
  auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
  size_t total = 0;
  auto take = [&](size_t bytes) { size_t o = total; total += al(bytes); return o; };
  size_t off_dup[5] = {0, 0, 0, 0, 0};
  size_t off_q[3][10], off_hit[4], off_sq[5], off_vb[8], off_pb[6], off_list[2], off_cp[3], off_nee[4], off_map[5];
  for (int k = 0; k < (dup ? 3 : 2); k++) {
    off_q[k][0] = take(P * 4);
    for (int j = 1; j < 10; j++) off_q[k][j] = take(P * 8);
  }
  if (dup) {
    off_dup[0] = take(P);
    off_dup[1] = take(PN * sizeof(Contrib));
    off_dup[2] = take(PN + 4);
    off_dup[3] = take(PN * 4);
    off_dup[4] = take(PN * sizeof(SRay));
  }
  off_hit[0] = take(P * 4);
  off_hit[1] = take(P * 8);
  off_hit[2] = take(P * 4);
  off_hit[3] = take(P * 4);
  off_sq[0] = take(PN * 4);
  off_sq[1] = take(PN * sizeof(SRay));
  off_vb[3] = take(P);
  off_vb[4] = take(PN * sizeof(Contrib));
  off_vb[7] = take(PN + 4);
  off_vb[5] = take(PN * 4);
  for (int j = 0; j < 3; j++) off_nee[j] = take(P * 8);
  off_nee[3] = take(P * 16);
  for (int j = 0; j < 6; j++) off_pb[j] = take(P * 8);
  for (int k = 0; k < 2; k++) off_list[k] = take(P * 4);
  off_map[0] = take(P * 8);
  off_map[1] = 0;
  off_map[2] = take((P + 1) * 4);
  off_map[3] = take((P / 256 + 2) * 4);
  off_map[4] = take((P / 256 + 2) * 4);
  off_cp[0] = take(P);
  off_cp[1] = take((P / 256 + 2) * 4);
  off_cp[2] = take((P / 256 + 2) * 4);
  HIPCHK(hipMalloc(&c->slab, total));
  char *b = (char *)c->slab;
  for (int k = 0; k < (dup ? 3 : 2); k++) {
    c->q[k].path = (int32_t *)(b + off_q[k][0]);
    for (int j = 0; j < 3; j++) {
      c->q[k].o[j] = (double *)(b + off_q[k][1 + j]);
      c->q[k].d[j] = (double *)(b + off_q[k][4 + j]);
      c->q[k].T[j] = (double *)(b + off_q[k][7 + j]);
    }
  }
  c->hb.prim = (int32_t *)(b + off_hit[0]);
  c->hb.t = (double *)(b + off_hit[1]);
  c->vb.self_code = (int32_t *)(b + off_hit[2]);
  c->vb.self_dmin = (float *)(b + off_hit[3]);
  c->vb.pend_prim = (int32_t *)(b + off_vb[5]);
  c->sq.slot = (int32_t *)(b + off_sq[0]);
  c->sq.ray = (SRay *)(b + off_sq[1]);
  for (int j = 0; j < 3; j++) c->vb.hp[j] = nullptr;  // set per depth: the next ray queue's origin arrays
  c->vb.flag = (uint8_t *)(b + off_vb[3]);
  c->vb.contrib = (Contrib *)(b + off_vb[4]);
  c->vb.vis = (uint8_t *)(b + off_vb[7]);
  for (int j = 0; j < 3; j++) c->vb.nz[j] = (double *)(b + off_nee[j]);
  c->vb.nee = (uint4 *)(b + off_nee[3]);
  for (int j = 0; j < 3; j++) {
    c->pb.rad[j] = (double *)(b + off_pb[j]);
    c->pb.pend[j] = (double *)(b + off_pb[3 + j]);
  }
  c->pb.pix_smp = (int2 *)(b + off_map[0]);
  c->pix_off = (uint32_t *)(b + off_map[2]);
  c->plan_blocks = (uint32_t *)(b + off_map[3]);
  c->slot_pix = (uint32_t *)(b + off_map[4]);
  for (int k = 0; k < 2; k++) c->list[k] = (int32_t *)(b + off_list[k]);
  c->cp.pflag = (uint8_t *)(b + off_cp[0]);
  c->cp.cnt_next = (uint32_t *)(b + off_cp[1]);
  c->cp.cnt_shadow = (uint32_t *)(b + off_cp[2]);
  c->cap_paths = paths;
  c->cap_nls = nls;
  c->flag_p[0] = c->vb.flag; c->contrib_p[0] = c->vb.contrib; c->vis_p[0] = c->vb.vis; c->pend_prim_p[0] = c->vb.pend_prim; c->ray_p[0] = c->sq.ray;
  if (dup) {
    c->flag_p[1] = (uint8_t *)(b + off_dup[0]);
    c->contrib_p[1] = (Contrib *)(b + off_dup[1]);
    c->vis_p[1] = (uint8_t *)(b + off_dup[2]);
    c->pend_prim_p[1] = (int32_t *)(b + off_dup[3]);
    c->ray_p[1] = (SRay *)(b + off_dup[4]);
  }
  c->cap_pend = true;
  c->cap_dup = dup;
  return PTAMD_OK;
}

int setup_frame_params(const ptamd_context *c, const ptamd_render_params *p, FrameParams &fp) {
  if (!c->have_scene || !c->have_camera) {
    ptamd::set_error("scene and camera must be set before rendering");
    return PTAMD_ERR_STATE;
  }
  if (!p || p->w <= 0 || p->h <= 0 || p->ns_aa < 0 || p->samples_per_batch <= 0 || p->ns_area_light < 0 ||
      p->max_ray_depth < 0 || (int64_t)p->w * p->h > (int64_t)1 << 30) {
    ptamd::set_error("invalid render parameters");
    return PTAMD_ERR_ARG;
  }
  fp.w = p->w; fp.h = p->h;
  fp.max_ray_depth = p->max_ray_depth;
  fp.accum = p->accumulate_bounces ? 1 : 0;
  fp.ns_area_light = p->ns_area_light;
  fp.hemisphere = p->hemisphere ? 1 : 0;
  fp.specular = p->specular ? 1 : 0;
  fp.seed_lo = (uint32_t)p->seed;
  fp.seed_hi = (uint32_t)(p->seed >> 32);
  const ptamd_camera &cam = c->cam;
  // tan(radians(fov) * 0.5) with the host's libm, as the reference evaluates
  // it per ray (camera.cpp:197)
  fp.tan_h = tan(cam.hFov * (PT_PI / 180) * 0.5);
  fp.tan_v = tan(cam.vFov * (PT_PI / 180) * 0.5);
  fp.n_clip = cam.nClip;
  fp.f_clip = cam.fClip;
  fp.cam_pos.x = cam.pos[0]; fp.cam_pos.y = cam.pos[1]; fp.cam_pos.z = cam.pos[2];
  for (int k = 0; k < 3; k++) {
    fp.c2w.c[k].x = cam.c2w[3 * k]; fp.c2w.c[k].y = cam.c2w[3 * k + 1]; fp.c2w.c[k].z = cam.c2w[3 * k + 2];
  }
  return PTAMD_OK;
}

int grid_for(const ptamd_context *c, int per_cu) { return c->num_cu * per_cu; }

}  // namespace

extern "C" {

int ptamd_create(int32_t device_id, ptamd_context **out) {
  if (!out) { ptamd::set_error("null output pointer"); return PTAMD_ERR_ARG; }
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    ptamd::set_error("no HIP device available: the render hot path needs a GPU and has no CPU fallback");
    return PTAMD_ERR_DEVICE;
  }
  if (device_id < 0 || device_id >= n) { ptamd::set_error("device id out of range"); return PTAMD_ERR_ARG; }
  HIPCHK(hipSetDevice(device_id));
  ptamd_context *c = new ptamd_context();
  c->device = device_id;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) c->num_cu = prop.multiProcessorCount;
  if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) c->own_stream = nullptr;
  if (const char *e = getenv("PTAMD_TRACE_TWO")) c->trace_two = atoi(e) != 0;
  if (const char *e = getenv("PTAMD_FUSE_RESOLVE")) c->fuse_resolve = atoi(e) != 0;
  if (const char *e = getenv("PTAMD_FUSE_BELOW")) c->fuse_below = (uint32_t)strtoul(e, nullptr, 10);
  if (const char *e = getenv("PTAMD_PREDICT_ABOVE")) c->predict_above = std::max<long long>(0, atoll(e));
  if (const char *e = getenv("PTAMD_PLAN_MARGIN")) { const double m = atof(e); if (m >= 1.0 && m <= 8.0) c->plan_margin = m; }
  // The counters live apart from the work buffers: those are reallocated when a later round needs
  // more room, and a frame's statistics must survive that.
  if (hipMalloc((void **)&c->ctl, CTL_RING * sizeof(Ctl)) != hipSuccess || hipMemset(c->ctl, 0, CTL_RING * sizeof(Ctl)) != hipSuccess) {
    ptamd::set_error("cannot allocate the device-side counters");
    ptamd_destroy(c);
    return PTAMD_ERR_DEVICE;
  }
  // Address of this device's copy of the static scratch array (the device is current: hipSetDevice
  // above); taken by the source file that defines the array.
  {
    const int e = ptk::static_scratch_address(&c->static_block_count);
    if (e != 0) {
      ptamd::set_error(std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString((hipError_t)e));
      ptamd_destroy(c);
      return PTAMD_ERR_DEVICE;
    }
  }
  *out = c;
  return PTAMD_OK;
}

void ptamd_destroy(ptamd_context *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  collect_events(c);
  for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
#ifdef PTAMD_WAVELOG
  wavelog_dump(c);
  if (c->S.wavelog) (void)hipFree(c->S.wavelog);
#endif
  free_scene(c);
  free_frame_scratch(c);
  if (c->slab) (void)hipFree(c->slab);
  if (c->ctl) (void)hipFree(c->ctl);
  if (c->block_count) (void)hipFree(c->block_count);
  if (c->gstack) (void)hipFree(c->gstack);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

int ptamd_set_max_paths(ptamd_context *c, int64_t max_paths) {
  if (!c) { ptamd::set_error("null context"); return PTAMD_ERR_ARG; }
  if (max_paths <= 0) max_paths = 16 << 20;
  if (max_paths > ((int64_t)1 << 27)) max_paths = (int64_t)1 << 27;
  c->max_paths = max_paths;
  return PTAMD_OK;
}

int ptamd_set_profiling(ptamd_context *c, int32_t events, int32_t counters) {
  if (!c) { ptamd::set_error("null context"); return PTAMD_ERR_ARG; }
  c->prof_events = events != 0;
  c->prof_counters = counters != 0;
  return PTAMD_OK;
}

int ptamd_set_event_classes(ptamd_context *c, uint32_t mask) {
  if (!c) { ptamd::set_error("null context"); return PTAMD_ERR_ARG; }
  c->event_mask = mask;
  return PTAMD_OK;
}
int ptamd_upload_scene(ptamd_context *c, const ptamd_scene_desc *d) {
  if (!c || !d) { ptamd::set_error("null argument"); return PTAMD_ERR_ARG; }
  HIPCHK(hipSetDevice(c->device));
  if (d->n_prims <= 0 || d->n_nodes <= 0 || d->root < 0 || d->root >= d->n_nodes || d->n_materials <= 0 ||
      d->jitter_n <= 0 || !d->prim_type || !d->prim_pos || !d->prim_nrm || !d->prim_material || !d->node_bbox ||
      !d->node_left || !d->node_right || !d->node_start || !d->node_count || !d->mat_kind || !d->mat_param ||
      !d->jitter_table || (d->n_lights > 0 && (!d->light_kind || !d->light_param)) || d->n_lights < 0) {
    ptamd::set_error("scene description is incomplete");
    return PTAMD_ERR_ARG;
  }
  ptamd::Tree4 T;
  {
    const int rc4 = ptamd::build_tree4(d, T);
    if (rc4) return rc4;
  }
  const int NP = d->n_prims;
  const std::vector<ptk::DevNode> &nodes = T.nodes;
  const std::vector<double> &prim = T.prim, &nrm = T.nrm;
  const std::vector<int32_t> &info = T.info, &light_end = T.light_end, &light_tag = T.light_tag, &visit_order = T.visit_order;
  const std::vector<ptamd::Tree4::Word4> &prim32 = T.prim32;
  const int has_end = T.has_end, max_depth = T.stack_cap - 1;
  const int32_t root_code = T.root_code;
  const double max_coord = T.max_coord;
  free_scene(c);
  c->dev2ref = visit_order;
  c->n_prims = NP;
  DevScene S{};
  int rc;
  if ((rc = upload(c, nodes, &S.nodes))) return rc;
  if ((rc = upload(c, prim, &S.prim))) return rc;
  {
    const ptamd::Tree4::Word4 *d32 = nullptr;
    if ((rc = upload(c, prim32, &d32))) return rc;
    S.prim32 = reinterpret_cast<const uint4 *>(d32);
  }
  if ((rc = upload(c, info, &S.prim_info))) return rc;
  if ((rc = upload(c, nrm, &S.prim_nrm))) return rc;
  std::vector<int32_t> mk(d->mat_kind, d->mat_kind + d->n_materials);
  std::vector<double> mp(d->mat_param, d->mat_param + 3 * (size_t)d->n_materials);
  std::vector<int32_t> lk(d->light_kind, d->light_kind + d->n_lights);
  std::vector<double> lp(d->light_param, d->light_param + 16 * (size_t)d->n_lights);
  std::vector<double> jt(d->jitter_table, d->jitter_table + 2 * (size_t)d->jitter_n * d->jitter_n);
  std::vector<double> me(4 * (size_t)d->n_materials, 0.0);
  if (d->mat_extra) me.assign(d->mat_extra, d->mat_extra + 4 * (size_t)d->n_materials);
  if ((rc = upload(c, mk, &S.mat_kind))) return rc;
  if ((rc = upload(c, mp, &S.mat_param))) return rc;
  if ((rc = upload(c, me, &S.mat_extra))) return rc;
  if ((rc = upload(c, lk, &S.light_kind))) return rc;
  if ((rc = upload(c, light_end, &S.light_end))) return rc;
  if ((rc = upload(c, light_tag, &S.light_tag))) return rc;
  S.has_end = has_end;
  if ((rc = upload(c, lp, &S.light_param))) return rc;
  if ((rc = upload(c, jt, &S.jitter))) return rc;
  S.n_lights = d->n_lights;
  S.jitter_n = d->jitter_n;
  S.root_code = root_code;
  c->scene_max_coord = max_coord;
  {
    double far = max_coord;  // camera rays start at the camera, every other ray on a primitive
    if (c->have_camera) for (int k = 0; k < 3; k++) far = std::max(far, std::fabs(c->cam.pos[k]));
    S.coord_bound = (float)(1.01 * far + 1e-30);
  }
  S.stack_cap = max_depth + 1;
  c->S = S;
  c->have_scene = true;
  c->host_light_kinds = lk;
  return PTAMD_OK;
}

int ptamd_set_camera(ptamd_context *c, const ptamd_camera *cam) {
  if (!c || !cam) { ptamd::set_error("null argument"); return PTAMD_ERR_ARG; }
  c->cam = *cam;
  // camera rays start at the camera, every other ray on a primitive
  const double far = std::max(c->scene_max_coord, std::max(std::fabs(cam->pos[0]), std::max(std::fabs(cam->pos[1]), std::fabs(cam->pos[2]))));
  c->S.coord_bound = (float)(1.01 * far + 1e-30);
  c->have_camera = true;
  return PTAMD_OK;
}


int ptamd_frame_begin(ptamd_context *c, const ptamd_render_params *p) {
  if (!c) { ptamd::set_error("null context"); return PTAMD_ERR_ARG; }
  HIPCHK(hipSetDevice(c->device));
  int rc = setup_frame_params(c, p, c->fp);
  if (rc) return rc;
  c->params = *p;
  c->fp.nls = light_samples_per_vertex(c, p, c->host_light_kinds);
  c->fp.bpv = 1 + (c->fp.nls + 1) / 2;
  c->have_frame = true;
  if (!c->events.empty()) {
    HIPCHK(hipDeviceSynchronize());
    collect_events(c);  // timings of earlier frames are not part of this one
  }
  memset(&c->stats, 0, sizeof c->stats);
  // The device-side counters are cleared by the frame's first ptamd_trace_list, on ITS stream: a
  // memset issued here, on the null stream, would not be ordered against an earlier frame still
  // running on a non-blocking stream.
  c->clear_ctl = true;
  return PTAMD_OK;
}

// Launch shape of the traversal kernels: 256-thread workgroups, PTAMD_STACK_* stack
// levels per lane in LDS, the rest (if the tree is deeper) in a global strip.
struct TravLaunch {
  int block;
  size_t lds_extend, lds_shadow;
  int grid_extend, grid_shadow;  // grid-stride driver: a few workgroups per resident slot; persistent: exactly the resident ones
  uint2 *gstack;
};
static int traversal_launch(ptamd_context *c, TravLaunch *t) {
  t->block = TRAV_BLOCK;
  t->lds_extend = (size_t)t->block * (PTAMD_STACK_EXTEND + 1) * sizeof(uint2);  // + the scratch level
  t->lds_shadow = (size_t)t->block * (PTAMD_STACK_SHADOW + 1) * sizeof(uint2);
  // resident workgroups per CU: bounded by LDS (160 KiB per CU) and by registers
  const int lds_e = (int)((160 * 1024) / t->lds_extend), lds_s = (int)((160 * 1024) / t->lds_shadow);
  t->grid_extend = c->num_cu * std::max(1, std::min(lds_e, PTAMD_WAVES_EXTEND)) * (PTAMD_REFILL_EXTEND ? 1 : PTAMD_GRID_MULT);
  t->grid_shadow = c->num_cu * std::max(1, std::min(lds_s, PTAMD_WAVES_SHADOW)) * (PTAMD_REFILL_SHADOW ? 1 : PTAMD_GRID_MULT);
  const size_t levels = (size_t)std::max(0, c->S.stack_cap - std::min(PTAMD_STACK_EXTEND, PTAMD_STACK_SHADOW));
  const size_t need = levels * (size_t)std::max(t->grid_extend, t->grid_shadow) * t->block * sizeof(uint2);
  if (need > c->gstack_bytes) {
    if (c->gstack) { HIPCHK(hipDeviceSynchronize()); HIPCHK(hipFree(c->gstack)); c->gstack = nullptr; }
    HIPCHK(hipMalloc((void **)&c->gstack, need));
    c->gstack_bytes = need;
  }
  t->gstack = c->gstack;
  return PTAMD_OK;
}

namespace {
int check_list(const ptamd_pixel_list *l, const char *who) {
  if (!l || l->n_max < 0 || l->stride < 1 || l->offset < 0 || (!l->d_n && (l->n < 0 || l->n > l->n_max)) ||
      (l->n_max > 0 && !l->d_pixels) || l->n_max > (int64_t)0x7fffffff) {
    ptamd::set_error(std::string("invalid pixel list passed to ") + who);
    return PTAMD_ERR_ARG;
  }
  return PTAMD_OK;
}
PixView view_of(const ptamd_pixel_list *l) {
  return PixView{l->d_pixels, (const long long *)l->d_n, (long long)l->n, (uint32_t)l->offset, (uint32_t)l->stride};
}
// entries of the shard, at most
int64_t shard_max(const ptamd_pixel_list *l) {
  return l->n_max > l->offset ? (l->n_max - l->offset + l->stride - 1) / l->stride : 0;
}
}  // namespace

int ptamd_trace_list(ptamd_context *c, const ptamd_pixel_list *list, int32_t sample_base, int32_t n_samples,
                     double *d_state, void *stream) {
  if (!c || !c->have_frame) { ptamd::set_error("ptamd_frame_begin must be called first"); return PTAMD_ERR_STATE; }
  int rc = check_list(list, "ptamd_trace_list");
  if (rc) return rc;
  const int64_t n = shard_max(list);
  if (n_samples <= 0 || sample_base < -1 || (n > 0 && !d_state) ||
      (sample_base < 0 && n_samples % c->params.samples_per_batch != 0)) {
    ptamd::set_error("invalid arguments to ptamd_trace_list");
    return PTAMD_ERR_ARG;
  }
  if (n == 0) return PTAMD_OK;
  HIPCHK(hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)stream;
  const FrameParams &fp = c->fp;
  const int nls = std::max(fp.nls, 1);
  if (nls >= (1 << 16)) {
    ptamd::set_error("more than 65535 light samples per vertex are not supported");
    return PTAMD_ERR_ARG;
  }
  const FastDiv by_nls = make_fastdiv((uint32_t)nls);
  const uint2 nls_nsa = make_uint2((uint32_t)nls, (uint32_t)std::max(fp.ns_area_light, 0));  // light samples per vertex, per area light
  // A depth's light samples are settled beside the next depth's light sampling (k_light_resolve) where the shading
  // kernel of that next depth leaves the paths' sums alone: bounces accumulated (-o 1, else k_shade hands `pending`
  // on), no delta materials, importance sampling.
  const bool overlap = c->fuse_resolve && fp.accum && !fp.specular && !fp.hemisphere && fp.max_ray_depth > 1 && fp.nls > 0;
  // paths per launch: max_paths, but never more than a 24 GB slab of work buffers
  // (about 290 bytes per path + 70 per light-sample slot; 40 GB, 370 and 131 with the per-depth copies),
  // and never so many that a light-sample slot index (entry * nls + q, an int32) would pass 2^31
  const int64_t by_memory = std::min(overlap ? ((int64_t)40 << 30) / (370 + 131 * (int64_t)nls) : ((int64_t)24 << 30) / (290 + 70 * (int64_t)nls),
                                     (((int64_t)1 << 31) - 1) / nls);
  int64_t px_per_chunk = std::max<int64_t>(1, std::min(c->max_paths, by_memory) / n_samples);
  px_per_chunk = std::min(px_per_chunk, n);
  rc = ensure_buffers(c, px_per_chunk * n_samples, nls, !fp.accum, overlap);
  if (rc) return rc;
  TravLaunch tl;
  rc = traversal_launch(c, &tl);
  if (rc) return rc;
  if (c->clear_ctl) {
    HIPCHK(hipMemsetAsync(c->ctl, 0, CTL_RING * sizeof(Ctl), s));
    c->clear_ctl = false;
  }
  const int tblock = tl.block;
  const int egrid = tl.grid_extend, sgrid = tl.grid_shadow;
  const size_t elds = tl.lds_extend, slds = tl.lds_shadow;
  uint2 *gstack = tl.gstack;
  const int ggrid = grid_for(c, 8);
  // k_shade runs one workgroup per 256 list entries; no list of this call is longer than a chunk's paths
  const int shade_grid = (int)((px_per_chunk * n_samples + 255) / 256);
  const int light_grid = (int)(((int64_t)shade_grid * 256 * nls + 255) / 256);  // k_light: one workgroup per 256 light-sample slots
  PathBuf pb = c->pb;
  if (fp.accum) pb.pend[0] = pb.pend[1] = pb.pend[2] = nullptr;
  const PixView pv = view_of(list);
  // per-pixel prediction only where a launch is big enough for dropped samples to cost time
  const RoundPlan rp{d_state, c->params.max_tolerance, sample_base, n_samples, c->params.samples_per_batch, c->params.ns_aa,
                     n * n_samples >= c->predict_above ? 1 : 0, c->plan_margin};
  // every pixel its own number of samples: the path slots are packed (PathMap), so the launches hold no unused slot
  const bool packed = sample_base < 0;
  const PathMap pm{packed ? (const uint32_t *)c->pix_off : nullptr, (const uint32_t *)c->slot_pix, (uint32_t)n_samples};
  Launcher L{c, s};
  const int depths = std::max(1, fp.max_ray_depth);
  for (int64_t p0 = 0; p0 < n; p0 += px_per_chunk) {
    const uint32_t up0 = (uint32_t)p0;
    Ctl *const ctl1 = c->ctl + 1;  // the control block of depth 1 (depth d: block d & 3)
    L.begin(PTAMD_K_GENERATE);
    if (packed) {
      const uint32_t nb = (uint32_t)((std::min<int64_t>(px_per_chunk, n - p0) + 255) / 256);
      hipLaunchKernelGGL(k_plan_count, nb, 256, 0, s, pv, up0, (uint32_t)px_per_chunk, rp, c->plan_blocks);
      hipLaunchKernelGGL(k_plan_scan, 1, 1024, 0, s, c->plan_blocks, nb, ctl1, pv, up0, (uint32_t)px_per_chunk);
      hipLaunchKernelGGL(k_plan_offsets, nb, 256, 0, s, pv, up0, ctl1, rp, (const uint32_t *)c->plan_blocks, c->pix_off, c->slot_pix);
    } else {
      hipLaunchKernelGGL(k_set_counts, 1, 1, 0, s, ctl1, pv, up0, (uint32_t)px_per_chunk, (uint32_t)n_samples);
    }
    hipLaunchKernelGGL(k_generate, ggrid, 256, 0, s, fp, c->S, pv, up0, ctl1, rp, pm, c->q[0], pb);
    L.end();
    int cur = 0;   // list[cur]: this depth's list, list[cur ^ 1]: the next one's
    int qi = 0;    // q[qi]: this depth's rays, q[qo]: the next depth's (its origin arrays hold this depth's hit points)
    const int nq = overlap ? 3 : 2;
    ResolveArgs prev{};  // the depth whose light samples have been traced but not yet summed (overlap)
    bool traced_ahead = false;  // this depth's bounce rays went with the previous depth's shadow rays (k_trace2)
    for (int depth = 1; depth <= depths; depth++) {
      const int qo = (qi + 1) % nq, par = overlap ? (depth & 1) : 0;
      Ctl *const ctl = c->ctl + (depth & (CTL_RING - 1)), *const ctl_next = c->ctl + ((depth + 1) & (CTL_RING - 1));
      const double min_t = depth == 1 ? fp.n_clip : PT_EPS_D;
      const double max_t = depth == 1 ? fp.f_clip : (double)INFINITY;
      const int32_t *lst = depth == 1 ? nullptr : c->list[cur];
      if (!traced_ahead) {
        L.begin(PTAMD_K_EXTEND);
        PTAMD_WAVELOG_NEXT(c, 0, egrid);
        if (c->prof_counters) hipLaunchKernelGGL((k_extend<true>), egrid, tblock, elds, s, c->S, c->q[qi], lst, c->hb, ctl, min_t, max_t, gstack, (const int32_t *)c->vb.self_code, (const float *)c->vb.self_dmin, fp.cam_pos);
        else hipLaunchKernelGGL((k_extend<false>), egrid, tblock, elds, s, c->S, c->q[qi], lst, c->hb, ctl, min_t, max_t, gstack, (const int32_t *)c->vb.self_code, (const float *)c->vb.self_dmin, fp.cam_pos);
        L.end();
      }
      traced_ahead = false;
      VertexBuf vb = c->vb;
      for (int k = 0; k < 3; k++) vb.hp[k] = c->q[qo].o[k];  // one copy of the hit point serves both uses
      vb.flag = c->flag_p[par]; vb.contrib = c->contrib_p[par]; vb.vis = c->vis_p[par]; vb.pend_prim = c->pend_prim_p[par];
      ShadowQ sq = c->sq;
      sq.ray = c->ray_p[par];
      L.begin(PTAMD_K_SHADE);
      if (fp.hemisphere)
        hipLaunchKernelGGL(k_shade<true>, shade_grid, 256, 0, s, fp, c->S, c->q[qi], lst, c->hb, c->q[qo], sq, vb, pb,
                           c->cp, ctl, depth);
      else
        hipLaunchKernelGGL(k_shade<false>, shade_grid, 256, 0, s, fp, c->S, c->q[qi], lst, c->hb, c->q[qo], sq, vb, pb,
                           c->cp, ctl, depth);
      L.end();
      if (!fp.hemisphere && fp.max_ray_depth > 0 && fp.nls > 0) {
        L.begin(PTAMD_K_LIGHT);
        const uint32_t rng_base = 1u + (uint32_t)(depth - 1) * (uint32_t)fp.bpv;
        if (overlap && depth > 1)  // with the previous depth's k_resolve: at most light_grid + shade_grid workgroups have work
          hipLaunchKernelGGL(k_light_resolve, light_grid + shade_grid, 256, 0, s, fp, c->S, vb, sq, c->cp, ctl, by_nls, rng_base, prev);
        else
          hipLaunchKernelGGL(k_light, light_grid, 256, 0, s, fp, c->S, vb, sq, c->cp, ctl, by_nls, rng_base);
        L.end();
      }
      L.begin(PTAMD_K_OTHER);
      hipLaunchKernelGGL(k_scan2, 1, 1024, 0, s, c->cp, ctl, ctl_next);
      hipLaunchKernelGGL(k_scatter, ggrid, 256, 0, s, c->cp, vb, c->list[cur ^ 1], sq, ctl, fp.max_ray_depth > 0 ? fp.nls : 0);
      L.end();
      if (fp.max_ray_depth > 0 && fp.nls > 0) {
        uint32_t shadow_skip = 0;  // k_shadow skips lists that k_trace2 has traced: those shorter than this
        if (depth < depths && c->trace_two) {
          // the next depth's bounce rays (queue qo, just written) together with this depth's shadow rays;
          // the hit records of this depth have been consumed by k_shade
          L.begin(PTAMD_K_TRACE2);
          PTAMD_WAVELOG_NEXT(c, 1, egrid);
          const bool hemi = fp.hemisphere != 0, cnt = c->prof_counters;
#define PTAMD_LAUNCH_TRACE2(H, C) hipLaunchKernelGGL((k_trace2<H, C>), egrid, tblock, elds, s, c->S, c->q[qo], (const int32_t *)c->list[cur ^ 1], c->hb, sq, vb, ctl, by_nls, nls_nsa, gstack, c->fuse_below)
          if (hemi) { if (cnt) PTAMD_LAUNCH_TRACE2(true, true); else PTAMD_LAUNCH_TRACE2(true, false); }
          else { if (cnt) PTAMD_LAUNCH_TRACE2(false, true); else PTAMD_LAUNCH_TRACE2(false, false); }
#undef PTAMD_LAUNCH_TRACE2
          L.end();
          traced_ahead = true;
          shadow_skip = c->fuse_below;
          // No list of this chunk can be longer than its path slots (bounce rays) and nls times that (shadow rays):
          // when even those bounds stay below the threshold, k_trace2 always takes the shadow rays along and the
          // launch of k_shadow, which would only find that out and return, is left out.
          if (std::min<int64_t>(px_per_chunk, n - p0) * n_samples * (1 + (int64_t)nls) < (int64_t)c->fuse_below) shadow_skip = 0xffffffffu;
        }
        if (shadow_skip != 0xffffffffu) {  // (all lists fused: not even the empty launch)
          L.begin(PTAMD_K_SHADOW);
          PTAMD_WAVELOG_NEXT(c, 2, sgrid);
          if (fp.hemisphere) {
            if (c->prof_counters) hipLaunchKernelGGL((k_shadow<true, true>), sgrid, tblock, slds, s, c->S, sq, vb, ctl, by_nls, nls_nsa, gstack, shadow_skip);
            else hipLaunchKernelGGL((k_shadow<true, false>), sgrid, tblock, slds, s, c->S, sq, vb, ctl, by_nls, nls_nsa, gstack, shadow_skip);
          } else {
            if (c->prof_counters) hipLaunchKernelGGL((k_shadow<false, true>), sgrid, tblock, slds, s, c->S, sq, vb, ctl, by_nls, nls_nsa, gstack, shadow_skip);
            else hipLaunchKernelGGL((k_shadow<false, false>), sgrid, tblock, slds, s, c->S, sq, vb, ctl, by_nls, nls_nsa, gstack, shadow_skip);
          }
          L.end();
        }
        if (overlap && depth < depths) {
          // summed by the next depth's k_light_resolve
          prev = ResolveArgs{c->q[qi], lst, vb, (const SRay *)sq.ray, pb, depth};
        } else {
          L.begin(PTAMD_K_RESOLVE);
          hipLaunchKernelGGL(k_resolve, shade_grid, 256, 0, s, fp, c->S, c->q[qi], lst, vb, (const SRay *)sq.ray, pb, ctl, depth);
          L.end();
        }
      }
      cur ^= 1;
      qi = qo;
    }
    L.begin(PTAMD_K_ACCUMULATE);
    hipLaunchKernelGGL(k_accumulate, ggrid, 256, 0, s, pv, up0, ctl1, rp, pm, pb, d_state);
    L.end();
  }
  HIPCHK(hipGetLastError());
  return PTAMD_OK;
}

int ptamd_trace_pixels(ptamd_context *c, const int32_t *d_pixels, int64_t n, int32_t sample_base, int32_t n_samples,
                       double *d_state, void *stream) {
  if (n < 0 || (n > 0 && !d_pixels)) { ptamd::set_error("invalid arguments to ptamd_trace_pixels"); return PTAMD_ERR_ARG; }
  const ptamd_pixel_list l{d_pixels, nullptr, n, n, 0, 1};
  return ptamd_trace_list(c, &l, sample_base, n_samples, d_state, stream);
}

int ptamd_filter_list(ptamd_context *c, const ptamd_pixel_list *in, int32_t n_done, const double *d_state,
                      int32_t *d_out, int64_t *d_n_out, void *stream) {
  if (!c || !c->have_frame) { ptamd::set_error("ptamd_frame_begin must be called first"); return PTAMD_ERR_STATE; }
  int rc = check_list(in, "ptamd_filter_list");
  if (rc) return rc;
  const int64_t n = shard_max(in);
  if ((n_done < 1 && n_done != -1) || !d_n_out || (n > 0 && (!d_out || !d_state))) {
    ptamd::set_error("invalid arguments to ptamd_filter_list");
    return PTAMD_ERR_ARG;
  }
  HIPCHK(hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)stream;
  const uint32_t nb = (uint32_t)((n + 255) / 256);
  if (nb > c->block_count_cap) {
    if (c->block_count) { HIPCHK(hipDeviceSynchronize()); HIPCHK(hipFree(c->block_count)); }
    c->block_count_cap = std::max<size_t>(nb, 4096);
    HIPCHK(hipMalloc((void **)&c->block_count, c->block_count_cap * sizeof(uint32_t)));
  }
  const PixView pv = view_of(in);
  Launcher L{c, s};
  L.begin(PTAMD_K_FILTER);
  if (nb) hipLaunchKernelGGL(k_filter_count, nb, 256, 0, s, pv, n_done, c->params.max_tolerance, c->params.samples_per_batch,
                             c->params.ns_aa, d_state, c->block_count);
  hipLaunchKernelGGL(k_filter_scan, 1, 1024, 0, s, c->block_count, nb, (long long *)d_n_out);
  if (nb) hipLaunchKernelGGL(k_filter_scatter, nb, 256, 0, s, pv, n_done, c->params.max_tolerance, c->params.samples_per_batch,
                             c->params.ns_aa, d_state, c->block_count, d_out);
  L.end();
  HIPCHK(hipGetLastError());
  return PTAMD_OK;
}

// ptamd_filter_list with the per-block counts in the static device array instead of a buffer
// the library allocates: no hipMalloc / hipFree / event creation on any path, so it may be
// called where allocation is not allowed (under stream capture, in a serving loop). The
// kernels are the same three; they get the array through the pointer taken at ptamd_create.
// Nothing in the array has to start at zero: k_filter_count writes every count that
// k_filter_scan and k_filter_scatter read. Calls that overlap on one GPU share the array and race.
int ptamd_filter_list_static(ptamd_context *c, const ptamd_pixel_list *in, int32_t n_done, const double *d_state,
                             int32_t *d_out, int64_t *d_n_out, void *stream) {
  if (!c || !c->have_frame) { ptamd::set_error("ptamd_frame_begin must be called first"); return PTAMD_ERR_STATE; }
  int rc = check_list(in, "ptamd_filter_list_static");
  if (rc) return rc;
  const int64_t n = shard_max(in);
  if ((n_done < 1 && n_done != -1) || !d_n_out || (n > 0 && (!d_out || !d_state))) {
    ptamd::set_error("invalid arguments to ptamd_filter_list_static");
    return PTAMD_ERR_ARG;
  }
  if (n > FILTER_STATIC_MAX_PIXELS) {
    ptamd::set_error("ptamd_filter_list_static: the list is longer than PTAMD_FILTER_STATIC_MAX_PIXELS; "
                     "use ptamd_filter_list");
    return PTAMD_ERR_ARG;
  }
  if (!c->static_block_count) { ptamd::set_error("static scratch array is not set up"); return PTAMD_ERR_STATE; }
  HIPCHK(hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)stream;
  const uint32_t nb = (uint32_t)((n + 255) / 256);
  uint32_t *counts = c->static_block_count;
  const PixView pv = view_of(in);
  c->stats.launches[PTAMD_K_FILTER]++;
  if (nb) hipLaunchKernelGGL(k_filter_count, nb, 256, 0, s, pv, n_done, c->params.max_tolerance, c->params.samples_per_batch,
                             c->params.ns_aa, d_state, counts);
  hipLaunchKernelGGL(k_filter_scan, 1, 1024, 0, s, counts, nb, (long long *)d_n_out);
  if (nb) hipLaunchKernelGGL(k_filter_scatter, nb, 256, 0, s, pv, n_done, c->params.max_tolerance, c->params.samples_per_batch,
                             c->params.ns_aa, d_state, counts, d_out);
  HIPCHK(hipGetLastError());
  return PTAMD_OK;
}

int ptamd_adaptive_filter(ptamd_context *c, const int32_t *d_in, int64_t n, int32_t n_done, const double *d_state,
                          int32_t *d_out, int64_t *d_n_out, void *stream) {
  if (n < 0 || (n > 0 && !d_in)) { ptamd::set_error("invalid arguments to ptamd_adaptive_filter"); return PTAMD_ERR_ARG; }
  const ptamd_pixel_list l{d_in, nullptr, n, n, 0, 1};
  return ptamd_filter_list(c, &l, n_done, d_state, d_out, d_n_out, stream);
}

int ptamd_pack_rows(ptamd_context *c, const ptamd_pixel_list *list, const double *d_state, double *d_rows, void *stream) {
  if (!c) { ptamd::set_error("null context"); return PTAMD_ERR_ARG; }
  int rc = check_list(list, "ptamd_pack_rows");
  if (rc) return rc;
  if (shard_max(list) == 0) return PTAMD_OK;
  if (!d_state || !d_rows) { ptamd::set_error("null buffer"); return PTAMD_ERR_ARG; }
  HIPCHK(hipSetDevice(c->device));
  Launcher L{c, (hipStream_t)stream};
  L.begin(PTAMD_K_OTHER);
  hipLaunchKernelGGL(k_pack_rows, grid_for(c, 4), 256, 0, (hipStream_t)stream, view_of(list), d_state, d_rows);
  L.end();
  HIPCHK(hipGetLastError());
  return PTAMD_OK;
}

int ptamd_unpack_rows(ptamd_context *c, const ptamd_pixel_list *list, int32_t world, int32_t skip_rank,
                      const double *d_rows, int64_t rows_per_rank, double *d_state, void *stream) {
  if (!c) { ptamd::set_error("null context"); return PTAMD_ERR_ARG; }
  int rc = check_list(list, "ptamd_unpack_rows");
  if (rc) return rc;
  if (world < 1 || list->offset != 0 || list->stride != 1 || rows_per_rank < (list->n_max + world - 1) / world) {
    ptamd::set_error("invalid arguments to ptamd_unpack_rows");
    return PTAMD_ERR_ARG;
  }
  if (list->n_max == 0) return PTAMD_OK;
  if (!d_state || !d_rows) { ptamd::set_error("null buffer"); return PTAMD_ERR_ARG; }
  HIPCHK(hipSetDevice(c->device));
  Launcher L{c, (hipStream_t)stream};
  L.begin(PTAMD_K_OTHER);
  hipLaunchKernelGGL(k_unpack_rows, grid_for(c, 4), 256, 0, (hipStream_t)stream, view_of(list), (uint32_t)world,
                     skip_rank < 0 ? 0xffffffffu : (uint32_t)skip_rank, d_rows, (uint32_t)rows_per_rank, d_state);
  L.end();
  HIPCHK(hipGetLastError());
  return PTAMD_OK;
}

int ptamd_frame_resolve(ptamd_context *c, const double *d_state, double *d_hdr, int32_t *d_count, void *stream) {
  if (!c || !c->have_frame) { ptamd::set_error("ptamd_frame_begin must be called first"); return PTAMD_ERR_STATE; }
  if (!d_state || !d_hdr || !d_count) { ptamd::set_error("null buffer"); return PTAMD_ERR_ARG; }
  HIPCHK(hipSetDevice(c->device));
  Launcher L{c, (hipStream_t)stream};
  L.begin(PTAMD_K_OTHER);
  hipLaunchKernelGGL(k_frame_resolve, grid_for(c, 4), 256, 0, (hipStream_t)stream, d_state,
                     (uint32_t)(c->params.w * c->params.h), d_hdr, d_count);
  L.end();
  HIPCHK(hipGetLastError());
  return PTAMD_OK;
}

int ptamd_get_stats(ptamd_context *c, ptamd_stats *out) {
  if (!c || !out) { ptamd::set_error("null argument"); return PTAMD_ERR_ARG; }
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipDeviceSynchronize());
  collect_events(c);
  if (c->ctl) {
    Ctl ring[CTL_RING], h{};  // the statistics of a frame are spread over the depths' control blocks
    HIPCHK(hipMemcpy(ring, c->ctl, sizeof ring, hipMemcpyDeviceToHost));
    for (const Ctl &r : ring) {
      h.st_rays += r.st_rays; h.st_shadow += r.st_shadow; h.st_vertices += r.st_vertices; h.st_nodes += r.st_nodes;
      h.st_tests += r.st_tests; h.st_exact += r.st_exact; h.st_samples += r.st_samples; h.st_dead += r.st_dead;
      h.st_two_ext += r.st_two_ext; h.st_two_shadow += r.st_two_shadow;
    }
    c->stats.rays = h.st_rays - h.st_dead;  // path slots a round left unused are not rays
    c->stats.shadow_rays = h.st_shadow;
    c->stats.path_vertices = h.st_vertices;
    c->stats.nodes_visited = h.st_nodes;
    c->stats.prim_tests = h.st_tests;
    c->stats.exact_tests = h.st_exact;
    c->stats.samples = h.st_samples;
    c->stats.two_ext_rays = h.st_two_ext;
    c->stats.two_shadow_rays = h.st_two_shadow;
    c->stats.unused_slots = h.st_dead;
  }
  *out = c->stats;
  return PTAMD_OK;
}

namespace {
// Buffers of ptamd_render / ptamd_render_list, kept by the context between frames (grown on demand).
int ensure_frame_scratch(ptamd_context *c, int64_t npix, int64_t nlist, bool outputs) {
  FrameScratch &f = c->fs;
  if (!f.d_n) {
    HIPCHK(hipMalloc((void **)&f.d_n, 2 * sizeof(int64_t)));
    HIPCHK(hipHostMalloc((void **)&f.h_n, 2 * sizeof(int64_t), hipHostMallocDefault));
    HIPCHK(hipEventCreateWithFlags(&f.ev[0], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&f.ev[1], hipEventDisableTiming));
  }
  if (nlist > f.cap_list) {
    HIPCHK(hipDeviceSynchronize());
    for (int k = 0; k < 3; k++) { if (f.d_list[k]) HIPCHK(hipFree(f.d_list[k])); f.d_list[k] = nullptr; }
    for (int k = 0; k < 3; k++) HIPCHK(hipMalloc((void **)&f.d_list[k], std::max<int64_t>(nlist, 1) * sizeof(int32_t)));
    f.cap_list = nlist;
  }
  if (outputs && npix > f.cap_pix) {
    HIPCHK(hipDeviceSynchronize());
    if (f.d_state) HIPCHK(hipFree(f.d_state));
    if (f.d_hdr) HIPCHK(hipFree(f.d_hdr));
    if (f.d_count) HIPCHK(hipFree(f.d_count));
    f.d_state = f.d_hdr = nullptr; f.d_count = nullptr;
    HIPCHK(hipMalloc((void **)&f.d_state, npix * 6 * sizeof(double)));
    HIPCHK(hipMalloc((void **)&f.d_hdr, npix * 3 * sizeof(double)));
    HIPCHK(hipMalloc((void **)&f.d_count, npix * sizeof(int32_t)));
    f.cap_pix = npix;
  }
  return PTAMD_OK;
}
void free_frame_scratch(ptamd_context *c) {
  FrameScratch &f = c->fs;
  for (int k = 0; k < 3; k++) if (f.d_list[k]) (void)hipFree(f.d_list[k]);
  if (f.d_state) (void)hipFree(f.d_state);
  if (f.d_hdr) (void)hipFree(f.d_hdr);
  if (f.d_count) (void)hipFree(f.d_count);
  if (f.d_n) (void)hipFree(f.d_n);
  if (f.h_n) (void)hipHostFree(f.h_n);
  for (hipEvent_t e : f.ev) if (e) (void)hipEventDestroy(e);
  f = FrameScratch{};
}
}  // namespace

int ptamd_render_list(ptamd_context *c, const ptamd_render_params *p, const int32_t *d_pixels, int64_t n, double *d_state,
                      void *stream) {
  if (!c || !p || n < 0 || (n > 0 && (!d_pixels || !d_state))) { ptamd::set_error("invalid arguments to ptamd_render_list"); return PTAMD_ERR_ARG; }
  int rc = ptamd_frame_begin(c, p);
  if (rc) return rc;
  if (n == 0) return PTAMD_OK;
  rc = ensure_frame_scratch(c, 0, n, false);
  if (rc) return rc;
  FrameScratch &f = c->fs;
  hipStream_t s = (hipStream_t)stream;
  {
    const int64_t init[2] = {n, n};
    HIPCHK(hipMemcpyAsync(f.d_n, init, sizeof init, hipMemcpyHostToDevice, s));
  }
  // Round r traces the survivors of the adaptive test after round r - 1. Their
  // number stays on the device; the host only needs an upper bound to size the
  // launches. A long list's bound is the count of one round earlier, read back
  // asynchronously, so the GPU does not wait for the host between big rounds; a
  // list of up to sync_below pixels is counted before its round is sized (the GPU
  // idles for tens of microseconds; a round sized by an old count looks less far
  // ahead than it could, which costs rounds).
  //
  // A round too small to fill the GPU traces several batches per pixel ahead of
  // their adaptive tests (k_accumulate applies them and drops what was not
  // needed). Frames of many batches per pixel do it PER PIXEL: every pixel goes
  // on from its own count, as far as its running statistics predict it needs
  // (RoundPlan::planned), and no further than `growth` times the look-ahead of
  // the rounds before, because a prediction from one or two batches is rough.
  // Measured, 1024 samples in batches of 32: 1 to 17 % faster than all pixels in
  // lockstep, more so the fewer pixels a GPU has; 256 samples: 3 to 10 % slower.
  const int batch = std::max(1, p->samples_per_batch);
  bool per_pixel = p->ns_aa >= 16 * batch;
  if (const char *e = getenv("PTAMD_PER_PIXEL")) per_pixel = atoi(e) != 0;
  const int64_t sync_below = 200000;
  const int growth = 4;
  const int32_t *lists[2] = {d_pixels, nullptr};  // current list: the caller's, then the context's two in turn
  int32_t *own[2] = {f.d_list[0], f.d_list[1]};
  int64_t n_known = n;
  // per_pixel: `done` is the least count a listed pixel can have (every round gives a pixel a batch at least)
  int cur = 0, done = 0, ahead = 0;
  for (int r = 0; done < p->ns_aa && n_known > 0; r++) {
    ptamd_pixel_list l{lists[cur], f.d_n + cur, 0, n_known, 0, 1};
    if (r > 0) {
      int32_t *out = own[r & 1];
      rc = ptamd_filter_list(c, &l, per_pixel ? -1 : done, d_state, out, f.d_n + (cur ^ 1), s);
      if (rc) return rc;
      cur ^= 1;
      lists[cur] = out;
      HIPCHK(hipMemcpyAsync(f.h_n + (r & 1), f.d_n + cur, sizeof(int64_t), hipMemcpyDeviceToHost, s));
      HIPCHK(hipEventRecord(f.ev[r & 1], s));
      if (r > 1) {
        HIPCHK(hipEventSynchronize(f.ev[(r - 1) & 1]));
        n_known = std::min<int64_t>(n_known, f.h_n[(r - 1) & 1]);
      }
      if (n_known <= sync_below) {
        HIPCHK(hipEventSynchronize(f.ev[r & 1]));
        n_known = std::min<int64_t>(n_known, f.h_n[r & 1]);
      }
      if (n_known == 0) break;
      l = ptamd_pixel_list{lists[cur], f.d_n + cur, 0, n_known, 0, 1};
    }
    const int64_t paths = n_known * batch;
    // (most pixels stop at their first test, so the first round traces ahead only when really small)
    const int64_t target = r == 0 ? (4 << 20) : (16 << 20);
    const int64_t left = p->ns_aa - done;
    int64_t k = std::max<int64_t>(1, std::min<int64_t>({target / std::max<int64_t>(paths, 1), 32, (left + batch - 1) / batch}));
    int ns = (int)(k * batch);
    if (!per_pixel) ns = (int)std::min<int64_t>(ns, left);
    else if (ahead > 0) ns = std::max(batch, std::min(ns, growth * ahead / batch * batch));
    ahead += ns;
    rc = ptamd_trace_list(c, &l, per_pixel ? -1 : done, ns, d_state, s);
    if (rc) return rc;
    done += per_pixel ? batch : ns;
  }
  return PTAMD_OK;
}

int ptamd_render(ptamd_context *c, const ptamd_render_params *p, double *hdr_out, int32_t *count_out) {
  if (!c || !p || !hdr_out || !count_out) { ptamd::set_error("null argument"); return PTAMD_ERR_ARG; }
  HIPCHK(hipSetDevice(c->device));
  const int x0 = std::max(0, p->x0), y0 = std::max(0, p->y0), x1 = std::min(p->w, p->x1), y1 = std::min(p->h, p->y1);
  const int64_t npix = (int64_t)p->w * p->h;
  const int64_t nrect = (x1 > x0 && y1 > y0) ? (int64_t)(x1 - x0) * (y1 - y0) : 0;
  if (npix <= 0) { ptamd::set_error("empty frame"); return PTAMD_ERR_ARG; }
  int rc = ensure_frame_scratch(c, npix, nrect, true);
  if (rc) return rc;
  FrameScratch &f = c->fs;
  hipStream_t s = c->own_stream;
  HIPCHK(hipMemsetAsync(f.d_state, 0, npix * 6 * sizeof(double), s));
  if (nrect) hipLaunchKernelGGL(k_iota, grid_for(c, 4), 256, 0, s, f.d_list[2], (uint32_t)nrect, p->w, x0, y0, x1 - x0);
  HIPCHK(hipStreamSynchronize(s));
  auto t0 = std::chrono::steady_clock::now();
  rc = ptamd_render_list(c, p, f.d_list[2], nrect, f.d_state, s);
  if (rc) return rc;
  rc = ptamd_frame_resolve(c, f.d_state, f.d_hdr, f.d_count, s);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(s));
  c->stats.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  HIPCHK(hipMemcpy(hdr_out, f.d_hdr, npix * 3 * sizeof(double), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(count_out, f.d_count, npix * sizeof(int32_t), hipMemcpyDeviceToHost));
  return PTAMD_OK;
}

int ptamd_trace_primary(ptamd_context *c, int64_t n, const double *sx, const double *sy, int32_t *prim_out,
                        double *t_out) {
  if (!c || n < 0 || (n > 0 && (!sx || !sy || !prim_out || !t_out))) { ptamd::set_error("invalid arguments"); return PTAMD_ERR_ARG; }
  ptamd_render_params p{};
  p.w = 1; p.h = 1; p.x1 = 1; p.y1 = 1; p.ns_aa = 1; p.samples_per_batch = 1;
  // (a frame that is open keeps its own parameters: these are local)
  FrameParams fp{};
  int rc = setup_frame_params(c, &p, fp);
  if (rc) return rc;
  HIPCHK(hipSetDevice(c->device));
  const int64_t chunk = std::min<int64_t>(std::max<int64_t>(n, 1), c->max_paths);
  rc = ensure_buffers(c, chunk, std::max(c->cap_nls, 1), false, false);
  if (rc) return rc;
  TravLaunch tl;
  rc = traversal_launch(c, &tl);
  if (rc) return rc;
  const int tblock = tl.block, tgrid = tl.grid_extend;
  const size_t lds = tl.lds_extend;
  hipStream_t s = c->own_stream;
  // the queue's throughput arrays are unused here and double as staging for sx, sy
  double *d_sx = c->q[0].T[0], *d_sy = c->q[0].T[1];
  std::vector<int32_t> prim((size_t)chunk);
  for (int64_t b = 0; b < n; b += chunk) {
    const int64_t m = std::min(chunk, n - b);
    HIPCHK(hipMemcpyAsync(d_sx, sx + b, m * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d_sy, sy + b, m * sizeof(double), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_set_counts, 1, 1, 0, s, c->ctl, PixView{nullptr, nullptr, (long long)m, 0u, 1u}, 0u, (uint32_t)m, 1u);
    hipLaunchKernelGGL(k_generate_primary, grid_for(c, 4), 256, 0, s, fp, d_sx, d_sy, (uint32_t)m, c->q[0]);
    hipLaunchKernelGGL((k_extend<false>), tgrid, tblock, lds, s, c->S, c->q[0], (const int32_t *)nullptr, c->hb, c->ctl, fp.n_clip, fp.f_clip, tl.gstack, (const int32_t *)nullptr, (const float *)nullptr, fp.cam_pos);
    hipLaunchKernelGGL(k_primary_t, grid_for(c, 4), 256, 0, s, c->S, c->q[0], (const int32_t *)c->hb.prim, (uint32_t)m, fp.n_clip, c->hb.t, fp.cam_pos);
    HIPCHK(hipMemcpyAsync(prim.data(), c->hb.prim, m * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(t_out + b, c->hb.t, m * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    for (int64_t i = 0; i < m; i++) {
      prim_out[b + i] = prim[i] >= 0 ? c->dev2ref[prim[i]] : -1;
      if (prim[i] < 0) t_out[b + i] = -1.0;
    }
  }
  return PTAMD_OK;
}

}  // extern "C"
