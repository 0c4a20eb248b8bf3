// The GPUs of one node rendering one frame together, from C++: the counterpart of the
// reference's worker pool (src/pathtracer/raytraced_renderer.cpp:297-302, 326-328, 644-658:
// numWorkerThreads threads taking 32x32 tiles from a queue).
//
// One host thread per GPU. The scene is replicated; the pixels of the frame's rectangle are
// dealt round-robin (pixel k of the rectangle goes to GPU k % n), which spreads the expensive
// regions of a frame evenly, so no queue is needed inside the frame. Every GPU runs the whole
// adaptive loop on its own pixels (ptamd_render_list) without talking to the others; at the
// end the per-pixel sums are gathered on GPU 0 with RCCL (one send per GPU, point to point
// over xGMI), where the frame is resolved. Results do not depend on n: the random numbers of a
// path are keyed by (seed, pixel, sample).
//
// Built into libptamd_multi.so, so that single-GPU users of libptamd.so do not load RCCL.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ptamd.h"
#include "error_state.h"

// PTAMD_MULTI_TEST_HOOKS is defined only by the TEST build of this file (tests/_build/libptamd_multi_fake.so,
// `make -C pathtracer_amd/csrc test_multi`), which also swaps RCCL for an in-process stand-in: several logical
// ranks may then share one physical GPU, and a test can make one rank's part of a frame fail. libptamd_multi.so
// is built without it.
#ifdef PTAMD_MULTI_TEST_HOOKS
#include <atomic>
static std::atomic<int> g_test_fail_rank{-1};
#endif

// Where the host threads of a frame meet before the exchange: a GPU whose part failed must not
// leave the others waiting in a send or receive that will never be matched.
struct Rendezvous {
  std::mutex mu;
  std::condition_variable cv;
  int waiting = 0;
  unsigned generation = 0;
  bool failed = false, result = false;
  // blocks until all n threads have arrived; true if any of them reported a failure
  bool arrive(bool i_failed, int n) {
    std::unique_lock<std::mutex> lk(mu);
    failed = failed || i_failed;
    if (++waiting == n) {
      result = failed;
      failed = false;
      waiting = 0;
      generation++;
      cv.notify_all();
      return result;
    }
    const unsigned g = generation;
    cv.wait(lk, [&] { return generation != g; });
    return result;
  }
};

struct ptamd_multi {
  Rendezvous meet;
  std::vector<int32_t> devices;
  std::vector<ptamd_context *> ctx;
  std::vector<ncclComm_t> comm;
  std::vector<hipStream_t> stream;
  // per GPU: the whole rectangle's pixel list, this GPU's own pixels, state, its packed rows
  struct PerGpu {
    int32_t *d_all = nullptr, *d_mine = nullptr;
    double *d_state = nullptr, *d_send = nullptr;
    int64_t cap_all = 0, cap_mine = 0, cap_pix = 0, cap_rows = 0;
  };
  std::vector<PerGpu> gpu;
  // GPU 0: everybody's rows, the resolved frame
  double *d_recv = nullptr, *d_hdr = nullptr;
  int32_t *d_count = nullptr;
  int64_t cap_recv = 0, cap_hdr = 0, cap_count = 0;
  double seconds = 0;
};

namespace {

std::string hip_msg(const char *what, hipError_t e) { return std::string(what) + ": " + hipGetErrorString(e); }
std::string nccl_msg(const char *what, ncclResult_t e) { return std::string(what) + ": " + ncclGetErrorString(e); }

#define M_HIP(expr)                                    \
  do {                                                 \
    hipError_t e_ = (expr);                            \
    if (e_ != hipSuccess) { err = hip_msg(#expr, e_); return PTAMD_ERR_DEVICE; } \
  } while (0)
#define M_NCCL(expr)                                   \
  do {                                                 \
    ncclResult_t e_ = (expr);                          \
    if (e_ != ncclSuccess) { err = nccl_msg(#expr, e_); return PTAMD_ERR_DEVICE; } \
  } while (0)
#define M_PT(expr)                                     \
  do {                                                 \
    int rc_ = (expr);                                  \
    if (rc_) { err = ptamd_last_error(); return rc_; } \
  } while (0)

template <typename T>
int grow(T **p, int64_t *cap, int64_t need, std::string &err) {
  if (need <= *cap) return PTAMD_OK;
  if (*p) M_HIP(hipFree(*p));
  *p = nullptr;
  M_HIP(hipMalloc((void **)p, (size_t)std::max<int64_t>(need, 1) * sizeof(T)));
  *cap = need;
  return PTAMD_OK;
}

// GPU `r`'s own part of a frame: its pixels traced, its rows packed (all queued on its stream);
// errors come back as text (ptamd_last_error is per thread)
int trace_rank(ptamd_multi *m, int r, const ptamd_render_params *p, const std::vector<int32_t> &all, std::string &err) {
  const int n = (int)m->ctx.size();
  ptamd_multi::PerGpu &g = m->gpu[r];
  hipStream_t s = m->stream[r];
  M_HIP(hipSetDevice(m->devices[r]));
#ifdef PTAMD_MULTI_TEST_HOOKS
  {
    int want = r;
    if (g_test_fail_rank.compare_exchange_strong(want, -1)) { err = "failure injected by the test"; return PTAMD_ERR_DEVICE; }
  }
#endif
  const int64_t npix = (int64_t)p->w * p->h, n_all = (int64_t)all.size();
  const int64_t n_mine = ptamd_multi_shard_size(n_all, r, n), rows = ptamd_multi_rows_per_rank(n_all, n);
  if (int rc = grow(&g.d_all, &g.cap_all, n_all, err)) return rc;
  if (int rc = grow(&g.d_mine, &g.cap_mine, n_mine, err)) return rc;
  if (int rc = grow(&g.d_state, &g.cap_pix, npix * 6, err)) return rc;
  if (int rc = grow(&g.d_send, &g.cap_rows, rows * 6, err)) return rc;
  std::vector<int32_t> mine((size_t)n_mine);
  for (int64_t k = 0; k < n_mine; k++) mine[(size_t)k] = all[(size_t)(r + k * n)];
  if (n_all) M_HIP(hipMemcpyAsync(g.d_all, all.data(), (size_t)n_all * sizeof(int32_t), hipMemcpyHostToDevice, s));
  if (n_mine) M_HIP(hipMemcpyAsync(g.d_mine, mine.data(), (size_t)n_mine * sizeof(int32_t), hipMemcpyHostToDevice, s));
  M_HIP(hipMemsetAsync(g.d_state, 0, (size_t)npix * 6 * sizeof(double), s));
  M_HIP(hipStreamSynchronize(s));  // `mine` is about to go out of use on the host side of the copy
  M_PT(ptamd_render_list(m->ctx[r], p, g.d_mine, n_mine, g.d_state, s));
  if (n > 1 && n_all > 0 && r > 0) {
    const ptamd_pixel_list shard{g.d_all, nullptr, n_all, n_all, r, n};
    M_HIP(hipMemsetAsync(g.d_send, 0, (size_t)rows * 6 * sizeof(double), s));
    M_PT(ptamd_pack_rows(m->ctx[r], &shard, g.d_state, g.d_send, s));
  }
  return PTAMD_OK;
}

// the one exchange of the frame (every GPU's finished rows to GPU 0) and, on GPU 0, the frame
int gather_rank(ptamd_multi *m, int r, const ptamd_render_params *p, const std::vector<int32_t> &all,
                double *hdr_out, int32_t *count_out, std::string &err) {
  const int n = (int)m->ctx.size();
  ptamd_multi::PerGpu &g = m->gpu[r];
  hipStream_t s = m->stream[r];
  const int64_t npix = (int64_t)p->w * p->h, n_all = (int64_t)all.size();
  const int64_t rows = ptamd_multi_rows_per_rank(n_all, n);
  const ptamd_pixel_list whole{g.d_all, nullptr, n_all, n_all, 0, 1};
  if (n > 1 && n_all > 0) {
    M_NCCL(ncclGroupStart());
    if (r == 0) {
      for (int peer = 1; peer < n; peer++)
        M_NCCL(ncclRecv(m->d_recv + (size_t)peer * rows * 6, (size_t)rows * 6, ncclDouble, peer, m->comm[0], s));
    } else {
      M_NCCL(ncclSend(g.d_send, (size_t)rows * 6, ncclDouble, 0, m->comm[r], s));
    }
    M_NCCL(ncclGroupEnd());
    if (r == 0) M_PT(ptamd_unpack_rows(m->ctx[0], &whole, n, 0, m->d_recv, rows, g.d_state, s));
  }
  if (r == 0) {
    M_PT(ptamd_frame_resolve(m->ctx[0], g.d_state, m->d_hdr, m->d_count, s));
    M_HIP(hipMemcpyAsync(hdr_out, m->d_hdr, (size_t)npix * 3 * sizeof(double), hipMemcpyDeviceToHost, s));
    M_HIP(hipMemcpyAsync(count_out, m->d_count, (size_t)npix * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  }
  M_HIP(hipStreamSynchronize(s));
  return PTAMD_OK;
}

// runs f(rank) on one thread per GPU; the first failure's code and text are the result
template <class F>
int on_every_gpu(ptamd_multi *m, F f) {
  const int n = (int)m->ctx.size();
  std::vector<int> rc((size_t)n, PTAMD_OK);
  std::vector<std::string> err((size_t)n);
  if (n == 1) {
    rc[0] = f(0, err[0]);
  } else {
    std::vector<std::thread> th;
    for (int r = 0; r < n; r++) th.emplace_back([&, r]() { rc[(size_t)r] = f(r, err[(size_t)r]); });
    for (auto &t : th) t.join();
  }
  for (int r = 0; r < n; r++)
    if (rc[(size_t)r]) {
      ptamd::set_error("GPU " + std::to_string(m->devices[(size_t)r]) + ": " + err[(size_t)r]);
      return rc[(size_t)r];
    }
  return PTAMD_OK;
}

}  // namespace

extern "C" {

int64_t ptamd_multi_shard_size(int64_t n, int32_t rank, int32_t world) {
  if (world < 1 || rank < 0 || rank >= world || n <= rank) return 0;
  return (n - rank + world - 1) / world;
}

int64_t ptamd_multi_rows_per_rank(int64_t n, int32_t world) {
  if (world < 1 || n <= 0) return 0;
  return (n + world - 1) / world;
}

int ptamd_multi_create(const int32_t *device_ids, int32_t n, ptamd_multi **out) {
  if (!device_ids || n < 1 || !out) { ptamd::set_error("invalid arguments to ptamd_multi_create"); return PTAMD_ERR_ARG; }
#ifndef PTAMD_MULTI_TEST_HOOKS
  for (int a = 0; a < n; a++)
    for (int b = a + 1; b < n; b++)
      if (device_ids[a] == device_ids[b]) { ptamd::set_error("ptamd_multi_create: a device is listed twice"); return PTAMD_ERR_ARG; }
#endif
  ptamd_multi *m = new ptamd_multi;
  m->devices.assign(device_ids, device_ids + n);
  m->gpu.resize((size_t)n);
  for (int r = 0; r < n; r++) {
    ptamd_context *c = nullptr;
    const int rc = ptamd_create(device_ids[r], &c);
    if (rc) { ptamd_multi_destroy(m); return rc; }
    m->ctx.push_back(c);
    hipStream_t s = nullptr;
    if (hipSetDevice(device_ids[r]) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
      ptamd::set_error("ptamd_multi_create: cannot create a stream");
      ptamd_multi_destroy(m);
      return PTAMD_ERR_DEVICE;
    }
    m->stream.push_back(s);
  }
  if (n > 1) {
    m->comm.resize((size_t)n);
    const ncclResult_t e = ncclCommInitAll(m->comm.data(), n, device_ids);
    if (e != ncclSuccess) {
      m->comm.clear();
      ptamd::set_error(nccl_msg("ncclCommInitAll", e));
      ptamd_multi_destroy(m);
      return PTAMD_ERR_DEVICE;
    }
  }
  *out = m;
  return PTAMD_OK;
}

void ptamd_multi_destroy(ptamd_multi *m) {
  if (!m) return;
  for (size_t r = 0; r < m->ctx.size(); r++) {
    (void)hipSetDevice(m->devices[r]);
    if (r < m->stream.size() && m->stream[r]) { (void)hipStreamSynchronize(m->stream[r]); }
    if (r < m->comm.size()) (void)ncclCommDestroy(m->comm[r]);
    ptamd_multi::PerGpu &g = m->gpu[r];
    (void)hipFree(g.d_all); (void)hipFree(g.d_mine); (void)hipFree(g.d_state); (void)hipFree(g.d_send);
    if (r == 0) { (void)hipFree(m->d_recv); (void)hipFree(m->d_hdr); (void)hipFree(m->d_count); }
    if (r < m->stream.size() && m->stream[r]) (void)hipStreamDestroy(m->stream[r]);
    ptamd_destroy(m->ctx[r]);
  }
  delete m;
}

int32_t ptamd_multi_size(const ptamd_multi *m) { return m ? (int32_t)m->ctx.size() : 0; }

int ptamd_multi_upload_scene(ptamd_multi *m, const ptamd_scene_desc *d) {
  if (!m || !d) { ptamd::set_error("null argument"); return PTAMD_ERR_ARG; }
  return on_every_gpu(m, [&](int r, std::string &err) {
    const int rc = ptamd_upload_scene(m->ctx[(size_t)r], d);
    if (rc) err = ptamd_last_error();
    return rc;
  });
}

int ptamd_multi_set_camera(ptamd_multi *m, const ptamd_camera *cam) {
  if (!m || !cam) { ptamd::set_error("null argument"); return PTAMD_ERR_ARG; }
  for (ptamd_context *c : m->ctx) {
    const int rc = ptamd_set_camera(c, cam);
    if (rc) return rc;
  }
  return PTAMD_OK;
}

int ptamd_multi_set_profiling(ptamd_multi *m, int32_t events, int32_t counters) {
  if (!m) { ptamd::set_error("null argument"); return PTAMD_ERR_ARG; }
  for (ptamd_context *c : m->ctx) {
    const int rc = ptamd_set_profiling(c, events, counters);
    if (rc) return rc;
  }
  return PTAMD_OK;
}

int ptamd_multi_render(ptamd_multi *m, const ptamd_render_params *p, double *hdr_out, int32_t *count_out) {
  if (!m || !p || !hdr_out || !count_out || p->w <= 0 || p->h <= 0) { ptamd::set_error("invalid arguments to ptamd_multi_render"); return PTAMD_ERR_ARG; }
  const int n = (int)m->ctx.size();
  const int x0 = std::max(0, p->x0), y0 = std::max(0, p->y0), x1 = std::min(p->w, p->x1), y1 = std::min(p->h, p->y1);
  std::vector<int32_t> all;
  if (x1 > x0 && y1 > y0) {
    all.reserve((size_t)(x1 - x0) * (size_t)(y1 - y0));
    for (int y = y0; y < y1; y++)
      for (int x = x0; x < x1; x++) all.push_back(y * p->w + x);
  }
  const int64_t npix = (int64_t)p->w * p->h, rows = ptamd_multi_rows_per_rank((int64_t)all.size(), n);
  {
    std::string err;
    if (hipSetDevice(m->devices[0]) != hipSuccess) { ptamd::set_error("hipSetDevice failed"); return PTAMD_ERR_DEVICE; }
    int rc = grow(&m->d_recv, &m->cap_recv, (int64_t)n * rows * 6, err);
    if (!rc) rc = grow(&m->d_hdr, &m->cap_hdr, npix * 3, err);
    if (!rc) rc = grow(&m->d_count, &m->cap_count, npix, err);
    if (rc) { ptamd::set_error(err); return rc; }
  }
  const auto t0 = std::chrono::steady_clock::now();
  const int rc = on_every_gpu(m, [&](int r, std::string &err) {
    const int rc_trace = trace_rank(m, r, p, all, err);
    // nobody enters the exchange unless everybody can
    if (m->meet.arrive(rc_trace != PTAMD_OK, n)) {
      (void)hipStreamSynchronize(m->stream[(size_t)r]);
      if (rc_trace == PTAMD_OK) err = "another GPU's part of the frame failed";
      return rc_trace != PTAMD_OK ? rc_trace : (int)PTAMD_ERR_DEVICE;
    }
    return gather_rank(m, r, p, all, hdr_out, count_out, err);
  });
  m->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return rc;
}

int ptamd_multi_get_stats(ptamd_multi *m, int32_t rank, ptamd_stats *out) {
  if (!m || !out || rank < 0 || rank >= (int32_t)m->ctx.size()) { ptamd::set_error("invalid arguments to ptamd_multi_get_stats"); return PTAMD_ERR_ARG; }
  const int rc = ptamd_get_stats(m->ctx[(size_t)rank], out);
  if (!rc) out->seconds = m->seconds;  // the frame's wall time, uploads of the pixel lists and the gather included
  return rc;
}

#ifdef PTAMD_MULTI_TEST_HOOKS
// the next frame's part of logical rank `rank` fails before it starts (-1: nobody's)
void ptamd_multi_test_fail_rank(int32_t rank) { g_test_fail_rank.store(rank); }
// Rendezvous::arrive with n threads over `generations` meetings, thread `failing` (or nobody: -1) reporting a
// failure in meeting `failing_generation`: returns 0 if every thread of every meeting got that meeting's
// verdict (true exactly in the failing one), i.e. a failure releases everybody and does not leak into the next
int32_t ptamd_multi_test_rendezvous(int32_t n, int32_t generations, int32_t failing, int32_t failing_generation) {
  Rendezvous meet;
  std::atomic<int> wrong{0};
  std::vector<std::thread> th;
  for (int t = 0; t < n; t++)
    th.emplace_back([&, t]() {
      for (int g = 0; g < generations; g++) {
        const bool verdict = meet.arrive(t == failing && g == failing_generation, n);
        if (verdict != (failing >= 0 && failing < n && g == failing_generation)) wrong++;
      }
    });
  for (auto &t : th) t.join();
  return wrong.load();
}
#endif

}  // extern "C"
