/* ptamd — C ABI of the MI355X render hot path.
 *
 * The reference (Sh-Anand/pathtracer) has no FFI; its seam is internal C++:
 * RaytracedRenderer owns a PathTracer and a BVHAccel and its worker threads call
 * pt->raytrace_pixel(x, y) per pixel (src/pathtracer/raytraced_renderer.cpp:
 * 268-329, 585-610, 644-678). This header is the boundary a maintainer would
 * bind instead: plain C structs, pointers and sizes; every function returns an
 * int status (0 = ok) and ptamd_last_error() explains a failure. Nothing here
 * ever calls exit().
 *
 * Conventions shared by every entry point
 *  - all geometry is FP64, in the reference's BVH primitive order;
 *  - pixel (x, y) has index x + y*w with y = 0 the BOTTOM row, as in the
 *    reference's sampleBuffer (src/pathtracer/pathtracer.cpp:270-271);
 *  - "hdr" is 3 doubles per pixel (mean radiance), "count" one int32 per pixel
 *    (samples actually taken, src/pathtracer/pathtracer.cpp:271).
 */
#ifndef PTAMD_H
#define PTAMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTAMD_OK 0
#define PTAMD_ERR_ARG 1     /* bad argument / shape mismatch            */
#define PTAMD_ERR_IO 2      /* file missing or unparsable               */
#define PTAMD_ERR_SCENE 3   /* scene violates a loader/builder invariant */
#define PTAMD_ERR_DEVICE 4  /* HIP call failed or no GPU                */
#define PTAMD_ERR_STATE 5   /* call order wrong (e.g. render before upload) */

/* primitive / material / light kinds used in ptamd_scene_desc */
#define PTAMD_PRIM_TRIANGLE 0
#define PTAMD_PRIM_SPHERE 1
#define PTAMD_MAT_DIFFUSE 0
#define PTAMD_MAT_EMISSION 1
#define PTAMD_MAT_MIRROR 2
#define PTAMD_MAT_MICROFACET 3
#define PTAMD_MAT_REFRACTION 4
#define PTAMD_MAT_GLASS 5
#define PTAMD_LIGHT_DIRECTIONAL 0
#define PTAMD_LIGHT_INF_HEMISPHERE 1
#define PTAMD_LIGHT_POINT 2
#define PTAMD_LIGHT_SPOT 3
#define PTAMD_LIGHT_AREA 4

/* Flattened static scene. Replaces the pointer graph the reference builds in
 * RaytracedRenderer::set_scene / build_accel (raytraced_renderer.cpp:106-128,
 * 351-375): Scene{objects, lights}, Triangle/Sphere objects, BVHAccel::nodes.
 * The caller keeps ownership of every array. */
typedef struct ptamd_scene_desc {
  int32_t n_prims, n_nodes, n_materials, n_lights, root, jitter_n;
  const int32_t *prim_type;     /* [n_prims]                                          */
  const double *prim_pos;       /* [n_prims*9] tri: p1,p2,p3; sphere: c.xyz, r, r*r    */
  const double *prim_nrm;       /* [n_prims*9] tri: n1,n2,n3 (not renormalised)        */
  const int32_t *prim_material; /* [n_prims] index into mat_*                          */
  const double *node_bbox;      /* [n_nodes*6] min.xyz, max.xyz                        */
  const int32_t *node_left;     /* [n_nodes] -1 for leaves                             */
  const int32_t *node_right;    /* [n_nodes]                                           */
  const int32_t *node_start;    /* [n_nodes] leaf: first primitive                     */
  const int32_t *node_count;    /* [n_nodes] leaf: primitive count (>0), interior: 0   */
  const int32_t *mat_kind;      /* [n_materials]                                       */
  const double *mat_param;      /* [n_materials*3] reflectance | radiance | ...        */
  const int32_t *light_kind;    /* [n_lights]                                          */
  const double *light_param;    /* [n_lights*16] radiance, position, direction,
                                   dim_x, dim_y (3 each), area                         */
  const double *jitter_table;   /* [jitter_n*jitter_n*2] pre-jittered sub-pixel table
                                   (src/pathtracer/sampler.h:78-89)                    */
  const double *mat_extra;      /* [n_materials*4] or NULL: what the reference parses into
                                   its specular BSDFs and never uses (bsdf.h MirrorBSDF /
                                   RefractionBSDF / GlassBSDF fields): glass reflectance rgb,
                                   ior. mat_param holds mirror reflectance or refraction /
                                   glass transmittance. Read only when
                                   ptamd_render_params.specular is set.                  */
} ptamd_scene_desc;

/* Replaces the Camera fields generate_ray reads (src/pathtracer/camera.cpp:190-203). */
typedef struct ptamd_camera {
  double pos[3];
  double c2w[9]; /* column-major: c2w[3*col + row] */
  double hFov, vFov, nClip, fClip; /* degrees */
} ptamd_camera;

/* Replaces the PathTracer public configuration fields
 * (src/pathtracer/pathtracer.h:73-83) plus the frame/cell rectangle of
 * RaytracedRenderer (raytraced_renderer.cpp:159-177, 303-321). */
typedef struct ptamd_render_params {
  int32_t w, h;               /* frame size                                         */
  int32_t x0, y0, x1, y1;     /* half-open pixel rectangle to render                */
  int32_t ns_aa;              /* -s samples per pixel                               */
  int32_t max_ray_depth;      /* -m                                                 */
  int32_t accumulate_bounces; /* -o                                                 */
  int32_t ns_area_light;      /* -l                                                 */
  int32_t samples_per_batch;  /* -a (first word)                                    */
  int32_t hemisphere;         /* -H direct lighting by uniform hemisphere sampling  */
  double max_tolerance;       /* -a (second word)                                   */
  uint64_t seed;              /* key of the counter-based per-pixel RNG             */
  int32_t specular;           /* 0: Mirror / Refraction / Glass are the black stubs of
                                 the reference (advanced_bsdf.cpp:21-27,99-106,124-134).
                                 1: they are the ideal specular BSDFs those stubs were
                                 meant to become (see "specular mode" below). No
                                 counterpart in the reference; not a parity mode.      */
  int32_t reserved;
} ptamd_render_params;

/* Specular mode (ptamd_render_params.specular = 1). In the shading frame (z = shading
 * normal, wo = direction back along the ray):
 *   reflect(wo)       = (-wo.x, -wo.y, wo.z)
 *   refract(wo, ior)  : eta = wo.z > 0 ? 1/ior : ior; c2 = 1 - eta^2 (1 - wo.z^2);
 *                       c2 < 0: total internal reflection; else
 *                       wi = (-eta wo.x, -eta wo.y, wo.z > 0 ? -sqrt(c2) : sqrt(c2))
 *   Mirror     : wi = reflect, pdf 1, f = reflectance / |wi.z|
 *   Refraction : TIR -> f = 0 (path ends); else pdf 1, f = transmittance / |wi.z| / eta^2
 *   Glass      : TIR -> as Mirror; else R = R0 + (1 - R0)(1 - |wo.z|)^5, R0 = ((1-ior)/(1+ior))^2;
 *                with probability R: reflect, pdf R, f = R reflectance / |wi.z|;
 *                else refract, pdf 1 - R, f = (1 - R) transmittance / |wi.z| / eta^2
 * (the first BSDF random number of the vertex decides). These BSDFs are Dirac deltas:
 * a vertex on one takes no light samples, and an emitter hit by a ray that left such
 * a vertex adds its radiance (light sampling could not have accounted for it). In this
 * mode a vertex whose BSDF is identically zero (an emitter, the microfacet stub) ends
 * its path. Russian roulette, depth limits and the -o switch work as in the reference. */

typedef struct ptamd_stats {
  double seconds;          /* wall time of the last render (device work only)     */
  uint64_t samples;        /* sum of per-pixel sample counts                      */
  uint64_t rays;           /* camera + extension + shadow rays traced             */
  uint64_t shadow_rays;
  uint64_t path_vertices;  /* shaded hits                                         */
  uint64_t nodes_visited;  /* BVH nodes popped (only when counters are enabled)   */
  uint64_t prim_tests;     /* primitive tests  (only when counters are enabled)   */
  uint64_t launches[12];   /* kernel launches per kernel class                    */
  double kernel_ms[12];    /* summed HIP-event time per kernel class (profiling)  */
  uint64_t two_ext_rays;   /* bounce rays traced by launches of class PTAMD_K_TRACE2 */
  uint64_t two_shadow_rays;/* shadow rays traced by launches of class PTAMD_K_TRACE2 */
  uint64_t exact_tests;    /* primitive tests the FP32 pre-test could not decide (counters enabled) */
  uint64_t unused_slots;   /* path slots of a round's launches that held no sample (a pixel traced fewer batches than its neighbours) */
} ptamd_stats;

/* kernel classes for ptamd_stats.launches / kernel_ms */
#define PTAMD_K_GENERATE 0
#define PTAMD_K_EXTEND 1
#define PTAMD_K_SHADE 2
#define PTAMD_K_SHADOW 3
#define PTAMD_K_RESOLVE 4
#define PTAMD_K_ACCUMULATE 5
#define PTAMD_K_FILTER 6
#define PTAMD_K_OTHER 7
#define PTAMD_K_TRACE2 8 /* one launch: bounce rays of depth d + 1 and shadow rays of depth d */
#define PTAMD_K_LIGHT 9  /* light samples of the shaded vertices, one lane per sample */
#define PTAMD_K_CLASSES 12

typedef struct ptamd_scene ptamd_scene;     /* host scene (parsed + built) */
typedef struct ptamd_context ptamd_context; /* one GPU                     */

const char *ptamd_last_error(void);

/* ---- host scene pipeline -------------------------------------------------
 * ptamd_scene_load replaces Collada::ColladaParser::load
 * (src/scene/collada/collada.cpp:133-227), Application::load
 * (src/application/application.cpp:229-305) and
 * RaytracedRenderer::build_accel (raytraced_renderer.cpp:351-375). */
int ptamd_scene_load(const char *dae_path, ptamd_scene **out);
void ptamd_scene_free(ptamd_scene *scene);
/* Pointers in *out stay valid until ptamd_scene_free. */
int ptamd_scene_get_desc(const ptamd_scene *scene, ptamd_scene_desc *out);
/* Camera for a w x h frame; replaces Application::resize -> Camera::set_screen_size
 * (application.cpp:189-201, camera.cpp:83-89). w = h = 0 keeps 800x600. */
int ptamd_scene_get_camera(ptamd_scene *scene, int32_t w, int32_t h, ptamd_camera *out);
/* Replaces Camera::load_settings (camera.cpp:170-185), the -c option. */
int ptamd_scene_load_camera_file(ptamd_scene *scene, const char *path);
/* Camera::dump_settings (camera.cpp:151-168): the same file, written. */
int ptamd_scene_dump_camera_file(ptamd_scene *scene, const char *path);
/* BVH build wall time in seconds (what build_accel prints). */
double ptamd_scene_build_seconds(const ptamd_scene *scene);

/* ---- device layer ---------------------------------------------------------
 * Replaces the worker pool: start_raytracing / worker_thread / raytrace_tile
 * (raytraced_renderer.cpp:268-329, 644-678, 585-610) and everything below
 * PathTracer::raytrace_pixel (src/pathtracer/pathtracer.cpp:234-272). */
int ptamd_create(int32_t device_id, ptamd_context **out);
void ptamd_destroy(ptamd_context *ctx);
/* Copies the scene to the device (replaces set_scene + build_accel's hand-over).
 * Limits (PTAMD_ERR_SCENE otherwise): leaves of at most 4 primitives (the reference builds with
 * max_leaf_size = 4), fewer than 2^24 - 1 primitives, coordinates below 2^20 in magnitude. */
int ptamd_upload_scene(ptamd_context *ctx, const ptamd_scene_desc *desc);
/* Replaces RaytracedRenderer::set_camera (raytraced_renderer.cpp:136-149). */
int ptamd_set_camera(ptamd_context *ctx, const ptamd_camera *cam);
/* Bound on paths in flight per wavefront pass (0 = default, 16 Mi). The work buffers are sized by it:
 * up to 370 + 131 x light samples per vertex bytes per path (about 15 GB at the default and -l 4). */
int ptamd_set_max_paths(ptamd_context *ctx, int64_t max_paths);
/* Enables per-launch HIP-event timing and traversal counters (slower). */
int ptamd_set_profiling(ptamd_context *ctx, int32_t events, int32_t counters);
/* Restricts the HIP events of ptamd_set_profiling to the kernel classes whose bit
 * (1 << PTAMD_K_x) is set in `mask` (default: all). An event pair costs about
 * 10 us of idle GPU between two launches, so a timed run records only the
 * classes it reports. */
int ptamd_set_event_classes(ptamd_context *ctx, uint32_t mask);

/* Whole frame on this GPU: replaces render_to_file's start_raytracing +
 * wait-for-DONE (raytraced_renderer.cpp:331-336). hdr_out/count_out are HOST
 * buffers of w*h*3 doubles / w*h int32; pixels outside the rectangle are 0. */
int ptamd_render(ptamd_context *ctx, const ptamd_render_params *p, double *hdr_out, int32_t *count_out);

/* Closest hit of camera rays through normalised sensor positions (sx, sy) in
 * [0,1]^2: Camera::generate_ray + BVHAccel::intersect (camera.cpp:190-203,
 * src/scene/bvh.cpp:164-200). prim_out is the primitive's index in desc order
 * (-1 = miss); t_out the hit distance. Host buffers of n entries. Uses the context's
 * work buffers: it must not run while a ptamd_trace_list of the same context is still
 * in flight (an open frame's parameters are left alone). */
int ptamd_trace_primary(ptamd_context *ctx, int64_t n, const double *sx, const double *sy,
                        int32_t *prim_out, double *t_out);

/* ---- round-level API (multi-GPU drivers shard pixels between rounds) -------
 * A frame is rendered in rounds of `samples_per_batch` samples per still-active
 * pixel, as raytrace_pixel's loop does (pathtracer.cpp:249-267). All pointers in
 * this group are DEVICE pointers; `stream` is a hipStream_t (NULL = default).
 *
 * Per-pixel state is 6 doubles per pixel, [w*h*6]: s1, s2, sum.r, sum.g, sum.b,
 * n (sample count as a double), indexed by pixel. */
int ptamd_frame_begin(ptamd_context *ctx, const ptamd_render_params *p);
/* Traces samples [sample_base, sample_base + n_samples) of the n listed pixels
 * and adds them to `state`. pixels: int32 pixel indices. */
int ptamd_trace_pixels(ptamd_context *ctx, const int32_t *d_pixels, int64_t n, int32_t sample_base,
                       int32_t n_samples, double *d_state, void *stream);
/* The adaptive test of pathtracer.cpp:250-258 at i = n_done for the n listed
 * pixels; survivors are written, in order, to d_pixels_out and their number to
 * *d_n_out (one int64 on the device). */
int ptamd_adaptive_filter(ptamd_context *ctx, const int32_t *d_pixels_in, int64_t n, int32_t n_done,
                          const double *d_state, int32_t *d_pixels_out, int64_t *d_n_out, void *stream);
/* A shard of a device-resident pixel list: entries offset, offset + stride,
 * offset + 2*stride, ... below the list's length. The length is read ON THE
 * DEVICE from *d_n when d_n is not NULL, so the host can queue the next round
 * before the previous round's adaptive test has been read back; n_max is the
 * host's upper bound of the length (it sizes launches and buffers). With
 * d_n == NULL the length is n. Rank r of `world` GPUs passes offset = r,
 * stride = world: that is the round-robin deal that replaces the reference's
 * tile queue (src/util/work_queue.h:11-51, raytraced_renderer.cpp:297-302). */
typedef struct ptamd_pixel_list {
  const int32_t *d_pixels;
  const int64_t *d_n;
  int64_t n;
  int64_t n_max;
  int32_t offset, stride;
} ptamd_pixel_list;
/* ptamd_trace_pixels for a shard of a list whose length lives on the device.
 * n_samples may span several batches of samples_per_batch: the adaptive test of
 * pathtracer.cpp:250-258 is then applied between the batches while the samples
 * are summed, and samples traced past a pixel's stop are dropped, so the state
 * is what one call per batch with a filter in between would have left. Tracing
 * ahead like this is how small rounds are kept from leaving the GPU idle.
 *
 * sample_base = -1 (n_samples then a multiple of samples_per_batch): every pixel
 * goes on from its OWN count (state n) and is traced ahead by as many batches, at
 * most n_samples / samples_per_batch, as its running mean and variance predict
 * it still needs before the adaptive test passes; the state is again what the
 * reference's per-pixel loop would have left, only the moment a sample is traced
 * changes. Pair it with ptamd_filter_list(n_done = -1).
 *
 * A frame's calls must reach the device in order (one stream, or streams the
 * caller orders); ptamd_frame_begin itself touches no device memory. */
int ptamd_trace_list(ptamd_context *ctx, const ptamd_pixel_list *list, int32_t sample_base,
                     int32_t n_samples, double *d_state, void *stream);
/* ptamd_adaptive_filter for such a list (offset 0, stride 1). A pixel survives if
 * it used all n_done samples traced for it so far (it did not stop at a test
 * inside an earlier multi-batch round) and the test at i = n_done lets it go on.
 * n_done = -1: every pixel is judged at its own count, as raytrace_pixel's loop
 * would (done at ns_aa samples, tested at every positive multiple of the batch). */
int ptamd_filter_list(ptamd_context *ctx, const ptamd_pixel_list *in, int32_t n_done, const double *d_state,
                      int32_t *d_pixels_out, int64_t *d_n_out, void *stream);
/* ptamd_filter_list for callers that cannot allocate device memory: code under HIP
 * stream capture, tight serving loops, harnesses that forbid hipMalloc. (No counterpart
 * in the reference; it exists for those callers only.) Same arguments, same result,
 * bit for bit; the per-block counts its kernels need live in a static device array of
 * PTAMD_FILTER_STATIC_MAX_PIXELS / 256 words (64 KiB per GPU, held until the process
 * ends) instead of a buffer the library grows on demand, and the call makes no
 * hipMalloc / hipFree / event creation. Lists longer than PTAMD_FILTER_STATIC_MAX_PIXELS
 * entries (n_max, after offset / stride) are rejected with PTAMD_ERR_ARG. All calls on one
 * GPU share that array: two calls that overlap in time on the same GPU race. Its launches are counted in
 * ptamd_stats.launches[PTAMD_K_FILTER] but, creating no events, add nothing to kernel_ms. */
#define PTAMD_FILTER_STATIC_MAX_PIXELS (4 * 1024 * 1024)
int ptamd_filter_list_static(ptamd_context *ctx, const ptamd_pixel_list *in, int32_t n_done,
                             const double *d_state, int32_t *d_pixels_out, int64_t *d_n_out, void *stream);
/* The exchange after a round, the counterpart of the reference's shared
 * sampleBuffer that every worker thread writes into (pathtracer.cpp:268-271):
 * pack gathers the state rows (6 doubles) of the shard's pixels into a dense
 * block d_rows[k*6..]; after an all-gather of the ranks' blocks (rows_per_rank
 * rows each), unpack writes row (i / world) of block (i % world) to the state of
 * entry i of the whole list. Blocks of rank skip_rank are left alone (-1: none;
 * a rank's own rows are already in place). */
int ptamd_pack_rows(ptamd_context *ctx, const ptamd_pixel_list *list, const double *d_state, double *d_rows,
                    void *stream);
int ptamd_unpack_rows(ptamd_context *ctx, const ptamd_pixel_list *list, int32_t world, int32_t skip_rank,
                      const double *d_rows, int64_t rows_per_rank, double *d_state, void *stream);
/* pixel_sum / n and the sample count (pathtracer.cpp:268-271) for every pixel. */
int ptamd_frame_resolve(ptamd_context *ctx, const double *d_state, double *d_hdr_out, int32_t *d_count_out,
                        void *stream);

/* The whole adaptive loop of a frame (PathTracer::raytrace_pixel, pathtracer.cpp:234-272, for
 * every listed pixel) in one call: ptamd_frame_begin, then rounds of ptamd_filter_list /
 * ptamd_trace_list until every listed pixel has stopped, the per-pixel sums left in d_state
 * ([w*h][6], zeroed by the caller; rows of pixels not listed stay untouched). d_pixels: n
 * distinct pixel indices in device memory. The work is queued on `stream`; the call returns
 * when the last round has been queued (it waits for earlier rounds' list lengths only).
 * This is what one rank of a multi-GPU render runs on its share of the pixels, and what
 * ptamd_render runs on the whole rectangle. Rounds too small to fill the GPU trace several
 * batches per pixel ahead of the adaptive tests that decide whether they are needed (what lies
 * past a pixel's stop is dropped: the frame is the one-batch-per-round frame, bit for bit);
 * frames of 16 or more batches per pixel do so per pixel, as far as each pixel's running mean
 * and variance predict it needs (ptamd_trace_list with sample_base = -1). */
int ptamd_render_list(ptamd_context *ctx, const ptamd_render_params *params, const int32_t *d_pixels, int64_t n,
                      double *d_state, void *stream);
/* Ordering: ptamd_render_list returns with its last rounds still queued and keeps, per context, the
 * survivor lists, their lengths and the work buffers those rounds use. Calls of ptamd_render_list,
 * ptamd_render, ptamd_trace_list and ptamd_trace_primary on ONE context must therefore reach the device in
 * order: issue them on the same stream, or synchronise with the earlier call's stream before the next
 * call. (A later call that needs larger buffers synchronises the whole device before it frees the old ones.) */

int ptamd_get_stats(ptamd_context *ctx, ptamd_stats *out);

/* ---- several GPUs of one node (libptamd_multi.so: links RCCL) ---------------
 * The counterpart of the reference's worker pool (numWorkerThreads threads taking tiles
 * from a queue, raytraced_renderer.cpp:297-302, 326-328, 644-658), one host thread per
 * GPU: the scene is replicated, pixel k of the frame's rectangle goes to GPU k % n, every
 * GPU runs ptamd_render_list on its pixels, and the per-pixel sums are gathered on the first
 * GPU with RCCL (one send per GPU) where the frame is resolved. Same frame for any n. */
typedef struct ptamd_multi ptamd_multi;
int ptamd_multi_create(const int32_t *device_ids, int32_t n, ptamd_multi **out);
void ptamd_multi_destroy(ptamd_multi *m);
int32_t ptamd_multi_size(const ptamd_multi *m);
int ptamd_multi_upload_scene(ptamd_multi *m, const ptamd_scene_desc *desc);
int ptamd_multi_set_camera(ptamd_multi *m, const ptamd_camera *cam);
/* ptamd_set_profiling on every GPU (the CLI's -C together with -g) */
int ptamd_multi_set_profiling(ptamd_multi *m, int32_t events, int32_t counters);
/* as ptamd_render */
int ptamd_multi_render(ptamd_multi *m, const ptamd_render_params *params, double *hdr_out, int32_t *count_out);
/* statistics of one GPU's share; seconds = wall time of the whole frame */
int ptamd_multi_get_stats(ptamd_multi *m, int32_t rank, ptamd_stats *out);
/* the deal's arithmetic: pixels of GPU `rank` out of n dealt to `world`, and the rows every
 * GPU's block of the exchange holds (the largest share) */
int64_t ptamd_multi_shard_size(int64_t n, int32_t rank, int32_t world);
int64_t ptamd_multi_rows_per_rank(int64_t n, int32_t world);

/* ---- output stage (host) --------------------------------------------------
 * HDRImageBuffer::toColor + ImageBuffer::update_pixel (src/util/image.h:183-198,
 * 53-62): gamma 2.2, exposure sqrt(2), truncation to 8 bits, R in the low byte. */
int ptamd_tonemap_rgba8(const double *hdr, int32_t w, int32_t h, uint32_t *rgba_out);
/* save_image (raytraced_renderer.cpp:680-718): vertical flip, alpha = 255, PNG. */
int ptamd_write_png(const char *path, const uint32_t *rgba_bottom_up, int32_t w, int32_t h);
/* Same encoder without the flip (used for the sampling-rate map). */
int ptamd_write_png_top_down(const char *path, const uint32_t *rgba_top_down, int32_t w, int32_t h);
/* save_sampling_rate_image (raytraced_renderer.cpp:720-751) as RGBA8, top row first. */
int ptamd_sampling_rate_rgba8(const int32_t *count, int32_t w, int32_t h, int32_t ns_aa, uint32_t *rgba_top_down);

#ifdef __cplusplus
}
#endif
#endif /* PTAMD_H */
