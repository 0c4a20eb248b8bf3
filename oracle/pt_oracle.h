/* TEST INFRASTRUCTURE — CPU restatement ("oracle") of the reference's render
 * hot path, in plain C, FP64, no FMA contraction.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the product never links, imports or calls it.
 *
 * What is restated 1:1 from the reference (file:line cited at each function in
 * pt_oracle.c): camera ray generation, BVH closest-hit traversal in the
 * reference's node order, bbox / triangle / sphere tests, coordinate frames,
 * BSDF and light sampling, the recursive radiance estimators with their Russian
 * roulette rules, the per-pixel adaptive-sampling loop.
 *
 * What differs on purpose: random numbers. The reference draws from one
 * unsynchronised global mt19937, which makes its images irreproducible; the
 * oracle (and the GPU path it checks) uses a counter-based generator,
 * Philox-4x32-10, keyed by (seed, pixel) with counter (sample, block):
 *
 *   block 0              words 0,1 -> the two draws of the jittered sampler
 *   vertex at depth d    base = 1 + (d-1)*BPV,  BPV = 1 + ceil(NLS/2),
 *                        NLS = number of light samples taken per vertex
 *     block base         word 0 -> Russian-roulette coin; words 1,2 -> BSDF sample
 *     block base+1+q/2   words 2*(q%2), 2*(q%2)+1 -> the q-th light sample
 *
 * Every 32-bit word w becomes clamp(w / (2^32-1), 1e-7, 0.99999999), the
 * reference's random_uniform() (src/util/random_util.h:20-22).
 *
 * Parity status: pinned against the reference itself. tests/golden holds
 * outputs of the reference's own sources (built by `make -C oracle ref`) for
 * the pure functions (bit-exact), primary-ray hits (bit-exact ids and t) and
 * converged images (statistical: different RNG).
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include <stdint.h>

#include "../include/ptamd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_stats {
  double seconds;
  uint64_t samples, rays, shadow_rays, path_vertices, nodes_visited, prim_tests;
  uint64_t shadow_nodes_visited, shadow_prim_tests; /* the share of the two above spent on shadow rays */
  int32_t threads;
} oracle_stats;

/* Renders the rectangle in p into hdr [w*h*3] and count [w*h] (y up). */
int oracle_render(const ptamd_scene_desc *scene, const ptamd_camera *cam, const ptamd_render_params *p,
                  int32_t threads, double *hdr, int32_t *count, oracle_stats *stats);

/* Camera rays through normalised (sx, sy); closest hit in desc primitive order. */
int oracle_trace_primary(const ptamd_scene_desc *scene, const ptamd_camera *cam, int64_t n, const double *sx,
                         const double *sy, int32_t *prim_out, double *t_out);

/* Radiance of ONE path sample, for fine-grained comparisons. */
int oracle_sample_radiance(const ptamd_scene_desc *scene, const ptamd_camera *cam, const ptamd_render_params *p,
                           int32_t x, int32_t y, int32_t sample, double rgb_out[3]);

/* Pure functions on scripted inputs (layouts as in tests/golden/kat.ptar). */
void oracle_kat_bbox(int64_t n, const double *in /*n*14*/, double *out /*n*3*/);
void oracle_kat_triangle(int64_t n, const double *in /*n*26*/, double *out /*n*6*/);
void oracle_kat_sphere(int64_t n, const double *in /*n*12*/, double *out /*n*6*/);
void oracle_kat_frame(int64_t n, const double *in /*n*3*/, double *out /*n*9*/);
void oracle_kat_tonemap(int64_t n, const double *in /*n*3*/, int32_t *out /*n*/);

/* The specular-mode BSDFs (include/ptamd.h, "specular mode") on n scripted inputs:
 * kind, mat_param (3), mat_extra (4), wo (3) and the random number per input. */
void oracle_kat_specular(int64_t n, const int32_t *kind, const double *param3, const double *extra4, const double *wo,
                         const double *u, double *wi_out, double *f_out, double *pdf_out);

/* Philox-4x32-10 words, for cross-checking the GPU generator. */
void oracle_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]);

#ifdef __cplusplus
}
#endif
#endif
