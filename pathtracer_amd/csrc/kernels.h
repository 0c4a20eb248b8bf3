// Declarations of the device layer's kernels. Each is defined in exactly one translation unit
// (trace_kernels.hip, shade_kernels.hip, frame_kernels.hip: the library is built without relocatable
// device code, so a kernel and everything it calls live in one unit) and launched from
// ptamd_device.hip through the host stub the defining unit exports. The templates are instantiated
// explicitly where they are defined.
#pragma once
#include "device_types.h"

namespace ptk {

// ---- trace_kernels.hip
template <bool COUNT>
__global__ void k_extend(DevScene S, RayQ q, const int32_t *__restrict__ list, HitBuf hb, Ctl *ctl, double min_t, double max_t,
                         uint2 *gstack, const int32_t *__restrict__ self_code, const float *__restrict__ self_dmin, v3 cam_o);
template <bool HEMI, bool COUNT>
__global__ void k_shadow(DevScene S, ShadowQ sq, VertexBuf vb, Ctl *ctl, FastDiv by_nls, uint2 nls_nsa, uint2 *gstack, uint32_t fuse_below);
template <bool HEMI, bool COUNT>
__global__ void k_trace2(DevScene S, RayQ q, const int32_t *__restrict__ list, HitBuf hb, ShadowQ sq, VertexBuf vb, Ctl *ctl,
                         FastDiv by_nls, uint2 nls_nsa, uint2 *gstack, uint32_t fuse_below);

// ---- shade_kernels.hip
__global__ void k_set_counts(Ctl *ctl, PixView pv, uint32_t p0, uint32_t chunk, uint32_t ns);
__global__ void k_plan_count(PixView pv, uint32_t p0, uint32_t chunk, RoundPlan rp, uint32_t *__restrict__ block_count);
__global__ void k_plan_scan(uint32_t *block_count, uint32_t n_blocks, Ctl *ctl, PixView pv, uint32_t p0, uint32_t chunk);
__global__ void k_plan_offsets(PixView pv, uint32_t p0, Ctl *ctl, RoundPlan rp, const uint32_t *__restrict__ block_offset,
                               uint32_t *__restrict__ off, uint32_t *__restrict__ slot_pix);
__global__ void k_generate(FrameParams fp, DevScene S, PixView pv, uint32_t p0, Ctl *ctl, RoundPlan rp, PathMap pm, RayQ q, PathBuf pb);
__global__ void k_generate_primary(FrameParams fp, const double *__restrict__ sx, const double *__restrict__ sy, uint32_t n, RayQ q);
__global__ void k_primary_t(DevScene S, RayQ q, const int32_t *__restrict__ prim, uint32_t n, double min_t, double *__restrict__ t_out, v3 cam_o);
template <bool HEMI>
__global__ void k_shade(FrameParams fp, DevScene S, RayQ q, const int32_t *__restrict__ list, HitBuf hb, RayQ qn, ShadowQ sq,
                        VertexBuf vb, PathBuf pb, Compact cp, Ctl *ctl, int32_t depth);
__global__ void k_light(FrameParams fp, DevScene S, VertexBuf vb, ShadowQ sq, Compact cp, Ctl *ctl, FastDiv by_nls, uint32_t rng_base);
__global__ void k_scan2(Compact cp, Ctl *ctl, Ctl *next);
__global__ void k_scatter(Compact cp, VertexBuf vb, int32_t *__restrict__ list_next, ShadowQ sq, Ctl *ctl, int32_t nls);
__global__ void k_resolve(FrameParams fp, DevScene S, RayQ q, const int32_t *__restrict__ list, VertexBuf vb,
                          const SRay *__restrict__ rays, PathBuf pb, Ctl *ctl, int32_t depth);
__global__ void k_accumulate(PixView pv, uint32_t p0, Ctl *ctl, RoundPlan rp, PathMap pm, PathBuf pb, double *__restrict__ state);
__global__ void k_light_resolve(FrameParams fp, DevScene S, VertexBuf vb, ShadowQ sq, Compact cp, Ctl *ctl, FastDiv by_nls, uint32_t rng_base,
                                ResolveArgs prev);

// ---- frame_kernels.hip
__global__ void k_filter_count(PixView pv, int32_t n_done, double tol, int32_t batch, int32_t ns_aa,
                               const double *__restrict__ state, uint32_t *__restrict__ block_count);
__global__ void k_filter_scan(uint32_t *block_count, uint32_t n_blocks, long long *total_out);
__global__ void k_filter_scatter(PixView pv, int32_t n_done, double tol, int32_t batch, int32_t ns_aa,
                                 const double *__restrict__ state, const uint32_t *__restrict__ block_offset, int32_t *__restrict__ out);
__global__ void k_frame_resolve(const double *__restrict__ state, uint32_t n_pix, double *__restrict__ hdr, int32_t *__restrict__ count);
__global__ void k_iota(int32_t *out, uint32_t n, int32_t w, int32_t x0, int32_t y0, int32_t rw);
__global__ void k_pack_rows(PixView pv, const double *__restrict__ state, double *__restrict__ rows);
__global__ void k_unpack_rows(PixView pv, uint32_t world, uint32_t skip_rank, const double *__restrict__ rows, uint32_t m,
                              double *__restrict__ state);
// most pixels a list given to ptamd_filter_list_static may hold
constexpr int64_t FILTER_STATIC_MAX_PIXELS = PTAMD_FILTER_STATIC_MAX_PIXELS;
// address of the current device's copy of the static block-count array; 0 on success, else a hipError_t
int static_scratch_address(uint32_t **out);

}  // namespace ptk
