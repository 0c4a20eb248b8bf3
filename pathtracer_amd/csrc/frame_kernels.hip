// Frame-level kernels: the ordered filter of the active-pixel list (count, scan, scatter), the final
// division by the sample count, the pixel list of a rectangle, and the packing of state rows for
// the exchange between GPUs; and the static device array of the one entry point that allocates nothing.
#include "kernels.h"
#include "pixel_test.h"

namespace ptk {

// Ordered stream compaction of the still-active pixels, three small kernels:
// per-block counts, scan of the counts, scatter.
__global__ void k_filter_count(PixView pv, int32_t n_done, double tol, int32_t batch, int32_t ns_aa,
                               const double *__restrict__ state, uint32_t *__restrict__ block_count) {
  __shared__ uint32_t wave_cnt[4];
  const uint32_t n = pv.count();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool keep = i < n && pixel_survives(state + (size_t)pv.at(i) * 6, n_done, tol, batch, ns_aa);
  const unsigned long long m = __ballot(keep);
  if (__lane_id() == 0) wave_cnt[threadIdx.x >> 6] = (uint32_t)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) block_count[blockIdx.x] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}
__global__ void k_filter_scan(uint32_t *block_count, uint32_t n_blocks, long long *total_out) {
  // one block; exclusive scan in place
  __shared__ uint32_t carry;
  __shared__ uint32_t tmp[1024];
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (uint32_t base = 0; base < n_blocks; base += 1024) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < n_blocks ? block_count[i] : 0;
    tmp[threadIdx.x] = v;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
      const uint32_t a = threadIdx.x >= off ? tmp[threadIdx.x - off] : 0;
      __syncthreads();
      tmp[threadIdx.x] += a;
      __syncthreads();
    }
    if (i < n_blocks) block_count[i] = carry + tmp[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += tmp[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) *total_out = (long long)carry;
}
__global__ void k_filter_scatter(PixView pv, int32_t n_done, double tol, int32_t batch, int32_t ns_aa,
                                 const double *__restrict__ state, const uint32_t *__restrict__ block_offset,
                                 int32_t *__restrict__ out) {
  __shared__ uint32_t wave_cnt[4];
  const uint32_t n = pv.count();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool keep = i < n && pixel_survives(state + (size_t)pv.at(i) * 6, n_done, tol, batch, ns_aa);
  const unsigned long long m = __ballot(keep);
  const int lane = __lane_id(), wv = threadIdx.x >> 6;
  if (lane == 0) wave_cnt[wv] = (uint32_t)__popcll(m);
  __syncthreads();
  uint32_t off = block_offset[blockIdx.x];
  for (int k = 0; k < wv; k++) off += wave_cnt[k];
  if (keep) out[off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = pv.at(i);
}

// pixel_sum / n and the sample count, pathtracer.cpp:268-271
__global__ void k_frame_resolve(const double *__restrict__ state, uint32_t n_pix, double *__restrict__ hdr,
                                int32_t *__restrict__ count) {
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < n_pix; p += stride) {
    const double *st = state + (size_t)p * 6;
    const double nn = st[5];
    if (nn > 0) {
      const double r = 1. / nn;
      hdr[3 * (size_t)p] = st[2] * r; hdr[3 * (size_t)p + 1] = st[3] * r; hdr[3 * (size_t)p + 2] = st[4] * r;
    } else {
      hdr[3 * (size_t)p] = 0; hdr[3 * (size_t)p + 1] = 0; hdr[3 * (size_t)p + 2] = 0;
    }
    count[p] = (int32_t)nn;
  }
}

__global__ void k_iota(int32_t *out, uint32_t n, int32_t w, int32_t x0, int32_t y0, int32_t rw) {
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = (y0 + (int32_t)(i / (uint32_t)rw)) * w + x0 + (int32_t)(i % (uint32_t)rw);
}

// State rows of a shard's pixels, gathered into a dense block (what a rank sends
// after a round) and scattered back from the blocks of all ranks: entry i of the
// whole list was traced by rank i % world and is row i / world of its block.
__global__ void k_pack_rows(PixView pv, const double *__restrict__ state, double *__restrict__ rows) {
  const uint32_t n = pv.count();
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n * 6u; t += stride) {
    const uint32_t k = t / 6u, f = t % 6u;
    rows[t] = state[(size_t)pv.at(k) * 6 + f];
  }
}
__global__ void k_unpack_rows(PixView pv, uint32_t world, uint32_t skip_rank, const double *__restrict__ rows, uint32_t m,
                              double *__restrict__ state) {
  const uint32_t n = pv.count();
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n * 6u; t += stride) {
    const uint32_t i = t / 6u, f = t % 6u;
    const uint32_t r = i % world, k = i / world;
    if (r != skip_rank) state[(size_t)pv.at(i) * 6 + f] = rows[((size_t)r * m + k) * 6 + f];
  }
}

// Block counts of ptamd_filter_list_static: the one entry point that allocates nothing. Each GPU
// the library runs on holds its own copy of this array, at its own address, until the process
// ends; host code reaches it only through the pointer hipGetSymbolAddress returned on that GPU.
constexpr size_t FILTER_STATIC_BLOCKS = (size_t)(FILTER_STATIC_MAX_PIXELS / 256);
static __device__ __align__(256) uint32_t scratch[FILTER_STATIC_BLOCKS];

}  // namespace ptk

// Address of this device's copy of the static scratch array (the device must be current). The
// array's name is only a host-side handle: kernels and memsets get the pointer returned here, and
// hipGetSymbolAddress has to be called in the source file that defines the array.
int ptk::static_scratch_address(uint32_t **out) {
  void *sp = nullptr;
  const hipError_t e = hipGetSymbolAddress(&sp, HIP_SYMBOL(ptk::scratch));
  if (e != hipSuccess || !sp) return (int)(e != hipSuccess ? e : hipErrorInvalidSymbol);
  *out = (uint32_t *)sp;
  return 0;
}
