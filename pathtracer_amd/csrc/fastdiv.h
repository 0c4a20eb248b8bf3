// Exact unsigned division by a divisor that is fixed for a whole launch.
//
// For 0 <= n < 2^31 and 1 <= d < 2^16:   n / d == (uint64(n) * m) >> sh
// with l = ceil(log2 d), sh = 31 + l, m = ceil(2^sh / d)  (Granlund & Montgomery,
// "Division by invariant integers using multiplication", theorem 4.2 with N = 31).
// m lies in [2^31, 2^32), so the product needs one 32x32 -> 64 bit multiply. The
// device's generic 32-bit division is a long instruction sequence that holds a
// dozen registers; a float quotient is NOT exact above 2^24.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define PT_HD __host__ __device__
#else
#define PT_HD
#endif

namespace ptdev {

struct FastDiv {
  uint32_t m, sh;
  PT_HD inline uint32_t div(uint32_t n) const { return (uint32_t)(((uint64_t)n * m) >> sh); }
};

inline FastDiv make_fastdiv(uint32_t d) {
  uint32_t l = 0;
  while ((1u << l) < d) l++;
  const uint64_t p = 1ull << (31 + l);
  return FastDiv{(uint32_t)((p + d - 1) / d), 31 + l};
}

}  // namespace ptdev
