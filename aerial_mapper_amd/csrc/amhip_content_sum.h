// amhip_content_sum.h -- the session's content sums (amhip_session.hip): per cell
//   x = mix(bits + K (g + 1)),  a += x,  b += mix2(x);   g = i + j * map rows.
// Both mixes are bijections of 64-bit words (xor-shift, odd multiplier), so ONE changed cell
// always changes both sums; they are not affine in the bits, so no relation d1 w1 + d2 w2 = 0
// between two edits cancels in either, let alone in both.  The sums are order-free: host threads
// and GPU lanes each take any part.  Shared by the device kernel, the host threads' scalar loop
// and the AVX-512 loop (amhip_hostsum.cc).
#ifndef AMHIP_CONTENT_SUM_H_
#define AMHIP_CONTENT_SUM_H_

#include <cstddef>

#if defined(__HIPCC__)
#define AMHIP_SUM_HD __host__ __device__ inline
#else
#define AMHIP_SUM_HD inline
#endif

namespace amhip {

constexpr unsigned long long kHashK = 0x9E3779B97F4A7C15ull;
constexpr unsigned long long kHashC1 = 0xBF58476D1CE4E5B9ull;
constexpr unsigned long long kHashC2 = 0x94D049BB133111EBull;

AMHIP_SUM_HD void cell_mix(unsigned bits, unsigned long long g, unsigned long long* a,
                           unsigned long long* b) {
  unsigned long long x = (unsigned long long)bits + kHashK * (g + 1ull);
  x ^= x >> 29;
  x *= kHashC1;
  x ^= x >> 32;
  unsigned long long y = x * kHashC2;
  y ^= y >> 31;
  *a += x;
  *b += y;
}

// *a, *b += the sums of cells col[0 .. n) at positions g0 .. g0 + n - 1 (host; picks the AVX-512
// loop where the CPU has avx512dq -- 6 x the scalar loop per thread on Zen 5 -- unless
// tuning knob session_scalar_sums is set).  host_sum_is_vectorized(): which one.
void host_column_sum(const unsigned* col, size_t n, unsigned long long g0, unsigned long long* a,
                     unsigned long long* b);
bool host_sum_is_vectorized();

}  // namespace amhip

#endif  // AMHIP_CONTENT_SUM_H_
