// amhip_hostsum.cc -- host side of the session's content sums (amhip_content_sum.h): the scalar
// loop and an AVX-512 loop of the same arithmetic (8 cells per step: vpmovzxdq, vpmullq, shifts),
// chosen once at run time.  Host-only translation unit: nothing here runs on the GPU.
// tools/ubench/host_sum_avx512.cc: 5.2 -> 33 GB/s per thread on the GPU boxes' EPYC 9575F, which
// matters under the pod's CPU quota (16 CPUs' worth of time: amhip_session.hip, usable_cpus).
#if defined(__x86_64__) || defined(__i386__)
#define AMHIP_HOSTSUM_X86 1
#include <immintrin.h>
#else
#define AMHIP_HOSTSUM_X86 0   // (aarch64 hosts: the portable scalar loop, host_sum_is_vectorized() == false)
#endif

#include <cstdlib>

#include "amhip_content_sum.h"
#include "amhip_tuning.h"

namespace amhip {

static void sum_scalar(const unsigned* col, size_t n, unsigned long long g0, unsigned long long* a,
                       unsigned long long* b) {
  unsigned long long ha = 0, hb = 0;
  for (size_t i = 0; i < n; ++i) cell_mix(col[i], g0 + i, &ha, &hb);
  *a += ha;
  *b += hb;
}

#if AMHIP_HOSTSUM_X86
__attribute__((target("avx512f,avx512dq"))) static void sum_avx512(const unsigned* col, size_t n,
                                                                   unsigned long long g0,
                                                                   unsigned long long* a,
                                                                   unsigned long long* b) {
  const __m512i c1 = _mm512_set1_epi64((long long)kHashC1);
  const __m512i c2 = _mm512_set1_epi64((long long)kHashC2);
  const __m512i step = _mm512_set1_epi64((long long)(kHashK * 8ull));
  const __m512i lane = _mm512_set_epi64(7, 6, 5, 4, 3, 2, 1, 0);
  // K (g + 1) for the eight cells of a step; + 8 K per step (everything modulo 2^64)
  __m512i kg = _mm512_mullo_epi64(_mm512_add_epi64(_mm512_set1_epi64((long long)(g0 + 1ull)), lane),
                                  _mm512_set1_epi64((long long)kHashK));
  __m512i va = _mm512_setzero_si512(), vb = _mm512_setzero_si512();
  size_t i = 0;
  for (; i + 8 <= n; i += 8) {
    __m512i x = _mm512_add_epi64(
        _mm512_cvtepu32_epi64(_mm256_loadu_si256(reinterpret_cast<const __m256i*>(col + i))), kg);
    kg = _mm512_add_epi64(kg, step);
    x = _mm512_xor_si512(x, _mm512_srli_epi64(x, 29));
    x = _mm512_mullo_epi64(x, c1);
    x = _mm512_xor_si512(x, _mm512_srli_epi64(x, 32));
    __m512i y = _mm512_mullo_epi64(x, c2);
    y = _mm512_xor_si512(y, _mm512_srli_epi64(y, 31));
    va = _mm512_add_epi64(va, x);
    vb = _mm512_add_epi64(vb, y);
  }
  unsigned long long ha = (unsigned long long)_mm512_reduce_add_epi64(va);
  unsigned long long hb = (unsigned long long)_mm512_reduce_add_epi64(vb);
  for (; i < n; ++i) cell_mix(col[i], g0 + i, &ha, &hb);
  *a += ha;
  *b += hb;
}

#endif  // AMHIP_HOSTSUM_X86

using SumFn = void (*)(const unsigned*, size_t, unsigned long long, unsigned long long*,
                       unsigned long long*);

static SumFn pick() {
#if AMHIP_HOSTSUM_X86
  if (tuning_on("session_scalar_sums")) return sum_scalar;
  __builtin_cpu_init();
  return (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq")) ? sum_avx512
                                                                                    : sum_scalar;
#else
  return sum_scalar;
#endif
}

static SumFn resolved() {
  static const SumFn fn = pick();
  return fn;
}

void host_column_sum(const unsigned* col, size_t n, unsigned long long g0, unsigned long long* a,
                     unsigned long long* b) {
  resolved()(col, n, g0, a, b);
}

bool host_sum_is_vectorized() {
#if AMHIP_HOSTSUM_X86
  return resolved() == sum_avx512;
#else
  return false;
#endif
}

}  // namespace amhip
