#include <immintrin.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
constexpr unsigned long long kHashK = 0x9E3779B97F4A7C15ull;
static inline void cell_mix(unsigned bits, unsigned long long g, unsigned long long* a, unsigned long long* b) {
  unsigned long long x = (unsigned long long)bits + kHashK * (g + 1ull);
  x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
  unsigned long long y = x * 0x94D049BB133111EBull; y ^= y >> 31;
  *a += x; *b += y;
}
static void sum_scalar(const unsigned* col, size_t n, unsigned long long g0, unsigned long long* a, unsigned long long* b) {
  unsigned long long ha = 0, hb = 0;
  for (size_t i = 0; i < n; ++i) cell_mix(col[i], g0 + i, &ha, &hb);
  *a += ha; *b += hb;
}
__attribute__((target("avx512f,avx512dq")))
static void sum_avx512(const unsigned* col, size_t n, unsigned long long g0, unsigned long long* a, unsigned long long* b) {
  const __m512i c1 = _mm512_set1_epi64((long long)0xBF58476D1CE4E5B9ull);
  const __m512i c2 = _mm512_set1_epi64((long long)0x94D049BB133111EBull);
  const __m512i step = _mm512_set1_epi64((long long)(kHashK * 8ull));
  const __m512i lane = _mm512_set_epi64(7, 6, 5, 4, 3, 2, 1, 0);
  __m512i kg = _mm512_mullo_epi64(_mm512_add_epi64(_mm512_set1_epi64((long long)(g0 + 1ull)), lane),
                                  _mm512_set1_epi64((long long)kHashK));
  __m512i va = _mm512_setzero_si512(), vb = _mm512_setzero_si512();
  size_t i = 0;
  for (; i + 8 <= n; i += 8) {
    __m512i x = _mm512_add_epi64(_mm512_cvtepu32_epi64(_mm256_loadu_si256((const __m256i*)(col + i))), kg);
    kg = _mm512_add_epi64(kg, step);
    x = _mm512_xor_si512(x, _mm512_srli_epi64(x, 29));
    x = _mm512_mullo_epi64(x, c1);
    x = _mm512_xor_si512(x, _mm512_srli_epi64(x, 32));
    __m512i y = _mm512_mullo_epi64(x, c2);
    y = _mm512_xor_si512(y, _mm512_srli_epi64(y, 31));
    va = _mm512_add_epi64(va, x);
    vb = _mm512_add_epi64(vb, y);
  }
  unsigned long long ha = _mm512_reduce_add_epi64(va), hb = _mm512_reduce_add_epi64(vb);
  for (; i < n; ++i) cell_mix(col[i], g0 + i, &ha, &hb);
  *a += ha; *b += hb;
}
int main() {
  size_t n = (1ull << 26) + 5;
  std::vector<unsigned> v(n);
  for (size_t i = 0; i < n; ++i) v[i] = (unsigned)(i * 2654435761u) ^ 0x7FC00000u;
  printf("avx512dq: %d\n", __builtin_cpu_supports("avx512dq"));
  for (int rep = 0; rep < 2; ++rep) {
    unsigned long long a1 = 0, b1 = 0, a2 = 0, b2 = 0;
    auto t0 = std::chrono::steady_clock::now();
    sum_scalar(v.data(), n, 12345678901ull, &a1, &b1);
    auto t1 = std::chrono::steady_clock::now();
    sum_avx512(v.data(), n, 12345678901ull, &a2, &b2);
    auto t2 = std::chrono::steady_clock::now();
    printf("scalar %.1f ms, avx512 %.1f ms, equal %d\n", std::chrono::duration<double, std::milli>(t1 - t0).count(),
           std::chrono::duration<double, std::milli>(t2 - t1).count(), a1 == a2 && b1 == b2);
  }
}
