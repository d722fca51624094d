// Throughput of the session's host-side content sums (amhip_session.hip: host_hashes) on the
// GPU box's CPUs: M map-sized matrices, T threads over column ranges, the matrices visited
// column by column (order 0: all matrices per column, as the session did up to round 4) or one
// matrix after the other (order 1).   g++ -O2 -pthread host_sum_bench.cc && ./a.out T M side order
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
constexpr unsigned long long kHashK = 0x9E3779B97F4A7C15ull;
static inline void cell_mix(unsigned bits, unsigned long long g, unsigned long long* a, unsigned long long* b) {
  unsigned long long x = (unsigned long long)bits + kHashK * (g + 1ull);
  x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
  unsigned long long y = x * 0x94D049BB133111EBull; y ^= y >> 31;
  *a += x; *b += y;
}
int main(int argc, char** argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 1, M = argc > 2 ? atoi(argv[2]) : 1;
  const int side = argc > 3 ? atoi(argv[3]) : 16384, order = argc > 4 ? atoi(argv[4]) : 0;
  const size_t n = (size_t)side * side;
  std::vector<unsigned*> mats(M);
  for (int m = 0; m < M; ++m) {
    mats[m] = (unsigned*)malloc(n * 4);
    for (size_t i = 0; i < n; ++i) mats[m][i] = 0x7FC00000u;  // (first touch: this thread)
  }
  for (int rep = 0; rep < 3; ++rep) {
    auto t0 = std::chrono::steady_clock::now();
    std::vector<unsigned long long> A(T * 8), B(T * 8);
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back([&, t]() {
      const int c0 = (int)((long long)side * t / T), c1 = (int)((long long)side * (t + 1) / T);
      unsigned long long a = 0, b = 0;
      if (order == 0) {
        for (int j = c0; j < c1; ++j)
          for (int m = 0; m < M; ++m) {
            const unsigned* col = mats[m] + (size_t)j * side;
            for (int i = 0; i < side; ++i) cell_mix(col[i], (size_t)j * side + i, &a, &b);
          }
      } else {
        for (int m = 0; m < M; ++m)
          for (int j = c0; j < c1; ++j) {
            const unsigned* col = mats[m] + (size_t)j * side;
            for (int i = 0; i < side; ++i) cell_mix(col[i], (size_t)j * side + i, &a, &b);
          }
      }
      A[t * 8] = a; B[t * 8] = b; });
    for (auto& x : th) x.join();
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("T=%d M=%d side=%d order=%d: %.1f ms %.1f GB/s (%llx)\n", T, M, side, order, dt * 1e3,
           n * 4 * M / dt / 1e9, A[0] ^ B[(T - 1) * 8]);
  }
}
