// Host build of the session's content sums for tests/test_host_content_sum.py: the dispatching
// entry (AVX-512 where the CPU has it) and a plain loop over cell_mix to hold it against.
#include <cstddef>

#include "amhip_content_sum.h"

extern "C" {

void amt_sum_dispatch(const unsigned* col, long n, unsigned long long g0, unsigned long long* ab) {
  amhip::host_column_sum(col, (size_t)n, g0, &ab[0], &ab[1]);
}

void amt_sum_plain(const unsigned* col, long n, unsigned long long g0, unsigned long long* ab) {
  for (long i = 0; i < n; ++i) amhip::cell_mix(col[i], g0 + (unsigned long long)i, &ab[0], &ab[1]);
}

int amt_sum_vectorized() { return amhip::host_sum_is_vectorized() ? 1 : 0; }

}
