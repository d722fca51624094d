// Host build of amhip_atan_cr.h for tests/test_atan_cr.py: a C entry point over arrays.
#include "amhip_atan_cr.h"

extern "C" void amt_atan_cr(const double* x, long n, double* out) {
  for (long k = 0; k < n; ++k) out[k] = amhip::atan_cr(x[k]);
}

// the fast path: out[k] = its value, okflag[k] = its own verdict
extern "C" void amt_atan_cr_fast(const double* x, long n, double* out, unsigned char* okflag) {
  for (long k = 0; k < n; ++k) {
    bool ok = false;
    out[k] = amhip::atan_cr_fast(x[k], &ok);
    okflag[k] = ok ? 1 : 0;
  }
}

extern "C" void amt_atan_device(const double* x, long n, double* out) {
  for (long k = 0; k < n; ++k) out[k] = amhip::atan_device(x[k]);
}
