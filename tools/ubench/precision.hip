// Accuracy of gfx950's FP64 reciprocal / reciprocal-square-root seeds and of what one and two
// refinement steps make of them (the mosaic's write-back uses them: amhip_ortho_fold.h).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off precision.hip -o precision
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void k_probe(const double* x, double* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = x[i];
  // reciprocal: seed, one Newton step, two
  double r0 = __builtin_amdgcn_rcp(v);
  double r1 = fma(fma(-v, r0, 1.0), r0, r0);
  double r2 = fma(fma(-v, r1, 1.0), r1, r1);
  // square root from rsq: Goldschmidt, one and two iterations
  const double y = __builtin_amdgcn_rsq(v);
  double g = v * y, h = 0.5 * y;
  double r = fma(-h, g, 0.5);
  double g1 = fma(g, r, g);
  double h1 = fma(h, r, h);
  r = fma(-h1, g1, 0.5);
  double g2 = fma(g1, r, g1);
  out[7 * i + 0] = r0;
  out[7 * i + 1] = r1;
  out[7 * i + 2] = r2;
  out[7 * i + 3] = g;
  out[7 * i + 4] = g1;
  out[7 * i + 5] = g2;
  out[7 * i + 6] = y;
}

int main() {
  const int n = 1 << 22;
  std::vector<double> hx(n), ho(7 * (size_t)n);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const double u = (double)(s >> 11) * 0x1p-53;            // [0, 1)
    const int e = (int)((s >> 3) % 41) - 20;                  // 2^-20 .. 2^20
    hx[i] = std::ldexp(1.0 + u, e);
  }
  double *dx, *dout;
  hipMalloc(&dx, n * sizeof(double));
  hipMalloc(&dout, 7 * (size_t)n * sizeof(double));
  hipMemcpy(dx, hx.data(), n * sizeof(double), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_probe, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
  hipMemcpy(ho.data(), dout, 7 * (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
  double worst[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    const long double v = hx[i];
    const long double rc = 1.0L / v, sq = sqrtl(v), rs = 1.0L / sqrtl(v);
    const long double ref[7] = {rc, rc, rc, sq, sq, sq, rs};
    for (int k = 0; k < 7; ++k) {
      const double e = (double)fabsl(((long double)ho[7 * (size_t)i + k] - ref[k]) / ref[k]);
      if (e > worst[k]) worst[k] = e;
    }
  }
  const char* names[7] = {"v_rcp_f64", "rcp + 1 Newton", "rcp + 2 Newton", "x * rsq", "sqrt, 1 iteration",
                          "sqrt, 2 iterations", "v_rsq_f64"};
  for (int k = 0; k < 7; ++k)
    printf("%-20s max relative error %.3e = 2^%.1f\n", names[k], worst[k], std::log2(worst[k]));
  return 0;
}
