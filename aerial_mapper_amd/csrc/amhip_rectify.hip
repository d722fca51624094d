// amhip_rectify.hip -- planar rectification of a stereo pair on MI355X (gfx950).
//
// Replaces stereo::Rectifier::rectifyStereoPair + computeMask
//   aerial_mapper_dense_pcl/src/rectifier.cpp:34-128  (Fusiello, Trucco, Verri 2000)
// the step in front of the block matcher of the dense-point-cloud pipeline (SURVEY 8f rank 3;
// its other half, the reprojection of the disparities, is amhip_densify.hip):
//   host    the rectified rotation (x = baseline, y = z_1 x x, z = x x y), K R_rect,
//           T_k = (K R_rect) (K R_k^T)^-1 and their inverses cast to float -- a few hundred
//           flops, in the order of operations the oracle adopts for Eigen's
//           (built with -ffp-contract=off);
//   GPU     one lane per rectified pixel: [x y w]^T = T_inv [u v 1]^T in float, the two map
//           values x / w, y / w (IEEE division), cv::remap's bilinear sample (5 fractional
//           bits, weights (32 - fx)(32 - fy) 32 ..., (sum + 2^14) >> 15, border 0) of both
//           images, and the mask = the closed quadrilateral of the four projected corners.
// Byte / float streaming work: HBM bound (reads 2 B, writes 3 B + optionally 16 B of maps per
// pixel), no MFMA.  Bit-exact against the oracle (tests/test_gpu_rectify.py).
#include <cmath>
#include <cstring>

#include "amhip_common.h"

namespace amhip {

struct RectifyParams {
  float T1[9], T2[9];   // inverse rectifying transformations, row-major
  int cx[4], cy[4];     // mask corners
  int width, height;
  size_t left_step, right_step;
};

__device__ __forceinline__ unsigned char remap_sample(const unsigned char* __restrict__ src,
                                                      size_t step, int W, int H, float mx, float my) {
  double dx = (double)mx * 32.0, dy = (double)my * 32.0;
  dx = fmax(-2147483648.0, fmin(2147483647.0, dx));
  dy = fmax(-2147483648.0, fmin(2147483647.0, dy));
  const int sx = (int)rint(dx), sy = (int)rint(dy);   // cvRound: to nearest even
  int ix = sx >> 5, iy = sy >> 5;
  ix = min(max(ix, -32768), 32767);                    // (a short in OpenCV's fixed-point map)
  iy = min(max(iy, -32768), 32767);
  const int fx = sx & 31, fy = sy & 31;
  const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32,
            w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
  auto px = [&](int xx, int yy) -> int {
    if (xx < 0 || yy < 0 || xx >= W || yy >= H) return 0;
    return (int)src[(size_t)yy * step + (size_t)xx];
  };
  const int acc = w00 * px(ix, iy) + w01 * px(ix + 1, iy) + w10 * px(ix, iy + 1) + w11 * px(ix + 1, iy + 1);
  return (unsigned char)((acc + (1 << 14)) >> 15);
}

__global__ void __launch_bounds__(256)
k_rectify(RectifyParams p, const unsigned char* __restrict__ left,
          const unsigned char* __restrict__ right, float* __restrict__ maps,
          unsigned char* __restrict__ out_left, unsigned char* __restrict__ out_right,
          unsigned char* __restrict__ mask, unsigned* __restrict__ dev_err) {
  const int u = blockIdx.x * 64 + (threadIdx.x & 63);
  const int v = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (u >= p.width || v >= p.height) return;
  const float fu = (float)u, fv = (float)v;
  const size_t o = (size_t)v * p.width + u, n = (size_t)p.width * p.height;
  // rectifier.cpp:92-102 (float; products and sums separately rounded)
  const float x1 = (p.T1[0] * fu + p.T1[1] * fv) + p.T1[2] * 1.0f;
  const float y1 = (p.T1[3] * fu + p.T1[4] * fv) + p.T1[5] * 1.0f;
  const float w1 = (p.T1[6] * fu + p.T1[7] * fv) + p.T1[8] * 1.0f;
  const float x2 = (p.T2[0] * fu + p.T2[1] * fv) + p.T2[2] * 1.0f;
  const float y2 = (p.T2[3] * fu + p.T2[4] * fv) + p.T2[5] * 1.0f;
  const float w2 = (p.T2[6] * fu + p.T2[7] * fv) + p.T2[8] * 1.0f;
  if (w1 == 0.0f || w2 == 0.0f) atomicOr(dev_err, kDevErrRectifyZeroW);  // CHECK_NE(xyw(2), 0.0)
  const float mx1 = x1 / w1, my1 = y1 / w1, mx2 = x2 / w2, my2 = y2 / w2;
  if (maps) {
    maps[o] = mx1;
    maps[n + o] = my1;
    maps[2 * n + o] = mx2;
    maps[3 * n + o] = my2;
  }
  if (out_left) out_left[o] = remap_sample(left, p.left_step, p.width, p.height, mx1, my1);
  if (out_right) out_right[o] = remap_sample(right, p.right_step, p.width, p.height, mx2, my2);
  if (mask) {
    bool pos = true, neg = true;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int q = (k + 1) & 3;
      const long long e = (long long)(p.cx[q] - p.cx[k]) * (v - p.cy[k]) -
                          (long long)(p.cy[q] - p.cy[k]) * (u - p.cx[k]);
      pos = pos && e >= 0;
      neg = neg && e <= 0;
    }
    mask[o] = (pos || neg) ? 255 : 0;
  }
}

// ---- host: the 3x3 algebra in front of the per-pixel work -------------------------------
struct H33 {
  double m[3][3];
};
static H33 h_mul(const H33& a, const H33& b) {
  H33 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      r.m[i][j] = (a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j]) + a.m[i][2] * b.m[2][j];
  return r;
}
static H33 h_transpose(const H33& a) {
  H33 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[j][i];
  return r;
}
static double h_cof(const H33& a, int i, int j) {
  const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
  return a.m[i1][j1] * a.m[i2][j2] - a.m[i1][j2] * a.m[i2][j1];
}
static H33 h_inverse(const H33& a) {  // by cofactors, like Eigen's fixed 3x3 inverse
  const double c00 = h_cof(a, 0, 0), c10 = h_cof(a, 1, 0), c20 = h_cof(a, 2, 0);
  const double det = (c00 * a.m[0][0] + c10 * a.m[1][0]) + c20 * a.m[2][0];
  const double invdet = 1.0 / det;
  H33 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.m[i][j] = h_cof(a, j, i) * invdet;
  return r;
}
static double h_norm(const double* v) { return std::sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]); }
static void h_cross(const double* a, const double* b, double* r) {
  r[0] = a[1] * b[2] - a[2] * b[1];
  r[1] = a[2] * b[0] - a[0] * b[2];
  r[2] = a[0] * b[1] - a[1] * b[0];
}

}  // namespace amhip

using namespace amhip;

extern "C" {

int amhip_rectify_stereo_pair_dev(amhip_ctx* h, const double* K, const double* R_G_C1,
                                  const double* R_G_C2, const double* t_G_C1, const double* t_G_C2,
                                  int width, int height, const uint8_t* dev_left, size_t left_step,
                                  const uint8_t* dev_right, size_t right_step, double* R_G_C_out,
                                  double* baseline_out, float* dev_maps, uint8_t* dev_rect_left,
                                  uint8_t* dev_rect_right, uint8_t* dev_mask) {
  if (!h) return arg_failure("null context");
  if (!K || !R_G_C1 || !R_G_C2 || !t_G_C1 || !t_G_C2 || width <= 0 || height <= 0)
    return arg_failure("amhip_rectify_stereo_pair_dev: bad argument");
  if ((dev_rect_left && (!dev_left || left_step < (size_t)width)) ||
      (dev_rect_right && (!dev_right || right_step < (size_t)width)))
    return arg_failure("amhip_rectify_stereo_pair_dev: image missing / step smaller than a row");
  Ctx* c = &h->impl;
  int rc = ctx_use_device(c);
  if (rc) return rc;
  H33 k, r1, r2;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      k.m[i][j] = K[3 * i + j];
      r1.m[i][j] = R_G_C1[3 * i + j];
      r2.m[i][j] = R_G_C2[3 * i + j];
    }
  // rectifier.cpp:45-58: new axes, rows of the rectified rotation
  double x[3], y[3], z[3];
  for (int q = 0; q < 3; ++q) x[q] = t_G_C2[q] - t_G_C1[q];
  const double baseline = h_norm(x);
  const double z1[3] = {r1.m[0][2], r1.m[1][2], r1.m[2][2]};
  h_cross(z1, x, y);
  h_cross(x, y, z);
  const double nx = h_norm(x), ny = h_norm(y), nz = h_norm(z);
  H33 R;
  for (int j = 0; j < 3; ++j) {
    R.m[0][j] = x[j] / nx;
    R.m[1][j] = y[j] / ny;
    R.m[2][j] = z[j] / nz;
  }
  if (baseline_out) *baseline_out = baseline;
  if (R_G_C_out)
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) R_G_C_out[3 * i + j] = R.m[i][j];
  // :63-78
  const H33 P33 = h_mul(k, R);
  const H33 T1 = h_mul(P33, h_inverse(h_mul(k, h_transpose(r1))));
  const H33 T2 = h_mul(P33, h_inverse(h_mul(k, h_transpose(r2))));
  const H33 I1 = h_inverse(T1), I2 = h_inverse(T2);
  RectifyParams p;
  std::memset(&p, 0, sizeof(p));
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      p.T1[3 * i + j] = (float)I1.m[i][j];
      p.T2[3 * i + j] = (float)I2.m[i][j];
    }
  // :116-128 the image corners through T1, truncated like cv::Point(double, double)
  const double corner[4][2] = {{0.0, 0.0}, {width - 1.0, 0.0}, {width - 1.0, height - 1.0}, {0.0, height - 1.0}};
  for (int q = 0; q < 4; ++q) {
    double hh[3];
    for (int i = 0; i < 3; ++i) hh[i] = (T1.m[i][0] * corner[q][0] + T1.m[i][1] * corner[q][1]) + T1.m[i][2] * 1.0;
    p.cx[q] = (int)(hh[0] / hh[2]);
    p.cy[q] = (int)(hh[1] / hh[2]);
  }
  p.width = width;
  p.height = height;
  p.left_step = left_step;
  p.right_step = right_step;
  ScopedTimer t(c, AMHIP_K_MISC);
  hipLaunchKernelGGL(k_rectify, dim3((unsigned)((width + 63) / 64), (unsigned)((height + 3) / 4)),
                     dim3(256), 0, c->stream, p, dev_left, dev_right, dev_maps, dev_rect_left,
                     dev_rect_right, dev_mask, c->dev_err);
  AMHIP_TRY(hipGetLastError());
  return AMHIP_OK;
}

}  // extern "C"
