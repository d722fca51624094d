/*
 * oracle/amo_rectify.h -- TEST INFRASTRUCTURE ONLY (CPU oracle for stereo::Rectifier).
 *
 * The arithmetic stereo::Rectifier::rectifyStereoPair
 * (aerial_mapper_dense_pcl/src/rectifier.cpp:34-114, Fusiello et al. 2000) is made of, in the
 * order of operations ADOPTED for the Eigen / OpenCV calls it makes (both libraries are absent
 * from /root/reference and from this image: "parity unpinned" for these pieces; the FLOW of
 * rectifier.cpp around them is pinned by compiling that file unchanged over oracle/refkit,
 * whose Eigen / OpenCV stand-ins forward to the functions below):
 *   Eigen   fixed-size products coefficient by coefficient, sum ((p0 + p1) + p2), no fused
 *           multiply-add; norm = sqrt((x0^2 + x1^2) + x2^2); normalized = v / norm (three
 *           divisions); cross in the textbook order; 3x3 inverse by cofactors,
 *           inverse(i, j) = cofactor(j, i) * (1 / det), det = (c00 m00 + c10 m10) + c20 m20
 *           (Eigen/src/LU/InverseImpl.h, compute_inverse size 3)
 *   OpenCV  cv::remap(CV_32FC1 maps, INTER_LINEAR, BORDER_CONSTANT 0) on 8UC1: sx =
 *           cvRound(map * 32), 5 fractional bits, weights (32 - fx)(32 - fy) * 32 ..., value
 *           (sum + 2^14) >> 15, taps outside the image contribute the border value;
 *           cv::drawContours(.., CV_FILLED) of the four projected corners: the closed
 *           quadrilateral through the TRUNCATED integer corner coordinates, evaluated at the
 *           pixel centres (boundary pixels included, as fillPoly draws them).
 */
#ifndef ORACLE_AMO_RECTIFY_H_
#define ORACLE_AMO_RECTIFY_H_
#include <climits>
#include <cmath>
#include <cstddef>
#include <cstdint>

namespace amo {
namespace rect {

struct M3 {
  double m[3][3];
};
struct V3d {
  double v[3];
};

static inline V3d sub(const V3d& a, const V3d& b) {
  V3d r = {{a.v[0] - b.v[0], a.v[1] - b.v[1], a.v[2] - b.v[2]}};
  return r;
}
static inline V3d neg(const V3d& a) {
  V3d r = {{-a.v[0], -a.v[1], -a.v[2]}};
  return r;
}
static inline double norm(const V3d& a) {
  return std::sqrt((a.v[0] * a.v[0] + a.v[1] * a.v[1]) + a.v[2] * a.v[2]);
}
static inline V3d normalized(const V3d& a) {
  const double n = norm(a);
  V3d r = {{a.v[0] / n, a.v[1] / n, a.v[2] / n}};
  return r;
}
static inline V3d cross(const V3d& a, const V3d& b) {
  V3d r = {{a.v[1] * b.v[2] - a.v[2] * b.v[1], a.v[2] * b.v[0] - a.v[0] * b.v[2],
            a.v[0] * b.v[1] - a.v[1] * b.v[0]}};
  return r;
}
static inline V3d col(const M3& a, int j) {
  V3d r = {{a.m[0][j], a.m[1][j], a.m[2][j]}};
  return r;
}
static inline M3 transpose(const M3& a) {
  M3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[j][i];
  return r;
}
static inline M3 mul(const M3& a, const M3& b) {
  M3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      r.m[i][j] = (a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j]) + a.m[i][2] * b.m[2][j];
  return r;
}
static inline V3d mul(const M3& a, const V3d& x) {
  V3d r;
  for (int i = 0; i < 3; ++i) r.v[i] = (a.m[i][0] * x.v[0] + a.m[i][1] * x.v[1]) + a.m[i][2] * x.v[2];
  return r;
}
static inline double cofactor(const M3& a, int i, int j) {
  const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
  return a.m[i1][j1] * a.m[i2][j2] - a.m[i1][j2] * a.m[i2][j1];
}
static inline M3 inverse(const M3& a) {
  const double c00 = cofactor(a, 0, 0), c10 = cofactor(a, 1, 0), c20 = cofactor(a, 2, 0);
  const double det = (c00 * a.m[0][0] + c10 * a.m[1][0]) + c20 * a.m[2][0];
  const double invdet = 1.0 / det;
  M3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.m[i][j] = cofactor(a, j, i) * invdet;
  return r;
}

// What rectifyStereoPair computes before it touches a pixel.
struct Plan {
  double baseline;
  M3 R_G_C;        // rectified rotation of both cameras
  M3 T1, T2;       // rectifying image transformations (double)
  float T1_inv[3][3], T2_inv[3][3];
};

static inline Plan make_plan(const M3& K, const M3& R1, const M3& R2, const V3d& t1, const V3d& t2) {
  Plan p;
  const V3d x = sub(t2, t1);                 // rectifier.cpp:45
  p.baseline = norm(x);                      // :46
  const V3d y = cross(col(R1, 2), x);        // :49
  const V3d z = cross(x, y);                 // :52
  const V3d xn = normalized(x), yn = normalized(y), zn = normalized(z);
  for (int j = 0; j < 3; ++j) {              // :55-58 (columns x, y, z; transposed)
    p.R_G_C.m[0][j] = xn.v[j];
    p.R_G_C.m[1][j] = yn.v[j];
    p.R_G_C.m[2][j] = zn.v[j];
  }
  // P_rect.block<3,3>(0,0) = K * R_rect (:63-70; the 4th column is never used)
  const M3 P33 = mul(K, p.R_G_C);
  const M3 Q1 = mul(K, transpose(R1));       // :73
  const M3 Q2 = mul(K, transpose(R2));       // :74
  p.T1 = mul(P33, inverse(Q1));              // :75
  p.T2 = mul(P33, inverse(Q2));              // :76
  const M3 i1 = inverse(p.T1), i2 = inverse(p.T2);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {            // :77-78 cast<float>()
      p.T1_inv[i][j] = static_cast<float>(i1.m[i][j]);
      p.T2_inv[i][j] = static_cast<float>(i2.m[i][j]);
    }
  return p;
}

// :92-102: [x y w]^T = T_inv (float) * [u v 1]^T, map = (x / w, y / w), all in float
static inline bool map_pixel(const float T[3][3], int u, int v, float* mx, float* my) {
  const float fu = static_cast<float>(u), fv = static_cast<float>(v);
  const float x = (T[0][0] * fu + T[0][1] * fv) + T[0][2] * 1.0f;
  const float y = (T[1][0] * fu + T[1][1] * fv) + T[1][2] * 1.0f;
  const float w = (T[2][0] * fu + T[2][1] * fv) + T[2][2] * 1.0f;
  *mx = x / w;
  *my = y / w;
  return w != 0.0f;  // CHECK_NE(xyw(2), 0.0)
}

static inline int cv_round(double v) {  // cvRound = lrint, saturated
  v = std::fmax(static_cast<double>(INT_MIN), std::fmin(static_cast<double>(INT_MAX), v));
  return static_cast<int>(std::nearbyint(v));
}

// one pixel of cv::remap(8UC1, CV_32FC1 maps, INTER_LINEAR, BORDER_CONSTANT 0)
static inline uint8_t remap_bilinear(const uint8_t* src, size_t step, int W, int H, float mx, float my) {
  const int sx = cv_round(static_cast<double>(mx) * 32.0);
  const int sy = cv_round(static_cast<double>(my) * 32.0);
  // (the integer part is stored as a short in OpenCV's fixed-point map)
  int ix = sx >> 5, iy = sy >> 5;
  ix = ix < -32768 ? -32768 : (ix > 32767 ? 32767 : ix);
  iy = iy < -32768 ? -32768 : (iy > 32767 ? 32767 : iy);
  const int fx = sx & 31, fy = sy & 31;
  const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32,
            w11 = fx * fy * 32;
  auto px = [&](int xx, int yy) -> int {
    if (xx < 0 || yy < 0 || xx >= W || yy >= H) return 0;
    return src[static_cast<size_t>(yy) * step + static_cast<size_t>(xx)];
  };
  const int acc = w00 * px(ix, iy) + w01 * px(ix + 1, iy) + w10 * px(ix, iy + 1) + w11 * px(ix + 1, iy + 1);
  return static_cast<uint8_t>((acc + (1 << 14)) >> 15);
}

// :116-128 the four image corners through T1 (double), truncated to cv::Point
static inline void mask_corners(const M3& T1, int W, int H, int cx[4], int cy[4]) {
  const double corner[4][2] = {{0.0, 0.0}, {W - 1.0, 0.0}, {W - 1.0, H - 1.0}, {0.0, H - 1.0}};
  for (int k = 0; k < 4; ++k) {
    const V3d c = {{corner[k][0], corner[k][1], 1.0}};
    const V3d h = mul(T1, c);
    cx[k] = static_cast<int>(h.v[0] / h.v[2]);
    cy[k] = static_cast<int>(h.v[1] / h.v[2]);
  }
}

// closed quadrilateral at the pixel centre: all four edge functions of one sign (or zero)
static inline bool in_quad(const int cx[4], const int cy[4], int x, int y) {
  bool pos = true, negv = true;
  for (int k = 0; k < 4; ++k) {
    const int n = (k + 1) & 3;
    const long long e = static_cast<long long>(cx[n] - cx[k]) * (y - cy[k]) -
                        static_cast<long long>(cy[n] - cy[k]) * (x - cx[k]);
    pos = pos && e >= 0;
    negv = negv && e <= 0;
  }
  return pos || negv;
}

}  // namespace rect
}  // namespace amo
#endif  // ORACLE_AMO_RECTIFY_H_
