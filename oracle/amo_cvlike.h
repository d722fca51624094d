/*
 * oracle/amo_cvlike.h -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * The pieces of OpenCV, aslam_cv2, minkindr and Eigen that ortho::OrthoForwardHomography is
 * made of, restated from their published algorithms (the list, and what is known to deviate,
 * is in the header of amo_forward.cc).  PARITY UNPINNED for all of it: none of those
 * libraries is in /root/reference or in this image.  Shared by the restated oracle
 * (amo_forward.cc) and by the stand-in headers the reference's own
 * ortho-forward-homography.cc is compiled against (refkit/), so that both sides of that
 * comparison use the same definitions of the externals.
 */
#ifndef AMO_CVLIKE_H_
#define AMO_CVLIKE_H_

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "amo_compat.h"
#include "amo_types.h"

namespace amo {

// --- Eigen::Quaterniond::toRotationMatrix (minkindr getRotationMatrix) --------
static inline void rotation_matrix(const Quat& q, double R[9]) {
  const double tx = 2.0 * q.x, ty = 2.0 * q.y, tz = 2.0 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0] = 1.0 - (tyy + tzz);
  R[1] = txy - twz;
  R[2] = txz + twy;
  R[3] = txy + twz;
  R[4] = 1.0 - (txx + tzz);
  R[5] = tyz - twx;
  R[6] = txz - twy;
  R[7] = tyz + twx;
  R[8] = 1.0 - (txx + tyy);
}

// --- aslam distortion undistort: Gauss-Newton on distort(y) = y_distorted -----
static inline void undistort_normalized(const amo_camera& c, double* px, double* py) {
  if (c.distortion == AMO_DIST_NONE) return;
  const double yx = *px, yy = *py;
  double bx = yx, by = yy;
  for (int it = 0; it < 5; ++it) {
    double dx = bx, dy = by;
    distort(c, &dx, &dy);
    // Jacobian by central differences of the same distort()
    const double h = 1e-6;
    double ax = bx + h, ay = by, cx = bx - h, cy = by;
    distort(c, &ax, &ay);
    distort(c, &cx, &cy);
    const double j00 = (ax - cx) / (2.0 * h), j10 = (ay - cy) / (2.0 * h);
    ax = bx; ay = by + h; cx = bx; cy = by - h;
    distort(c, &ax, &ay);
    distort(c, &cx, &cy);
    const double j01 = (ax - cx) / (2.0 * h), j11 = (ay - cy) / (2.0 * h);
    const double ex = yx - dx, ey = yy - dy;
    // du = (J^T J)^-1 J^T e
    const double a = j00 * j00 + j10 * j10, b = j00 * j01 + j10 * j11,
                 d = j01 * j01 + j11 * j11;
    const double gx = j00 * ex + j10 * ey, gy = j01 * ex + j11 * ey;
    const double det = a * d - b * b;
    bx = bx + (d * gx - b * gy) / det;
    by = by + (a * gy - b * gx) / det;
    if (ex * ex + ey * ey <= 1e-8) break;
  }
  *px = bx;
  *py = by;
}

// --- the image -> mosaic homography of one frame --------------------------------
// ortho-forward-homography.cc:85-112 (updateOrthomosaic) / :143-170 (batch).
// `batch_quirk`: batch() adds width/2 to BOTH ground coordinates (:155-158).
static inline int solve8(double A[8][9]) {  // Gaussian elimination, partial pivoting
  for (int col = 0; col < 8; ++col) {
    int piv = col;
    for (int r = col + 1; r < 8; ++r)
      if (std::fabs(A[r][col]) > std::fabs(A[piv][col])) piv = r;
    if (A[piv][col] == 0.0) return 1;
    if (piv != col)
      for (int k = 0; k < 9; ++k) std::swap(A[piv][k], A[col][k]);
    for (int r = col + 1; r < 8; ++r) {
      const double f = A[r][col] / A[col][col];
      if (f == 0.0) continue;
      for (int k = col; k < 9; ++k) A[r][k] = A[r][k] - f * A[col][k];
    }
  }
  for (int r = 7; r >= 0; --r) {
    double s = A[r][8];
    for (int k = r + 1; k < 8; ++k) s = s - A[r][k] * A[k][8];
    A[r][8] = s / A[r][r];
  }
  return 0;
}

// cv::getPerspectiveTransform(src, dst): the 8x8 system of the four correspondences
// (float points widened to double), M[8] = 1.
static inline bool get_perspective_transform(const float src[4][2], const float dst[4][2],
                                             double M[9]) {
  double A[8][9];
  for (int i = 0; i < 4; ++i) {
    const double sx = src[i][0], sy = src[i][1], dx = dst[i][0], dy = dst[i][1];
    const double r0[9] = {sx, sy, 1.0, 0.0, 0.0, 0.0, -sx * dx, -sy * dx, dx};
    const double r1[9] = {0.0, 0.0, 0.0, sx, sy, 1.0, -sx * dy, -sy * dy, dy};
    std::memcpy(A[i], r0, sizeof(r0));
    std::memcpy(A[i + 4], r1, sizeof(r1));
  }
  if (solve8(A)) return false;
  for (int k = 0; k < 8; ++k) M[k] = A[k][8];
  M[8] = 1.0;
  return true;
}

struct MosaicDesc {
  int width, height;  // settings.width_mosaic_pixels / height_mosaic_pixels
  double ground;      // ground_plane_elevation_m
  double origin[3];
};

static inline int frame_homography(const amo_camera& cam, const MosaicDesc& ds, const Pose& T_G_C,
                            bool batch_quirk, double M[9]) {
  const double W1 = static_cast<double>(cam.width - 1), H1 = static_cast<double>(cam.height - 1);
  const double kp[4][2] = {{0.0, 0.0}, {W1, 0.0}, {W1, H1}, {0.0, H1}};  // :33-39
  double R[9];
  rotation_matrix(T_G_C.q, R);
  float src[4][2], dst[4][2];
  for (int k = 0; k < 4; ++k) {
    // PinholeCamera::backProject3
    double rx = (kp[k][0] - cam.cu) / cam.fu;
    double ry = (kp[k][1] - cam.cv) / cam.fv;
    undistort_normalized(cam, &rx, &ry);
    const double ray[3] = {rx, ry, 1.0};
    const double rz = (R[6] * ray[0] + R[7] * ray[1]) + R[8] * ray[2];
    const double scale = -(T_G_C.t.z - ds.ground) / rz;
    double S[9];
    for (int e = 0; e < 9; ++e) S[e] = scale * R[e];  // Eigen: (scale * R) * C_ray
    const double vx = (S[0] * ray[0] + S[1] * ray[1]) + S[2] * ray[2];
    const double vy = (S[3] * ray[0] + S[4] * ray[1]) + S[5] * ray[2];
    const double gx = (T_G_C.t.x + vx) - ds.origin[0];
    const double gy = (T_G_C.t.y + vy) - ds.origin[1];
    const double off_x = static_cast<double>(ds.width) / 2.0;
    const double off_y = static_cast<double>(batch_quirk ? ds.width : ds.height) / 2.0;
    dst[k][0] = static_cast<float>(gy + off_x);  // cv::Point2f(G(1) + w/2, G(0) + h/2)
    dst[k][1] = static_cast<float>(gx + off_y);
    src[k][0] = static_cast<float>(kp[k][0]);
    src[k][1] = static_cast<float>(kp[k][1]);
  }
  return get_perspective_transform(src, dst, M) ? AMO_OK : AMO_ERR_ARG;
}

// cv::invert of a 3x3 double matrix (closed form; DECOMP_LU special case)
static inline bool invert3(const double S[9], double D[9]) {
  double d = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) +
             S[2] * (S[3] * S[7] - S[4] * S[6]);
  if (d == 0.0) return false;
  d = 1.0 / d;
  double t[9];
  t[0] = (S[4] * S[8] - S[5] * S[7]) * d;
  t[1] = (S[2] * S[7] - S[1] * S[8]) * d;
  t[2] = (S[1] * S[5] - S[2] * S[4]) * d;
  t[3] = (S[5] * S[6] - S[3] * S[8]) * d;
  t[4] = (S[0] * S[8] - S[2] * S[6]) * d;
  t[5] = (S[2] * S[3] - S[0] * S[5]) * d;
  t[6] = (S[3] * S[7] - S[4] * S[6]) * d;
  t[7] = (S[1] * S[6] - S[0] * S[7]) * d;
  t[8] = (S[0] * S[4] - S[1] * S[3]) * d;
  std::memcpy(D, t, sizeof(t));
  return true;
}

static inline int round_half_even_sat(double v) {  // saturate_cast<int>(double) = lrint
  v = std::max(static_cast<double>(INT_MIN), std::min(static_cast<double>(INT_MAX), v));
  return static_cast<int>(std::nearbyint(v));
}

struct Image8 {
  const uint8_t* data;
  size_t step;
  int width, height, channels;
};

// aslam MappedUndistorter::processImage (cv::remap INTER_LINEAR, BORDER_CONSTANT)
static inline void undistort_image(const amo_camera& cam, const Image8& in, std::vector<uint8_t>* out) {
  const int W = in.width, H = in.height, ch = in.channels;
  out->assign(static_cast<size_t>(W) * H * ch, 0);
  for (int v = 0; v < H; ++v) {
    for (int u = 0; u < W; ++u) {
      double x = (static_cast<double>(u) - cam.cu) / cam.fu;
      double y = (static_cast<double>(v) - cam.cv) / cam.fv;
      distort(cam, &x, &y);
      const float mx = static_cast<float>(cam.fu * x + cam.cu);
      const float my = static_cast<float>(cam.fv * y + cam.cv);
      const int sx = round_half_even_sat(static_cast<double>(mx) * 32.0);
      const int sy = round_half_even_sat(static_cast<double>(my) * 32.0);
      const int ix = sx >> 5, iy = sy >> 5, fx = sx & 31, fy = sy & 31;
      const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32,
                w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
      for (int c = 0; c < ch; ++c) {
        auto px = [&](int xx, int yy) -> int {
          if (xx < 0 || yy < 0 || xx >= W || yy >= H) return 0;
          return in.data[static_cast<size_t>(yy) * in.step + static_cast<size_t>(xx) * ch + c];
        };
        const int acc = w00 * px(ix, iy) + w01 * px(ix + 1, iy) + w10 * px(ix, iy + 1) +
                        w11 * px(ix + 1, iy + 1);
        (*out)[(static_cast<size_t>(v) * W + u) * ch + c] =
            static_cast<uint8_t>((acc + (1 << 14)) >> 15);
      }
    }
  }
}

// cv::warpPerspective(src, dst, M, Size(w, h), INTER_NEAREST, BORDER_CONSTANT)
// dst: h x w x ch bytes.
static inline bool warp_nearest(const Image8& src, const double Mfwd[9], int w, int h,
                         std::vector<uint8_t>* dst) {
  double M[9];
  dst->assign(static_cast<size_t>(w) * h * src.channels, 0);
  if (!invert3(Mfwd, M)) return false;
  const int bh0a = std::min(32 / 2, h);
  const int bw0 = std::min(32 * 32 / bh0a, w);
  for (int y = 0; y < h; ++y) {
    for (int x = 0; x < w; x += bw0) {
      const int bw = std::min(bw0, w - x);
      const double X0 = M[0] * x + M[1] * y + M[2];
      const double Y0 = M[3] * x + M[4] * y + M[5];
      const double W0 = M[6] * x + M[7] * y + M[8];
      for (int x1 = 0; x1 < bw; ++x1) {
        double Wd = W0 + M[6] * x1;
        Wd = Wd != 0.0 ? 1.0 / Wd : 0.0;
        int X = round_half_even_sat((X0 + M[0] * x1) * Wd);
        int Y = round_half_even_sat((Y0 + M[3] * x1) * Wd);
        X = std::max(-32768, std::min(32767, X));  // saturate_cast<short>
        Y = std::max(-32768, std::min(32767, Y));
        if (X < 0 || Y < 0 || X >= src.width || Y >= src.height) continue;
        for (int c = 0; c < src.channels; ++c)
          (*dst)[(static_cast<size_t>(y) * w + (x + x1)) * src.channels + c] =
              src.data[static_cast<size_t>(Y) * src.step + static_cast<size_t>(X) * src.channels + c];
      }
    }
  }
  return true;
}

// cv::distanceTransform(mask, dist, DIST_L1, 3): two-pass 3x3 chamfer, a = 1,
// b = 2, one-pixel "infinite" border.
static inline void distance_l1(const std::vector<uint8_t>& mask, int w, int h, std::vector<float>* out) {
  const int INF = INT_MAX >> 2;
  const int step = w + 2;
  std::vector<int> t(static_cast<size_t>(step) * (h + 2), INF);
  for (int y = 0; y < h; ++y) {
    int* row = &t[static_cast<size_t>(y + 1) * step + 1];
    for (int x = 0; x < w; ++x) {
      if (!mask[static_cast<size_t>(y) * w + x]) {
        row[x] = 0;
      } else {
        int v = row[x - step - 1] + 2;
        v = std::min(v, row[x - step] + 1);
        v = std::min(v, row[x - step + 1] + 2);
        v = std::min(v, row[x - 1] + 1);
        row[x] = v;
      }
    }
  }
  out->resize(static_cast<size_t>(w) * h);
  for (int y = h - 1; y >= 0; --y) {
    int* row = &t[static_cast<size_t>(y + 1) * step + 1];
    for (int x = w - 1; x >= 0; --x) {
      int v = row[x];
      if (v > 1) {
        v = std::min(v, row[x + step + 1] + 2);
        v = std::min(v, row[x + step] + 1);
        v = std::min(v, row[x + step - 1] + 2);
        v = std::min(v, row[x + 1] + 1);
        row[x] = v;
      }
      (*out)[static_cast<size_t>(y) * w + x] = static_cast<float>(v);
    }
  }
}

// cv::detail::FeatherBlender over the whole mosaic (dst_roi = mosaic rect,
// every feed at tl = (0, 0)).
struct Feather {
  int w, h;
  std::vector<int16_t> dst;       // CV_16SC3
  std::vector<float> dst_weight;  // CV_32F
  void prepare(int w_, int h_) {
    w = w_;
    h = h_;
    dst.assign(static_cast<size_t>(w) * h * 3, 0);
    dst_weight.assign(static_cast<size_t>(w) * h, 0.0f);
  }
  void feed(const std::vector<int16_t>& img, const std::vector<uint8_t>& mask) {
    std::vector<float> weight;
    distance_l1(mask, w, h, &weight);
    const float sharpness = 0.02f;
    for (size_t k = 0; k < weight.size(); ++k) {
      float v = weight[k] * sharpness;  // multiply(weight, sharpness, tmp)
      if (v > 1.0f) v = 1.0f;           // threshold(..., 1.f, 1.f, THRESH_TRUNC)
      weight[k] = v;
    }
    for (size_t k = 0; k < weight.size(); ++k) {
      for (int c = 0; c < 3; ++c)
        dst[3 * k + c] = static_cast<int16_t>(
            dst[3 * k + c] + static_cast<int16_t>(static_cast<float>(img[3 * k + c]) * weight[k]));
      dst_weight[k] += weight[k];
    }
  }
  void blend(std::vector<int16_t>* result, std::vector<uint8_t>* result_mask) {
    const float eps = 1e-5f;
    result_mask->resize(static_cast<size_t>(w) * h);
    for (size_t k = 0; k < dst_weight.size(); ++k) {
      for (int c = 0; c < 3; ++c)
        dst[3 * k + c] =
            static_cast<int16_t>(static_cast<float>(dst[3 * k + c]) / (dst_weight[k] + eps));
      const bool on = dst_weight[k] > eps;
      (*result_mask)[k] = on ? 255 : 0;
      if (!on) dst[3 * k] = dst[3 * k + 1] = dst[3 * k + 2] = 0;
    }
    *result = dst;
  }
};

// addImage(cv::Mat image_warped) (:42-58): GRAY2RGB, ->16SC3, mask = gray(img > 0.1)
static inline void to_16sc3_and_mask(const std::vector<uint8_t>& warped, int w, int h, int ch,
                              std::vector<int16_t>* img, std::vector<uint8_t>* mask) {
  const size_t n = static_cast<size_t>(w) * h;
  img->resize(3 * n);
  mask->resize(n);
  for (size_t k = 0; k < n; ++k) {
    int m[3];
    for (int c = 0; c < 3; ++c) {
      const int v = warped[k * ch + (ch == 1 ? 0 : c)];
      (*img)[3 * k + c] = static_cast<int16_t>(v);
      m[c] = (static_cast<double>(v) > 0.1) ? 255 : 0;
    }
    (*mask)[k] = static_cast<uint8_t>((m[0] * 4899 + m[1] * 9617 + m[2] * 1868 + (1 << 13)) >> 14);
  }
}

}  // namespace amo

#endif  // AMO_CVLIKE_H_
