// amhip_forward.hip -- homography-based forward orthomosaic on MI355X.
//
// Replaces ortho::OrthoForwardHomography (aerial_mapper_ortho/src/
// ortho-forward-homography.cc): per frame the four image corners are
// intersected with the ground plane (:85-112 / :143-170), the image ->
// mosaic homography is solved (cv::getPerspectiveTransform), the frame is
// warped into the mosaic with nearest-neighbour sampling (cv::warpPerspective,
// INTER_NEAREST, BORDER_CONSTANT) and fed to an OpenCV FeatherBlender
// (L1 distance transform of the frame's mask x 0.02, truncated at 1 = the
// blend weight); blend() divides the weighted sums by the summed weights.
//
// Mapping to the GPU (everything per mosaic pixel is independent except the
// distance transform, which is separable):
//   k_fwd_undistort  only for cameras with a distortion model (remap, bilinear)
//   k_fwd_warp       up to 64 frames at once, each on the bounding region of its
//                    footprint only: inverse homography per pixel in double with
//                    OpenCV's evaluation order, u8 sample + mask
//   k_fwd_dt_rows    one wave per (row, frame): distance to the nearest zero of
//                    the row = prefix-max / suffix-min of zero positions
//   k_fwd_dt_cols    one lane per (column, frame): min-plus sweep down and up;
//                    rows + columns = the exact L1 transform (what OpenCV's 3x3
//                    chamfer with a = 1, b = 2 computes); distances saturate at
//                    255, the weight saturates at 50
//   k_fwd_feed       per pixel, frames of the batch in ascending order:
//                    dst += (short)(src * w), weight += w  (float sums in the
//                    reference's order, 16-bit wrap like cv::Point3_<short>)
//   k_fwd_blend      normalizeUsingWeightMap + mask + Blender::blend
// The 8x8 solve and the 3x3 inverse run on the host in double (a few hundred
// flops per frame).
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>
#include <vector>

#include "amhip_common.h"

namespace amhip {

struct FwdFrame {
  double m[9];  // mosaic pixel -> image pixel (inverse of the frame's homography)
  // Region of the mosaic the frame can touch: the bounding box of its four
  // ground points (+ margin), clipped.  Outside of it the warped image is the
  // border value 0, i.e. mask 0, distance 0, feather weight 0 -- the frame
  // contributes nothing there, so every kernel works on the region only.  One
  // ring of the region is guaranteed to be outside the footprint (or the
  // mosaic ends there, which is how cv::distanceTransform treats its border).
  int x0, y0, w, h;
  unsigned long long off;  // first pixel of the region in the per-batch buffers
};

struct Mosaic {
  amhip_mosaic_desc desc;
  amhip_camera cam;
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  size_t pixels = 0;
  // blender state (cv::detail::FeatherBlender dst_ / dst_weight_map_)
  int16_t* dst16 = nullptr;   // H x W x 3
  float* dst_weight = nullptr;
  // result_ / result_mask_
  int16_t* result16 = nullptr;
  uint8_t* result_mask = nullptr;
  // per-batch workspaces
  uint8_t* warped = nullptr;   // G x H x W x ch
  size_t warped_cap = 0;
  uint8_t* mask = nullptr;     // G x H x W
  size_t mask_cap = 0;
  uint8_t* dist = nullptr;     // G x H x W
  size_t dist_cap = 0;
  uint8_t* undist = nullptr;   // G undistorted frames
  size_t undist_cap = 0;
  uint8_t* stage = nullptr;    // host frames staged to the device
  size_t stage_cap = 0;
  FwdFrame* frames = nullptr;      // per-frame table of one call, device
  size_t frames_cap = 0;
  FwdFrame* host_frames = nullptr; // its pinned staging copy
  size_t host_frames_cap = 0;
  hipEvent_t frames_event = nullptr;
  bool frames_pending = false;
};

static int fwd_arg_fail(const char* msg) {
  set_last_error(msg);
  return AMHIP_ERR_ARG;
}

// ---------------------------------------------------------------------------
// host: the homography of one frame
// ---------------------------------------------------------------------------
namespace {

struct Q {
  double w, x, y, z;
};

void rotation_matrix(const Q& q, double R[9]) {  // Eigen::Quaterniond::toRotationMatrix
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

void distort_host(const amhip_camera& c, double* px, double* py) {
  double x = *px, y = *py;
  if (c.distortion == AMHIP_DIST_RADTAN) {
    const double k1 = c.dist[0], k2 = c.dist[1], p1 = c.dist[2], p2 = c.dist[3];
    const double mx2 = x * x, my2 = y * y, mxy = x * y;
    const double rho2 = mx2 + my2;
    const double rad = k1 * rho2 + k2 * rho2 * rho2;
    const double nx = x + (x * rad + 2.0 * p1 * mxy + p2 * (rho2 + 2.0 * mx2));
    const double ny = y + (y * rad + 2.0 * p2 * mxy + p1 * (rho2 + 2.0 * my2));
    x = nx;
    y = ny;
  } else if (c.distortion == AMHIP_DIST_EQUIDISTANT) {
    const double r = std::sqrt(x * x + y * y);
    const double theta = std::atan(r);
    const double th2 = theta * theta, th4 = th2 * th2, th6 = th4 * th2, th8 = th4 * th4;
    const double thetad =
        theta * (1.0 + c.dist[0] * th2 + c.dist[1] * th4 + c.dist[2] * th6 + c.dist[3] * th8);
    const double scaling = (r > 1e-8) ? thetad / r : 1.0;
    x = x * scaling;
    y = y * scaling;
  }
  *px = x;
  *py = y;
}

// aslam distortion undistort(): Gauss-Newton on distort(y) = y_d, 5 iterations
void undistort_normalized(const amhip_camera& c, double* px, double* py) {
  if (c.distortion == AMHIP_DIST_NONE) return;
  const double yx = *px, yy = *py;
  double bx = yx, by = yy;
  for (int it = 0; it < 5; ++it) {
    double dx = bx, dy = by;
    distort_host(c, &dx, &dy);
    const double h = 1e-6;
    double ax = bx + h, ay = by, cx = bx - h, cy = by;
    distort_host(c, &ax, &ay);
    distort_host(c, &cx, &cy);
    const double j00 = (ax - cx) / (2.0 * h), j10 = (ay - cy) / (2.0 * h);
    ax = bx; ay = by + h; cx = bx; cy = by - h;
    distort_host(c, &ax, &ay);
    distort_host(c, &cx, &cy);
    const double j01 = (ax - cx) / (2.0 * h), j11 = (ay - cy) / (2.0 * h);
    const double ex = yx - dx, ey = yy - dy;
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

bool solve8(double A[8][9]) {  // Gaussian elimination with partial pivoting
  for (int col = 0; col < 8; ++col) {
    int piv = col;
    for (int r = col + 1; r < 8; ++r)
      if (std::fabs(A[r][col]) > std::fabs(A[piv][col])) piv = r;
    if (A[piv][col] == 0.0) return false;
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
  return true;
}

// ortho-forward-homography.cc:85-112 (updateOrthomosaic) / :143-170 (batch; it
// offsets BOTH ground coordinates by width/2 -- `batch_quirk`).
bool frame_homography(const amhip_camera& cam, const amhip_mosaic_desc& ds, const double* T,
                      bool batch_quirk, double M[9]) {
  const double W1 = static_cast<double>(cam.width - 1), H1 = static_cast<double>(cam.height - 1);
  const double kp[4][2] = {{0.0, 0.0}, {W1, 0.0}, {W1, H1}, {0.0, H1}};
  const Q q = {T[3], T[4], T[5], T[6]};
  double R[9];
  rotation_matrix(q, R);
  float src[4][2], dst[4][2];
  for (int k = 0; k < 4; ++k) {
    double rx = (kp[k][0] - cam.cu) / cam.fu;  // PinholeCamera::backProject3
    double ry = (kp[k][1] - cam.cv) / cam.fv;
    undistort_normalized(cam, &rx, &ry);
    const double ray[3] = {rx, ry, 1.0};
    const double rz = (R[6] * ray[0] + R[7] * ray[1]) + R[8] * ray[2];
    const double scale = -(T[2] - ds.ground_plane_elevation_m) / rz;
    double S[9];
    for (int e = 0; e < 9; ++e) S[e] = scale * R[e];
    const double vx = (S[0] * ray[0] + S[1] * ray[1]) + S[2] * ray[2];
    const double vy = (S[3] * ray[0] + S[4] * ray[1]) + S[5] * ray[2];
    const double gx = (T[0] + vx) - ds.origin[0];
    const double gy = (T[1] + vy) - ds.origin[1];
    const double off_x = static_cast<double>(ds.width_mosaic_pixels) / 2.0;
    const double off_y =
        static_cast<double>(batch_quirk ? ds.width_mosaic_pixels : ds.height_mosaic_pixels) / 2.0;
    dst[k][0] = static_cast<float>(gy + off_x);
    dst[k][1] = static_cast<float>(gx + off_y);
    src[k][0] = static_cast<float>(kp[k][0]);
    src[k][1] = static_cast<float>(kp[k][1]);
  }
  double A[8][9];
  for (int i = 0; i < 4; ++i) {
    const double sx = src[i][0], sy = src[i][1], dx = dst[i][0], dy = dst[i][1];
    const double r0[9] = {sx, sy, 1.0, 0.0, 0.0, 0.0, -sx * dx, -sy * dx, dx};
    const double r1[9] = {0.0, 0.0, 0.0, sx, sy, 1.0, -sx * dy, -sy * dy, dy};
    std::memcpy(A[i], r0, sizeof(r0));
    std::memcpy(A[i + 4], r1, sizeof(r1));
  }
  if (!solve8(A)) return false;
  for (int k = 0; k < 8; ++k) M[k] = A[k][8];
  M[8] = 1.0;
  return true;
}

bool invert3(const double S[9], double D[9]) {  // cv::invert, 3x3 closed form
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

}  // namespace

// ---------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------
struct FwdGeom {
  int mw, mh;          // mosaic
  int iw, ih, ch;      // frames
  int bw0;             // cv::warpPerspective block width (X0 is evaluated at block starts)
  size_t frame_stride; // bytes between frames
  size_t row_step;     // bytes between rows of a frame
};

__device__ __forceinline__ int round_half_even_sat(double v) {
  v = fmax((double)INT_MIN, fmin((double)INT_MAX, v));
  return (int)rint(v);
}

__device__ __forceinline__ void distort_dev(const amhip_camera& c, double* px, double* py) {
  double x = *px, y = *py;
  if (c.distortion == AMHIP_DIST_RADTAN) {
    const double k1 = c.dist[0], k2 = c.dist[1], p1 = c.dist[2], p2 = c.dist[3];
    const double mx2 = x * x, my2 = y * y, mxy = x * y;
    const double rho2 = mx2 + my2;
    const double rad = k1 * rho2 + k2 * rho2 * rho2;
    const double nx = x + (x * rad + 2.0 * p1 * mxy + p2 * (rho2 + 2.0 * mx2));
    const double ny = y + (y * rad + 2.0 * p2 * mxy + p1 * (rho2 + 2.0 * my2));
    x = nx;
    y = ny;
  } else if (c.distortion == AMHIP_DIST_EQUIDISTANT) {
    const double r = sqrt(x * x + y * y);
    const double theta = atan(r);
    const double th2 = theta * theta, th4 = th2 * th2, th6 = th4 * th2, th8 = th4 * th4;
    const double thetad =
        theta * (1.0 + c.dist[0] * th2 + c.dist[1] * th4 + c.dist[2] * th6 + c.dist[3] * th8);
    const double scaling = (r > 1e-8) ? thetad / r : 1.0;
    x = x * scaling;
    y = y * scaling;
  }
  *px = x;
  *py = y;
}

// aslam MappedUndistorter::processImage = cv::remap(INTER_LINEAR, BORDER_CONSTANT)
// with 1/32-pixel coordinates and 15-bit weights.  out: G x ih x iw x ch, packed.
__global__ void __launch_bounds__(256)
k_fwd_undistort(amhip_camera cam, FwdGeom g, const uint8_t* __restrict__ frames, int G,
                uint8_t* __restrict__ out) {
  const int u = blockIdx.x * 256 + threadIdx.x;
  const int v = blockIdx.y;
  const int f = blockIdx.z;
  if (u >= g.iw || f >= G) return;
  double x = ((double)u - cam.cu) / cam.fu;
  double y = ((double)v - cam.cv) / cam.fv;
  distort_dev(cam, &x, &y);
  const float mx = (float)(cam.fu * x + cam.cu);
  const float my = (float)(cam.fv * y + cam.cv);
  const int sx = round_half_even_sat((double)mx * 32.0);
  const int sy = round_half_even_sat((double)my * 32.0);
  const int ix = sx >> 5, iy = sy >> 5, fx = sx & 31, fy = sy & 31;
  const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32,
            w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
  const uint8_t* src = frames + (size_t)f * g.frame_stride;
  for (int c = 0; c < g.ch; ++c) {
    auto px = [&](int xx, int yy) -> int {
      if (xx < 0 || yy < 0 || xx >= g.iw || yy >= g.ih) return 0;
      return src[(size_t)yy * g.row_step + (size_t)xx * g.ch + c];
    };
    const int acc =
        w00 * px(ix, iy) + w01 * px(ix + 1, iy) + w10 * px(ix, iy + 1) + w11 * px(ix + 1, iy + 1);
    out[(((size_t)f * g.ih + v) * g.iw + u) * g.ch + c] = (uint8_t)((acc + (1 << 14)) >> 15);
  }
}

// cv::warpPerspective(INTER_NEAREST, BORDER_CONSTANT) of G frames + addImage's mask,
// evaluated on every frame's region only.
__global__ void __launch_bounds__(256)
k_fwd_warp(FwdGeom g, const FwdFrame* __restrict__ fr, const uint8_t* __restrict__ frames, int G,
           uint8_t* __restrict__ warped, uint8_t* __restrict__ mask) {
  const int f = blockIdx.z;
  const FwdFrame& F = fr[f];
  const int rx = blockIdx.x * 256 + threadIdx.x;
  const int ry = blockIdx.y;
  if (rx >= F.w || ry >= F.h) return;
  const int x = F.x0 + rx, y = F.y0 + ry;
  const double* M = F.m;
  const int xs = (x / g.bw0) * g.bw0;  // start of OpenCV's block
  const int x1 = x - xs;
  const double X0 = M[0] * xs + M[1] * y + M[2];
  const double Y0 = M[3] * xs + M[4] * y + M[5];
  const double W0 = M[6] * xs + M[7] * y + M[8];
  double W = W0 + M[6] * x1;
  W = W != 0.0 ? 1.0 / W : 0.0;
  int X = round_half_even_sat((X0 + M[0] * x1) * W);
  int Y = round_half_even_sat((Y0 + M[3] * x1) * W);
  X = max(-32768, min(32767, X));  // saturate_cast<short>
  Y = max(-32768, min(32767, Y));
  const bool inside = X >= 0 && Y >= 0 && X < g.iw && Y < g.ih;
  const size_t at = (size_t)F.off + (size_t)ry * F.w + rx;
  const uint8_t* src = frames + (size_t)f * g.frame_stride + (size_t)Y * g.row_step +
                       (size_t)X * g.ch;
  bool any = false;
  for (int c = 0; c < g.ch; ++c) {
    const uint8_t v = inside ? src[c] : (uint8_t)0;
    warped[at * g.ch + c] = v;
    any = any || v > 0;  // (img > 0.1) per channel, RGB2GRAY of 0/255 is non-zero iff any is
  }
  mask[at] = any ? 255 : 0;
}

// A "plane" is one W x H raster of the distance transform: a frame's region, or
// the whole mosaic (feed(result_, result_mask_) of updateOrthomosaic).
struct DtPlane {
  int w, h;
  unsigned long long off;
};
__device__ __forceinline__ DtPlane dt_plane(const FwdFrame* fr, int f, int mw, int mh) {
  DtPlane p;
  if (fr) {
    p.w = fr[f].w;
    p.h = fr[f].h;
    p.off = fr[f].off;
  } else {
    p.w = mw;
    p.h = mh;
    p.off = 0;
  }
  return p;
}

// rows: distance to the nearest zero of the same row (saturated at 255).
// One wave per row; grid = (ceil(max rows / 4), planes).
__global__ void __launch_bounds__(256)
k_fwd_dt_rows(const FwdFrame* __restrict__ fr, int mw, int mh, const uint8_t* __restrict__ mask,
              uint8_t* __restrict__ dist) {
  const DtPlane P = dt_plane(fr, blockIdx.y, mw, mh);
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= P.h) return;
  const int w = P.w;
  const uint8_t* m = mask + P.off + (size_t)row * w;
  uint8_t* d = dist + P.off + (size_t)row * w;
  const int kFar = 1 << 24;
  int last = -kFar;
  for (int x0 = 0; x0 < w; x0 += 64) {
    const int x = x0 + lane;
    int v = (x < w && m[x] == 0) ? x : -kFar;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(v, off, 64);
      if (lane >= off) v = max(v, o);
    }
    v = max(v, last);
    if (x < w) d[x] = (uint8_t)min(x - v, 255);
    last = __shfl(v, 63, 64);
  }
  int next = kFar;
  for (int x0 = ((w - 1) / 64) * 64; x0 >= 0; x0 -= 64) {
    const int x = x0 + lane;
    int v = (x < w && m[x] == 0) ? x : kFar;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_down(v, off, 64);
      if (lane + off < 64) v = min(v, o);
    }
    v = min(v, next);
    if (x < w) d[x] = (uint8_t)min((int)d[x], min(v - x, 255));
    next = __shfl(v, 0, 64);
  }
}

// columns: d(x, y) = min_y' (|y - y'| + row distance(x, y')), in place.
// One lane per column; grid = (ceil(max width / 256), planes).
__global__ void __launch_bounds__(256)
k_fwd_dt_cols(const FwdFrame* __restrict__ fr, int mw, int mh, uint8_t* __restrict__ dist) {
  const DtPlane P = dt_plane(fr, blockIdx.y, mw, mh);
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= P.w) return;
  uint8_t* col = dist + P.off + x;
  const size_t w = (size_t)P.w;
  // the chain d = min(v[y], d + 1) is cheap; what costs is the latency of the
  // loads, so eight rows are fetched at a time before the chain runs over them
  constexpr int U = 8;
  int d = 255;
  int y = 0;
  for (; y + U <= P.h; y += U) {
    int v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) v[k] = col[(size_t)(y + k) * w];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      d = min(v[k], min(d + 1, 255));
      col[(size_t)(y + k) * w] = (uint8_t)d;
    }
  }
  for (; y < P.h; ++y) {
    d = min((int)col[(size_t)y * w], min(d + 1, 255));
    col[(size_t)y * w] = (uint8_t)d;
  }
  d = 255;
  y = P.h - 1;
  for (; y - U + 1 >= 0; y -= U) {
    int v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) v[k] = col[(size_t)(y - k) * w];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      d = min(v[k], min(d + 1, 255));
      col[(size_t)(y - k) * w] = (uint8_t)d;
    }
  }
  for (; y >= 0; --y) {
    d = min((int)col[(size_t)y * w], min(d + 1, 255));
    col[(size_t)y * w] = (uint8_t)d;
  }
}

__device__ __forceinline__ float feather_weight(int d) {
  // createWeightMap: multiply(dist, 0.02f) then threshold(1, THRESH_TRUNC);
  // a saturated distance (>= 255) is far beyond the truncation point (50)
  const float v = (float)d * 0.02f;
  return v > 1.0f ? 1.0f : v;
}

// FeatherBlender::feed for the G frames of a batch, ascending.  The launch
// covers the union (ux0, uy0, uw, uh) of the frames' regions; a workgroup owns a
// strip of 256 pixels of one mosaic row, first lists (in ascending order) the
// frames whose region meets the strip, then every pixel walks that short list.
// A frame whose region does not contain the pixel would add (short)(0 * 0) and
// weight 0.
__global__ void __launch_bounds__(256)
k_fwd_feed(const FwdFrame* __restrict__ fr, const uint8_t* __restrict__ warped,
           const uint8_t* __restrict__ dist, int ch, int G, int mw, int ux0, int uy0, int uw,
           int16_t* __restrict__ dst16, float* __restrict__ dst_weight) {
  __shared__ int s_roi[64][4];
  __shared__ unsigned long long s_off[64];
  __shared__ int s_n;
  const int xs = ux0 + blockIdx.x * 256;  // the strip: [xs, xs + 256) x {y}
  const int y = uy0 + blockIdx.y;
  if (threadIdx.x < 64) {  // G <= 64: one wave builds the list with a ballot
    bool meets = false;
    int x0 = 0, y0 = 0, w = 0, h = 0;
    if ((int)threadIdx.x < G) {
      x0 = fr[threadIdx.x].x0;
      y0 = fr[threadIdx.x].y0;
      w = fr[threadIdx.x].w;
      h = fr[threadIdx.x].h;
      meets = w > 0 && y >= y0 && y < y0 + h && x0 < xs + 256 && x0 + w > xs;
    }
    const unsigned long long m = __ballot(meets);
    if (meets) {
      const int k = __popcll(m & ((1ull << threadIdx.x) - 1ull));
      s_roi[k][0] = x0;
      s_roi[k][1] = y0;
      s_roi[k][2] = w;
      s_roi[k][3] = h;
      s_off[k] = fr[threadIdx.x].off;
    }
    if (threadIdx.x == 0) s_n = __popcll(m);
  }
  __syncthreads();
  const int n = s_n;
  const int rx = blockIdx.x * 256 + threadIdx.x;
  if (n == 0 || rx >= uw) return;
  const int x = ux0 + rx;
  const size_t k = (size_t)y * mw + x;
  int16_t a0 = 0, a1 = 0, a2 = 0;
  float ws = 0.0f;
  bool loaded = false;
  for (int f = 0; f < n; ++f) {
    const int px = x - s_roi[f][0], py = y - s_roi[f][1];
    if (px < 0 || px >= s_roi[f][2]) continue;
    if (!loaded) {
      a0 = dst16[3 * k + 0];
      a1 = dst16[3 * k + 1];
      a2 = dst16[3 * k + 2];
      ws = dst_weight[k];
      loaded = true;
    }
    const size_t at = (size_t)s_off[f] + (size_t)py * s_roi[f][2] + px;
    const float w = feather_weight(dist[at]);
    const uint8_t* s = warped + at * ch;
    const int v0 = s[0], v1 = s[ch == 1 ? 0 : 1], v2 = s[ch == 1 ? 0 : 2];
    a0 = (int16_t)(a0 + (int16_t)(int)((float)v0 * w));
    a1 = (int16_t)(a1 + (int16_t)(int)((float)v1 * w));
    a2 = (int16_t)(a2 + (int16_t)(int)((float)v2 * w));
    ws += w;
  }
  if (!loaded) return;
  dst16[3 * k + 0] = a0;
  dst16[3 * k + 1] = a1;
  dst16[3 * k + 2] = a2;
  dst_weight[k] = ws;
}

// feed(result_, result_mask_) of updateOrthomosaic (:118): 16-bit source
__global__ void __launch_bounds__(256)
k_fwd_feed16(const int16_t* __restrict__ src, const uint8_t* __restrict__ dist, size_t pixels,
             int16_t* __restrict__ dst16, float* __restrict__ dst_weight) {
  const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= pixels) return;
  const float w = feather_weight(dist[k]);
  for (int c = 0; c < 3; ++c)
    dst16[3 * k + c] = (int16_t)(dst16[3 * k + c] + (int16_t)(int)((float)src[3 * k + c] * w));
  dst_weight[k] += w;
}

// FeatherBlender::blend (+ the "unobserved pixels" pass of batch(), :178-186)
__global__ void __launch_bounds__(256)
k_fwd_blend(int16_t* __restrict__ dst16, const float* __restrict__ dst_weight, size_t pixels,
            int batch_tail, int16_t* __restrict__ result, uint8_t* __restrict__ result_mask) {
  const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= pixels) return;
  const float eps = 1e-5f;
  const float wsum = dst_weight[k];
  const float den = wsum + eps;
  int16_t r[3];
  for (int c = 0; c < 3; ++c) {
    // float division, correctly rounded (double quotient of two floats rounds
    // to the same float), then static_cast<short>
    const float q = (float)((double)(float)dst16[3 * k + c] / (double)den);
    r[c] = (int16_t)(int)q;
  }
  const bool on = wsum > eps;
  if (!on) r[0] = r[1] = r[2] = 0;
  for (int c = 0; c < 3; ++c) dst16[3 * k + c] = r[c];  // normalizeUsingWeightMap is in place
  if (batch_tail && !(r[0] > 0 && r[1] > 0 && r[2] > 0)) r[0] = r[1] = r[2] = 0;
  for (int c = 0; c < 3; ++c) result[3 * k + c] = r[c];
  result_mask[k] = on ? 255 : 0;
}

// ---------------------------------------------------------------------------
// host driver
// ---------------------------------------------------------------------------
static int fwd_use(Mosaic* m) {
  AMHIP_TRY(hipSetDevice(m->device));
  return AMHIP_OK;
}

static int fwd_prepare_blender(Mosaic* m) {  // prepareBlenderForNextImage()
  AMHIP_TRY(hipMemsetAsync(m->dst16, 0, m->pixels * 3 * sizeof(int16_t), m->stream));
  AMHIP_TRY(hipMemsetAsync(m->dst_weight, 0, m->pixels * sizeof(float), m->stream));
  return AMHIP_OK;
}

static FwdGeom fwd_geom(const Mosaic* m, int ch, size_t frame_stride, size_t row_step) {
  FwdGeom g;
  g.mw = m->desc.width_mosaic_pixels;
  g.mh = m->desc.height_mosaic_pixels;
  g.iw = m->cam.width;
  g.ih = m->cam.height;
  g.ch = ch;
  const int bh0 = std::min(32 / 2, g.mh);
  g.bw0 = std::min(32 * 32 / bh0, g.mw);
  g.frame_stride = frame_stride;
  g.row_step = row_step;
  return g;
}

// distance transform of the regions of G frames (fr != null) or of one
// mosaic-sized plane (fr == null)
static int fwd_distance(Mosaic* m, const FwdFrame* fr, int G, int max_w, int max_h,
                        const uint8_t* mask, uint8_t* dist) {
  const int mw = m->desc.width_mosaic_pixels, mh = m->desc.height_mosaic_pixels;
  hipLaunchKernelGGL(k_fwd_dt_rows, dim3((unsigned)((max_h + 3) / 4), (unsigned)G), dim3(256), 0,
                     m->stream, fr, mw, mh, mask, dist);
  hipLaunchKernelGGL(k_fwd_dt_cols, dim3((unsigned)((max_w + 255) / 256), (unsigned)G), dim3(256),
                     0, m->stream, fr, mw, mh, dist);
  AMHIP_TRY(hipGetLastError());
  return AMHIP_OK;
}

// The region of the mosaic a frame can touch (see FwdFrame).  Mfwd maps image
// to mosaic pixels; the source rectangle grown by one pixel is mapped forward
// and its bounding box grown by two mosaic pixels.  If the rectangle reaches
// the homography's vanishing line the footprint is unbounded: whole mosaic.
static void fwd_region(const Mosaic* m, const double Mfwd[9], FwdFrame* out) {
  const int mw = m->desc.width_mosaic_pixels, mh = m->desc.height_mosaic_pixels;
  const double xs[2] = {-1.0, (double)m->cam.width}, ys[2] = {-1.0, (double)m->cam.height};
  double lo_x = 1e300, hi_x = -1e300, lo_y = 1e300, hi_y = -1e300;
  bool bounded = true;
  double w_ref = 0.0;
  for (int a = 0; a < 2 && bounded; ++a)
    for (int b = 0; b < 2; ++b) {
      const double w = Mfwd[6] * xs[a] + Mfwd[7] * ys[b] + Mfwd[8];
      if (a == 0 && b == 0) w_ref = w;
      if (!(w * w_ref > 0.0) || !(std::fabs(w) > 1e-9 * std::fabs(Mfwd[8]))) {
        bounded = false;
        break;
      }
      const double X = (Mfwd[0] * xs[a] + Mfwd[1] * ys[b] + Mfwd[2]) / w;
      const double Y = (Mfwd[3] * xs[a] + Mfwd[4] * ys[b] + Mfwd[5]) / w;
      if (!(std::fabs(X) < 1e9 && std::fabs(Y) < 1e9)) {
        bounded = false;
        break;
      }
      lo_x = std::min(lo_x, X);
      hi_x = std::max(hi_x, X);
      lo_y = std::min(lo_y, Y);
      hi_y = std::max(hi_y, Y);
    }
  int x0 = 0, y0 = 0, x1 = mw - 1, y1 = mh - 1;
  if (bounded) {
    x0 = std::max(0, (int)std::floor(lo_x) - 2);
    y0 = std::max(0, (int)std::floor(lo_y) - 2);
    x1 = std::min(mw - 1, (int)std::ceil(hi_x) + 2);
    y1 = std::min(mh - 1, (int)std::ceil(hi_y) + 2);
  }
  out->x0 = x0;
  out->y0 = y0;
  out->w = std::max(0, x1 - x0 + 1);
  out->h = std::max(0, y1 - y0 + 1);
}

constexpr int kFwdMaxFrames = 64;                       // frames per launch (k_fwd_feed's table)
constexpr size_t kFwdChunkBytes = size_t(512) << 20;    // warped + mask + dist per launch

// feed `F` device-resident frames, ascending, in chunks
static int fwd_feed_frames(Mosaic* m, const double* T_G_C, size_t F, const uint8_t* frames,
                           size_t frame_stride, size_t row_step, int ch, bool quirk) {
  if (F == 0) return AMHIP_OK;
  // ---- plan: homography + region of every frame, cut into chunks ------------------
  struct Chunk {
    size_t first;
    int count, max_w, max_h, ux0, uy0, ux1, uy1;
    size_t px;
  };
  int rc;
  // the pinned staging table of the previous call may still be in flight
  if (m->frames_pending) {
    AMHIP_TRY(hipEventSynchronize(m->frames_event));
    m->frames_pending = false;
  }
  if (m->host_frames_cap < F) {
    if (m->host_frames) AMHIP_TRY(hipHostFree(m->host_frames));
    m->host_frames = nullptr;
    m->host_frames_cap = 0;
    const size_t want = F + F / 4 + 64;
    AMHIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&m->host_frames), want * sizeof(FwdFrame), 0));
    m->host_frames_cap = want;
  }
  std::vector<Chunk> chunks;
  Chunk cur = {0, 0, 0, 0, INT_MAX, INT_MAX, -1, -1, 0};
  for (size_t f = 0; f < F; ++f) {
    FwdFrame fr;
    double M[9];
    if (!frame_homography(m->cam, m->desc, T_G_C + 7 * f, quirk, M) || !invert3(M, fr.m))
      return fwd_arg_fail("forward homography: degenerate frame (camera parallel to the ground?)");
    fwd_region(m, M, &fr);
    const size_t rpx = (size_t)fr.w * fr.h;
    if (cur.count > 0 &&
        (cur.count >= kFwdMaxFrames || (cur.px + rpx) * (size_t)(ch + 2) > kFwdChunkBytes)) {
      chunks.push_back(cur);
      cur = Chunk{f, 0, 0, 0, INT_MAX, INT_MAX, -1, -1, 0};
    }
    fr.off = cur.px;
    cur.px += rpx;
    ++cur.count;
    if (rpx) {
      cur.max_w = std::max(cur.max_w, fr.w);
      cur.max_h = std::max(cur.max_h, fr.h);
      cur.ux0 = std::min(cur.ux0, fr.x0);
      cur.uy0 = std::min(cur.uy0, fr.y0);
      cur.ux1 = std::max(cur.ux1, fr.x0 + fr.w - 1);
      cur.uy1 = std::max(cur.uy1, fr.y0 + fr.h - 1);
    }
    m->host_frames[f] = fr;
  }
  chunks.push_back(cur);
  size_t max_px = 0;
  int max_count = 0;
  for (const Chunk& c : chunks) {
    max_px = std::max(max_px, c.px);
    max_count = std::max(max_count, c.count);
  }
  if (max_px == 0) return AMHIP_OK;  // no frame reaches the mosaic
  if ((rc = ensure_capacity(&m->warped, &m->warped_cap, max_px * ch))) return rc;
  if ((rc = ensure_capacity(&m->mask, &m->mask_cap, max_px))) return rc;
  if ((rc = ensure_capacity(&m->dist, &m->dist_cap, max_px))) return rc;
  {
    void* p = m->frames;
    size_t cap = m->frames_cap * sizeof(FwdFrame);
    if ((rc = ensure_bytes(&p, &cap, F * sizeof(FwdFrame)))) return rc;
    m->frames = static_cast<FwdFrame*>(p);
    m->frames_cap = cap / sizeof(FwdFrame);
  }
  FwdGeom g0 = fwd_geom(m, ch, frame_stride, row_step);
  const size_t fbytes = (size_t)g0.iw * g0.ih * ch;
  if (m->cam.distortion != AMHIP_DIST_NONE &&
      (rc = ensure_capacity(&m->undist, &m->undist_cap, (size_t)max_count * fbytes)))
    return rc;
  // ONE pinned -> device copy of the whole table: nothing below waits on the host
  AMHIP_TRY(hipMemcpyAsync(m->frames, m->host_frames, F * sizeof(FwdFrame), hipMemcpyHostToDevice,
                           m->stream));
  AMHIP_TRY(hipEventRecord(m->frames_event, m->stream));
  m->frames_pending = true;
  // ---- run the chunks ----------------------------------------------------------------
  for (const Chunk& c : chunks) {
    if (c.px == 0) continue;  // none of the chunk's frames reaches the mosaic
    const int G = c.count;
    const FwdFrame* table = m->frames + c.first;
    FwdGeom g = g0;
    const uint8_t* src = frames + c.first * frame_stride;
    if (m->cam.distortion != AMHIP_DIST_NONE) {
      hipLaunchKernelGGL(k_fwd_undistort,
                         dim3((unsigned)((g.iw + 255) / 256), (unsigned)g.ih, (unsigned)G),
                         dim3(256), 0, m->stream, m->cam, g, src, G, m->undist);
      src = m->undist;
      g.frame_stride = fbytes;
      g.row_step = (size_t)g.iw * ch;
    }
    hipLaunchKernelGGL(k_fwd_warp,
                       dim3((unsigned)((c.max_w + 255) / 256), (unsigned)c.max_h, (unsigned)G),
                       dim3(256), 0, m->stream, g, table, src, G, m->warped, m->mask);
    AMHIP_TRY(hipGetLastError());
    if ((rc = fwd_distance(m, table, G, c.max_w, c.max_h, m->mask, m->dist))) return rc;
    const int uw = c.ux1 - c.ux0 + 1, uh = c.uy1 - c.uy0 + 1;
    hipLaunchKernelGGL(k_fwd_feed, dim3((unsigned)((uw + 255) / 256), (unsigned)uh), dim3(256), 0,
                       m->stream, table, m->warped, m->dist, ch, G, g.mw, c.ux0, c.uy0, uw,
                       m->dst16, m->dst_weight);
    AMHIP_TRY(hipGetLastError());
  }
  return AMHIP_OK;
}

static int fwd_blend(Mosaic* m, bool batch_tail) {
  hipLaunchKernelGGL(k_fwd_blend, dim3((unsigned)((m->pixels + 255) / 256)), dim3(256), 0,
                     m->stream, m->dst16, m->dst_weight, m->pixels, batch_tail ? 1 : 0,
                     m->result16, m->result_mask);
  AMHIP_TRY(hipGetLastError());
  return AMHIP_OK;
}

static int fwd_batch_dev(Mosaic* m, const double* T_G_C, size_t F, const uint8_t* frames,
                         size_t frame_stride, size_t row_step, int ch) {
  int rc = fwd_feed_frames(m, T_G_C, F, frames, frame_stride, row_step, ch, /*quirk=*/true);
  if (rc) return rc;
  return fwd_blend(m, /*batch_tail=*/true);
}

static int fwd_update_dev(Mosaic* m, const double* T_G_C7, const uint8_t* frame, size_t row_step,
                          int ch) {
  int rc;
  if ((rc = fwd_feed_frames(m, T_G_C7, 1, frame, 0, row_step, ch, /*quirk=*/false))) return rc;
  if ((rc = fwd_blend(m, false))) return rc;       // blender_->blend(result_, result_mask_)
  if ((rc = fwd_prepare_blender(m))) return rc;    // prepareBlenderForNextImage()
  if ((rc = ensure_capacity(&m->dist, &m->dist_cap, m->pixels))) return rc;
  if ((rc = fwd_distance(m, nullptr, 1, m->desc.width_mosaic_pixels, m->desc.height_mosaic_pixels,
                         m->result_mask, m->dist)))
    return rc;  // addImage(result_, result_mask_)
  hipLaunchKernelGGL(k_fwd_feed16, dim3((unsigned)((m->pixels + 255) / 256)), dim3(256), 0,
                     m->stream, m->result16, m->dist, m->pixels, m->dst16, m->dst_weight);
  AMHIP_TRY(hipGetLastError());
  return AMHIP_OK;
}

static int fwd_download(Mosaic* m, int16_t* result, uint8_t* mask) {
  if (result)
    AMHIP_TRY(hipMemcpyAsync(result, m->result16, m->pixels * 3 * sizeof(int16_t),
                             hipMemcpyDeviceToHost, m->stream));
  if (mask)
    AMHIP_TRY(hipMemcpyAsync(mask, m->result_mask, m->pixels, hipMemcpyDeviceToHost, m->stream));
  AMHIP_TRY(hipStreamSynchronize(m->stream));
  return AMHIP_OK;
}

static int fwd_stage(Mosaic* m, const uint8_t* const* images, const size_t* steps, size_t F,
                     int ch, size_t* frame_bytes) {
  const size_t row = (size_t)m->cam.width * ch;
  const size_t fb = row * (size_t)m->cam.height;
  int rc;
  if ((rc = ensure_capacity(&m->stage, &m->stage_cap, F * fb))) return rc;
  for (size_t f = 0; f < F; ++f) {
    if (!images[f]) return fwd_arg_fail("null image");
    if (steps[f] < row) return fwd_arg_fail("image step smaller than a row");
    if (steps[f] == row)  // dense rows: one linear copy
      AMHIP_TRY(hipMemcpyAsync(m->stage + f * fb, images[f], fb, hipMemcpyHostToDevice, m->stream));
    else
      AMHIP_TRY(hipMemcpy2DAsync(m->stage + f * fb, row, images[f], steps[f], row,
                                 (size_t)m->cam.height, hipMemcpyHostToDevice, m->stream));
  }
  *frame_bytes = fb;
  return AMHIP_OK;
}

}  // namespace amhip

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
using namespace amhip;

struct amhip_mosaic {
  Mosaic impl;
};

extern "C" {

int amhip_mosaic_homography(const amhip_mosaic_desc* desc, const amhip_camera* cam,
                            const double* T_G_C7, int batch_quirk, double* M9) {
  if (!desc || !cam || !T_G_C7 || !M9) return fwd_arg_fail("amhip_mosaic_homography: null argument");
  if (!frame_homography(*cam, *desc, T_G_C7, batch_quirk != 0, M9))
    return fwd_arg_fail("forward homography: degenerate frame");
  return AMHIP_OK;
}

int amhip_mosaic_create(const amhip_mosaic_desc* desc, const amhip_camera* cam, int device,
                        amhip_mosaic** out) {
  if (!desc || !cam || !out) return fwd_arg_fail("amhip_mosaic_create: null argument");  // CHECK(ncameras_)
  if (desc->width_mosaic_pixels <= 0 || desc->height_mosaic_pixels <= 0 || cam->width <= 0 ||
      cam->height <= 0)
    return fwd_arg_fail("amhip_mosaic_create: empty mosaic or image");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    (void)hipGetLastError();
    set_last_error("no HIP device available (libaerial_mapper_hip has no CPU fallback)");
    return AMHIP_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= ndev) return fwd_arg_fail("amhip_mosaic_create: bad device index");
  amhip_mosaic* h = new amhip_mosaic();
  Mosaic* m = &h->impl;
  m->desc = *desc;
  m->cam = *cam;
  m->device = device;
  m->pixels = (size_t)desc->width_mosaic_pixels * (size_t)desc->height_mosaic_pixels;
  int rc = AMHIP_OK;
  do {
    if ((rc = fwd_use(m))) break;
    hipError_t e = hipStreamCreateWithFlags(&m->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      rc = hip_fail(e, "hipStreamCreate", __FILE__, __LINE__);
      break;
    }
    m->stream = m->own_stream;
    e = hipEventCreateWithFlags(&m->frames_event, hipEventDisableTiming);
    if (e != hipSuccess) {
      rc = hip_fail(e, "hipEventCreate", __FILE__, __LINE__);
      break;
    }
    void* p = nullptr;
    size_t cap = 0;
    if ((rc = ensure_bytes(&p, &cap, m->pixels * 3 * sizeof(int16_t)))) break;
    m->dst16 = static_cast<int16_t*>(p);
    p = nullptr; cap = 0;
    if ((rc = ensure_bytes(&p, &cap, m->pixels * sizeof(float)))) break;
    m->dst_weight = static_cast<float*>(p);
    p = nullptr; cap = 0;
    if ((rc = ensure_bytes(&p, &cap, m->pixels * 3 * sizeof(int16_t)))) break;
    m->result16 = static_cast<int16_t*>(p);
    p = nullptr; cap = 0;
    if ((rc = ensure_bytes(&p, &cap, m->pixels))) break;
    m->result_mask = static_cast<uint8_t*>(p);
    if ((rc = fwd_prepare_blender(m))) break;  // constructor (:27)
    hipError_t e2 = hipMemsetAsync(m->result16, 0, m->pixels * 3 * sizeof(int16_t), m->stream);
    if (e2 == hipSuccess) e2 = hipMemsetAsync(m->result_mask, 0, m->pixels, m->stream);
    if (e2 == hipSuccess) e2 = hipStreamSynchronize(m->stream);
    if (e2 != hipSuccess) rc = hip_fail(e2, "mosaic init", __FILE__, __LINE__);
  } while (0);
  if (rc) {
    amhip_mosaic_destroy(h);
    return rc;
  }
  *out = h;
  return AMHIP_OK;
}

int amhip_mosaic_destroy(amhip_mosaic* h) {
  if (!h) return AMHIP_OK;
  Mosaic* m = &h->impl;
  (void)hipSetDevice(m->device);
  if (m->stream) (void)hipStreamSynchronize(m->stream);
  void* bufs[] = {m->dst16, m->dst_weight, m->result16, m->result_mask, m->warped, m->mask,
                  m->dist,  m->undist,     m->stage,    m->frames};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  if (m->host_frames) (void)hipHostFree(m->host_frames);
  if (m->frames_event) (void)hipEventDestroy(m->frames_event);
  if (m->own_stream) (void)hipStreamDestroy(m->own_stream);
  delete h;
  return AMHIP_OK;
}

int amhip_mosaic_set_stream(amhip_mosaic* h, void* hip_stream) {
  if (!h) return fwd_arg_fail("null mosaic");
  Mosaic* m = &h->impl;
  int rc = fwd_use(m);
  if (rc) return rc;
  AMHIP_TRY(hipStreamSynchronize(m->stream));
  m->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : m->own_stream;
  return AMHIP_OK;
}

int amhip_mosaic_synchronize(amhip_mosaic* h) {
  if (!h) return fwd_arg_fail("null mosaic");
  Mosaic* m = &h->impl;
  int rc = fwd_use(m);
  if (rc) return rc;
  AMHIP_TRY(hipStreamSynchronize(m->stream));
  return AMHIP_OK;
}

int amhip_mosaic_reset(amhip_mosaic* h) {
  if (!h) return fwd_arg_fail("null mosaic");
  Mosaic* m = &h->impl;
  int rc = fwd_use(m);
  if (rc) return rc;
  if ((rc = fwd_prepare_blender(m))) return rc;
  AMHIP_TRY(hipMemsetAsync(m->result16, 0, m->pixels * 3 * sizeof(int16_t), m->stream));
  AMHIP_TRY(hipMemsetAsync(m->result_mask, 0, m->pixels, m->stream));
  return AMHIP_OK;
}

int amhip_mosaic_batch_dev(amhip_mosaic* h, const double* T_G_C, size_t num_frames,
                           const void* dev_frames, size_t frame_stride, size_t row_step,
                           int channels) {
  if (!h || !T_G_C || !dev_frames) return fwd_arg_fail("amhip_mosaic_batch_dev: null argument");
  if (channels != 1 && channels != 3) return fwd_arg_fail("channels must be 1 (8UC1) or 3 (8UC3)");
  Mosaic* m = &h->impl;
  if (row_step < (size_t)m->cam.width * channels) return fwd_arg_fail("row_step smaller than a row");
  int rc = fwd_use(m);
  if (rc) return rc;
  return fwd_batch_dev(m, T_G_C, num_frames, static_cast<const uint8_t*>(dev_frames), frame_stride,
                       row_step, channels);
}

int amhip_mosaic_batch(amhip_mosaic* h, const double* T_G_C, size_t num_frames,
                       const void* const* images, const size_t* steps, int channels,
                       int16_t* result, uint8_t* result_mask) {
  if (!h || !T_G_C || !images || !steps) return fwd_arg_fail("amhip_mosaic_batch: null argument");
  if (channels != 1 && channels != 3) return fwd_arg_fail("channels must be 1 (8UC1) or 3 (8UC3)");
  Mosaic* m = &h->impl;
  int rc = fwd_use(m);
  if (rc) return rc;
  // frames are staged to the device in chunks of bounded size
  const size_t fbytes = (size_t)m->cam.width * m->cam.height * channels;
  size_t gmax = (size_t(1) << 30) / (fbytes ? fbytes : 1);
  if (gmax < 1) gmax = 1;
  for (size_t f0 = 0; f0 < num_frames; f0 += gmax) {
    const size_t G = std::min<size_t>(gmax, num_frames - f0);
    size_t fb = 0;
    if ((rc = fwd_stage(m, reinterpret_cast<const uint8_t* const*>(images) + f0, steps + f0, G,
                        channels, &fb)))
      return rc;
    if ((rc = fwd_feed_frames(m, T_G_C + 7 * f0, G, m->stage, fb,
                              (size_t)m->cam.width * channels, channels, /*quirk=*/true)))
      return rc;
    AMHIP_TRY(hipStreamSynchronize(m->stream));  // the staging buffer is reused
  }
  if ((rc = fwd_blend(m, /*batch_tail=*/true))) return rc;
  return fwd_download(m, result, result_mask);
}

int amhip_mosaic_update_dev(amhip_mosaic* h, const double* T_G_C7, const void* dev_frame,
                            size_t row_step, int channels) {
  if (!h || !T_G_C7 || !dev_frame) return fwd_arg_fail("amhip_mosaic_update_dev: null argument");
  if (channels != 1 && channels != 3) return fwd_arg_fail("channels must be 1 (8UC1) or 3 (8UC3)");
  Mosaic* m = &h->impl;
  if (row_step < (size_t)m->cam.width * channels) return fwd_arg_fail("row_step smaller than a row");
  int rc = fwd_use(m);
  if (rc) return rc;
  return fwd_update_dev(m, T_G_C7, static_cast<const uint8_t*>(dev_frame), row_step, channels);
}

int amhip_mosaic_update(amhip_mosaic* h, const double* T_G_C7, const void* image, size_t step,
                        int channels, int16_t* result, uint8_t* result_mask) {
  if (!h || !T_G_C7 || !image) return fwd_arg_fail("amhip_mosaic_update: null argument");
  if (channels != 1 && channels != 3) return fwd_arg_fail("channels must be 1 (8UC1) or 3 (8UC3)");
  Mosaic* m = &h->impl;
  int rc = fwd_use(m);
  if (rc) return rc;
  size_t fb = 0;
  const uint8_t* img = static_cast<const uint8_t*>(image);
  if ((rc = fwd_stage(m, &img, &step, 1, channels, &fb))) return rc;
  if ((rc = fwd_update_dev(m, T_G_C7, m->stage, (size_t)m->cam.width * channels, channels)))
    return rc;
  return fwd_download(m, result, result_mask);
}

int amhip_mosaic_download(amhip_mosaic* h, int16_t* result, uint8_t* result_mask) {
  if (!h) return fwd_arg_fail("null mosaic");
  Mosaic* m = &h->impl;
  int rc = fwd_use(m);
  if (rc) return rc;
  return fwd_download(m, result, result_mask);
}

int amhip_mosaic_device_ptr(amhip_mosaic* h, void** result_16sc3, void** result_mask) {
  if (!h) return fwd_arg_fail("null mosaic");
  if (result_16sc3) *result_16sc3 = h->impl.result16;
  if (result_mask) *result_mask = h->impl.result_mask;
  return AMHIP_OK;
}

}  // extern "C"
