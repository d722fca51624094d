// amhip_ortho_fold.h -- arithmetic of one (cell, frame) pair of the backward-grid
// fold, ortho-backward-grid.cc:144-208, in two forms:
//
//   exact_view()   the reference's own operations, one for one (minkindr
//                  transform in Eigen's order, aslam pinhole project3, the
//                  squares under asin's sqrt), double, no contraction;
//   fast_*         a bounded-error evaluation (pose as a 3x4 matrix, box test
//                  without the division) whose every DECISION carries a margin
//                  that covers the worst-case distance to the reference's
//                  doubles; a cell with a decision inside a margin is replayed in
//                  the reference's arithmetic (exact_refold / exact_finish).  Outcomes are therefore the
//                  reference's, the cost is ~30 instead of ~85 FP64 operations
//                  for almost every pair.
//
// Host + device: the kernel (amhip_ortho.hip) and the CPU emulation the unit
// tests drive (tests/cpp/ortho_fold_emul.cc) compile the same functions.
//
// Error budget (u = 2^-53; |q|^2 = 1 within 1e-6; V = |L|_2, D = |L - p|_2 with
// p the camera centre in the ground frame; per component of the camera-frame
// point c):
//   reference   c_r = fl(q (x) L) + t, Eigen's _transformVector order:
//                 cross      2 u V          (two products + the difference)
//                 doubling   exact          -> |d(uv)| <= 4 u V, |uv| <= 2 V
//                 w * uv     4 u V + 2 u V
//                 v + w uv   3 u V
//                 cross 2    sqrt(2) 4 u V + 4 u V
//                 + cross 2  u V            -> |c_r - c*| <= 19.7 u V  (+ u |c| for + t)
//   here        c_a = M (L - p): M and p from long double (<= 0.5 u each after
//               rounding), three fma:  |c_a - c*| <= 0.5 u (V + D) + 9.7 u D
//               (+ 1.5 u V where a caller steps L.y through a slab as
//               ly0 + c * dly instead of taking the grid's rounded value)
//   => |c_a - c_r| <= 21.7 u V + 10.2 u D;  eps = 32 u V1 + 16 u D1  with the
//      1-norms V1 >= V, D1 >= D  (2^-48 V1 + 2^-49 D1): 1.5x headroom.
#ifndef AMHIP_ORTHO_FOLD_H_
#define AMHIP_ORTHO_FOLD_H_

#include <cmath>
#include <cstring>
#include <limits>

#if defined(__HIPCC__)
#define AMHIP_HD __host__ __device__ __forceinline__
#else
#define AMHIP_HD inline
#endif

namespace amhip {

// Per-frame inverse pose T_C_G = T_G_C^-1 (minkindr inverse()).
struct FramePose {
  double qw, qx, qy, qz;
  double tx, ty, tz;
  double _pad;
};

// The same pose as the linear map of Eigen's _transformVector formula
//   v + w (2 q x v) + q x (2 q x v)  =  M v,
//   M = (1 - 2|q_v|^2) I + 2 q_v q_v^T + 2 w [q_v]x
// (which is what that formula computes for ANY quaternion, unit or not), around
// the camera centre:  M L + t = M (L - p).  Both correctly rounded from a long
// double evaluation, so that the large coordinates cancel BEFORE anything is
// rounded at their magnitude.
struct FrameFast {
  double m[9];  // row-major
  double p[3];  // camera centre in the ground frame: M p + t = 0
  double _pad[4];
};

struct V3 {
  double x, y, z;
};

AMHIP_HD V3 cross3(const V3& a, const V3& b) {
  V3 r;
  r.x = a.y * b.z - a.z * b.y;
  r.y = a.z * b.x - a.x * b.z;
  r.z = a.x * b.y - a.y * b.x;
  return r;
}

// Eigen::Quaternion::_transformVector followed by the translation
// (kindr::minimal::QuatTransformation::transform).
AMHIP_HD V3 transform_point(const FramePose& T, const V3& v) {
  const V3 qv = {T.qx, T.qy, T.qz};
  V3 uv = cross3(qv, v);
  uv.x = uv.x + uv.x;
  uv.y = uv.y + uv.y;
  uv.z = uv.z + uv.z;
  const V3 c2 = cross3(qv, uv);
  V3 r;
  r.x = (v.x + T.qw * uv.x) + c2.x;
  r.y = (v.y + T.qw * uv.y) + c2.y;
  r.z = (v.z + T.qw * uv.z) + c2.z;
  r.x = r.x + T.tx;
  r.y = r.y + T.ty;
  r.z = r.z + T.tz;
  return r;
}

inline bool make_frame_fast(const FramePose& T, FrameFast* o) {
  typedef long double ld;
  static_assert(sizeof(FrameFast) == 128, "frame table layout");
  const ld w = T.qw, x = T.qx, y = T.qy, z = T.qz;
  ld m[9];
  m[0] = 1.0L - 2.0L * (y * y + z * z);
  m[1] = 2.0L * (x * y - w * z);
  m[2] = 2.0L * (x * z + w * y);
  m[3] = 2.0L * (x * y + w * z);
  m[4] = 1.0L - 2.0L * (x * x + z * z);
  m[5] = 2.0L * (y * z - w * x);
  m[6] = 2.0L * (x * z - w * y);
  m[7] = 2.0L * (y * z + w * x);
  m[8] = 1.0L - 2.0L * (x * x + y * y);
  // p = -M^-1 t by cofactors (M is a rotation up to 1e-6: well conditioned)
  const ld c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8],
           c02 = m[3] * m[7] - m[4] * m[6];
  const ld det = m[0] * c00 + m[1] * c01 + m[2] * c02;
  const ld c10 = m[2] * m[7] - m[1] * m[8], c11 = m[0] * m[8] - m[2] * m[6],
           c12 = m[1] * m[6] - m[0] * m[7];
  const ld c20 = m[1] * m[5] - m[2] * m[4], c21 = m[2] * m[3] - m[0] * m[5],
           c22 = m[0] * m[4] - m[1] * m[3];
  const ld tx = T.tx, ty = T.ty, tz = T.tz;
  const ld px = -(c00 * tx + c10 * ty + c20 * tz) / det;
  const ld py = -(c01 * tx + c11 * ty + c21 * tz) / det;
  const ld pz = -(c02 * tx + c12 * ty + c22 * tz) / det;
  for (int k = 0; k < 9; ++k) o->m[k] = (double)m[k];
  o->p[0] = (double)px;
  o->p[1] = (double)py;
  o->p[2] = (double)pz;
  o->_pad[0] = o->_pad[1] = o->_pad[2] = o->_pad[3] = 0.0;
  const ld n = w * w + x * x + y * y + z * z;
  // the error budget above assumes a unit quaternion, finite numbers and a
  // 64-bit significand for this function
  const bool finite = std::fabs(o->p[0]) < 1e300 && std::fabs(o->p[1]) < 1e300 &&
                      std::fabs(o->p[2]) < 1e300;
  return std::numeric_limits<long double>::digits >= 64 && std::fabs((double)(n - 1.0L)) < 1e-6 &&
         finite;
}

// One frame against the bounding sphere (centre, radius) of a tile's landmarks.
//   keep   false only if no landmark of the sphere can be visible (entirely
//          behind the camera or outside one side plane of the view pyramid)
//   full   every landmark of the sphere is visible, by a clear margin, in the
//          reference's arithmetic (undistorted pinhole only)
//   tmin, tmax   bounds of tan(theta), theta = angle between the optical axis
//          and the ray to a landmark (= pi/2 - the view angle), over the
//          landmarks of the sphere in front of the camera
// pl: unit inward normals of the four side planes through the optical centre.
struct FrameBounds {
  bool keep, full;
  double tmin, tmax;
};

// r_in > 0 (cameras with a distortion model): `full` <=> the whole sphere lies
// inside the inner cone |p| <= r_in z instead of inside the four planes (which
// are only an OUTER bound of the view then).
AMHIP_HD FrameBounds frame_bounds(const double (*pl)[3], const FramePose& T, const V3& centre,
                                  double radius, double slack, double r_in = 0.0) {
  const V3 cc = transform_point(T, centre);
  FrameBounds b;
  b.keep = !(cc.z < -radius);
  double dmin = HUGE_VAL;
  for (int k = 0; k < 4; ++k) {
    const double d = pl[k][0] * cc.x + pl[k][1] * cc.y + pl[k][2] * cc.z;
    if (d < -radius) b.keep = false;
    dmin = fmin(dmin, d);
  }
  // inside every plane by `slack` metres and at least that far in front:
  // u = d * |n| / z >= slack * |n| / z pixels inside the box, orders of
  // magnitude above the rounding of the reference's u (callers scale slack
  // with the coordinate magnitudes)
  const double zlo = cc.z - radius;
  b.full = b.keep && (dmin - radius > slack) && (zlo > slack) && (zlo > 1e-3);
  const double rxy = sqrt(cc.x * cc.x + cc.y * cc.y);
  const double zhi = cc.z + radius;
  // directed rounding by hand: 1e-12 relative dwarfs the few ulps of the
  // sqrt / divisions (whatever their implementation)
  b.tmin = zhi > 0.0 ? (fmax(rxy - radius, 0.0) / zhi) * (1.0 - 1e-12) : HUGE_VAL;
  b.tmax = zlo > 0.0 ? ((rxy + radius) / zlo) * (1.0 + 1e-12) : HUGE_VAL;
  if (r_in > 0.0) b.full = b.keep && (zlo > slack) && (zlo > 1e-3) && (b.tmax <= r_in * (1.0 - 1e-9));
  return b;
}

// Frame f can be left out of a tile's fold when some frame g that is fully
// visible over the tile beats it at every landmark by a clear margin:
//   tan(theta_f) >= tmin_f > tmax_g + 1e-5 (1 + tmin_f^2) >= tan(theta_g) + ...
// i.e. theta_f - theta_g > ~1e-5 rad, 80x the float spacing of the stored
// angle: f is never the view the fold ends on, whatever the order of the
// frames and whatever the layer held before (g is accepted over f or blocks
// it; any third frame that matters beats f by the same margin).  Only the
// COUNT of accepted updates changes, so callers prune only while
// num_observations is known to be zero everywhere (`+= itself` keeps it zero).
AMHIP_HD bool dominated(double tmin_f, double best_tmax) {
  return tmin_f - best_tmax > 1e-5 * (1.0 + tmin_f * tmin_f);
}

// Camera constants of the fast path.  cam[] layout of the device table entry
// that follows the frames: fu, fv, cu, cv, W, H.
struct FoldCam {
  double fu, fv, cu, cv;
  double wcu, hcv;  // W - cu, H - cv
  double kuv;       // box-test margin per unit of eps
  double kround;    // keypoint error per unit of eps / z (fold_finish)
  double uv_abs;    // its absolute part
};

inline FoldCam make_fold_cam(double fu, double fv, double cu, double cv, int width, int height) {
  FoldCam k;
  k.fu = fu;
  k.fv = fv;
  k.cu = cu;
  k.cv = cv;
  k.wcu = (double)width - cu;
  k.hcv = (double)height - cv;
  // Box test, e.g. u >= 0 for z_r > 0:  the reference's
  //   u_ref = fl(fl(fu * fl(x_r * fl(1 / z_r))) + cu) >= 0
  //   <=>  fu x_r (1 + th) + cu z_r >= 0,  |th| <= 3.01 u
  // and ours  s = fu x_a + cu z_a  differs from that by at most
  //   (|fu| + |cu|) eps + 3.01 u |fu| |x_r| + 2 u (|fu x_a| + |cu z_a|)
  //   <= (|fu| + |cu|) (eps + 5.1 u D) <= 1.4 (|fu| + |cu|) eps      (eps >= 16 u D);
  // u < W adds W u z_r.  kuv = 4 * (sum of all of them): ~3x that.
  k.kuv = 4.0 * (std::fabs(fu) + std::fabs(fv) + std::fabs(cu) + std::fabs(cv) + (double)width +
                 (double)height);
  // Keypoint of the winning view from the approximate point (fold_finish):
  //   |kx_a - kx_r| <= (eps / z) (1 + |kx|) + 4u |kx|
  //   |u_a - u_ref| <= |fu| |kx_a - kx_r| + 2u (|fu kx| + |cu|)
  // kround = 4 (|fu| + |fv|) per (eps / z) (1 + |kx| + |ky|); uv_abs = 2^-46
  // (everything) covers the roundings (|u| < W, |v| < H there).
  k.kround = 4.0 * (std::fabs(fu) + std::fabs(fv));
  k.uv_abs = 0x1p-46 * (std::fabs(fu) + std::fabs(fv) + std::fabs(cu) + std::fabs(cv) +
                        (double)width + (double)height);
  return k;
}

// What the reference computes for one pair.  n2 < 0: not visible.
struct ExactView {
  double u, v;   // keypoint (aslam project3)
  double absz;   // |C_p.z|
  double n2;     // C_p.x^2 + C_p.y^2 + C_p.z^2 in Eigen's order
};

// aslam::PinholeCamera::project3 without distortion + the visibility test of
// ortho-backward-grid.cc:164-171 + the terms of the view angle (:173-176).
AMHIP_HD ExactView exact_view_inline(const double* cam, const FramePose& T, double lx, double ly,
                                     double lz) {
  const V3 L = {lx, ly, lz};
  const V3 c = transform_point(T, L);
  const double rz = 1.0 / c.z;
  const double kx = c.x * rz;
  const double ky = c.y * rz;
  ExactView e;
  e.u = cam[0] * kx + cam[2];
  e.v = cam[1] * ky + cam[3];
  const bool in_box = (e.u >= 0.0) && (e.v >= 0.0) && (e.u < cam[4]) && (e.v < cam[5]);
  const double zz = c.z * c.z;
  e.absz = std::fabs(c.z);
  e.n2 = c.x * c.x + c.y * c.y + zz;
  // status not in {POINT_BEHIND_CAMERA, PROJECTION_INVALID} <=> z > 1e-10
  if (!(in_box && c.z > 1e-10)) e.n2 = -1.0;
  return e;
}

// Relative half-width of the band of squared sines inside which the float
// rounding of the stored angle may decide (ortho-backward-grid.cc:180 compares
// the new asin with the FLOAT layer value).  Outside it the order of the
// angles is the order of the squared sines: d(asin)/ds >= 1 and asin(s) <=
// (pi/2) s, so a relative gap of 1e-6 in s is > 10x the 6e-8 rounding of the
// angle.  4e-6 on the approximate squares leaves >= 2.5e-6 on the true ones
// (their relative error is <= 7e-7 in the worst admissible geometry).
constexpr double kSineBand = 4e-6;

// ---- small numeric helpers (device: hardware seed + one refinement step; host: libm) -------
// tools/ubench/precision.hip on gfx950: v_rcp_f64 / v_rsq_f64 are good to 2^-24.4 / 2^-24.2, ONE
// refinement step brings the reciprocal to 2.3e-15 and the square root to 4.2e-15 (relative), a
// second one to 1.1e-16 / 2.3e-16.  fold_finish's error bounds carry slack in the 1e-14 .. 1e-12
// range (uv_abs, da): one step is enough there.
AMHIP_HD double fold_rcp1(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  const double r = __builtin_amdgcn_rcp(x);
  return fma(fma(-x, r, 1.0), r, r);
#else
  return 1.0 / x;
#endif
}

AMHIP_HD double fold_sqrt1(double x) {  // x > 0, normal
#if defined(__HIP_DEVICE_COMPILE__)
  const double y = __builtin_amdgcn_rsq(x);
  const double g = x * y, h = 0.5 * y;
  return fma(g, fma(-h, g, 0.5), g);
#else
  return std::sqrt(x);
#endif
}

AMHIP_HD float fold_rcpf(float x) {  // ~1 ulp (only picks a table entry)
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rcpf(x);
#else
  return 1.0f / x;
#endif
}

AMHIP_HD double fold_fract(double x) {  // x - floor(x), x >= 0
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_fract(x);
#else
  return x - std::floor(x);
#endif
}

// atan(i / 8), i = 0 .. 16, correctly rounded; lives behind the camera in the
// device frame table (doubles 8 .. 24 of the slots that follow the frames).
constexpr int kAtanTabSize = 17;
inline void make_atan_table(double* tab) {
  for (int i = 0; i < kAtanTabSize; ++i) tab[i] = std::atan((double)i / 8.0);
}

// Fold state of one cell.
struct CellFold {
  // Camera-frame point of the best view so far, within eps of the reference's
  // (sin^2 of its angle = bz^2 / |b|^2); for the angle `a` a layer held before
  // this call: (cos a, 0, sin a).
  double bx, by, bz;
  bool redo;     // some decision fell inside a margin: the cell is folded again
                 // in the reference's arithmetic (exact_refold)
  int best_f;    // frame of the best view accepted in THIS call (-1: none)
  int accepted;  // accepted updates (num_observations += itself, :183)
};

// Start from the layer's current angle (0 on a fresh map; the maximum left by
// earlier batches in incremental mode).
AMHIP_HD void fold_init(CellFold* s, float layer_angle) {
  s->redo = false;
  s->by = 0.0;
  if (layer_angle >= 1.5707964f || layer_angle != layer_angle) {
    // no asin exceeds (float)(pi/2), `alpha > NaN` never holds: sin^2 = 1 sends
    // every view that is not clearly lower to the reference's comparison
    s->bx = 0.0;
    s->bz = 1.0;
  } else if (layer_angle > 0.0f) {
    s->bx = cos((double)layer_angle);
    s->bz = sin((double)layer_angle);
  } else {
    s->bx = 1.0;  // every visible view wins (alpha > 0)
    s->bz = 0.0;
  }
  s->best_f = -1;
  s->accepted = 0;
}

// One frame folded into one cell.  (cx, cy, cz) = M (L - p) from the fma chain
// over FrameFast; epsL = 2^-48 V1 with V1 >= |L|_1 (the D part of eps is added
// here from the point itself).
// `valid` false: the cell has no elevation (NaN): never visible.
// The steps follow ortho-backward-grid.cc:164-208.  A pair whose visibility or
// whose comparison with the running best falls inside a margin (or involves
// non-finite numbers) marks the cell `redo`; nothing else happens for it then
// -- exact_refold() replays the cell later, out of the hot loop.
// Written with predicates rather than early exits: almost every pair that
// survives the tile's frame cull is visible, and the wave executes all of it
// anyway.
AMHIP_HD void fold_pair(CellFold* s, int f, const FoldCam& k, bool valid, double cx, double cy,
                        double cz, double epsL) {
  const double eps = fma(0x1p-49, fabs(cx) + fabs(cy) + fabs(cz), epsL);
  // z_r > 1e-10 is certain above zthr, where the relative error of cz is <= 2^-24
  const double zthr = fma(0x1p+24, eps, 1e-10);
  const double muv = k.kuv * eps;
  // box test without the division: for z > 0
  //   u >= 0 <=> fu x + cu z >= 0,   u < W <=> (W - cu) z - fu x > 0   (same for v)
  const double a = k.fu * cx;
  const double b = k.fv * cy;
  const double d1 = fma(k.cu, cz, a);
  const double d2 = fma(k.wcu, cz, -a);
  const double e1 = fma(k.cv, cz, b);
  const double e2 = fma(k.hcv, cz, -b);
  // (all four are finite whenever zok: a non-finite elevation makes eps, hence
  // zthr, infinite or NaN; poses with non-finite entries never reach the fast path)
  const double g = fmin(fmin(d1, d2), fmin(e1, e2));
  const double zz = cz * cz;
  const double n2 = fma(cx, cx, fma(cy, cy, zz));
  // (below -zthr it is certainly false)
  const bool zok = cz > zthr;
  const bool vis = valid & zok & (g > muv);
  const bool invis = !valid | (cz < -zthr) | (zok & (g < -muv));
  const double zzb = s->bz * s->bz;
  const double n2b = fma(s->bx, s->bx, fma(s->by, s->by, zzb));
  const double lhs = zz * n2b;
  const double rhs = zzb * n2;
  const bool accept = vis & (lhs > rhs * (1.0 + kSineBand));
  const bool tie = vis & !accept & !(lhs < rhs * (1.0 - kSineBand));
  if (!(vis | invis) | tie) s->redo = true;
  if (accept) {
    s->bx = cx;
    s->by = cy;
    s->bz = cz;
    s->best_f = f;
    s->accepted++;
  }
}

// After the last frame: the winner's keypoint (:186-193) and its angle as the
// layer stores it (:181), from the approximate point when that provably gives
// the reference's result:
//   keypoint  round(u) is safe when u is farther from k + 1/2 than its error
//   angle     (float)alpha is safe when alpha is farther from the float
//             rounding boundaries than its error:  |alpha_a - alpha_ref| <=
//             eps / z (direction of the ray) + 2u / tan(theta) (the rounding of
//             |z| / ||p|| under the reference's asin) + 1e-14 (everything else)
// epsL: as in fold_pair.  Returns
//   kFoldNone    no view was accepted: the cell keeps its values
//   kFoldDone    *kp_x, *kp_y, *angle are the reference's
//   kFoldFinish  the winner (s->best_f) is right, keypoint / angle need
//                exact_finish()
//   kFoldRedo    the whole cell needs exact_refold()
enum { kFoldNone = 0, kFoldDone = 1, kFoldFinish = 2, kFoldRedo = 3 };

AMHIP_HD int fold_finish(const CellFold* s, const FoldCam& k, const double* atan_tab,
                         double epsL, int width, int height, int* kp_x, int* kp_y,
                         float* angle) {
  // Written without early exits: the four cells of a lane are finished side by side, and a
  // branch per cell would keep their dependent chains from overlapping; a cell that is to be
  // replayed or that no view was accepted for runs the arithmetic on the point (1, 0, 1) and
  // drops the result.
  const bool live = !s->redo && s->accepted != 0;
  const double bx = live ? s->bx : 1.0, by = live ? s->by : 0.0, bz = live ? s->bz : 1.0;
  // tan(theta) = rho / z, theta = angle between the optical axis and the ray; the stored angle is
  // pi/2 - atan(tan theta) with atan from the table of atan(i / 8) plus a 6-term series in
  //   y = (r - c) / (1 + r c) = (rho - c z) / (z + rho c),   c = i / 8 nearest to r,
  // so that ONE reciprocal, of z (z + rho c), serves the series AND the keypoint's 1 / z.
  // (rho = 0, a view exactly along the axis: NaN from here on, `in_range` false.)
  const double rho = fold_sqrt1(fma(bx, bx, by * by));
  // which table entry: single precision is plenty (|y| <= 1/16 + 1e-6 either way)
  const float rf = (float)rho * fold_rcpf((float)bz);
  const bool in_range = rf >= 1.001e-3f && rf <= 1.999f;
  const float fi = in_range ? rintf(rf * 8.0f) : 8.0f;
  const double c = (double)(fi * 0.125f);
  const double num = fma(-c, bz, rho);
  const double den = fma(c, rho, bz);
  const double rc = fold_rcp1(bz * den);
  const double rcz = den * rc;  // 1 / z to within 2.6e-15 (relative)
  const double y = num * (bz * rc);
  const double kx = bx * rcz;
  const double ky = by * rcz;
  const double u = fma(k.fu, kx, k.cu);
  const double v = fma(k.fv, ky, k.cv);
  const double eps = fma(0x1p-49, fabs(bx) + fabs(by) + fabs(bz), epsL);
  const double ez = eps * rcz;  // the direction of the ray is known to within this
  // (uv_abs = 128 u (|fu| + ... + W + H) also covers the 24 u |fu kx| of the one-step reciprocal)
  const double duv = fma(k.kround * ez, 1.0 + fabs(kx) + fabs(ky), k.uv_abs);
  // std::round of a non-negative number = floor(x + 1/2) away from the ties; a winner is inside
  // the image box, so x + 1/2 > 0 and the conversion's truncation is the floor
  const double tu = u + 0.5, tv = v + 0.5;
  const double ru = fold_fract(tu), rv = fold_fract(tv);
  const double hi = 1.0 - duv;
  bool ok = in_range & (ru > duv) & (ru < hi) & (rv > duv) & (rv < hi);
  const int kx_i = (int)tu, ky_i = (int)tv;
  *kp_y = ky_i < height - 1 ? ky_i : height - 1;
  *kp_x = kx_i < width - 1 ? kx_i : width - 1;
  // atan(r) - atan(c) = atan(y): truncation y^13 / 13 < 2e-17; what the one-step square root and
  // reciprocal leave in y is < 3e-15 (absolute), the table entry is correctly rounded
  const double y2 = y * y;
  double q = fma(y2, -1.0 / 11.0, 1.0 / 9.0);
  q = fma(y2, q, -1.0 / 7.0);
  q = fma(y2, q, 1.0 / 5.0);
  q = fma(y2, q, -1.0 / 3.0);
  const double theta = atan_tab[(int)fi] + fma(y * y2, q, y);
  const double alpha = (1.5707963267948966 - theta) + 6.123233995736766e-17;
  // |alpha - alpha_ref| <= 2 eps / z (direction of the ray) + 2u / tan(theta) (the rounding of
  // |z| / ||p|| under the reference's asin: <= 2^-49 / r, r >= 1e-3) + 2e-12 (everything else,
  // the 5e-15 of this routine's own arithmetic included)
  const double da = 2.0 * ez + 2e-12;
  const float fl = (float)alpha;
  const double e = alpha - (double)fl;
  // half the spacing of the floats around fl: 2^(exponent - 24), straight from fl's bits
  // (alpha in [0.46, 1.57]: 2^-24 in [1, 2), 2^-25 in [0.5, 1), 2^-26 below)
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned flb = __float_as_uint(fl);
  const double half = __hiloint2double((int)(((flb >> 23) + 872u) << 20), 0);
#else
  unsigned flb;
  std::memcpy(&flb, &fl, 4);
  const double half = std::ldexp(1.0, (int)(flb >> 23) - 127 - 24);
#endif
  // (a float with an all-zero mantissa sits where the spacing halves below it: stay clear)
  ok = ok & (fabs(e) < half - da) & ((flb & 0x7FFFFFu) != 0u) & (fl > 0.26f);
  *angle = fl;
  return s->redo ? kFoldRedo : (s->accepted == 0 ? kFoldNone : (ok ? kFoldDone : kFoldFinish));
}

// ---- the reference's arithmetic, for the cells the margins could not settle ----
struct FoldResult {
  float best;   // elevation_angle
  int best_f;   // observation_index (-1: no view accepted, the cell keeps its values)
  int accepted;
  int kp_x, kp_y;
  int bad_alpha;  // CHECK(alpha > 0.0) would have fired
};

AMHIP_HD double exact_angle(double absz, double n2) {
  const double norm = sqrt(n2);
  return asin(absz / norm);
}

// keypoint and stored angle of the known winner `pose`
AMHIP_HD FoldResult exact_finish(const double* cam, const FramePose& pose, double lx, double ly,
                                 double lz, int best_f, int accepted) {
  FoldResult r;
  const ExactView e = exact_view_inline(cam, pose, lx, ly, lz);
  const double alpha = exact_angle(e.absz, e.n2);
  r.bad_alpha = !(alpha > 0.0);
  r.best = (float)alpha;
  r.best_f = best_f;
  r.accepted = accepted;
  const int ky = (int)round(e.v), kx = (int)round(e.u);
  const int w = (int)cam[4], h = (int)cam[5];
  r.kp_y = ky < h - 1 ? ky : h - 1;
  r.kp_x = kx < w - 1 ? kx : w - 1;
  return r;
}

// ortho-backward-grid.cc:144-208 for one cell, literally, over the frames
// cand[0 .. n) (ascending; cand == nullptr: frames 0 .. n-1).  layer_angle:
// what the elevation_angle layer held before this call.
AMHIP_HD FoldResult exact_refold(const double* cam, const FramePose* poses, const int* cand, int n,
                                 double lx, double ly, double lz, float layer_angle) {
  FoldResult r;
  r.best = layer_angle;
  r.best_f = -1;
  r.accepted = 0;
  r.kp_x = r.kp_y = 0;
  r.bad_alpha = 0;
  const int w = (int)cam[4], h = (int)cam[5];
  for (int k = 0; k < n; ++k) {
    const int f = cand ? cand[k] : k;
    const ExactView e = exact_view_inline(cam, poses[f], lx, ly, lz);
    if (e.n2 < 0.0) continue;
    const double alpha = exact_angle(e.absz, e.n2);
    if (!(alpha > 0.0)) r.bad_alpha = 1;
    if (alpha > (double)r.best) {
      r.best = (float)alpha;
      r.best_f = f;
      r.accepted++;
      const int ky = (int)round(e.v), kx = (int)round(e.u);
      r.kp_y = ky < h - 1 ? ky : h - 1;
      r.kp_x = kx < w - 1 ? kx : w - 1;
    }
  }
  return r;
}

}  // namespace amhip

#endif  // AMHIP_ORTHO_FOLD_H_
