// amhip_ortho_fold.h -- arithmetic of one (cell, frame) pair of the backward-grid
// fold, ortho-backward-grid.cc:144-208, in two forms:
//
//   exact_view()   the reference's own operations, one for one (minkindr
//                  transform in Eigen's order, aslam pinhole project3, the
//                  squares under asin's sqrt), double, no contraction;
//   fast_*         a bounded-error evaluation (pose as a 3x4 matrix, box test
//                  without the division) whose every DECISION carries a margin
//                  that covers the worst-case distance to the reference's
//                  doubles; a pair whose decision falls inside the margin is
//                  handed to exact_view().  Outcomes are therefore the
//                  reference's, the cost is ~30 instead of ~85 FP64 operations
//                  for almost every pair.
//
// Host + device: the kernel (amhip_ortho.hip) and the CPU emulation the unit
// tests drive (tests/cpp/ortho_fold_emul.cc) compile the same functions.
//
// Error budget (u = 2^-53, |q| = 1 within 1e-6, S = |L|_1, mag = S + |t|_1):
//   reference   c_r = fl(q (x) L + t): 33 operations, |c_r - c*| <= 49 u mag
//   here        c_a = fma chain over M(q), |M - M*| <= 8u per entry:
//               |c_a - c*| <= 11 u mag
//   => |c_a - c_r| <= 60 u mag; every margin below assumes eps = 128 u mag
//      = 2^-46 mag and then doubles again.
#ifndef AMHIP_ORTHO_FOLD_H_
#define AMHIP_ORTHO_FOLD_H_

#include <cmath>

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
// (which is what that formula computes for ANY quaternion, unit or not).
struct FrameFast {
  double m[9];  // row-major
  double t[3];
  double tmag;  // |tx| + |ty| + |tz|
  double _pad[3];
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
  const double w = T.qw, x = T.qx, y = T.qy, z = T.qz;
  o->m[0] = 1.0 - 2.0 * (y * y + z * z);
  o->m[1] = 2.0 * (x * y - w * z);
  o->m[2] = 2.0 * (x * z + w * y);
  o->m[3] = 2.0 * (x * y + w * z);
  o->m[4] = 1.0 - 2.0 * (x * x + z * z);
  o->m[5] = 2.0 * (y * z - w * x);
  o->m[6] = 2.0 * (x * z - w * y);
  o->m[7] = 2.0 * (y * z + w * x);
  o->m[8] = 1.0 - 2.0 * (x * x + y * y);
  o->t[0] = T.tx;
  o->t[1] = T.ty;
  o->t[2] = T.tz;
  o->tmag = std::fabs(T.tx) + std::fabs(T.ty) + std::fabs(T.tz);
  o->_pad[0] = o->_pad[1] = o->_pad[2] = 0.0;
  const double n = w * w + x * x + y * y + z * z;
  // the error budget above assumes a unit quaternion and finite numbers
  return std::fabs(n - 1.0) < 1e-6 && o->tmag < 1e300;
}

// Camera constants of the fast path.  cam[] layout of the device table entry
// that follows the frames: fu, fv, cu, cv, W, H.
struct FoldCam {
  double fu, fv, cu, cv;
  double wcu, hcv;  // W - cu, H - cv
  double kuv;       // box-test margin per unit of mag
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
  //   <= (|fu| + |cu|) (2^-46 + 2^-50) mag ;
  // u < W adds W u z_r.  kuv = 2^-44 * (sum of all of them): > 4x that.
  k.kuv = 0x1p-44 * (std::fabs(fu) + std::fabs(fv) + std::fabs(cu) + std::fabs(cv) +
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

// Fold state of one cell.  sin^2 of the best view so far = zb2 / n2b.
struct CellFold {
  double zb2, n2b;
  float best;    // its angle as the layer stores it; valid iff have_f
  bool have_f;
  int best_f;    // frame of the best view accepted in THIS call (-1: none)
  int accepted;  // accepted updates (num_observations += itself, :183)
};

// Start from the layer's current angle (0 on a fresh map; the maximum left by
// earlier batches in incremental mode).
AMHIP_HD void fold_init(CellFold* s, float layer_angle) {
  s->best = layer_angle;
  s->have_f = true;
  s->n2b = 1.0;
  if (layer_angle >= 1.5707964f) {
    s->zb2 = HUGE_VAL;  // no asin can exceed (float)(pi/2): nothing is accepted
  } else if (layer_angle > 0.0f) {
    const double sn = sin((double)layer_angle);
    s->zb2 = sn * sn;
  } else if (layer_angle == layer_angle) {
    s->zb2 = 0.0;       // every visible view wins (alpha > 0)
  } else {
    s->zb2 = HUGE_VAL;  // NaN in the layer: `alpha > NaN` never holds
  }
  s->best_f = -1;
  s->accepted = 0;
}

// One frame folded into one cell.  (cx, cy, cz) is the camera-frame point from
// the fma chain over FrameFast, mag >= |L|_1 + |t|_1 of the pair,
//   zthr = 1e-10 + 2^-22 mag,  muv = kuv * mag;
// `valid` false: the cell has no (finite-or-not) elevation at all -- NaN, never
// visible.  `ex` supplies the reference's arithmetic for this cell:
//   ExactView ex.view(int frame);   double ex.angle(double absz, double n2);
// The steps follow ortho-backward-grid.cc:164-208; *bad_alpha <=> CHECK(alpha > 0).
// Written with predicates rather than early exits: almost every pair that
// survives the tile's frame cull is visible, and the wave executes all of it
// anyway.
template <class Exact>
AMHIP_HD void fold_pair(CellFold* s, int f, const FoldCam& k, bool valid, double cx, double cy,
                        double cz, double zthr, double muv, const Exact& ex, bool* bad_alpha) {
  // box test without the division: for z > 0
  //   u >= 0 <=> fu x + cu z >= 0,   u < W <=> (W - cu) z - fu x > 0   (same for v)
  const double a = k.fu * cx;
  const double b = k.fv * cy;
  const double d1 = fma(k.cu, cz, a);
  const double d2 = fma(k.wcu, cz, -a);
  const double e1 = fma(k.cv, cz, b);
  const double e2 = fma(k.hcv, cz, -b);
  // (all four are finite whenever zok: a non-finite elevation makes mag, hence
  // zthr, infinite; poses with non-finite entries never reach the fast path)
  const double g = fmin(fmin(d1, d2), fmin(e1, e2));
  double zz = cz * cz;
  double n2 = fma(cx, cx, fma(cy, cy, zz));
  // z_r > 1e-10 is certain above zthr (eps = 2^-46 mag) and the relative error
  // of cz there is <= 2^-24; below -zthr it is certainly false
  const bool zok = cz > zthr;
  bool vis = valid & zok & (g > muv);
  const bool invis = !valid | (cz < -zthr) | (zok & (g < -muv));
  ExactView e;
  bool have_e = false;
  if (!(vis | invis)) {
    // inside a margin (or NaN somewhere): the reference's arithmetic decides
    e = ex.view(f);
    have_e = true;
    vis = e.n2 >= 0.0;
    zz = e.absz * e.absz;
    n2 = e.n2;
  }
  const double lhs = zz * s->n2b;
  const double rhs = s->zb2 * n2;
  bool accept = vis & (lhs > rhs * (1.0 + kSineBand));
  const bool tie = vis & !accept & !(lhs < rhs * (1.0 - kSineBand));
  bool exact = false;
  if (tie) {
    // near tie: compare the angles like ortho-backward-grid.cc:180 does
    if (!s->have_f) {
      const ExactView w = ex.view(s->best_f);
      s->best = (float)ex.angle(w.absz, w.n2);
      s->have_f = true;
    }
    if (!have_e) e = ex.view(f);
    const double alpha = ex.angle(e.absz, e.n2);
    if (!(alpha > 0.0)) *bad_alpha = true;
    if (alpha > (double)s->best) {
      s->best = (float)alpha;
      zz = e.absz * e.absz;
      n2 = e.n2;
      accept = true;
      exact = true;
    }
  }
  if (accept) {
    s->have_f = exact;
    s->zb2 = zz;
    s->n2b = n2;
    s->best_f = f;
    s->accepted++;
  }
}

// After the last frame: the winner's angle and keypoint, evaluated exactly like
// the reference does (:181, :186-193).  false: no view was accepted.
template <class Exact>
AMHIP_HD bool fold_finish(CellFold* s, const Exact& ex, int width, int height, int* kp_x,
                          int* kp_y, bool* bad_alpha) {
  if (s->accepted == 0) return false;
  const ExactView e = ex.view(s->best_f);
  if (!s->have_f) {
    const double alpha = ex.angle(e.absz, e.n2);
    if (!(alpha > 0.0)) *bad_alpha = true;
    s->best = (float)alpha;
    s->have_f = true;
  }
  const int ky = (int)round(e.v);
  const int kx = (int)round(e.u);
  *kp_y = ky < height - 1 ? ky : height - 1;
  *kp_x = kx < width - 1 ? kx : width - 1;
  return true;
}

}  // namespace amhip

#endif  // AMHIP_ORTHO_FOLD_H_
