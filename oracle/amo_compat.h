/*
 * oracle/amo_compat.h -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Stand-ins for the three EXTERNAL, un-vendored dependencies whose arithmetic
 * sits on the hot path (grid_map_core, minkindr, aslam_cv2).  None of them is
 * present in /root/reference (install/dependencies_https.rosinstall:1-21
 * lists git URLs without version pins) nor in this container, so the formulas
 * below are the DEFINITIONS the oracle adopts.
 *
 *   PARITY UNPINNED for everything in this file: the reference ships no
 *   tests, golden vectors or fixtures that would pin these formulas.
 *
 * All arithmetic is IEEE double without fused multiply-add (the reference is
 * built for baseline x86-64; this directory is compiled with
 * -ffp-contract=off), one operation per C++ operator, in the order written.
 */
#ifndef AMO_COMPAT_H_
#define AMO_COMPAT_H_

#include <cmath>
#include <cstdint>
#include <cstring>

#include "amo_types.h"

namespace amo {

// ---------------------------------------------------------------------------
// grid_map_core  (call sites: dsm.cc:28,59,116,125;
//                 ortho-backward-grid.cc:48-58,145-150;
//                 aerial-mapper-grid-map.cc:30-33)
// ---------------------------------------------------------------------------

// GridMap::setGeometry(Length, resolution, Position):
//   size = (int)round(length / resolution); length := size * resolution.
inline amo_grid make_grid(double length_x, double length_y, double resolution,
                          double pos_x, double pos_y) {
  amo_grid g;
  g.rows = static_cast<int>(std::round(length_x / resolution));
  g.cols = static_cast<int>(std::round(length_y / resolution));
  g.resolution = resolution;
  g.length_x = static_cast<double>(g.rows) * resolution;
  g.length_y = static_cast<double>(g.cols) * resolution;
  g.pos_x = pos_x;
  g.pos_y = pos_y;
  return g;
}

// GridMap::getPosition(index, position) with startIndex == 0:
//   position = (mapPosition + vectorToFirstCell) + resolution * (-index)
//   vectorToFirstCell = 0.5 * length - 0.5 * resolution
inline void cell_position(const amo_grid& g, int i, int j, double* x,
                          double* y) {
  const double off_x = 0.5 * g.length_x - 0.5 * g.resolution;
  const double off_y = 0.5 * g.length_y - 0.5 * g.resolution;
  const double base_x = g.pos_x + off_x;
  const double base_y = g.pos_y + off_y;
  *x = base_x + g.resolution * (-static_cast<double>(i));
  *y = base_y + g.resolution * (-static_cast<double>(j));
}

// GridMapIterator: linear index -> (i, j), column-major.
inline void linear_to_index(const amo_grid& g, size_t lin, int* i, int* j) {
  *i = static_cast<int>(lin % static_cast<size_t>(g.rows));
  *j = static_cast<int>(lin / static_cast<size_t>(g.rows));
}

// grid_map::colorVectorToValue(Vector3f, float&):
//   Vector3i t = (c * 255.0).cast<int>(); value bits = t0<<16 | t1<<8 | t2.
inline float color_vector_to_value(float c0, float c1, float c2) {
  const int t0 = static_cast<int>(c0 * 255.0f);
  const int t1 = static_cast<int>(c1 * 255.0f);
  const int t2 = static_cast<int>(c2 * 255.0f);
  const uint32_t bits = (static_cast<uint32_t>(t0) << 16) |
                        (static_cast<uint32_t>(t1) << 8) |
                        static_cast<uint32_t>(t2);
  float out;
  std::memcpy(&out, &bits, sizeof(out));
  return out;
}

// ---------------------------------------------------------------------------
// minkindr  kindr::minimal::QuatTransformation
// (call sites: ortho-backward-grid.cc:157-158,232)
// Hamilton unit quaternion (w,x,y,z) + translation.
// ---------------------------------------------------------------------------
struct Vec3 {
  double x, y, z;
};

inline Vec3 cross(const Vec3& a, const Vec3& b) {
  Vec3 r;
  r.x = a.y * b.z - a.z * b.y;
  r.y = a.z * b.x - a.x * b.z;
  r.z = a.x * b.y - a.y * b.x;
  return r;
}

struct Quat {
  double w, x, y, z;
};

// Eigen::Quaternion::_transformVector:
//   uv = q.vec x v; uv += uv; return v + q.w * uv + q.vec x uv
inline Vec3 rotate(const Quat& q, const Vec3& v) {
  const Vec3 qv = {q.x, q.y, q.z};
  Vec3 uv = cross(qv, v);
  uv.x = uv.x + uv.x;
  uv.y = uv.y + uv.y;
  uv.z = uv.z + uv.z;
  const Vec3 c2 = cross(qv, uv);
  Vec3 r;
  r.x = (v.x + q.w * uv.x) + c2.x;
  r.y = (v.y + q.w * uv.y) + c2.y;
  r.z = (v.z + q.w * uv.z) + c2.z;
  return r;
}

inline Quat conjugate(const Quat& q) {
  Quat r = {q.w, -q.x, -q.y, -q.z};
  return r;
}

// Eigen quaternion product (Hamilton).
inline Quat qmul(const Quat& a, const Quat& b) {
  Quat r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return r;
}

struct Pose {
  Quat q;   // q_A_B
  Vec3 t;   // A_t_A_B
};

// Layout used across the C boundary: tx,ty,tz,qw,qx,qy,qz
// (the reference's pose text format, aerial-mapper-io.cc:103-121).
inline Pose pose_from7(const double* p) {
  Pose r;
  r.t.x = p[0];
  r.t.y = p[1];
  r.t.z = p[2];
  r.q.w = p[3];
  r.q.x = p[4];
  r.q.y = p[5];
  r.q.z = p[6];
  return r;
}

inline void pose_to7(const Pose& p, double* o) {
  o[0] = p.t.x;
  o[1] = p.t.y;
  o[2] = p.t.z;
  o[3] = p.q.w;
  o[4] = p.q.x;
  o[5] = p.q.y;
  o[6] = p.q.z;
}

// transform(p) = q (x) p + t
inline Vec3 transform(const Pose& T, const Vec3& p) {
  const Vec3 r = rotate(T.q, p);
  Vec3 o = {r.x + T.t.x, r.y + T.t.y, r.z + T.t.z};
  return o;
}

// inverse() = (q*, -(q* (x) t))
inline Pose inverse(const Pose& T) {
  Pose r;
  r.q = conjugate(T.q);
  const Vec3 rt = rotate(r.q, T.t);
  r.t.x = -rt.x;
  r.t.y = -rt.y;
  r.t.z = -rt.z;
  return r;
}

// A * B = (qA qB, tA + qA (x) tB)
inline Pose compose(const Pose& A, const Pose& B) {
  Pose r;
  r.q = qmul(A.q, B.q);
  const Vec3 rt = rotate(A.q, B.t);
  r.t.x = A.t.x + rt.x;
  r.t.y = A.t.y + rt.y;
  r.t.z = A.t.z + rt.z;
  return r;
}

// ---------------------------------------------------------------------------
// aslam_cv2  aslam::PinholeCamera::project3 + ProjectionResult
// (call sites: ortho-backward-grid.cc:160-171,186-193)
// ---------------------------------------------------------------------------
enum ProjectionStatus {
  KEYPOINT_VISIBLE = 0,
  KEYPOINT_OUTSIDE_IMAGE_BOX = 1,
  POINT_BEHIND_CAMERA = 2,
  PROJECTION_INVALID = 3
};

static const double kMinimumDepth = 1e-10;

inline void distort(const amo_camera& c, double* px, double* py) {
  double& x = *px;
  double& y = *py;
  if (c.distortion == AMO_DIST_RADTAN) {
    const double k1 = c.dist[0], k2 = c.dist[1], p1 = c.dist[2],
                 p2 = c.dist[3];
    const double mx2 = x * x;
    const double my2 = y * y;
    const double mxy = x * y;
    const double rho2 = mx2 + my2;
    const double rad = k1 * rho2 + k2 * rho2 * rho2;
    const double nx = x + (x * rad + 2.0 * p1 * mxy + p2 * (rho2 + 2.0 * mx2));
    const double ny = y + (y * rad + 2.0 * p2 * mxy + p1 * (rho2 + 2.0 * my2));
    x = nx;
    y = ny;
  } else if (c.distortion == AMO_DIST_EQUIDISTANT) {
    const double r = std::sqrt(x * x + y * y);
    const double theta = std::atan(r);
    const double th2 = theta * theta;
    const double th4 = th2 * th2;
    const double th6 = th4 * th2;
    const double th8 = th4 * th4;
    const double thetad =
        theta * (1.0 + c.dist[0] * th2 + c.dist[1] * th4 + c.dist[2] * th6 +
                 c.dist[3] * th8);
    const double scaling = (r > 1e-8) ? thetad / r : 1.0;
    x = x * scaling;
    y = y * scaling;
  }
}

inline ProjectionStatus project3(const amo_camera& c, const Vec3& p, double* u,
                                 double* v) {
  const double rz = 1.0 / p.z;
  double kx = p.x * rz;
  double ky = p.y * rz;
  distort(c, &kx, &ky);
  *u = c.fu * kx + c.cu;
  *v = c.fv * ky + c.cv;
  const bool in_box = (*u >= 0.0) && (*v >= 0.0) &&
                      (*u < static_cast<double>(c.width)) &&
                      (*v < static_cast<double>(c.height));
  if (in_box && (p.z > kMinimumDepth)) return KEYPOINT_VISIBLE;
  if (!in_box && (p.z > kMinimumDepth)) return KEYPOINT_OUTSIDE_IMAGE_BOX;
  if (p.z < 0.0) return POINT_BEHIND_CAMERA;
  return PROJECTION_INVALID;
}

}  // namespace amo

#endif  // AMO_COMPAT_H_
