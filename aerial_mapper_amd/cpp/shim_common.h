// Helpers shared by the drop-in classes: geometry extraction from a
// grid_map::GridMap, pose / camera description and the glog-CHECK-like failure path.
#ifndef AERIAL_MAPPER_HIP_SHIM_COMMON_H_
#define AERIAL_MAPPER_HIP_SHIM_COMMON_H_

#include <cstdio>
#include <cstdlib>

#include "aerial-mapper-deps.h"
#include "aerial_mapper_hip.h"

namespace amhip_shim {

// The reference aborts through glog CHECK / LOG(FATAL); so does the shim.
[[noreturn]] inline void fatal(const char* where, const char* what) {
  std::fprintf(stderr, "[aerial_mapper_hip] FATAL %s: %s\n", where, what);
  std::abort();
}

inline void check_status(int status, const char* where) {
  if (status != AMHIP_OK) fatal(where, amhip_last_error());
}

// AERIAL_MAPPER_HIP_DEVICE: the device the drop-in classes create their contexts on (default 0)
int default_device();

inline amhip_grid_desc describe(const grid_map::GridMap& map) {
  // the raw matrices are addressed from (0, 0): a map that was move()d (circular buffer start
  // index != 0) would be mis-addressed -- the hot path never moves the map, refuse if it was
  if (map.getStartIndex()(0) != 0 || map.getStartIndex()(1) != 0)
    fatal("describe(GridMap)", "map.getStartIndex() != (0, 0): moved maps are not supported");
  amhip_grid_desc g;
  g.rows = map.getSize()(0);
  g.cols = map.getSize()(1);
  g.resolution = map.getResolution();
  g.length_x = map.getLength()(0);
  g.length_y = map.getLength()(1);
  g.pos_x = map.getPosition()(0);
  g.pos_y = map.getPosition()(1);
  return g;
}

// (Re)create the context when the map's geometry is not the one it was made for
// (classes that keep a context of their own: OrthoFromPcl).
inline void ensure_context(amhip_ctx** ctx, int* rows, int* cols, double* geom,
                           const grid_map::GridMap& map, const char* where) {
  const amhip_grid_desc g = describe(map);
  if (*ctx && *rows == g.rows && *cols == g.cols && geom[0] == g.resolution &&
      geom[1] == g.pos_x && geom[2] == g.pos_y && geom[3] == g.length_x)
    return;
  if (*ctx) amhip_ctx_destroy(*ctx);
  *ctx = nullptr;
  check_status(amhip_ctx_create(&g, default_device(), ctx), where);
  *rows = g.rows;
  *cols = g.cols;
  geom[0] = g.resolution;
  geom[1] = g.pos_x;
  geom[2] = g.pos_y;
  geom[3] = g.length_x;
}

// One amhip_session per grid_map::GridMap (keyed by the map's address and geometry), shared
// by every dsm::Dsm / ortho::OrthoBackwardGrid working on that map: the layers stay on the
// device(s) between Dsm::process and OrthoBackwardGrid::process, and with
// AERIAL_MAPPER_HIP_DEVICES=0,1,... the map is cut into one window per listed device.
// acquire() returns the session of `map` (creating or re-creating it when the geometry
// changed) and releases `*held`; release() drops a reference (the last one destroys it).
amhip_session* acquire_session(const grid_map::GridMap& map, amhip_session* held, const char* where);
void release_session(amhip_session* held);

// kindr::minimal::QuatTransformation -> tx,ty,tz,qw,qx,qy,qz
inline void pose_to7(const kindr::minimal::QuatTransformation& T, double* o) {
  const Eigen::Vector3d& t = T.getPosition();
  const Eigen::Quaterniond& q = T.getRotation().toImplementation();
  o[0] = t(0);
  o[1] = t(1);
  o[2] = t(2);
  o[3] = q.w();
  o[4] = q.x();
  o[5] = q.y();
  o[6] = q.z();
}

inline amhip_camera describe_camera(const aslam::Camera& camera) {
  // The reference calls the virtual camera.project3() (ortho-backward-grid.cc:160-161), so it
  // works with every aslam camera; the kernels implement the pinhole projection with no /
  // radial-tangential / equidistant distortion.  Anything else must fail loudly -- treated as
  // an undistorted pinhole it would give a silently wrong mosaic.
  if (camera.getType() != aslam::Camera::Type::kPinhole)
    fatal("describe_camera", "only aslam::PinholeCamera is implemented on the GPU "
                             "(parameters fu, fv, cu, cv); UnifiedProjection and others are not");
  amhip_camera c;
  const Eigen::VectorXd& p = camera.getParameters();  // fu, fv, cu, cv
  if (p.size() != 4) fatal("describe_camera", "pinhole camera with other than 4 parameters");
  c.fu = p(0);
  c.fv = p(1);
  c.cu = p(2);
  c.cv = p(3);
  c.width = static_cast<int32_t>(camera.imageWidth());
  c.height = static_cast<int32_t>(camera.imageHeight());
  c._pad = 0;
  for (int k = 0; k < 4; ++k) c.dist[k] = 0.0;
  const aslam::Distortion& d = camera.getDistortion();
  switch (d.getType()) {
    case aslam::Distortion::Type::kNoDistortion:
      c.distortion = AMHIP_DIST_NONE;
      break;
    case aslam::Distortion::Type::kRadTan:
      c.distortion = AMHIP_DIST_RADTAN;
      break;
    case aslam::Distortion::Type::kEquidistant:
      c.distortion = AMHIP_DIST_EQUIDISTANT;
      break;
    default:
      fatal("describe_camera", "distortion model not implemented on the GPU (none, radtan and "
                               "equidistant are; e.g. fisheye / FOV is not)");
  }
  if (c.distortion != AMHIP_DIST_NONE) {
    const Eigen::VectorXd& dp = d.getParameters();
    if (dp.size() != 4) fatal("describe_camera", "distortion with other than 4 parameters");
    for (int k = 0; k < 4; ++k) c.dist[k] = dp(k);
  }
  return c;
}

}  // namespace amhip_shim

#endif  // AERIAL_MAPPER_HIP_SHIM_COMMON_H_
