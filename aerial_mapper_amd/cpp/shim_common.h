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

inline amhip_grid_desc describe(const grid_map::GridMap& map) {
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

// (Re)create the context when the map's geometry is not the one it was made for.
inline void ensure_context(amhip_ctx** ctx, int* rows, int* cols, double* geom,
                           const grid_map::GridMap& map, const char* where) {
  const amhip_grid_desc g = describe(map);
  if (*ctx && *rows == g.rows && *cols == g.cols && geom[0] == g.resolution &&
      geom[1] == g.pos_x && geom[2] == g.pos_y && geom[3] == g.length_x)
    return;
  if (*ctx) amhip_ctx_destroy(*ctx);
  *ctx = nullptr;
  int device = 0;
  if (const char* env = std::getenv("AERIAL_MAPPER_HIP_DEVICE")) device = std::atoi(env);
  check_status(amhip_ctx_create(&g, device, ctx), where);
  *rows = g.rows;
  *cols = g.cols;
  geom[0] = g.resolution;
  geom[1] = g.pos_x;
  geom[2] = g.pos_y;
  geom[3] = g.length_x;
}

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
  amhip_camera c;
  const Eigen::VectorXd& p = camera.getParameters();  // fu, fv, cu, cv
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
    case aslam::Distortion::Type::kRadTan:
      c.distortion = AMHIP_DIST_RADTAN;
      break;
    case aslam::Distortion::Type::kEquidistant:
      c.distortion = AMHIP_DIST_EQUIDISTANT;
      break;
    default:
      c.distortion = AMHIP_DIST_NONE;
      break;
  }
  if (c.distortion != AMHIP_DIST_NONE) {
    const Eigen::VectorXd& dp = d.getParameters();
    for (int k = 0; k < 4 && k < static_cast<int>(dp.size()); ++k) c.dist[k] = dp(k);
  }
  return c;
}


}  // namespace amhip_shim

#endif  // AERIAL_MAPPER_HIP_SHIM_COMMON_H_
