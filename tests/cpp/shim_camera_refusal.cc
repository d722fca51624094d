// shim_camera_refusal.cc -- the drop-in must REFUSE camera models the GPU path does not
// implement (the reference calls the virtual Camera::project3, ortho-backward-grid.cc:160-161,
// and handles them; mapping them to an undistorted pinhole would give a silently wrong mosaic).
// argv[1] selects the camera; the process is expected to abort() for every case but "ok".
#include <cstdio>
#include <cstring>

#include "../../aerial_mapper_amd/cpp/shim_common.h"

int main(int argc, char** argv) {
  const char* what = argc > 1 ? argv[1] : "ok";
  const int W = 160, H = 120;
  if (!std::strcmp(what, "ok")) {
    aslam::Camera cam(120.0, 120.0, 79.5, 59.5, W, H,
                      aslam::Distortion(aslam::Distortion::Type::kRadTan, -0.2, 0.05, 1e-4, -1e-4));
    const amhip_camera c = amhip_shim::describe_camera(cam);
    std::printf("distortion %d k1 %.3f\n", c.distortion, c.dist[0]);
    return c.distortion == AMHIP_DIST_RADTAN && c.dist[0] == -0.2 ? 0 : 1;
  }
  if (!std::strcmp(what, "fisheye")) {
    aslam::Camera cam(120.0, 120.0, 79.5, 59.5, W, H,
                      aslam::Distortion(aslam::Distortion::Type::kFisheye, 0.9, 0, 0, 0));
    (void)amhip_shim::describe_camera(cam);
  } else if (!std::strcmp(what, "unified")) {
    aslam::Camera cam(120.0, 120.0, 79.5, 59.5, W, H, aslam::Distortion(),
                      aslam::Camera::Type::kUnifiedProjection);
    (void)amhip_shim::describe_camera(cam);
  }
  std::printf("NOT REFUSED\n");
  return 0;
}
