// ortho::OrthoBackwardGrid over the C ABI
// (see include/aerial-mapper-ortho/ortho-backward-grid.h).
#include "aerial-mapper-ortho/ortho-backward-grid.h"

#include <cstdio>
#include <vector>

#include "shim_common.h"

namespace ortho {

using amhip_shim::describe_camera;
using amhip_shim::pose_to7;

OrthoBackwardGrid::OrthoBackwardGrid(const std::shared_ptr<aslam::NCamera> ncameras,
                                     const Settings& settings, grid_map::GridMap* map)
    : ncameras_(ncameras), settings_(settings), session_(nullptr) {
  if (!ncameras_) amhip_shim::fatal("OrthoBackwardGrid", "CHECK(ncameras_)");
  printParams();
  // The reference dereferences `map` here when use_multi_threads is set
  // (ortho-backward-grid.cc:30-39); a null map is simply deferred to process().
  if (map) ensureSession(*map);
}

OrthoBackwardGrid::~OrthoBackwardGrid() { amhip_shim::release_session(session_); }

void OrthoBackwardGrid::ensureSession(const grid_map::GridMap& map) const {
  session_ = amhip_shim::acquire_session(map, session_, "OrthoBackwardGrid");
}

void OrthoBackwardGrid::process(const Poses& T_G_Bs, const Images& images,
                                grid_map::GridMap* map) const {
  if (T_G_Bs.empty()) amhip_shim::fatal("OrthoBackwardGrid::process", "CHECK(!T_G_Bs.empty())");
  if (T_G_Bs.size() != images.size())
    amhip_shim::fatal("OrthoBackwardGrid::process", "CHECK(T_G_Bs.size() == images.size())");
  if (!map) amhip_shim::fatal("OrthoBackwardGrid::process", "CHECK(map)");
  std::fprintf(stderr, "[aerial_mapper_hip] Num. images = %zu\n", images.size());
  ensureSession(*map);

  const size_t F = T_G_Bs.size();
  std::vector<double> T_G_B(7 * F), T_G_C(7 * F);
  for (size_t f = 0; f < F; ++f) pose_to7(T_G_Bs[f], &T_G_B[7 * f]);
  double T_C_B[7];
  pose_to7(ncameras_->get_T_C_B(0u), T_C_B);
  // T_G_C = T_G_B * T_C_B^-1 (ortho-backward-grid.cc:230-233)
  amhip_compose_T_G_C(T_G_B.data(), T_C_B, F, T_G_C.data());

  const amhip_camera cam = describe_camera(ncameras_->getCamera(kFrameIdx));
  const int channels = settings_.colored_ortho ? 3 : 1;
  std::vector<const uint8_t*> data(F);
  std::vector<size_t> steps(F);
  for (size_t f = 0; f < F; ++f) {
    if (images[f].channels() != channels || images[f].rows != cam.height ||
        images[f].cols != cam.width)
      amhip_shim::fatal("OrthoBackwardGrid::process",
                        "image type/size does not match the camera and colored_ortho");
    data[f] = images[f].data;
    steps[f] = static_cast<size_t>(images[f].step);
  }
  amhip_shim::check_status(
      amhip_session_ortho_backward_process(
          session_, &cam, T_G_C.data(), F, data.data(), steps.data(), channels,
          settings_.colored_ortho ? 1 : 0, (*map)["elevation"].data(),
          (*map)["elevation_angle"].data(), (*map)["observation_index"].data(),
          (*map)["num_observations"].data(), (*map)["ortho"].data(),
          (*map)["colored_ortho"].data()),
      "OrthoBackwardGrid::process");
}

void OrthoBackwardGrid::printParams() const {
  std::fprintf(stderr,
               "**************************************************\n"
               "Orthomosaic parameters (MI355X / HIP):\n"
               "  Show orthomosaic opencv %d\n  Save orthomosaic jpg    %d\n"
               "  Orthomosaic filename    %s\n"
               "**************************************************\n",
               (int)settings_.show_orthomosaic_opencv, (int)settings_.save_orthomosaic_jpg,
               settings_.orthomosaic_jpg_filename.c_str());
}

}  // namespace ortho
