// ortho::OrthoBackwardGrid on MI355X -- drop-in for the reference class
// (aerial_mapper_ortho/include/aerial-mapper-ortho/ortho-backward-grid.h:32-50):
// same namespace, Settings and public signatures, so
// main-ortho-backward-grid.cc:135-141 and
// main-ortho-backward-grid-incremental.cc:122-157 compile against it unchanged.
// NOTE (as in the reference): two more headers define a struct named
// ortho::Settings; include only one of them per translation unit.
#ifndef AERIAL_MAPPER_HIP_ORTHO_BACKWARD_GRID_H_
#define AERIAL_MAPPER_HIP_ORTHO_BACKWARD_GRID_H_

#include <memory>
#include <string>

#include "aerial-mapper-deps.h"
#include "aerial-mapper-io/aerial-mapper-io.h"

struct amhip_session;

namespace ortho {

struct Settings {
  EIGEN_MAKE_ALIGNED_OPERATOR_NEW
  bool show_orthomosaic_opencv = true;
  bool save_orthomosaic_jpg = true;
  std::string orthomosaic_jpg_filename = "";
  double orthomosaic_elevation_m = 0.0;
  bool use_digital_elevation_map = true;
  bool colored_ortho = false;
  bool use_multi_threads = true;
};

class OrthoBackwardGrid {
 public:
  EIGEN_MAKE_ALIGNED_OPERATOR_NEW
  OrthoBackwardGrid(const std::shared_ptr<aslam::NCamera> ncameras, const Settings& settings,
                    grid_map::GridMap* map = nullptr);
  ~OrthoBackwardGrid();
  OrthoBackwardGrid(const OrthoBackwardGrid&) = delete;
  OrthoBackwardGrid& operator=(const OrthoBackwardGrid&) = delete;

  // Folds the images (ascending) into the map's elevation_angle /
  // observation_index / num_observations and ortho or colored_ortho layers,
  // reading its elevation layer.
  void process(const Poses& T_G_Bs, const Images& images, grid_map::GridMap* map) const;

 private:
  void ensureSession(const grid_map::GridMap& map) const;
  void printParams() const;

  std::shared_ptr<aslam::NCamera> ncameras_;
  static constexpr size_t kFrameIdx = 0u;
  Settings settings_;
  // the map's session, shared with dsm::Dsm on the same grid_map::GridMap
  mutable amhip_session* session_;
};

}  // namespace ortho

#endif  // AERIAL_MAPPER_HIP_ORTHO_BACKWARD_GRID_H_
