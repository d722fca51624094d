// ortho::OrthoFromPcl on MI355X -- drop-in for the reference class
// (aerial_mapper_ortho/include/aerial-mapper-ortho/ortho-from-pcl.h:28-52):
// same namespace, Settings and public signatures, so
// aerial_mapper_demos/src/ortho/main-ortho-from-pcl.cc:122-137 compiles against
// it unchanged.  NOTE (as in the reference): this header's `ortho::Settings`
// clashes with the one in ortho-backward-grid.h; include one per translation
// unit.
#ifndef AERIAL_MAPPER_HIP_ORTHO_FROM_PCL_H_
#define AERIAL_MAPPER_HIP_ORTHO_FROM_PCL_H_

#include <string>
#include <vector>

#include "aerial-mapper-deps.h"
#include "aerial-mapper-utils/utils-nearest-neighbor.h"

namespace ortho {

struct Settings {
  EIGEN_MAKE_ALIGNED_OPERATOR_NEW
  bool show_orthomosaic_opencv = false;
  int interpolation_radius = 2;  // SQUARED search radius, m^2
  bool use_adaptive_interpolation = false;
  bool save_orthomosaic_jpg = false;
  std::string orthomosaic_jpg_filename = "";
};

class OrthoFromPcl {
 public:
  EIGEN_MAKE_ALIGNED_OPERATOR_NEW

  OrthoFromPcl(const Settings& settings);

  // Interpolates the points' intensities into the map's "ortho" layer.
  void process(const AlignedType<std::vector, Eigen::Vector3d>::type& pointcloud,
               const std::vector<int>& intensities, grid_map::GridMap* map) const;

 private:
  void printParams() const;
  Settings settings_;
};

}  // namespace ortho

#endif  // AERIAL_MAPPER_HIP_ORTHO_FROM_PCL_H_
