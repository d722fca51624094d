/*
 * oracle/ref_loops_ortho_from_pcl.cc -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * C entry point over the reference's OWN ortho::OrthoFromPcl
 * (aerial_mapper_ortho/src/ortho-from-pcl.cc, compiled unchanged from /root/reference; see
 * refkit/refkit.h).  Same arguments as the restated oracle's amo_ortho_from_pcl_process.
 */
#include <aerial-mapper-ortho/ortho-from-pcl.h>

#include "ref_loops_common.h"

extern "C" {

int amr_ortho_from_pcl_process(const double* xyz, const int32_t* intensities, size_t n,
                               const amo_grid* grid, int radius_sq, int adaptive, float* ortho) {
  if (!grid || !ortho || !xyz || !intensities || n == 0) return AMO_ERR_ARG;
  refkit::check_reset();
  grid_map::GridMap map({"ortho"});
  ref_loops::set_geometry(*grid, &map);
  if (!ref_loops::same_geometry(map.geometry(), *grid)) return AMO_ERR_ARG;
  ref_loops::layer_in(ortho, &map["ortho"]);
  AlignedType<std::vector, Eigen::Vector3d>::type cloud;
  cloud.reserve(n);
  for (size_t k = 0; k < n; ++k)
    cloud.push_back(Eigen::Vector3d(xyz[3 * k + 0], xyz[3 * k + 1], xyz[3 * k + 2]));
  const std::vector<int> values(intensities, intensities + n);
  ortho::Settings settings;
  settings.interpolation_radius = radius_sq;
  settings.use_adaptive_interpolation = adaptive != 0;
  ortho::OrthoFromPcl mosaic(settings);
  mosaic.process(cloud, values, &map);
  ref_loops::layer_out(map["ortho"], ortho);
  return ref_loops::check_result();
}

}  // extern "C"
