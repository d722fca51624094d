/*
 * oracle/ref_loops_dsm.cc -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * C entry point over the reference's OWN dsm::Dsm (aerial_mapper_dsm/src/dsm.cc, compiled
 * unchanged from /root/reference next to this file; see refkit/refkit.h for what that
 * pins).  Same arguments as the restated oracle's amo_dsm_process (amo_dsm.cc), so a test
 * can run both on the same input and compare bit for bit.
 */
#include <aerial-mapper-dsm/dsm.h>

#include "ref_loops_common.h"

extern "C" {

/* grid_map::GridMap with an "elevation" layer -> dsm::Dsm(settings, &map) ->
 * Dsm::process(cloud, &map) (dsm.cc:186-201).  multi_thread selects
 * updateElevationLayerMultiThreaded (:113-184) or updateElevationLayer (:54-111). */
int amr_dsm_process(const double* xyz, size_t n, const amo_grid* grid, int radius_sq,
                    double center_easting, double center_northing, int multi_thread,
                    float* elevation, double* timing) {
  if (!grid || !elevation || (n && !xyz)) return AMO_ERR_ARG;
  refkit::check_reset();
  grid_map::GridMap map({"elevation"});
  ref_loops::set_geometry(*grid, &map);
  if (!ref_loops::same_geometry(map.geometry(), *grid)) return AMO_ERR_ARG;
  ref_loops::layer_in(elevation, &map["elevation"]);
  dsm::Settings settings;
  settings.interpolation_radius = radius_sq;
  settings.center_easting = center_easting;
  settings.center_northing = center_northing;
  settings.use_multi_threads = multi_thread != 0;
  AlignedType<std::vector, Eigen::Vector3d>::type cloud;
  cloud.reserve(n);
  for (size_t k = 0; k < n; ++k)
    cloud.push_back(Eigen::Vector3d(xyz[3 * k + 0], xyz[3 * k + 1], xyz[3 * k + 2]));
  const double t0 = ref_loops::now_s();
  dsm::Dsm digital_surface_map(settings, &map);
  const double t1 = ref_loops::now_s();
  digital_surface_map.process(cloud, &map);
  if (timing) {  // [0] constructor (one sample per cell, dsm.cc:20-34), [1] process()
    timing[0] = t1 - t0;
    timing[1] = ref_loops::now_s() - t1;
  }
  ref_loops::layer_out(map["elevation"], elevation);
  return ref_loops::check_result();
}

}  // extern "C"
