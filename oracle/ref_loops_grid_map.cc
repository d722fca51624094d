/*
 * oracle/ref_loops_grid_map.cc -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * C entry point over the reference's OWN grid_map::AerialGridMap
 * (aerial_mapper_grid_map/src/aerial-mapper-grid-map.cc, compiled unchanged from
 * /root/reference; see refkit/refkit.h): which layers a map starts with, their initial
 * values and the arguments of the geometry call (:23-48).
 */
#include <aerial-mapper-grid-map/aerial-mapper-grid-map.h>

#include "ref_loops_common.h"

extern "C" {

/* layers: 6 pointers (may be null) in the order ortho, elevation, elevation_angle,
 * num_observations, observation_index, colored_ortho; each rows*cols floats of the
 * geometry returned in *geometry (call once with null layers to learn the size). */
int amr_grid_map_initialize(double center_easting, double center_northing, double delta_easting,
                            double delta_northing, double resolution, amo_grid* geometry,
                            float* const* layers) {
  if (!geometry) return AMO_ERR_ARG;
  refkit::check_reset();
  grid_map::Settings settings;
  settings.center_easting = center_easting;
  settings.center_northing = center_northing;
  settings.delta_easting = delta_easting;
  settings.delta_northing = delta_northing;
  settings.resolution = resolution;
  grid_map::AerialGridMap map(settings);
  grid_map::GridMap* m = map.getMutable();
  *geometry = m->geometry();
  static const char* const kNames[6] = {"ortho", "elevation", "elevation_angle",
                                        "num_observations", "observation_index", "colored_ortho"};
  for (int k = 0; k < 6; ++k) {
    if (!m->exists(kNames[k])) return AMO_ERR_ARG;
    if (layers && layers[k]) ref_loops::layer_out((*m)[kNames[k]], layers[k]);
  }
  return ref_loops::check_result();
}

}  // extern "C"
