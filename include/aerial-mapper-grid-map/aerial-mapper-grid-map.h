// grid_map::AerialGridMap without its ROS publisher: creates the layered map
// exactly like aerial_mapper_grid_map/src/aerial-mapper-grid-map.cc:23-49
// (same layers, geometry call and init constants).  The reference's own
// class (with the publishers) keeps working unchanged against the GPU path;
// this header exists so examples/ and tests/cpp build without ROS.
#ifndef AERIAL_MAPPER_HIP_GRID_MAP_H_
#define AERIAL_MAPPER_HIP_GRID_MAP_H_

#include <cmath>

#include "aerial-mapper-deps.h"

namespace grid_map {

struct Settings {
  double center_easting;
  double center_northing;
  double delta_easting;
  double delta_northing;
  double resolution;
};

class AerialGridMap {
 public:
  explicit AerialGridMap(const Settings& settings) : settings_(settings) { initialize(); }
  grid_map::GridMap* getMutable() { return &map_; }

 private:
  void initialize() {
    map_ = grid_map::GridMap({"ortho", "elevation", "elevation_angle", "num_observations",
                              "elevation_angle_first_view", "delta", "observation_index",
                              "observation_index_first", "colored_ortho"});
    map_.setFrameId("world");
    map_.setGeometry(grid_map::Length(settings_.delta_easting, settings_.delta_northing),
                     settings_.resolution,
                     grid_map::Position(settings_.center_easting, settings_.center_northing));
    map_["ortho"].setConstant(255);
    map_["elevation"].setConstant(NAN);
    map_["elevation_angle"].setConstant(0.0);
    map_["elevation_angle_first_view"].setConstant(NAN);
    map_["num_observations"].setConstant(0);
    map_["observation_index"].setConstant(NAN);
    map_["observation_index_first"].setConstant(NAN);
    map_["delta"].setConstant(NAN);
    map_["colored_ortho"].setConstant(NAN);
  }
  grid_map::GridMap map_;
  Settings settings_;
};

}  // namespace grid_map

#endif  // AERIAL_MAPPER_HIP_GRID_MAP_H_
