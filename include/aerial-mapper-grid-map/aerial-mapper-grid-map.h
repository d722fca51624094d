// grid_map::AerialGridMap without its ROS publisher: creates the layered map
// exactly like aerial_mapper_grid_map/src/aerial-mapper-grid-map.cc:23-49
// (same layers, geometry call and init constants).  The reference's own
// class (with the publishers) keeps working unchanged against the GPU path;
// this header exists so examples/ and tests/cpp build without ROS.
#ifndef AERIAL_MAPPER_HIP_GRID_MAP_H_
#define AERIAL_MAPPER_HIP_GRID_MAP_H_

#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "aerial-mapper-deps.h"
#include "aerial_mapper_hip.h"

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

  // What publishOnce() / publishUntilShutdown() put on the wire
  // (aerial-mapper-grid-map.cc:51-72: setTimestamp(now) + GridMapRosConverter::toMessage +
  // publish), as the serialized grid_map_msgs/GridMap message (ROS 1 wire format): hand it to a
  // publisher of pre-serialized messages, a bag writer or a socket.
  std::vector<uint8_t> serializeMessage(uint64_t stamp_ns) {
    static const char* const kLayers[9] = {"ortho", "elevation", "elevation_angle",
                                           "num_observations", "elevation_angle_first_view",
                                           "delta", "observation_index",
                                           "observation_index_first", "colored_ortho"};
    amhip_grid_desc g;
    g.rows = map_.getSize()(0);
    g.cols = map_.getSize()(1);
    g.resolution = map_.getResolution();
    g.length_x = map_.getLength()(0);
    g.length_y = map_.getLength()(1);
    g.pos_x = map_.getPosition()(0);
    g.pos_y = map_.getPosition()(1);
    const std::string frame = "world";  // setFrameId("world"), :29
    std::vector<uint8_t> msg(amhip_grid_map_msg_bytes(&g, frame.c_str(), 9, kLayers));
    size_t at[9];
    if (amhip_grid_map_msg_layout(&g, stamp_ns, frame.c_str(), 9, kLayers, msg.data(), msg.size(),
                                  at) != AMHIP_OK)
      return std::vector<uint8_t>();
    for (int l = 0; l < 9; ++l)
      std::memcpy(msg.data() + at[l], map_[kLayers[l]].data(),
                  sizeof(float) * static_cast<size_t>(g.rows) * static_cast<size_t>(g.cols));
    return msg;
  }

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
