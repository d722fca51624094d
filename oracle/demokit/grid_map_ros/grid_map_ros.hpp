// oracle/demokit: <grid_map_ros/grid_map_ros.hpp> (see ../demokit.h).
#ifndef ORACLE_DEMOKIT_GRID_MAP_ROS_HPP_
#define ORACLE_DEMOKIT_GRID_MAP_ROS_HPP_
#include <grid_map_core/GridMap.hpp>
#include <grid_map_msgs/GridMap.h>
#include "../demokit.h"
namespace grid_map {
struct GridMapRosConverter {
  static void toMessage(const GridMap& map, grid_map_msgs::GridMap& message) {
    const GridMap* p = &map;
    message.dump = [p]() { ::demokit::dump_layers(*p); };
  }
};
}  // namespace grid_map
#endif  // ORACLE_DEMOKIT_GRID_MAP_ROS_HPP_
