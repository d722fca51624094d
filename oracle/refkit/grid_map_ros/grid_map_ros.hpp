// oracle/refkit: stand-in for <grid_map_ros/grid_map_ros.hpp> (see ../refkit.h): the message
// conversion the (never called here) publishing functions mention.  TEST INFRASTRUCTURE ONLY.
#ifndef ORACLE_REFKIT_GRID_MAP_ROS_HPP_
#define ORACLE_REFKIT_GRID_MAP_ROS_HPP_
#include <grid_map_core/GridMap.hpp>
#include <grid_map_msgs/GridMap.h>
namespace grid_map {
struct GridMapRosConverter {
  static void toMessage(const GridMap&, grid_map_msgs::GridMap&) {}
};
}  // namespace grid_map
#endif  // ORACLE_REFKIT_GRID_MAP_ROS_HPP_
