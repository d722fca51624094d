// oracle/refkit: stand-in for <sensor_msgs/PointCloud2.h> (see ../refkit.h): the two fields the
// densifier's reprojection loop writes through.  TEST INFRASTRUCTURE ONLY.
#ifndef ORACLE_REFKIT_SENSOR_MSGS_POINTCLOUD2_H_
#define ORACLE_REFKIT_SENSOR_MSGS_POINTCLOUD2_H_
#include <cstdint>
#include <vector>
namespace sensor_msgs {
struct PointCloud2 {
  std::vector<uint8_t> data;
  uint32_t point_step = 0;
};
}  // namespace sensor_msgs
#endif  // ORACLE_REFKIT_SENSOR_MSGS_POINTCLOUD2_H_
