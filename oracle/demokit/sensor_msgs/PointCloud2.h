// oracle/demokit: <sensor_msgs/PointCloud2.h> (included by main-dsm.cc, never used).
#ifndef ORACLE_DEMOKIT_SENSOR_MSGS_POINTCLOUD2_H_
#define ORACLE_DEMOKIT_SENSOR_MSGS_POINTCLOUD2_H_
namespace sensor_msgs {
struct PointCloud2 {};
}  // namespace sensor_msgs
#endif
