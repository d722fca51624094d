// oracle/refkit: stand-in for <geometry_msgs/PolygonStamped.h> (see ../refkit.h).  TEST INFRASTRUCTURE ONLY.
#ifndef ORACLE_REFKIT_GEOMETRY_MSGS_POLYGONSTAMPED_H_
#define ORACLE_REFKIT_GEOMETRY_MSGS_POLYGONSTAMPED_H_
#include <std_msgs/Header.h>
namespace geometry_msgs {
struct PolygonStamped {
  std_msgs::Header header;
};
}  // namespace geometry_msgs
#endif  // ORACLE_REFKIT_GEOMETRY_MSGS_POLYGONSTAMPED_H_
