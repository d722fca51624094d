// oracle/refkit: stand-in for <std_msgs/Header.h> (see ../refkit.h).  TEST INFRASTRUCTURE ONLY.
#ifndef ORACLE_REFKIT_STD_MSGS_HEADER_H_
#define ORACLE_REFKIT_STD_MSGS_HEADER_H_
#include <string>
#include <ros/ros.h>
namespace std_msgs {
struct Header {
  ros::Time stamp;
  std::string frame_id;
};
}  // namespace std_msgs
#endif  // ORACLE_REFKIT_STD_MSGS_HEADER_H_
