// oracle/refkit: stand-in for <sensor_msgs/Image.h> (see ../refkit.h): a message that is only
// ever handed to a publisher stub.  TEST INFRASTRUCTURE ONLY.
#ifndef ORACLE_REFKIT_SENSOR_MSGS_IMAGE_H_
#define ORACLE_REFKIT_SENSOR_MSGS_IMAGE_H_
#include <std_msgs/Header.h>
namespace sensor_msgs {
struct Image {
  std_msgs::Header header;
};
}  // namespace sensor_msgs
#endif  // ORACLE_REFKIT_SENSOR_MSGS_IMAGE_H_
