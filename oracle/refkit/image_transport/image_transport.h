// oracle/refkit: stand-in for <image_transport/image_transport.h> (see ../refkit.h): publishers
// that drop what they are given.  TEST INFRASTRUCTURE ONLY.
#ifndef ORACLE_REFKIT_IMAGE_TRANSPORT_H_
#define ORACLE_REFKIT_IMAGE_TRANSPORT_H_
#include <cstdint>
#include <string>
#include <ros/ros.h>
namespace image_transport {
class Publisher {
 public:
  template <typename Message>
  void publish(const Message&) const {}
};
class ImageTransport {
 public:
  explicit ImageTransport(const ros::NodeHandle&) {}
  Publisher advertise(const std::string&, uint32_t, bool = false) { return Publisher(); }
};
}  // namespace image_transport
#endif  // ORACLE_REFKIT_IMAGE_TRANSPORT_H_
