// oracle/refkit: stand-in for <sensor_msgs/fill_image.h> (see ../refkit.h).  TEST INFRASTRUCTURE ONLY.
#ifndef ORACLE_REFKIT_SENSOR_MSGS_FILL_IMAGE_H_
#define ORACLE_REFKIT_SENSOR_MSGS_FILL_IMAGE_H_
#include <cstdint>
#include <string>
#include <sensor_msgs/Image.h>
namespace sensor_msgs {
namespace image_encodings {
const std::string MONO8 = "mono8";
}  // namespace image_encodings
inline bool fillImage(Image&, const std::string&, uint32_t, uint32_t, uint32_t, const void*) { return true; }
}  // namespace sensor_msgs
#endif  // ORACLE_REFKIT_SENSOR_MSGS_FILL_IMAGE_H_
