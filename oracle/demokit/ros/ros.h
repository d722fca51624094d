// oracle/demokit: stand-in for <ros/ros.h> of the demo mains (see ../demokit.h): like refkit's,
// plus init / Time::init and a Publisher that hands what it is given to `published()`, found
// by argument-dependent lookup: the grid map message writes the layers and ends the process.
#ifndef ORACLE_DEMOKIT_ROS_ROS_H_
#define ORACLE_DEMOKIT_ROS_ROS_H_
#include <chrono>
#include <cstdint>
#include <iomanip>
#include <ostream>
#include <sstream>
#include <string>

#define ROS_INFO(...) \
  do {                \
  } while (0)

namespace ros {
inline void init(int&, char**, const std::string&) {}
struct Duration {
  double seconds;
};
inline std::ostream& operator<<(std::ostream& os, const Duration& d) { return os << d.seconds; }
struct Time {
  std::chrono::steady_clock::time_point t;
  static void init() {}
  static Time now() {
    Time r;
    r.t = std::chrono::steady_clock::now();
    return r;
  }
  uint64_t toNSec() const {
    return static_cast<uint64_t>(
        std::chrono::duration_cast<std::chrono::nanoseconds>(t.time_since_epoch()).count());
  }
};
inline Duration operator-(const Time& a, const Time& b) {
  Duration d;
  d.seconds = std::chrono::duration<double>(a.t - b.t).count();
  return d;
}
template <typename Message>
inline void published(const Message&) {}
class Publisher {
 public:
  template <typename Message>
  void publish(const Message& m) const {
    published(m);  // (ADL: grid_map_msgs::published writes the layers and exits)
  }
};
class NodeHandle {
 public:
  template <typename Message>
  Publisher advertise(const std::string&, uint32_t, bool = false) {
    return Publisher();
  }
};
struct Rate {
  explicit Rate(double) {}
  void sleep() {}
};
inline void spinOnce() {}
}  // namespace ros
#endif  // ORACLE_DEMOKIT_ROS_ROS_H_
