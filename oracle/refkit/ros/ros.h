// oracle/refkit: stand-in for <ros/ros.h> (see ../refkit.h) -- the reference files of the hot
// path only take wall-clock differences for their log lines and hold publishers they never
// need here.  TEST INFRASTRUCTURE ONLY.
#ifndef ORACLE_REFKIT_ROS_ROS_H_
#define ORACLE_REFKIT_ROS_ROS_H_

#include <chrono>
#include <iomanip>  // (the reference's files rely on ROS / OpenCV bringing these in)
#include <sstream>
#include <cstdint>
#include <ostream>
#include <string>

#define ROS_INFO(...) \
  do {                \
  } while (0)

namespace ros {

struct Duration {
  double seconds;
};
inline std::ostream& operator<<(std::ostream& os, const Duration& d) { return os << d.seconds; }

struct Time {
  std::chrono::steady_clock::time_point t;
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

class Publisher {
 public:
  template <typename Message>
  void publish(const Message&) const {}
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

#endif  // ORACLE_REFKIT_ROS_ROS_H_
