// oracle/refkit: stand-in for <ros/ros.h> (see ../refkit.h) -- the three files only take
// wall-clock differences for their log lines.  TEST INFRASTRUCTURE ONLY.
#ifndef ORACLE_REFKIT_ROS_ROS_H_
#define ORACLE_REFKIT_ROS_ROS_H_

#include <chrono>
#include <ostream>

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
};
inline Duration operator-(const Time& a, const Time& b) {
  Duration d;
  d.seconds = std::chrono::duration<double>(a.t - b.t).count();
  return d;
}

}  // namespace ros

#endif  // ORACLE_REFKIT_ROS_ROS_H_
