// oracle/refkit: see ros/ros.h.  TEST INFRASTRUCTURE ONLY.
#include <ros/ros.h>
