// oracle/refkit: stand-in for <opencv2/core/core.hpp> (see ../../refkit.h): cv::Mat as the
// other OpenCV stand-ins define it.  TEST INFRASTRUCTURE ONLY.
#ifndef ORACLE_REFKIT_OPENCV2_CORE_CORE_HPP_
#define ORACLE_REFKIT_OPENCV2_CORE_CORE_HPP_
#include "../highgui/highgui.hpp"
#endif
