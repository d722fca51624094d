// oracle/refkit: see opencv2/highgui/highgui.hpp.  TEST INFRASTRUCTURE ONLY.
#include <opencv2/highgui/highgui.hpp>
