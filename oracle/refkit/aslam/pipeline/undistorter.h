// oracle/refkit: see aslam/pipeline/undistorter-mapped.h.  TEST INFRASTRUCTURE ONLY.
#include <aslam/pipeline/undistorter-mapped.h>
