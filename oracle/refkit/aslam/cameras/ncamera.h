// oracle/refkit: see aslam/cameras/camera.h, which holds every aslam / minkindr name the
// three reference files mention.  TEST INFRASTRUCTURE ONLY.
#include <aslam/cameras/camera.h>
