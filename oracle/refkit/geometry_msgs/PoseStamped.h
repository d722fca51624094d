// oracle/refkit: the reference's dense-pcl headers include this header but the reprojection
// loop uses nothing from it (see refkit.h).  TEST INFRASTRUCTURE ONLY.
