// oracle/demokit: <aerial-mapper-dense-pcl/stereo.h> of the demo mains: the dense
// reconstruction (OpenCV block matching) is outside the path; the tests hand the mains a point
// cloud FILE, so Stereo::addFrames is never reached (it ends the process if it is).
#ifndef ORACLE_DEMOKIT_STEREO_H_
#define ORACLE_DEMOKIT_STEREO_H_
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <aerial-mapper-io/aerial-mapper-io.h>
namespace stereo {
struct Settings {
  int use_every_nth_image = 10;
};
struct BlockMatchingParameters {
  bool use_BM = true;
};
class Stereo {
 public:
  Stereo(const std::shared_ptr<aslam::NCamera>&, const Settings&, const BlockMatchingParameters&) {}
  void addFrames(const Poses&, const Images&,
                 AlignedType<std::vector, Eigen::Vector3d>::type*) {
    std::fprintf(stderr, "demokit: stereo::Stereo is not part of this test\n");
    std::exit(3);
  }
};
}  // namespace stereo
#endif  // ORACLE_DEMOKIT_STEREO_H_
