// oracle/demokit: <aerial-mapper-dense-pcl/stereo.h> of the demo mains: the dense
// reconstruction (OpenCV block matching) is outside the path.  The batch mains are handed a
// point cloud FILE, so Stereo::addFrames is never reached (it ends the process if it is).
// main-ortho-backward-grid-incremental.cc calls addFrame() per image (:143-146): here the k-th
// call (k >= 1; the first frame has no partner yet, like the reference's Stereo::addFrame)
// reads the cloud the test wrote for stereo pair k from $AMHIP_DEMO_STEREO_PREFIX<k>.txt
// (`x y z` per line; a missing file = an empty cloud).
#ifndef ORACLE_DEMOKIT_STEREO_H_
#define ORACLE_DEMOKIT_STEREO_H_
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <memory>
#include <string>
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
  Stereo(const std::shared_ptr<aslam::NCamera>&, const Settings&, const BlockMatchingParameters&)
      : frames_(0) {}
  void addFrames(const Poses&, const Images&,
                 AlignedType<std::vector, Eigen::Vector3d>::type*, std::vector<int>* = nullptr) {
    std::fprintf(stderr, "demokit: stereo::Stereo::addFrames is not part of this test\n");
    std::exit(3);
  }
  void addFrame(const Pose&, const Image&,
                AlignedType<std::vector, Eigen::Vector3d>::type* point_cloud,
                std::vector<int>* = nullptr) {
    const int k = frames_++;
    const char* prefix = std::getenv("AMHIP_DEMO_STEREO_PREFIX");
    if (k == 0 || !prefix) return;
    std::ifstream in(std::string(prefix) + std::to_string(k) + ".txt");
    double x, y, z;
    while (in >> x >> y >> z) point_cloud->push_back(Eigen::Vector3d(x, y, z));
  }

 private:
  int frames_;
};
}  // namespace stereo
#endif  // ORACLE_DEMOKIT_STEREO_H_
