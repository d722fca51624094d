// oracle/demokit: <aerial-mapper-io/aerial-mapper-io.h> of the demo mains (see ../demokit.h).
// Same typedefs and the four loader signatures main-dsm.cc / main-ortho-backward-grid.cc call
// (aerial_mapper_io/include/aerial-mapper-io/aerial-mapper-io.h:17-20,26-73); the formats are
// the test's own where the reference needs OpenCV / aslam_cv2:
//   poses        the reference's text format: x y z qw qx qy qz per line (aerial-mapper-io.cc:103-121)
//   point cloud  the reference's text format: x y z intensity, z <= -100 dropped (:309-347)
//   camera rig   one line: fu fv cu cv width height distortion(0 none,1 radtan,2 equidistant)
//                d0 d1 d2 d3  tx ty tz qw qx qy qz (T_C_B)          [reference: aslam YAML]
//   images       <prefix><i>.jpg holding a binary PGM (P5) / PPM (P6, stored BGR) [reference: imread]
// TEST INFRASTRUCTURE ONLY.
#ifndef ORACLE_DEMOKIT_IO_H_
#define ORACLE_DEMOKIT_IO_H_
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <memory>
#include <string>
#include <vector>

#ifdef DEMOKIT_DROPIN
#include "aerial-mapper-deps.h"
#include <glog/logging.h>  // (the real aslam headers bring glog in; main-ortho-backward-grid.cc relies on it)
#else
#include <Eigen/Core>
#include <aslam/cameras/ncamera.h>
#include <opencv2/highgui/highgui.hpp>
#endif
#include <aerial-mapper-utils/utils-nearest-neighbor.h>

typedef kindr::minimal::QuatTransformation Pose;
typedef std::vector<Pose> Poses;
typedef cv::Mat Image;
typedef std::vector<Image> Images;

namespace io {

enum PoseFormat { Standard, COLMAP, PIX4D, ROS };

class AerialMapperIO {
 public:
  AerialMapperIO() {}

  static Pose make_pose(const double* p /* x y z qw qx qy qz */) {
#ifdef DEMOKIT_DROPIN
    return Pose(kindr::minimal::RotationQuaternion(p[3], p[4], p[5], p[6]),
                Eigen::Vector3d(p[0], p[1], p[2]));
#else
    return Pose(amo::pose_from7(p));
#endif
  }

  void loadPosesFromFile(const PoseFormat&, const std::string& filename, Poses* T_G_Bs) {
    std::ifstream in(filename);
    if (!in) die("poses", filename);
    double p[7];
    while (in >> p[0] >> p[1] >> p[2] >> p[3] >> p[4] >> p[5] >> p[6]) T_G_Bs->push_back(make_pose(p));
  }

  std::shared_ptr<aslam::NCamera> loadCameraRigFromFile(const std::string& filename) {
    std::ifstream in(filename);
    if (!in) die("camera rig", filename);
    double fu, fv, cu, cv, d[4], t[7];
    int w, h, dist;
    in >> fu >> fv >> cu >> cv >> w >> h >> dist >> d[0] >> d[1] >> d[2] >> d[3];
    for (int k = 0; k < 7; ++k) in >> t[k];
    if (!in) die("camera rig (format)", filename);
#ifdef DEMOKIT_DROPIN
    aslam::Distortion::Type ty = dist == 1 ? aslam::Distortion::Type::kRadTan
                                 : dist == 2 ? aslam::Distortion::Type::kEquidistant
                                             : aslam::Distortion::Type::kNoDistortion;
    aslam::Camera cam(fu, fv, cu, cv, w, h, aslam::Distortion(ty, d[0], d[1], d[2], d[3]));
    return std::shared_ptr<aslam::NCamera>(new aslam::NCamera(cam, make_pose(t)));
#else
    amo_camera c;
    std::memset(&c, 0, sizeof(c));
    c.fu = fu; c.fv = fv; c.cu = cu; c.cv = cv;
    c.width = w; c.height = h; c.distortion = dist;
    for (int k = 0; k < 4; ++k) c.dist[k] = d[k];
    return std::shared_ptr<aslam::NCamera>(new aslam::NCamera(c, make_pose(t)));
#endif
  }

  // (load_colored_images: the reference's imread flag, aerial-mapper-io.h:34-35; the test's PGM /
  // PPM files carry their own channel count)
  void loadImagesFromFile(const std::string& filename_base, size_t num_poses, Images* images,
                          bool /*load_colored_images*/ = false) {
    for (size_t i = 0; i < num_poses; ++i) {
      const std::string name = filename_base + std::to_string(i) + ".jpg";
      FILE* f = std::fopen(name.c_str(), "rb");
      if (!f) die("image", name);
      char magic[3] = {0, 0, 0};
      int w = 0, h = 0, maxv = 0;
      if (std::fscanf(f, "%2s %d %d %d", magic, &w, &h, &maxv) != 4) die("image header", name);
      std::fgetc(f);
      const int ch = magic[1] == '6' ? 3 : 1;
#ifdef DEMOKIT_DROPIN
      Image img(h, w, ch);
#else
      Image img(h, w, ch == 3 ? CV_8UC3 : CV_8UC1);
#endif
      for (int r = 0; r < h; ++r)
        if (std::fread(img.data + static_cast<size_t>(r) * img.step, 1, static_cast<size_t>(w) * ch, f) !=
            static_cast<size_t>(w) * ch)
          die("image data", name);
      std::fclose(f);
      images->push_back(img);
    }
  }

  // aerial-mapper-io.h:47-50 (main-ortho-from-pcl.cc:104-105)
  void loadPointCloudFromFile(const std::string& filename,
                              AlignedType<std::vector, Eigen::Vector3d>::type* point_cloud_xyz,
                              std::vector<int>* point_cloud_intensities) {
    std::ifstream in(filename);
    if (!in) die("point cloud", filename);
    double x, y, z;
    int intensity;
    while (in >> x >> y >> z >> intensity)
      if (z > -100.0) {
        point_cloud_xyz->push_back(Eigen::Vector3d(x, y, z));
        point_cloud_intensities->push_back(intensity);
      }
  }

  void loadPointCloudFromFile(const std::string& filename,
                              AlignedType<std::vector, Eigen::Vector3d>::type* point_cloud_xyz) {
    std::ifstream in(filename);
    if (!in) die("point cloud", filename);
    double x, y, z;
    int intensity;
    while (in >> x >> y >> z >> intensity)
      if (z > -100.0) point_cloud_xyz->push_back(Eigen::Vector3d(x, y, z));
  }

 private:
  static void die(const char* what, const std::string& name) {
    std::fprintf(stderr, "demokit io: cannot read %s '%s'\n", what, name.c_str());
    std::exit(2);
  }
};

}  // namespace io

#ifdef DEMOKIT_REF
// main-ortho-forward-homography.cc has no map to publish: what the reference's class hands to
// cv::imwrite (refkit keeps the last one: cv::last_written()) is written to
// $AMHIP_DEMO_OUT/mosaic.i16 (rows cols channels in mosaic_shape.txt) when the process ends.
namespace demokit {
struct MosaicDumper {
  MosaicDumper() { (void)cv::last_written(); }  // (constructed first => destroyed after this)
  ~MosaicDumper() {
    const char* dir = std::getenv("AMHIP_DEMO_OUT");
    const cv::Mat& m = cv::last_written();
    if (!dir || m.rows <= 0 || m.cols <= 0 || m.type() != CV_16SC3) return;
    FILE* f = std::fopen((std::string(dir) + "/mosaic.i16").c_str(), "wb");
    if (!f) return;
    for (int r = 0; r < m.rows; ++r) std::fwrite(m.ptr<int16_t>(r), 2, 3 * static_cast<size_t>(m.cols), f);
    std::fclose(f);
    f = std::fopen((std::string(dir) + "/mosaic_shape.txt").c_str(), "w");
    if (f) {
      std::fprintf(f, "%d %d 3\n", m.rows, m.cols);
      std::fclose(f);
    }
  }
};
static MosaicDumper g_mosaic_dumper;
}  // namespace demokit
#endif
#endif  // ORACLE_DEMOKIT_IO_H_
