// oracle/refkit: stand-in for <opencv2/calib3d/calib3d.hpp> (see ../../refkit.h): the block
// matchers' types, which the densifier's constructor creates and the reprojection loop never
// touches (block matching itself is OpenCV's and out of scope).  TEST INFRASTRUCTURE ONLY.
#ifndef ORACLE_REFKIT_OPENCV2_CALIB3D_HPP_
#define ORACLE_REFKIT_OPENCV2_CALIB3D_HPP_

#include <opencv2/highgui/highgui.hpp>

namespace cv {

struct StereoBM {
  static Ptr<StereoBM> create(int, int) { return Ptr<StereoBM>(new StereoBM()); }
  void setMinDisparity(int) {}
  void setNumDisparities(int) {}
  void setPreFilterCap(int) {}
  void setUniquenessRatio(int) {}
  void setTextureThreshold(int) {}
  void setSpeckleWindowSize(int) {}
  void setSpeckleRange(int) {}
  void setBlockSize(int) {}
};

struct StereoSGBM {
  static Ptr<StereoSGBM> create(int, int, int) { return Ptr<StereoSGBM>(new StereoSGBM()); }
  void setMinDisparity(int) {}
  void setNumDisparities(int) {}
  void setPreFilterCap(int) {}
  void setUniquenessRatio(int) {}
  void setSpeckleWindowSize(int) {}
  void setSpeckleRange(int) {}
  void setDisp12MaxDiff(int) {}
  void setP1(int) {}
  void setP2(int) {}
  void setBlockSize(int) {}
};

}  // namespace cv

#endif  // ORACLE_REFKIT_OPENCV2_CALIB3D_HPP_
