// The four typedefs of aerial-mapper-io.h:17-20 that appear in the mosaic
// classes' signatures, and the text-format loaders of io::AerialMapperIO that
// sit directly in front of the hot path:
//   loadPointCloudFromFile (both overloads, aerial-mapper-io.cc:309-347) --
//     tokenised and parsed on the GPU (amhip_io_parse_point_cloud_text);
//   loadPosesFromFileStandard (:103-121) -- a few KB, read on the host;
//   subtractOriginFromPoses.
//   toGeoTiff / writeDataToDEMGeoTiffColor (:349-509) -- the same files without
//     GDAL (amhip_geotiff_write_u8: baseline TIFF + GeoTIFF tags).
// The image / camera-rig loaders (OpenCV imread, aslam YAML) and the format
// converters stay the reference's.
#ifndef AERIAL_MAPPER_HIP_IO_TYPES_H_
#define AERIAL_MAPPER_HIP_IO_TYPES_H_

#include <string>
#include <vector>

#include "aerial-mapper-deps.h"
#include "aerial-mapper-utils/utils-nearest-neighbor.h"

typedef kindr::minimal::QuatTransformation Pose;
typedef std::vector<Pose> Poses;
typedef cv::Mat Image;
typedef std::vector<Image> Images;

namespace io {

enum PoseFormat { Standard, COLMAP, PIX4D, ROS };

class AerialMapperIO {
 public:
  EIGEN_MAKE_ALIGNED_OPERATOR_NEW
  AerialMapperIO() {}

  void loadPosesFromFileStandard(const std::string& filename, Poses* T_G_Bs);

  void loadPointCloudFromFile(const std::string& filename_point_cloud,
                              AlignedType<std::vector, Eigen::Vector3d>::type* point_cloud_xyz,
                              std::vector<int>* point_cloud_intensities);

  void loadPointCloudFromFile(const std::string& filename_point_cloud,
                              AlignedType<std::vector, Eigen::Vector3d>::type* point_cloud_xyz);

  void subtractOriginFromPoses(const Eigen::Vector3d& origin, Poses* T_G_Bs);

  void writeDataToDEMGeoTiffColor(const cv::Mat& ortho_image, const Eigen::Vector2d& xy,
                                  const std::string& geotiff_filename);

  void toGeoTiff(const cv::Mat& orthomosaic, const Eigen::Vector2d& xy,
                 const std::string& geotiff_filename);

  // --- extension: keep the parsed cloud in HBM (3 * n doubles AoS + n int32,
  // the layout of amhip_dsm_process_dev / amhip_ortho_from_pcl_process_dev);
  // release both with amhip_io_free().
  void loadPointCloudFromFileToDevice(const std::string& filename_point_cloud, double** dev_xyz,
                                      int32_t** dev_intensities, size_t* num_points);

  // --- extension: the same cloud as a binary file (amhip_io_write_point_cloud_binary) and its
  // loader, staged through pinned buffers straight into HBM.
  void savePointCloudToBinaryFile(const std::string& filename,
                                  const AlignedType<std::vector, Eigen::Vector3d>::type& point_cloud_xyz,
                                  const std::vector<int>& point_cloud_intensities);
  void loadPointCloudFromBinaryFileToDevice(const std::string& filename, double** dev_xyz,
                                            int32_t** dev_intensities, size_t* num_points);
};

}  // namespace io

#endif  // AERIAL_MAPPER_HIP_IO_TYPES_H_
