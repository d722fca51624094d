// io::AerialMapperIO's text loaders over the C ABI
// (see include/aerial-mapper-io/aerial-mapper-io.h).
#include "aerial-mapper-io/aerial-mapper-io.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <vector>

#include "shim_common.h"

namespace io {

namespace {

// read-only mapping of a whole file (the parser wants the bytes, not a stream)
struct MappedFile {
  const char* data;
  size_t size;
  int fd;
  explicit MappedFile(const std::string& name) : data(nullptr), size(0), fd(-1) {
    fd = ::open(name.c_str(), O_RDONLY);
    if (fd < 0) return;
    struct stat st;
    if (::fstat(fd, &st) != 0 || st.st_size <= 0) return;
    void* p = ::mmap(nullptr, static_cast<size_t>(st.st_size), PROT_READ, MAP_PRIVATE, fd, 0);
    if (p == MAP_FAILED) return;
    data = static_cast<const char*>(p);
    size = static_cast<size_t>(st.st_size);
  }
  ~MappedFile() {
    if (data) ::munmap(const_cast<char*>(data), size);
    if (fd >= 0) ::close(fd);
  }
};

int device_index() {
  return amhip_shim::default_device();
}

}  // namespace

void AerialMapperIO::loadPointCloudFromFileToDevice(const std::string& filename_point_cloud,
                                                    double** dev_xyz, int32_t** dev_intensities,
                                                    size_t* num_points) {
  if (filename_point_cloud.empty())
    amhip_shim::fatal("loadPointCloudFromFile", "CHECK(filename_point_cloud != \"\")");
  if (!dev_xyz || !dev_intensities || !num_points)
    amhip_shim::fatal("loadPointCloudFromFile", "CHECK(point_cloud_xyz)");
  std::fprintf(stderr, "[aerial_mapper_hip] Loading pointcloud from: %s\n",
               filename_point_cloud.c_str());
  MappedFile file(filename_point_cloud);
  size_t slow = 0;
  amhip_shim::check_status(
      amhip_io_parse_point_cloud_text(device_index(), file.data, file.size, dev_xyz,
                                      dev_intensities, num_points, &slow),
      "loadPointCloudFromFile");
  if (*num_points == 0)
    amhip_shim::fatal("loadPointCloudFromFile", "CHECK(point_cloud_xyz->size() > 0)");
}

void AerialMapperIO::loadPointCloudFromFile(
    const std::string& filename_point_cloud,
    AlignedType<std::vector, Eigen::Vector3d>::type* point_cloud_xyz,
    std::vector<int>* point_cloud_intensities) {
  if (!point_cloud_xyz) amhip_shim::fatal("loadPointCloudFromFile", "CHECK(point_cloud_xyz)");
  if (!point_cloud_intensities)
    amhip_shim::fatal("loadPointCloudFromFile", "CHECK(point_cloud_intensities)");
  double* dxyz = nullptr;
  int32_t* dint = nullptr;
  size_t n = 0;
  loadPointCloudFromFileToDevice(filename_point_cloud, &dxyz, &dint, &n);
  // the reference push_back()s: append to whatever the vectors hold
  static_assert(sizeof(Eigen::Vector3d) == 3 * sizeof(double), "Vector3d must be 3 packed doubles");
  static_assert(sizeof(int) == sizeof(int32_t), "int must be 32 bits");
  const size_t at = point_cloud_xyz->size(), at_i = point_cloud_intensities->size();
  point_cloud_xyz->resize(at + n);
  point_cloud_intensities->resize(at_i + n);
  amhip_shim::check_status(
      amhip_io_download_point_cloud(dxyz, dint, n,
                                    reinterpret_cast<double*>(point_cloud_xyz->data() + at),
                                    reinterpret_cast<int32_t*>(point_cloud_intensities->data() + at_i)),
      "loadPointCloudFromFile");
  amhip_io_free(dxyz);
  amhip_io_free(dint);
}

void AerialMapperIO::loadPointCloudFromFile(
    const std::string& filename_point_cloud,
    AlignedType<std::vector, Eigen::Vector3d>::type* point_cloud_xyz) {
  std::vector<int> intensities;
  loadPointCloudFromFile(filename_point_cloud, point_cloud_xyz, &intensities);
}

void AerialMapperIO::loadPosesFromFileStandard(const std::string& filename, Poses* T_G_Bs) {
  if (!T_G_Bs) amhip_shim::fatal("loadPosesFromFileStandard", "CHECK(T_G_Bs)");
  if (filename.empty()) amhip_shim::fatal("loadPosesFromFileStandard", "Empty filename");
  std::fprintf(stderr, "[aerial_mapper_hip] Loading body poses from: %s\n", filename.c_str());
  std::ifstream infile(filename);
  double x, y, z, qw, qx, qy, qz;
  while (infile >> x >> y >> z >> qw >> qx >> qy >> qz) {
    T_G_Bs->push_back(Pose(kindr::minimal::RotationQuaternion(qw, qx, qy, qz),
                           Eigen::Vector3d(x, y, z)));
    if (infile.eof()) break;
  }
  if (T_G_Bs->empty()) amhip_shim::fatal("loadPosesFromFileStandard", "No poses loaded.");
}

void AerialMapperIO::subtractOriginFromPoses(const Eigen::Vector3d& origin, Poses* T_G_Bs) {
  if (!T_G_Bs) amhip_shim::fatal("subtractOriginFromPoses", "CHECK(T_G_Bs)");
  if (T_G_Bs->empty()) amhip_shim::fatal("subtractOriginFromPoses", "CHECK(T_G_Bs->size() > 0u)");
  for (Pose& T : *T_G_Bs) {
    const Eigen::Vector3d& t = T.getPosition();
    T = Pose(T.getRotation(), Eigen::Vector3d(t(0) - origin(0), t(1) - origin(1), t(2) - origin(2)));
  }
}

// aerial-mapper-io.cc:349-431: one byte band; the geotransform is the constant the reference
// hard-codes (`xy` is not used there either), UTM 32 north on WGS 84.
void AerialMapperIO::toGeoTiff(const cv::Mat& orthomosaic, const Eigen::Vector2d& /*xy*/,
                               const std::string& geotiff_filename) {
  if (orthomosaic.empty() || orthomosaic.channels() != 1)
    amhip_shim::fatal("toGeoTiff", "an 8UC1 image is expected (orthomosaic.at<uchar>)");
  const double gt[6] = {464499.00, 1.0, 0.0, 5.2727e+06, 0.0, -1.0};
  amhip_shim::check_status(
      amhip_geotiff_write_u8(geotiff_filename.c_str(), orthomosaic.data, orthomosaic.cols,
                             orthomosaic.rows, orthomosaic.step, 1, gt, 32, 1),
      "toGeoTiff");
}

// aerial-mapper-io.cc:433-509: three byte bands, band 1 / 2 / 3 = channel 2 / 0 / 1 of the
// cv::Vec3b pixel (the reference's "TODO: Fix color bands"), unit pixels anchored at xy.
void AerialMapperIO::writeDataToDEMGeoTiffColor(const cv::Mat& ortho_image,
                                                const Eigen::Vector2d& xy,
                                                const std::string& geotiff_filename) {
  if (ortho_image.empty() || ortho_image.channels() != 3)
    amhip_shim::fatal("writeDataToDEMGeoTiffColor", "an 8UC3 image is expected (at<cv::Vec3b>)");
  const int w = ortho_image.cols, h = ortho_image.rows;
  std::vector<uint8_t> bands(static_cast<size_t>(w) * h * 3);
  for (int y = 0; y < h; ++y) {
    const uint8_t* src = ortho_image.data + static_cast<size_t>(y) * ortho_image.step;
    uint8_t* dst = bands.data() + static_cast<size_t>(y) * w * 3;
    for (int x = 0; x < w; ++x) {
      dst[3 * x + 0] = src[3 * x + 2];
      dst[3 * x + 1] = src[3 * x + 0];
      dst[3 * x + 2] = src[3 * x + 1];
    }
  }
  const double gt[6] = {xy(0), 1.0, 0.0, xy(1), 0.0, -1.0};
  amhip_shim::check_status(
      amhip_geotiff_write_u8(geotiff_filename.c_str(), bands.data(), w, h,
                             static_cast<size_t>(w) * 3, 3, gt, 32, 1),
      "writeDataToDEMGeoTiffColor");
}

void AerialMapperIO::savePointCloudToBinaryFile(
    const std::string& filename,
    const AlignedType<std::vector, Eigen::Vector3d>::type& point_cloud_xyz,
    const std::vector<int>& point_cloud_intensities) {
  static_assert(sizeof(Eigen::Vector3d) == 3 * sizeof(double), "Vector3d must be 3 packed doubles");
  if (!point_cloud_intensities.empty() && point_cloud_intensities.size() != point_cloud_xyz.size())
    amhip_shim::fatal("savePointCloudToBinaryFile", "CHECK(xyz.size() == intensities.size())");
  amhip_shim::check_status(
      amhip_io_write_point_cloud_binary(
          filename.c_str(), reinterpret_cast<const double*>(point_cloud_xyz.data()),
          point_cloud_intensities.empty()
              ? nullptr
              : reinterpret_cast<const int32_t*>(point_cloud_intensities.data()),
          point_cloud_xyz.size()),
      "savePointCloudToBinaryFile");
}

void AerialMapperIO::loadPointCloudFromBinaryFileToDevice(const std::string& filename,
                                                          double** dev_xyz,
                                                          int32_t** dev_intensities,
                                                          size_t* num_points) {
  amhip_shim::check_status(
      amhip_io_load_point_cloud_binary(device_index(), filename.c_str(), dev_xyz, dev_intensities,
                                       num_points),
      "loadPointCloudFromBinaryFileToDevice");
  if (*num_points == 0)
    amhip_shim::fatal("loadPointCloudFromBinaryFileToDevice", "CHECK(point_cloud_xyz->size() > 0)");
}

}  // namespace io
