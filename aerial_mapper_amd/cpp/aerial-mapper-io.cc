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
  if (const char* env = std::getenv("AERIAL_MAPPER_HIP_DEVICE")) return std::atoi(env);
  return 0;
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

}  // namespace io
