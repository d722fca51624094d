// shim_io_parity.cc -- io::AerialMapperIO's text loaders (GPU parser behind
// the reference's signatures) against the reference's own iostream loops
// (oracle/amo_io.cc).  TEST ONLY: links oracle/liboracle.so.  Exit 0 = parity.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "aerial-mapper-io/aerial-mapper-io.h"

extern "C" {
size_t amo_io_load_point_cloud(const char*, size_t, double*, int32_t*, size_t);
size_t amo_io_load_poses(const char*, size_t, double*, size_t);
}

static uint64_t g_state = 0x9E3779B97F4A7C15ULL;
static double urand() {
  uint64_t z = (g_state += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  z ^= z >> 31;
  return (z >> 11) * (1.0 / 9007199254740992.0);
}

int main(int argc, char** argv) {
  const std::string dir = argc > 1 ? argv[1] : "/tmp";
  const std::string cloud_file = dir + "/cloud.txt", pose_file = dir + "/poses.txt";
  const size_t n = 120000;
  {
    std::ofstream f(cloud_file);
    f.precision(15);
    for (size_t k = 0; k < n; ++k) {
      const double x = 464000.0 + urand() * 900.0, y = 5.27e6 + urand() * 700.0;
      const double z = (k % 53 == 0) ? -250.0 : 380.0 + 40.0 * urand();
      f << x << " " << y << " " << z << " " << static_cast<int>(urand() * 255.0) << "\n";
    }
    std::ofstream p(pose_file);
    p.precision(17);
    for (int k = 0; k < 40; ++k)
      p << 100.0 * urand() << " " << 50.0 * urand() << " " << 400.0 + k << " " << urand() << " "
        << urand() << " " << urand() << " " << urand() << "\n";
  }
  io::AerialMapperIO loader;
  AlignedType<std::vector, Eigen::Vector3d>::type cloud, cloud_only;
  std::vector<int> intensities;
  loader.loadPointCloudFromFile(cloud_file, &cloud, &intensities);
  loader.loadPointCloudFromFile(cloud_file, &cloud_only);
  Poses poses;
  loader.loadPosesFromFileStandard(pose_file, &poses);

  std::ifstream in(cloud_file, std::ios::binary);
  std::stringstream ss;
  ss << in.rdbuf();
  const std::string text = ss.str();
  std::vector<double> want(3 * n);
  std::vector<int32_t> want_i(n);
  const size_t m = amo_io_load_point_cloud(text.data(), text.size(), want.data(), want_i.data(), n);
  bool ok = m == cloud.size() && m == intensities.size() && m == cloud_only.size() && m < n &&
            m > n / 2;
  size_t bad = 0;
  for (size_t k = 0; ok && k < m; ++k) {
    if (std::memcmp(&cloud[k], &want[3 * k], 24) != 0 || intensities[k] != want_i[k] ||
        std::memcmp(&cloud_only[k], &want[3 * k], 24) != 0)
      ++bad;
  }
  std::ifstream pin(pose_file, std::ios::binary);
  std::stringstream ps;
  ps << pin.rdbuf();
  const std::string ptext = ps.str();
  std::vector<double> wp(7 * 64);
  const size_t np = amo_io_load_poses(ptext.data(), ptext.size(), wp.data(), 64);
  ok = ok && bad == 0 && np == poses.size() && np == 40;
  for (size_t k = 0; ok && k < np; ++k) {
    const Eigen::Vector3d& t = poses[k].getPosition();
    const Eigen::Quaterniond& q = poses[k].getRotation().toImplementation();
    const double got[7] = {t(0), t(1), t(2), q.w(), q.x(), q.y(), q.z()};
    if (std::memcmp(got, &wp[7 * k], sizeof(got)) != 0) ok = false;
  }
  loader.subtractOriginFromPoses(Eigen::Vector3d(1.0, 2.0, 3.0), &poses);
  ok = ok && poses[0].getPosition()(2) == wp[2] - 3.0;
  std::printf("cloud: %zu points (%zu dropped by z > -100), %zu differ; poses: %zu\n", m, n - m, bad,
              np);
  std::printf(ok ? "PARITY OK\n" : "PARITY FAILED\n");
  return ok ? 0 : 1;
}
