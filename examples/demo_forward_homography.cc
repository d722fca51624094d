// demo_forward_homography.cc -- the flow of the reference's
// aerial_mapper_demos/src/ortho/main-ortho-forward-homography.cc:60-102 (load
// poses, build the mosaic with ortho::OrthoForwardHomography, batch or
// incremental) on synthetic inputs, through the drop-in C++ classes of this
// repository: io::AerialMapperIO::loadPosesFromFileStandard for the pose file,
// ortho::OrthoForwardHomography for the mosaic.  No ROS, no OpenCV, no oracle.
//
//   make -C examples && examples/demo_forward_homography [frames] [mosaic_px] [incremental]
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <memory>

#include "aerial-mapper-ortho/ortho-forward-homography.h"

static uint64_t g_state = 7;
static double urand() {
  uint64_t z = (g_state += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  z ^= z >> 31;
  return (z >> 11) * (1.0 / 9007199254740992.0);
}

static double now_s() {
  using namespace std::chrono;
  return duration_cast<duration<double> >(steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
  const int F = argc > 1 ? std::atoi(argv[1]) : 40;
  const int N = argc > 2 ? std::atoi(argv[2]) : 1200;
  const bool incremental = argc > 3 && std::atoi(argv[3]) != 0;
  const int W = 752, H = 480;
  const double f = 450.0, altitude = 600.0, ground = 414.0;

  // a lawn-mower pose file in the reference's format (x y z qw qx qy qz per line)
  const char* pose_file = "/tmp/demo_forward_poses.txt";
  {
    std::ofstream out(pose_file);
    out.precision(17);
    const double s45 = std::sqrt(0.5);
    const int lines = std::max(1, static_cast<int>(std::sqrt(F / 2.0)));
    const int per_line = (F + lines - 1) / lines;
    for (int k = 0; k < F; ++k) {
      const int ln = k / per_line, s = k % per_line;
      const double span = 0.7 * N;
      const double y = -span / 2 + (ln + 0.5) * span / lines;
      const double x = (-span / 2 + (s + 0.5) * span / per_line) * (ln % 2 ? -1.0 : 1.0);
      double q[4] = {0.02 * (urand() - 0.5), s45, s45 + 0.02 * (urand() - 0.5), 0.02 * (urand() - 0.5)};
      const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
      out << x << " " << y << " " << altitude << " " << q[0] / n << " " << q[1] / n << " " << q[2] / n
          << " " << q[3] / n << "\n";
    }
  }
  io::AerialMapperIO io_handler;
  Poses T_G_Bs;
  io_handler.loadPosesFromFileStandard(pose_file, &T_G_Bs);

  Images images;
  for (int k = 0; k < F; ++k) {
    Image img(H, W, 1);
    for (int v = 0; v < H; ++v)
      for (int u = 0; u < W; ++u)
        img.data[static_cast<size_t>(v) * img.step + u] =
            static_cast<uint8_t>(40 + ((u / 16 + v / 16 + k) % 2) * 150 + (u + v) % 7);
    images.push_back(img);
  }

  aslam::Camera cam(f, f, (W - 1) / 2.0, (H - 1) / 2.0, W, H);
  std::shared_ptr<aslam::NCamera> ncameras(new aslam::NCamera(
      cam, aslam::Transformation(kindr::minimal::RotationQuaternion(1, 0, 0, 0),
                                 Eigen::Vector3d(0.0, 0.0, 0.0))));

  // "Construct the mosaic by computing the homography that projects the image
  //  onto the ground plane." (main-ortho-forward-homography.cc:80-102)
  ortho::Settings settings;
  settings.batch = !incremental;
  settings.ground_plane_elevation_m = ground;
  settings.width_mosaic_pixels = N;
  settings.height_mosaic_pixels = N;
  settings.filename_mosaic_output = "/tmp/demo_forward_mosaic";
  const double t0 = now_s();
  ortho::OrthoForwardHomography mosaic(ncameras, settings);
  if (settings.batch) {
    mosaic.batch(T_G_Bs, images);
  } else {
    for (size_t i = 0u; i < images.size(); ++i) mosaic.updateOrthomosaic(T_G_Bs[i], images[i]);
  }
  const double t1 = now_s();

  size_t covered = 0;
  for (uint8_t m : mosaic.result_mask()) covered += m != 0;
  std::printf("%s: %d frames %dx%d -> %dx%d mosaic in %.1f ms (host images, PCIe included), "
              "%.1f %% covered, written to %s.ppm\n",
              incremental ? "updateOrthomosaic x F" : "batch", F, W, H, N, N, (t1 - t0) * 1e3,
              100.0 * covered / (static_cast<double>(N) * N), settings.filename_mosaic_output.c_str());
  return covered > 0 ? 0 : 1;
}
