// demo_backward_grid.cc -- the flow of the reference's
// aerial_mapper_demos/src/ortho/main-ortho-backward-grid.cc:118-141 (layered
// map -> dsm::Dsm::process -> ortho::OrthoBackwardGrid::process) on synthetic
// inputs, through the drop-in C++ classes of this repository.  No ROS, no file
// I/O, no oracle: product code only.
//
//   make -C examples && examples/demo_backward_grid [cells_per_side] [frames] [output_dir]
// With an output directory: the mosaic as a GeoTiff (io::AerialMapperIO::toGeoTiff) and the
// map as the serialized grid_map_msgs/GridMap message AerialGridMap::publishOnce would send.
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <memory>
#include <string>

#include "aerial-mapper-dsm/dsm.h"
#include "aerial-mapper-grid-map/aerial-mapper-grid-map.h"
#include "aerial-mapper-io/aerial-mapper-io.h"
#include "aerial-mapper-ortho/ortho-backward-grid.h"

static uint64_t g_state = 42;
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
  const int side = argc > 1 ? std::atoi(argv[1]) : 1000;
  const int F = argc > 2 ? std::atoi(argv[2]) : 12;
  const double res = 0.25, len = side * res;

  // "Initialize layered map."
  grid_map::Settings settings_aerial_grid_map;
  settings_aerial_grid_map.center_easting = 0.0;
  settings_aerial_grid_map.center_northing = 0.0;
  settings_aerial_grid_map.delta_easting = len;
  settings_aerial_grid_map.delta_northing = len;
  settings_aerial_grid_map.resolution = res;
  grid_map::AerialGridMap map(settings_aerial_grid_map);

  // a dense cloud: 8 points per square metre on a smooth terrain
  AlignedType<std::vector, Eigen::Vector3d>::type point_cloud;
  const size_t n = static_cast<size_t>(8.0 * (len + 8.0) * (len + 8.0));
  point_cloud.reserve(n);
  for (size_t k = 0; k < n; ++k) {
    const double x = (urand() - 0.5) * (len + 8.0), y = (urand() - 0.5) * (len + 8.0);
    point_cloud.push_back(Eigen::Vector3d(
        x, y, 400.0 + 10.0 * std::sin(0.01 * x) * std::cos(0.01 * y) + 0.1 * (urand() - 0.5)));
  }

  // "Create DSM (batch)."
  dsm::Settings settings_dsm;
  settings_dsm.center_easting = settings_aerial_grid_map.center_easting;
  settings_dsm.center_northing = settings_aerial_grid_map.center_northing;
  dsm::Dsm digital_surface_map(settings_dsm, map.getMutable());
  double t0 = now_s();
  digital_surface_map.process(point_cloud, map.getMutable());
  const double t_dsm = now_s() - t0;

  // camera rig, nadir poses over the map, random frames
  const int W = 1920, H = 1080;
  std::shared_ptr<aslam::NCamera> ncameras(new aslam::NCamera(
      aslam::Camera(1400.0, 1400.0, (W - 1) / 2.0, (H - 1) / 2.0, W, H),
      aslam::Transformation(kindr::minimal::RotationQuaternion(1, 0, 0, 0), Eigen::Vector3d(0, 0, 0))));
  Poses T_G_Bs;
  Images images;
  const double s45 = std::sqrt(0.5);
  for (int f = 0; f < F; ++f) {
    const double x = (urand() - 0.5) * len, y = (urand() - 0.5) * len;
    T_G_Bs.push_back(Pose(kindr::minimal::RotationQuaternion(0.0, s45, s45, 0.0),  // looking down
                          Eigen::Vector3d(x, y, 700.0)));
    Image img(H, W, 1);
    for (size_t b = 0; b < static_cast<size_t>(H) * img.step; ++b)
      img.data[b] = static_cast<uint8_t>(urand() * 256.0);
    images.push_back(img);
  }

  // "Construct the orthomosaic (batch)."
  ortho::Settings settings_ortho;
  ortho::OrthoBackwardGrid mosaic(ncameras, settings_ortho, map.getMutable());
  t0 = now_s();
  mosaic.process(T_G_Bs, images, map.getMutable());
  const double t_ortho = now_s() - t0;

  const grid_map::Matrix& elevation = (*map.getMutable())["elevation"];
  const grid_map::Matrix& index = (*map.getMutable())["observation_index"];
  size_t filled = 0, seen = 0;
  for (long k = 0; k < elevation.size(); ++k) {
    filled += !std::isnan(elevation.data()[k]);
    seen += !std::isnan(index.data()[k]);
  }
  const double cells = static_cast<double>(side) * side;
  std::printf("%d x %d cells, %zu points, %d frames\n", side, side, point_cloud.size(), F);
  std::printf("DSM   %.1f ms (host buffers in/out)   cells with a height: %.1f %%\n", 1e3 * t_dsm,
              100.0 * filled / cells);
  std::printf("ortho %.1f ms (host buffers in/out)   cells with a view:   %.1f %%\n", 1e3 * t_ortho,
              100.0 * seen / cells);
  if (argc > 3) {
    // the ortho layer as the 8-bit image grid_map_cv's toImage makes of it (image row = grid
    // index 0), written like aerial-mapper-io.cc:349-431 does, and the map as a ROS message
    const std::string dir = argv[3];
    const grid_map::Matrix& ortho_layer = (*map.getMutable())["ortho"];
    cv::Mat image(side, side, 1);
    for (int i = 0; i < side; ++i)
      for (int j = 0; j < side; ++j) {
        const float v = ortho_layer(i, j);
        image.at<uint8_t>(i, j) = std::isfinite(v) ? static_cast<uint8_t>(std::fmin(std::fmax(v, 0.f), 255.f)) : 0;
      }
    io::AerialMapperIO io_handler;
    io_handler.toGeoTiff(image, Eigen::Vector2d(0.0, 0.0), dir + "/orthomosaic.tif");
    const std::vector<uint8_t> msg = map.serializeMessage(0);
    std::ofstream(dir + "/grid_map.msg", std::ios::binary)
        .write(reinterpret_cast<const char*>(msg.data()), static_cast<std::streamsize>(msg.size()));
    std::printf("wrote %s/orthomosaic.tif and %s/grid_map.msg (%zu bytes)\n", dir.c_str(), dir.c_str(),
                msg.size());
  }
  return 0;
}
