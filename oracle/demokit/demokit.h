// oracle/demokit -- what the reference's OWN demo mains need besides the classes of the hot
// path, so that aerial_mapper_demos/src/dsm/main-dsm.cc and ortho/main-ortho-backward-grid.cc
// can be compiled UNCHANGED and run without ROS / gflags / OpenCV / aslam_cv2 / GDAL:
//   gflags/gflags.h                 DEFINE_* + ParseCommandLineFlags (--name=value)
//   glog/logging.h                  LOG / CHECK (refkit's) + InitGoogleLogging, InstallFailureSignalHandler
//   ros/ros.h, ros/publisher.h      init, Time, NodeHandle; a Publisher whose publish() of the
//                                   grid map WRITES THE LAYERS to $AMHIP_DEMO_OUT and exits
//                                   (the demos end in map.publishUntilShutdown(), a loop)
//   grid_map_msgs, grid_map_ros     the message = a handle on the map
//   aerial-mapper-io/...            poses / rig / images / cloud loaders for the test's own
//                                   on-disk formats (the reference's loaders need OpenCV
//                                   imread, aslam YAML)
//   aerial-mapper-dense-pcl/...     stereo::Stereo stub (never reached: the cloud comes from a file)
// Two builds of each main (oracle/Makefile, target `demos`):
//   *_ref      over oracle/refkit + the reference's dsm.cc / ortho-backward-grid.cc /
//              aerial-mapper-grid-map.cc (-DDEMOKIT_REF)
//   *_dropin   over include/ of this repository + libaerial_mapper_shim.so, with the
//              reference's aerial-mapper-grid-map.cc on the compat GridMap (-DDEMOKIT_DROPIN)
// TEST INFRASTRUCTURE ONLY (tests/test_gpu_reference_demos.py).
#ifndef ORACLE_DEMOKIT_H_
#define ORACLE_DEMOKIT_H_
#include <cstdio>
#include <cstdlib>
#include <string>

namespace demokit {

// raw float32, column-major (the Eigen matrix as it lies in memory) + a shape file
template <class GridMapT>
inline void dump_layers(const GridMapT& map) {
  const char* dir = std::getenv("AMHIP_DEMO_OUT");
  if (!dir) return;
  const char* names[] = {"ortho", "elevation", "elevation_angle", "num_observations",
                         "observation_index", "colored_ortho"};
  const int rows = map.getSize()(0), cols = map.getSize()(1);
  {
    const std::string p = std::string(dir) + "/shape.txt";
    FILE* f = std::fopen(p.c_str(), "w");
    if (f) {
      std::fprintf(f, "%d %d\n", rows, cols);
      std::fclose(f);
    }
  }
  for (const char* n : names) {
    const std::string p = std::string(dir) + "/" + n + ".f32";
    FILE* f = std::fopen(p.c_str(), "wb");
    if (!f) continue;
    std::fwrite(map[n].data(), sizeof(float), static_cast<size_t>(rows) * cols, f);
    std::fclose(f);
  }
}

}  // namespace demokit
#endif  // ORACLE_DEMOKIT_H_
