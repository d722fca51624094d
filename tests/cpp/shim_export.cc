// The drop-in io::AerialMapperIO GeoTiff writers and AerialGridMap::serializeMessage, host code
// only: writes <dir>/gray.tif, <dir>/colour.tif, <dir>/cloud.ampc, <dir>/map.msg and the inputs
// they were made from (<dir>/inputs.bin) for tests/test_export_formats.py to parse.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "aerial-mapper-grid-map/aerial-mapper-grid-map.h"
#include "aerial-mapper-io/aerial-mapper-io.h"

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  const std::string dir = argv[1];
  unsigned s = 12345u;
  auto next = [&]() { s = s * 1664525u + 1013904223u; return (uint8_t)(s >> 24); };
  const int H = 19, W = 23;
  cv::Mat gray(H, W, 1), colour(H, W, 3);
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      gray.at<uint8_t>(y, x) = next();
      for (int c = 0; c < 3; ++c) colour.data[(size_t)y * colour.step + 3 * x + c] = next();
    }
  io::AerialMapperIO io_handler;
  io_handler.toGeoTiff(gray, Eigen::Vector2d(1.0, 2.0), dir + "/gray.tif");
  io_handler.writeDataToDEMGeoTiffColor(colour, Eigen::Vector2d(464736.27, 5272359.16),
                                        dir + "/colour.tif");
  AlignedType<std::vector, Eigen::Vector3d>::type xyz;
  std::vector<int> inten;
  for (int k = 0; k < 100; ++k) {
    xyz.push_back(Eigen::Vector3d(k * 0.5, -k * 0.25, 400.0 + k));
    inten.push_back(k % 256);
  }
  io_handler.savePointCloudToBinaryFile(dir + "/cloud.ampc", xyz, inten);

  grid_map::Settings st;
  st.center_easting = 10.0;
  st.center_northing = -4.0;
  st.delta_easting = 3.0;
  st.delta_northing = 2.0;
  st.resolution = 0.5;
  grid_map::AerialGridMap map(st);
  grid_map::GridMap* m = map.getMutable();
  (*m)["elevation"](2, 1) = 412.5f;
  (*m)["ortho"](0, 3) = 17.0f;
  const std::vector<uint8_t> msg = map.serializeMessage(1506593812123456789ull);
  std::ofstream(dir + "/map.msg", std::ios::binary).write((const char*)msg.data(), msg.size());
  std::ofstream in(dir + "/inputs.bin", std::ios::binary);
  in.write((const char*)gray.data, (size_t)H * W);
  in.write((const char*)colour.data, (size_t)H * W * 3);
  std::printf("ok %d %d\n", H, W);
  return 0;
}
