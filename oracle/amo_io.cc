/*
 * oracle/amo_io.cc -- TEST INFRASTRUCTURE ONLY (CPU oracle for the text
 * formats of io::AerialMapperIO).
 *
 * The reference reads its point clouds and poses with std::ifstream and
 * operator>> (aerial_mapper_io/src/aerial-mapper-io.cc):
 *   loadPointCloudFromFile (xyz)              :309-325
 *   loadPointCloudFromFile (xyz + intensity)  :327-347
 *   loadPosesFromFileStandard                 :103-121
 * This file runs THE SAME extraction loops on an in-memory stream (a
 * std::istream over the caller's buffer), so tokenisation, number syntax,
 * rounding (glibc strtod: correctly rounded) and the stop-at-first-failure
 * behaviour are the C++ library's own -- nothing is restated.  PINNED by
 * construction for every input, including malformed ones.
 */
#include <cstdint>
#include <cstring>
#include <istream>
#include <streambuf>

namespace {

struct MemBuf : std::streambuf {
  MemBuf(const char* p, size_t n) {
    char* b = const_cast<char*>(p);
    setg(b, b, b + n);
  }
};

}  // namespace

extern "C" {

/* Both overloads of loadPointCloudFromFile (:309-347).  xyz: 3 * cap doubles,
 * intensities: cap ints (may be null = the xyz-only overload).  Returns the
 * number of points the reference would have pushed (the caller sizes `cap`
 * generously; points beyond it are counted but not stored). */
size_t amo_io_load_point_cloud(const char* text, size_t len, double* xyz, int32_t* intensities,
                               size_t cap) {
  MemBuf buf(text, len);
  std::istream infile(&buf);
  size_t n = 0;
  double x, y, z;
  int intensity;
  while (infile >> x >> y >> z >> intensity) {
    if (z > -100) {
      if (n < cap) {
        xyz[3 * n + 0] = x;
        xyz[3 * n + 1] = y;
        xyz[3 * n + 2] = z;
        if (intensities) intensities[n] = intensity;
      }
      ++n;
    }
    if (infile.eof()) break;
  }
  return n;
}

/* loadPosesFromFileStandard (:103-121): records x y z qw qx qy qz -> the
 * 7-double layout of the C boundary (tx,ty,tz,qw,qx,qy,qz). */
size_t amo_io_load_poses(const char* text, size_t len, double* poses7, size_t cap) {
  MemBuf buf(text, len);
  std::istream infile(&buf);
  size_t n = 0;
  double x, y, z, qw, qx, qy, qz;
  while (infile >> x >> y >> z >> qw >> qx >> qy >> qz) {
    if (n < cap) {
      double* o = poses7 + 7 * n;
      o[0] = x; o[1] = y; o[2] = z; o[3] = qw; o[4] = qx; o[5] = qy; o[6] = qz;
    }
    ++n;
    if (infile.eof()) break;
  }
  return n;
}

}  // extern "C"
