// amhip_export.hip -- the formats BEHIND the hot path:
//   * layer -> 8-bit image, as grid_map_cv::GridMapCvConverter::toImage<unsigned char, 1> builds
//     it (the include of aerial_mapper_grid_map/src/aerial-mapper-grid-map.cc:10), and the packed
//     colour layer -> BGR image: transposition kernels, the image leaves HBM already in the
//     raster order the GeoTiff writers want;
//   * the GeoTiff container of io::AerialMapperIO::toGeoTiff / writeDataToDEMGeoTiffColor
//     (aerial_mapper_io/src/aerial-mapper-io.cc:349-509) without GDAL: baseline TIFF + the
//     GeoTIFF tags of a north-up UTM raster;
//   * the grid_map_msgs/GridMap message of AerialGridMap::publishOnce / publishUntilShutdown
//     (aerial_mapper_grid_map/src/aerial-mapper-grid-map.cc:51-72) in ROS 1 wire format: its
//     layout here, the device -> message copies in amhip_session.hip;
//   * a binary point-cloud file (the reference has only the text format, :309-347) staged through
//     pinned double buffers.
// grid_map_cv, grid_map_ros, GDAL and roscpp are not in the reference tree: the adopted
// definitions are restated by the test oracle ("parity unpinned", like the other
// external-library formulas).
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "amhip_common.h"

namespace amhip {

// ---------------------------------------------------------------------------
// layer (column-major f32: (i, j) at i + j * rows) -> image (row-major: row i, column j)
// ---------------------------------------------------------------------------
// toImage(): image = zeros(size(0), size(1)); the layer is clamped to [lower, upper]; every cell
// with a finite value becomes (uchar)(((v - lower) / (upper - lower)) * (float)255) -- float
// arithmetic, truncating cast; the others stay 0.
__device__ __forceinline__ uint8_t to_image_u8(float v, float lower, float upper) {
  if (!isfinite(v)) return 0;
  v = fminf(fmaxf(v, lower), upper);
  const float t = ((v - lower) / (upper - lower)) * 255.0f;
  return (uint8_t)(int)t;
}

// One workgroup = 64 x 64 cells through LDS: lanes run along i on the way in (coalesced 4-byte
// reads of a layer column) and along j on the way out (coalesced bytes of an image row).
// kBgr: the layer holds grid_map's packed colours (bits R << 16 | G << 8 | B,
// ortho-backward-grid.cc:203-207); the image is 8UC3 in OpenCV's B, G, R order; NaN -> 0, 0, 0.
template <bool kBgr>
__global__ void __launch_bounds__(256)
k_layer_to_image(const float* __restrict__ layer, int rows, int cols, float lower, float upper,
                 uint8_t* __restrict__ image, size_t step) {
  __shared__ uint32_t s_tile[64][65];
  const int ti = blockIdx.x * 64, tj = blockIdx.y * 64;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int jj = w; jj < 64; jj += 4) {
    const int i = ti + lane, j = tj + jj;
    uint32_t out = 0;
    if (i < rows && j < cols) {
      const float v = layer[(size_t)i + (size_t)j * (size_t)rows];
      if (kBgr) out = (v == v) ? __float_as_uint(v) : 0u;
      else out = to_image_u8(v, lower, upper);
    }
    s_tile[jj][lane] = out;
  }
  __syncthreads();
  if (!kBgr && (step & 3u) == 0 && (reinterpret_cast<size_t>(image) & 3u) == 0) {
    // gray: a thread packs four neighbouring pixels of an image row into one 4-byte store
    // (16 threads per row of the tile, 16 rows per sweep)
    const int g = threadIdx.x & 15, r0 = threadIdx.x >> 4;
    for (int ii = r0; ii < 64; ii += 16) {
      const int i = ti + ii, j = tj + 4 * g;
      if (i >= rows || j >= cols) continue;
      uint8_t* px = image + (size_t)i * step + j;
      if (j + 3 < cols) {
        const uint32_t v = s_tile[4 * g][ii] | (s_tile[4 * g + 1][ii] << 8) |
                           (s_tile[4 * g + 2][ii] << 16) | (s_tile[4 * g + 3][ii] << 24);
        *reinterpret_cast<uint32_t*>(px) = v;
      } else {
        for (int q = 0; j + q < cols; ++q) px[q] = (uint8_t)s_tile[4 * g + q][ii];
      }
    }
    return;
  }
  for (int ii = w; ii < 64; ii += 4) {
    const int i = ti + ii, j = tj + lane;
    if (i < rows && j < cols) {
      const uint32_t v = s_tile[lane][ii];
      uint8_t* px = image + (size_t)i * step;
      if (kBgr) {
        px += (size_t)j * 3u;
        px[0] = (uint8_t)(v & 0xFFu);
        px[1] = (uint8_t)((v >> 8) & 0xFFu);
        px[2] = (uint8_t)((v >> 16) & 0xFFu);
      } else {
        px[j] = (uint8_t)v;
      }
    }
  }
}

int layer_to_image_dev(Ctx* c, int layer, int bgr, float lower, float upper, uint8_t* dev_image,
                       size_t step) {
  int rc = ctx_use_device(c);
  if (rc) return rc;
  if ((rc = ctx_materialize(c, layer))) return rc;
  const dim3 grid((c->win_rows + 63) / 64, (c->win_cols + 63) / 64);
  if (bgr)
    hipLaunchKernelGGL(k_layer_to_image<true>, grid, dim3(256), 0, c->stream, c->layers[layer],
                       c->win_rows, c->win_cols, lower, upper, dev_image, step);
  else
    hipLaunchKernelGGL(k_layer_to_image<false>, grid, dim3(256), 0, c->stream, c->layers[layer],
                       c->win_rows, c->win_cols, lower, upper, dev_image, step);
  AMHIP_TRY(hipGetLastError());
  return AMHIP_OK;
}

// ---------------------------------------------------------------------------
// GeoTIFF (little-endian classic TIFF, one IFD, uncompressed strips)
// ---------------------------------------------------------------------------
namespace {

struct TiffTag {
  uint16_t tag, type;
  uint32_t count;
  uint32_t value;  // the value, or the file offset of the values
};

enum { kShort = 3, kLong = 4, kAscii = 2, kDouble = 12 };

template <typename T>
void put(std::vector<uint8_t>* b, T v) {
  const uint8_t* p = reinterpret_cast<const uint8_t*>(&v);
  b->insert(b->end(), p, p + sizeof(T));
}

bool write_all(int fd, const void* data, size_t n) {
  const uint8_t* p = static_cast<const uint8_t*>(data);
  while (n) {
    const ssize_t w = ::write(fd, p, std::min<size_t>(n, (size_t)1 << 30));
    if (w < 0) {
      if (errno == EINTR) continue;
      return false;
    }
    p += w;
    n -= (size_t)w;
  }
  return true;
}

}  // namespace

}  // namespace amhip

using namespace amhip;

extern "C" {

int amhip_layer_to_image_dev(amhip_ctx* h, int layer, int bgr, float lower, float upper,
                             uint8_t* dev_image, size_t step) {
  if (!h || !dev_image || layer < 0 || layer >= AMHIP_NUM_LAYERS)
    return arg_failure("amhip_layer_to_image_dev: bad argument");
  Ctx* c = &h->impl;
  if (step < (size_t)c->win_cols * (bgr ? 3u : 1u))
    return arg_failure("amhip_layer_to_image_dev: step smaller than an image row");
  if (!bgr && !(upper > lower)) return arg_failure("amhip_layer_to_image_dev: upper <= lower");
  return layer_to_image_dev(c, layer, bgr, lower, upper, dev_image, step);
}

int amhip_layer_to_image(amhip_ctx* h, int layer, int bgr, float lower, float upper,
                         uint8_t* host_image, size_t step) {
  if (!h || !host_image || layer < 0 || layer >= AMHIP_NUM_LAYERS)
    return arg_failure("amhip_layer_to_image: bad argument");
  Ctx* c = &h->impl;
  const size_t row = (size_t)c->win_cols * (bgr ? 3u : 1u);
  if (step < row) return arg_failure("amhip_layer_to_image: step smaller than an image row");
  if (!bgr && !(upper > lower)) return arg_failure("amhip_layer_to_image: upper <= lower");
  int rc = ctx_use_device(c);
  if (rc) return rc;
  uint8_t* dev = nullptr;
  AMHIP_TRY(hipMalloc(reinterpret_cast<void**>(&dev), row * (size_t)c->win_rows));
  rc = layer_to_image_dev(c, layer, bgr, lower, upper, dev, row);
  if (rc == AMHIP_OK) {
    hipError_t e = hipMemcpy2DAsync(host_image, step, dev, row, row, (size_t)c->win_rows,
                                    hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) rc = hip_fail(e, "image download", __FILE__, __LINE__);
  }
  (void)hipFree(dev);
  return rc;
}

/* Host only. */
int amhip_geotiff_write_u8(const char* filename, const uint8_t* pixels, int width, int height,
                           size_t step, int bands, const double* geotransform, int utm_zone,
                           int northern) {
  if (!filename || !pixels || width <= 0 || height <= 0 || (bands != 1 && bands != 3) ||
      !geotransform || utm_zone < 1 || utm_zone > 60)
    return arg_failure("amhip_geotiff_write_u8: bad argument");
  const size_t row = (size_t)width * (size_t)bands;
  if (step < row) return arg_failure("amhip_geotiff_write_u8: step smaller than a row");
  if (geotransform[2] != 0.0 || geotransform[4] != 0.0 || !(geotransform[1] > 0.0) ||
      !(geotransform[5] < 0.0))
    return arg_failure("amhip_geotiff_write_u8: only north-up geotransforms (GT2 = GT4 = 0, "
                       "GT1 > 0, GT5 < 0)");
  const unsigned long long payload = (unsigned long long)row * (unsigned long long)height;
  if (payload > 0xFFF00000ull)
    return arg_failure("amhip_geotiff_write_u8: raster beyond classic TIFF's 4 GB");
  // strips of <= 64 MB
  const uint32_t rows_per_strip =
      (uint32_t)std::min<unsigned long long>((unsigned long long)height,
                                             std::max<unsigned long long>(1ull, (64ull << 20) / row));
  const uint32_t nstrips = ((uint32_t)height + rows_per_strip - 1) / rows_per_strip;

  // what SetProjCS(..) + SetWellKnownGeogCS("WGS84") + SetUTM(zone, north) describe
  // (aerial-mapper-io.cc:390-394, :467-474): EPSG 326zz / 327zz
  char citation[96];
  std::snprintf(citation, sizeof(citation), "UTM %d (WGS84) in %s hemisphere.|", utm_zone,
                northern ? "northern" : "southern");
  const std::string ascii = std::string(citation) + "WGS 84|";
  const uint16_t cit_len = (uint16_t)std::strlen(citation);
  const uint16_t geokeys[] = {
      1, 1, 0, 7,                                          // version, revision, minor, keys
      1024, 0, 1, 1,                                       // GTModelType = projected
      1025, 0, 1, 1,                                       // GTRasterType = PixelIsArea
      1026, 34737, cit_len, 0,                             // GTCitation
      2048, 0, 1, 4326,                                    // GeographicType = WGS 84
      2049, 34737, 7, cit_len,                             // GeogCitation "WGS 84|"
      3072, 0, 1, (uint16_t)((northern ? 32600 : 32700) + utm_zone),  // ProjectedCSType
      3076, 0, 1, 9001,                                    // ProjLinearUnits = metre
  };
  const double scale[3] = {geotransform[1], -geotransform[5], 0.0};
  const double tie[6] = {0.0, 0.0, 0.0, geotransform[0], geotransform[3], 0.0};

  // ---- layout: header, IFD, out-of-line values, strips ----
  const int ntags = 15;
  const uint32_t ifd_at = 8;
  uint32_t extra_at = ifd_at + 2 + ntags * 12 + 4;
  std::vector<uint8_t> extra;
  auto reserve = [&](const void* data, size_t n) {
    const uint32_t at = extra_at + (uint32_t)extra.size();
    const uint8_t* p = static_cast<const uint8_t*>(data);
    extra.insert(extra.end(), p, p + n);
    if (extra.size() & 1) extra.push_back(0);  // word alignment
    return at;
  };
  const uint16_t bits3[3] = {8, 8, 8}, fmt3[3] = {1, 1, 1};
  const uint32_t bits_at = bands == 3 ? reserve(bits3, 6) : 0;
  const uint32_t fmt_at = bands == 3 ? reserve(fmt3, 6) : 0;
  const uint32_t scale_at = reserve(scale, sizeof(scale));
  const uint32_t tie_at = reserve(tie, sizeof(tie));
  const uint32_t keys_at = reserve(geokeys, sizeof(geokeys));
  const uint32_t ascii_at = reserve(ascii.c_str(), ascii.size() + 1);
  // strip tables (out of line when there is more than one strip)
  const uint32_t offs_at = nstrips > 1 ? extra_at + (uint32_t)extra.size() : 0;
  if (nstrips > 1) extra.resize(extra.size() + 4u * nstrips);
  const uint32_t cnts_at = nstrips > 1 ? extra_at + (uint32_t)extra.size() : 0;
  if (nstrips > 1) extra.resize(extra.size() + 4u * nstrips);
  const uint32_t data_at = (extra_at + (uint32_t)extra.size() + 15u) & ~15u;
  std::vector<uint32_t> soff(nstrips), scnt(nstrips);
  for (uint32_t s = 0; s < nstrips; ++s) {
    const uint32_t r0 = s * rows_per_strip;
    const uint32_t nr = std::min(rows_per_strip, (uint32_t)height - r0);
    soff[s] = data_at + (uint32_t)((unsigned long long)r0 * row);
    scnt[s] = (uint32_t)((unsigned long long)nr * row);
  }
  if (nstrips > 1) {
    std::memcpy(extra.data() + (offs_at - extra_at), soff.data(), 4u * nstrips);
    std::memcpy(extra.data() + (cnts_at - extra_at), scnt.data(), 4u * nstrips);
  }
  const TiffTag tags[ntags] = {
      {256, kLong, 1, (uint32_t)width},
      {257, kLong, 1, (uint32_t)height},
      {258, kShort, (uint32_t)bands, bands == 3 ? bits_at : 8u},
      {259, kShort, 1, 1},                       // no compression
      {262, kShort, 1, bands == 3 ? 2u : 1u},    // RGB / BlackIsZero
      {273, kLong, nstrips, nstrips > 1 ? offs_at : soff[0]},
      {277, kShort, 1, (uint32_t)bands},
      {278, kLong, 1, rows_per_strip},
      {279, kLong, nstrips, nstrips > 1 ? cnts_at : scnt[0]},
      {284, kShort, 1, 1},                       // pixel interleaved
      {339, kShort, (uint32_t)bands, bands == 3 ? fmt_at : 1u},
      {33550, kDouble, 3, scale_at},
      {33922, kDouble, 6, tie_at},
      {34735, kShort, (uint32_t)(sizeof(geokeys) / 2), keys_at},
      {34737, kAscii, (uint32_t)ascii.size() + 1, ascii_at},
  };
  std::vector<uint8_t> head;
  head.push_back('I');
  head.push_back('I');
  put<uint16_t>(&head, 42);
  put<uint32_t>(&head, ifd_at);
  put<uint16_t>(&head, (uint16_t)ntags);
  for (const TiffTag& t : tags) {
    put<uint16_t>(&head, t.tag);
    put<uint16_t>(&head, t.type);
    put<uint32_t>(&head, t.count);
    put<uint32_t>(&head, t.value);
  }
  put<uint32_t>(&head, 0);  // no further IFD
  head.insert(head.end(), extra.begin(), extra.end());
  head.resize(data_at, 0);

  const int fd = ::open(filename, O_WRONLY | O_CREAT | O_TRUNC, 0644);
  if (fd < 0) {
    set_last_error(std::string("amhip_geotiff_write_u8: cannot open ") + filename + ": " +
                   std::strerror(errno));
    return AMHIP_ERR_ARG;
  }
  bool ok = write_all(fd, head.data(), head.size());
  if (ok) {
    if (step == row) {
      ok = write_all(fd, pixels, (size_t)payload);
    } else {
      for (int r = 0; r < height && ok; ++r) ok = write_all(fd, pixels + (size_t)r * step, row);
    }
  }
  if (::close(fd) != 0) ok = false;
  if (!ok) {
    set_last_error(std::string("amhip_geotiff_write_u8: write failed: ") + std::strerror(errno));
    return AMHIP_ERR_ARG;
  }
  return AMHIP_OK;
}

/* ---- grid_map_msgs/GridMap, ROS 1 wire format: layout -------------------------------------
 * info.header (uint32 seq, uint32 sec, uint32 nsec, string frame_id), float64 resolution,
 * length_x, length_y, pose (position xyz, orientation xyzw), string[] layers, string[]
 * basic_layers, Float32MultiArray[] data (layout.dim[2] {label, size, stride}, data_offset,
 * float32[] data), uint16 outer_start_index, inner_start_index.  A string / array is a uint32
 * count followed by its elements, everything little-endian, nothing padded. */
size_t amhip_grid_map_msg_bytes(const amhip_grid_desc* grid, const char* frame_id, int num_layers,
                                const char* const* layer_names) {
  if (!grid || !frame_id || num_layers < 0 || (num_layers && !layer_names)) return 0;
  size_t n = 4 + 4 + 4 + 4 + std::strlen(frame_id);  // header
  n += 3 * 8 + 7 * 8;                                // resolution, lengths, pose
  n += 4;                                            // layers[]
  for (int l = 0; l < num_layers; ++l) n += 4 + std::strlen(layer_names[l]);
  n += 4;                                            // basic_layers[] (empty)
  n += 4;                                            // data[]
  const size_t per = 4 + (4 + 12 + 4 + 4) + (4 + 9 + 4 + 4) + 4 + 4 +
                     4 * (size_t)grid->rows * (size_t)grid->cols;
  n += per * (size_t)num_layers;
  n += 2 + 2;
  return n;
}

/* Writes everything but the layers' float payloads; payload_offsets[l] = where layer l's
 * rows * cols floats (column-major, the Eigen matrix as it lies in memory) go. */
int amhip_grid_map_msg_layout(const amhip_grid_desc* grid, uint64_t stamp_ns, const char* frame_id,
                              int num_layers, const char* const* layer_names, uint8_t* out,
                              size_t cap, size_t* payload_offsets) {
  const size_t need = amhip_grid_map_msg_bytes(grid, frame_id, num_layers, layer_names);
  if (!need || !out || !payload_offsets || cap < need)
    return arg_failure("amhip_grid_map_msg_layout: bad argument or buffer too small");
  uint8_t* p = out;
  auto u32 = [&](uint32_t v) { std::memcpy(p, &v, 4); p += 4; };
  auto f64 = [&](double v) { std::memcpy(p, &v, 8); p += 8; };
  auto str = [&](const char* s) {
    const uint32_t n = (uint32_t)std::strlen(s);
    u32(n);
    std::memcpy(p, s, n);
    p += n;
  };
  u32(0);                                           // seq (the publisher's business)
  u32((uint32_t)(stamp_ns / 1000000000ull));        // ros::Time::fromNSec
  u32((uint32_t)(stamp_ns % 1000000000ull));
  str(frame_id);
  f64(grid->resolution);
  f64(grid->length_x);
  f64(grid->length_y);
  f64(grid->pos_x);
  f64(grid->pos_y);
  f64(0.0);
  f64(0.0);
  f64(0.0);
  f64(0.0);
  f64(1.0);
  u32((uint32_t)num_layers);
  for (int l = 0; l < num_layers; ++l) str(layer_names[l]);
  u32(0);  // basic_layers
  u32((uint32_t)num_layers);
  const uint32_t rows = (uint32_t)grid->rows, cols = (uint32_t)grid->cols;
  for (int l = 0; l < num_layers; ++l) {
    // matrixEigenCopyToMultiArrayMessage for a column-major matrix: the outer dimension first
    u32(2);
    str("column_index");
    u32(cols);
    u32(rows * cols);
    str("row_index");
    u32(rows);
    u32(rows);
    u32(0);  // data_offset
    u32(rows * cols);
    payload_offsets[l] = (size_t)(p - out);
    p += 4 * (size_t)rows * (size_t)cols;
  }
  // getStartIndex(): (0, 0) for every map the reference makes (the drop-in refuses moved maps)
  const uint16_t start[2] = {0, 0};
  std::memcpy(p, start, 4);
  p += 4;
  if ((size_t)(p - out) != need) return arg_failure("amhip_grid_map_msg_layout: internal size mismatch");
  return AMHIP_OK;
}

/* ---- binary point clouds ---------------------------------------------------------------------
 * "AMPCLD01" | uint64 n | uint32 flags (bit 0: intensities follow) | uint32 0 | uint64 0 |
 * n x (x, y, z) float64 | n x int32.  What loadPointCloudFromFile leaves in its vectors, as it
 * lies in memory (points with z <= -100 already dropped by whoever wrote the file). */
static const char kCloudMagic[8] = {'A', 'M', 'P', 'C', 'L', 'D', '0', '1'};

int amhip_io_write_point_cloud_binary(const char* filename, const double* host_xyz,
                                      const int32_t* host_intensities, size_t n) {
  if (!filename || (!host_xyz && n)) return arg_failure("amhip_io_write_point_cloud_binary: bad argument");
  const int fd = ::open(filename, O_WRONLY | O_CREAT | O_TRUNC, 0644);
  if (fd < 0) {
    set_last_error(std::string("cannot open ") + filename + ": " + std::strerror(errno));
    return AMHIP_ERR_ARG;
  }
  uint8_t head[32] = {};
  std::memcpy(head, kCloudMagic, 8);
  const uint64_t n64 = n;
  const uint32_t flags = host_intensities ? 1u : 0u;
  std::memcpy(head + 8, &n64, 8);
  std::memcpy(head + 16, &flags, 4);
  bool ok = write_all(fd, head, 32) && write_all(fd, host_xyz, 24 * n) &&
            (!host_intensities || write_all(fd, host_intensities, 4 * n));
  if (::close(fd) != 0) ok = false;
  if (!ok) {
    set_last_error(std::string("amhip_io_write_point_cloud_binary: write failed: ") + std::strerror(errno));
    return AMHIP_ERR_ARG;
  }
  return AMHIP_OK;
}

/* The file goes to HBM through two pinned staging buffers: while one is on its way over the link
 * (hipMemcpyAsync), read() fills the other.  *dev_intensities = NULL when the file has none. */
int amhip_io_load_point_cloud_binary(int device, const char* filename, double** dev_xyz,
                                     int32_t** dev_intensities, size_t* num_points) {
  if (!filename || !dev_xyz || !dev_intensities || !num_points)
    return arg_failure("amhip_io_load_point_cloud_binary: null argument");
  *dev_xyz = nullptr;
  *dev_intensities = nullptr;
  *num_points = 0;
  const int fd = ::open(filename, O_RDONLY);
  if (fd < 0) {
    set_last_error(std::string("cannot open ") + filename + ": " + std::strerror(errno));
    return AMHIP_ERR_ARG;
  }
  struct Closer {
    int fd;
    ~Closer() { ::close(fd); }
  } closer{fd};
  uint8_t head[32];
  if (::read(fd, head, 32) != 32 || std::memcmp(head, kCloudMagic, 8) != 0)
    return arg_failure("amhip_io_load_point_cloud_binary: not an AMPCLD01 file");
  uint64_t n64;
  uint32_t flags;
  std::memcpy(&n64, head + 8, 8);
  std::memcpy(&flags, head + 16, 4);
  struct stat st;
  // the header is untrusted: bound n by the file's size BEFORE anything is multiplied (2^61 points
  // would wrap 24 * n to a small number), refuse counts the DSM cannot index and unknown flags
  if (flags & ~1u) return arg_failure("amhip_io_load_point_cloud_binary: unknown flag bits in the header");
  if (::fstat(fd, &st) != 0 || st.st_size < 32)
    return arg_failure("amhip_io_load_point_cloud_binary: file shorter than its header says");
  const unsigned long long per_point = (flags & 1u) ? 28ull : 24ull;
  if (n64 > ((unsigned long long)st.st_size - 32ull) / per_point)
    return arg_failure("amhip_io_load_point_cloud_binary: file shorter than its header says");
  if (n64 >= 0x7FFFFFFFull)
    return arg_failure("amhip_io_load_point_cloud_binary: more than 2^31-1 points (the reference "
                       "indexes search results with int)");
  const size_t n = (size_t)n64;
  if (n == 0) return AMHIP_OK;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) {
    set_last_error("amhip_io_load_point_cloud_binary: no such device");
    return AMHIP_ERR_NO_DEVICE;
  }
  AMHIP_TRY(hipSetDevice(device));
  const size_t kStage = (size_t)32 << 20;
  uint8_t* stage[2] = {nullptr, nullptr};
  hipEvent_t done[2] = {nullptr, nullptr};
  hipStream_t stream = nullptr;
  double* dxyz = nullptr;
  int32_t* dint = nullptr;
  int rc = AMHIP_OK;
  auto fail = [&](hipError_t e, const char* what) { rc = hip_fail(e, what, __FILE__, __LINE__); };
  hipError_t e;
  if ((e = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking)) != hipSuccess) fail(e, "stream");
  for (int b = 0; b < 2 && rc == AMHIP_OK; ++b) {
    if ((e = hipHostMalloc(reinterpret_cast<void**>(&stage[b]), kStage, hipHostMallocDefault)) != hipSuccess)
      fail(e, "hipHostMalloc");
    else if ((e = hipEventCreateWithFlags(&done[b], hipEventDisableTiming)) != hipSuccess)
      fail(e, "event");
  }
  if (rc == AMHIP_OK && (e = hipMalloc(reinterpret_cast<void**>(&dxyz), 24 * n)) != hipSuccess) fail(e, "hipMalloc");
  if (rc == AMHIP_OK && (flags & 1u) &&
      (e = hipMalloc(reinterpret_cast<void**>(&dint), 4 * n)) != hipSuccess)
    fail(e, "hipMalloc");
  // the two sections of the file, chunk by chunk (a staging buffer is refilled only after the
  // transfer that last read it has completed -- across the sections too)
  int b = 0;
  bool used[2] = {false, false};
  size_t file_at = 32;  // behind the header
  for (int sec = 0; sec < 2 && rc == AMHIP_OK; ++sec) {
    if (sec == 1 && !(flags & 1u)) break;
    uint8_t* dst = sec == 0 ? reinterpret_cast<uint8_t*>(dxyz) : reinterpret_cast<uint8_t*>(dint);
    size_t left = sec == 0 ? 24 * n : 4 * n;
    while (left && rc == AMHIP_OK) {
      const size_t chunk = std::min(left, kStage);
      if (used[b] && (e = hipEventSynchronize(done[b])) != hipSuccess) {
        fail(e, "event wait");
        break;
      }
      // (one core copies out of the page cache at ~20 GB/s, the link takes 56: eight readers)
      {
        const int kReaders = 8;
        bool ok[kReaders];
        std::thread th[kReaders];
        const size_t part = (chunk + kReaders - 1) / kReaders;
        for (int t = 0; t < kReaders; ++t) {
          ok[t] = true;
          const size_t a = std::min(chunk, t * part), z = std::min(chunk, a + part);
          uint8_t* dstp = stage[b] + a;
          const off_t at = (off_t)(file_at + a);
          bool* okp = &ok[t];
          th[t] = std::thread([=]() {
            size_t got = 0;
            while (got < z - a) {
              const ssize_t r = ::pread(fd, dstp + got, z - a - got, at + (off_t)got);
              if (r < 0 && errno == EINTR) continue;
              if (r <= 0) {
                *okp = false;
                return;
              }
              got += (size_t)r;
            }
          });
        }
        for (int t = 0; t < kReaders; ++t) th[t].join();
        for (int t = 0; t < kReaders; ++t)
          if (!ok[t]) rc = arg_failure("amhip_io_load_point_cloud_binary: read failed");
      }
      if (rc != AMHIP_OK) break;
      file_at += chunk;
      if ((e = hipMemcpyAsync(dst, stage[b], chunk, hipMemcpyHostToDevice, stream)) != hipSuccess ||
          (e = hipEventRecord(done[b], stream)) != hipSuccess) {
        fail(e, "copy");
        break;
      }
      used[b] = true;
      dst += chunk;
      left -= chunk;
      b ^= 1;
    }
  }
  if (stream) {
    if ((e = hipStreamSynchronize(stream)) != hipSuccess && rc == AMHIP_OK) fail(e, "sync");
  }
  for (int b = 0; b < 2; ++b) {
    if (done[b]) (void)hipEventDestroy(done[b]);
    if (stage[b]) (void)hipHostFree(stage[b]);
  }
  if (stream) (void)hipStreamDestroy(stream);
  if (rc != AMHIP_OK) {
    if (dxyz) (void)hipFree(dxyz);
    if (dint) (void)hipFree(dint);
    return rc;
  }
  *dev_xyz = dxyz;
  *dev_intensities = dint;
  *num_points = n;
  return AMHIP_OK;
}

}  // extern "C"
