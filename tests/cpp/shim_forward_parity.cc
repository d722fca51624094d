// shim_forward_parity.cc -- drives the drop-in ortho::OrthoForwardHomography
// (over libaerial_mapper_hip.so) the way
// aerial_mapper_demos/src/ortho/main-ortho-forward-homography.cc:80-102 does --
// batch() when settings.batch, else one updateOrthomosaic() per frame -- and
// checks result_ / result_mask_ entry for entry against the CPU oracle
// (TEST ONLY: links oracle/liboracle.so).  Exit code 0 = parity.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <vector>

#include "aerial-mapper-ortho/ortho-forward-homography.h"

extern "C" {
#include "amo_types.h"
typedef struct amo_mosaic_desc {
  int32_t width_mosaic_pixels;
  int32_t height_mosaic_pixels;
  double ground_plane_elevation_m;
  double origin[3];
} amo_mosaic_desc;
void* amo_fwd_create(const amo_camera*, const amo_mosaic_desc*);
void amo_fwd_destroy(void*);
int amo_fwd_batch(void*, const double*, const double*, const void* const*, const size_t*, int,
                  size_t, int16_t*, uint8_t*);
int amo_fwd_update(void*, const double*, const double*, const void*, size_t, int, int16_t*,
                   uint8_t*);
}

static uint64_t g_state = 0x2545F4914F6CDD1DULL;
static double urand() {  // splitmix64 -> [0,1)
  uint64_t z = (g_state += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  z ^= z >> 31;
  return (z >> 11) * (1.0 / 9007199254740992.0);
}

int main(int argc, char** argv) {
  const bool incremental = argc > 1 && std::strcmp(argv[1], "incremental") == 0;
  const bool colored = argc > 2 && std::strcmp(argv[2], "colored") == 0;
  const int W = 160, H = 120, F = 8;
  aslam::Camera cam(120.0, 120.0, (W - 1) / 2.0, (H - 1) / 2.0, W, H);
  std::shared_ptr<aslam::NCamera> ncameras(new aslam::NCamera(
      cam, aslam::Transformation(kindr::minimal::RotationQuaternion(1, 0, 0, 0),
                                 Eigen::Vector3d(0.02, -0.01, 0.03))));
  ortho::Settings settings;
  settings.batch = !incremental;
  settings.ground_plane_elevation_m = 402.0;
  settings.width_mosaic_pixels = 360;
  settings.height_mosaic_pixels = 300;
  settings.origin = Eigen::Vector3d(5.0, -3.0, 0.0);
  settings.filename_mosaic_output = "";  // no file output in the test

  const double s45 = std::sqrt(0.5);
  Poses T_G_Bs;
  Images images;
  for (int f = 0; f < F; ++f) {
    const double px = 5.0 - 50.0 + 14.0 * f;
    const double py = -3.0 + ((f % 3) - 1) * 25.0;
    double q[4] = {0.01 * (f - 4), s45, s45 + 0.004 * f, 0.003 * (4 - f)};
    const double nq = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    T_G_Bs.push_back(Pose(kindr::minimal::RotationQuaternion(q[0] / nq, q[1] / nq, q[2] / nq, q[3] / nq),
                          Eigen::Vector3d(px, py, 500.0)));
    Image img(H, W, colored ? 3 : 1);
    for (size_t b = 0; b < static_cast<size_t>(H) * img.step; ++b) {
      const int v = static_cast<int>(urand() * 256.0);
      img.data[b] = static_cast<uint8_t>(v < 5 ? 0 : v);
    }
    images.push_back(img);
  }

  // ---- the demo's flow (main-ortho-forward-homography.cc:93-102) ------------------
  ortho::OrthoForwardHomography mosaic(ncameras, settings);
  if (settings.batch) {
    mosaic.batch(T_G_Bs, images);
  } else {
    for (size_t i = 0u; i < images.size(); ++i) mosaic.updateOrthomosaic(T_G_Bs[i], images[i]);
  }

  // ---- oracle on the same inputs ------------------------------------------------
  amo_camera oc;
  std::memset(&oc, 0, sizeof(oc));
  oc.fu = oc.fv = 120.0;
  oc.cu = (W - 1) / 2.0;
  oc.cv = (H - 1) / 2.0;
  oc.width = W;
  oc.height = H;
  amo_mosaic_desc od;
  od.width_mosaic_pixels = 360;
  od.height_mosaic_pixels = 300;
  od.ground_plane_elevation_m = 402.0;
  od.origin[0] = 5.0;
  od.origin[1] = -3.0;
  od.origin[2] = 0.0;
  void* h = amo_fwd_create(&oc, &od);
  if (!h) return 2;
  std::vector<double> tgb(7 * F);
  std::vector<const void*> ptrs(F);
  std::vector<size_t> steps(F);
  for (int f = 0; f < F; ++f) {
    const Eigen::Vector3d& t = T_G_Bs[f].getPosition();
    const Eigen::Quaterniond& q = T_G_Bs[f].getRotation().toImplementation();
    double* o = &tgb[7 * f];
    o[0] = t(0); o[1] = t(1); o[2] = t(2); o[3] = q.w(); o[4] = q.x(); o[5] = q.y(); o[6] = q.z();
    ptrs[f] = images[f].data;
    steps[f] = images[f].step;
  }
  const double tcb[7] = {0.02, -0.01, 0.03, 1, 0, 0, 0};
  const size_t n = 360u * 300u;
  std::vector<int16_t> want(3 * n);
  std::vector<uint8_t> want_mask(n);
  int rc = 0;
  if (settings.batch) {
    rc = amo_fwd_batch(h, tgb.data(), tcb, ptrs.data(), steps.data(), colored ? 3 : 1, F,
                       want.data(), want_mask.data());
  } else {
    for (int f = 0; f < F && !rc; ++f)
      rc = amo_fwd_update(h, &tgb[7 * f], tcb, ptrs[f], steps[f], colored ? 3 : 1, want.data(),
                          want_mask.data());
  }
  amo_fwd_destroy(h);
  if (rc) return 3;

  size_t bad = 0, bad_mask = 0, covered = 0;
  for (size_t k = 0; k < 3 * n; ++k) bad += mosaic.result()[k] != want[k];
  for (size_t k = 0; k < n; ++k) {
    bad_mask += mosaic.result_mask()[k] != want_mask[k];
    covered += want_mask[k] != 0;
  }
  std::printf("forward mosaic (%s, %s): %zu / %zu values differ, %zu / %zu mask pixels differ, "
              "coverage %.3f\n", incremental ? "incremental" : "batch", colored ? "colored" : "gray",
              bad, 3 * n, bad_mask, n, (double)covered / n);
  const cv::Mat img8 = mosaic.result8();
  bool ok = bad == 0 && bad_mask == 0 && covered > n / 10 && img8.rows == 300 && img8.cols == 360;
  std::printf(ok ? "PARITY OK\n" : "PARITY FAILED\n");
  return ok ? 0 : 1;
}
