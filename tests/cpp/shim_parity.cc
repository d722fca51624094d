// shim_parity.cc -- drives the drop-in C++ classes (dsm::Dsm,
// ortho::OrthoBackwardGrid over libaerial_mapper_hip.so) exactly the way
// aerial_mapper_demos/src/ortho/main-ortho-backward-grid.cc:118-141 does, and
// checks every layer cell-for-cell against the CPU oracle (TEST ONLY: links
// oracle/liboracle.so).  Exit code 0 = parity.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <vector>

#include "aerial-mapper-dsm/dsm.h"
#include "aerial-mapper-grid-map/aerial-mapper-grid-map.h"
#include "aerial-mapper-ortho/ortho-backward-grid.h"

extern "C" {
#include "amo_types.h"
int amo_dsm_process(const double*, size_t, const amo_grid*, int, double, double, int, int,
                    float*, double*);
int amo_ortho_backward_process(const amo_grid*, const amo_camera*, const double*, const double*,
                               const uint8_t* const*, const size_t*, int, size_t, int, int, int,
                               const float*, float*, float*, float*, float*, float*);
void amo_make_grid(double, double, double, double, double, amo_grid*);
}

static uint64_t g_state = 0x853c49e6748fea9bULL;
static double urand() {  // splitmix64 -> [0,1)
  uint64_t z = (g_state += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  z ^= z >> 31;
  return (z >> 11) * (1.0 / 9007199254740992.0);
}

static bool same_bits(const float* a, const float* b, size_t n, const char* name) {
  size_t bad = 0;
  for (size_t i = 0; i < n; ++i) {
    uint32_t x, y;
    std::memcpy(&x, a + i, 4);
    std::memcpy(&y, b + i, 4);
    if (x != y && !(std::isnan(a[i]) && std::isnan(b[i]))) ++bad;
  }
  std::printf("  %-18s %zu / %zu cells differ\n", name, bad, n);
  return bad == 0;
}

int main(int argc, char** argv) {
  const bool colored = argc > 1 && std::strcmp(argv[1], "colored") == 0;
  grid_map::Settings gs;
  gs.center_easting = 12.0;
  gs.center_northing = -7.0;
  gs.delta_easting = 120.0;
  gs.delta_northing = 90.0;
  gs.resolution = 0.5;
  grid_map::AerialGridMap map(gs);
  grid_map::GridMap* m = map.getMutable();
  const int rows = m->getSize()(0), cols = m->getSize()(1);
  const size_t cells = static_cast<size_t>(rows) * cols;

  // point cloud: ~4 pts/m^2 over the map (+ margin), smooth terrain + noise.
  // dsm.cc:42-43 subtracts center_NORTHING from x and center_EASTING from y,
  // so the cloud is placed where that quirk maps it onto the grid.
  AlignedType<std::vector, Eigen::Vector3d>::type cloud;
  const size_t n = 4 * 130 * 100;
  for (size_t k = 0; k < n; ++k) {
    const double x = gs.center_easting + (urand() - 0.5) * 130.0;
    const double y = gs.center_northing + (urand() - 0.5) * 100.0;
    const double z = 400.0 + 6.0 * std::sin(0.05 * x) * std::cos(0.04 * y) + 0.05 * urand();
    cloud.push_back(Eigen::Vector3d(x + gs.center_northing, y + gs.center_easting, z));
  }

  dsm::Settings sd;
  sd.center_easting = gs.center_easting;
  sd.center_northing = gs.center_northing;
  dsm::Dsm digital_surface_map(sd, m);
  digital_surface_map.process(cloud, m);

  // camera rig + poses + images
  const int W = 160, H = 120;
  aslam::Camera cam(120.0, 120.0, (W - 1) / 2.0, (H - 1) / 2.0, W, H);
  const double s45 = std::sqrt(0.5);
  std::shared_ptr<aslam::NCamera> ncameras(new aslam::NCamera(
      cam, aslam::Transformation(kindr::minimal::RotationQuaternion(1, 0, 0, 0),
                                 Eigen::Vector3d(0.02, -0.01, 0.03))));
  Poses T_G_Bs;
  Images images;
  const int F = 9;
  for (int f = 0; f < F; ++f) {
    const double px = gs.center_easting - 45.0 + 11.0 * f;
    const double py = gs.center_northing + ((f % 3) - 1) * 20.0;
    // nadir-looking: q = Rz(90deg) * Rx(180deg) = (0, s45, s45, 0), slightly perturbed
    double q[4] = {0.01 * (f - 4), s45, s45 + 0.005 * f, 0.004 * (4 - f)};
    const double nq = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    T_G_Bs.push_back(Pose(kindr::minimal::RotationQuaternion(q[0] / nq, q[1] / nq, q[2] / nq, q[3] / nq),
                          Eigen::Vector3d(px, py, 470.0)));
    Image img(H, W, colored ? 3 : 1);
    for (size_t b = 0; b < static_cast<size_t>(H) * img.step; ++b)
      img.data[b] = static_cast<uint8_t>(urand() * 256.0);
    images.push_back(img);
  }
  ortho::Settings so;
  so.colored_ortho = colored;
  ortho::OrthoBackwardGrid mosaic(ncameras, so, m);
  mosaic.process(T_G_Bs, images, m);

  // ---- oracle on the same inputs ----------------------------------------------
  amo_grid g;
  amo_make_grid(gs.delta_easting, gs.delta_northing, gs.resolution, gs.center_easting,
                gs.center_northing, &g);
  if (g.rows != rows || g.cols != cols) return 2;
  std::vector<float> elev(cells, NAN), angle(cells, 0.0f), idx(cells, NAN), nobs(cells, 0.0f),
      ortho_l(cells, 255.0f), col(cells, NAN);
  int rc = amo_dsm_process(reinterpret_cast<const double*>(cloud.data()), cloud.size(), &g, 1,
                           sd.center_easting, sd.center_northing, 1, 0, elev.data(), nullptr);
  if (rc) return 3;
  amo_camera oc;
  std::memset(&oc, 0, sizeof(oc));
  oc.fu = oc.fv = 120.0;
  oc.cu = (W - 1) / 2.0;
  oc.cv = (H - 1) / 2.0;
  oc.width = W;
  oc.height = H;
  std::vector<double> tgb(7 * F);
  std::vector<const uint8_t*> ptrs(F);
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
  // the GPU mosaic ran on the GPU DSM; give the oracle that very layer so the
  // two folds see identical input, and check the DSM separately (1e-4 m).
  const float* gpu_elev = (*m)["elevation"].data();
  double max_dh = 0.0;
  size_t nan_mismatch = 0;
  for (size_t i = 0; i < cells; ++i) {
    if (std::isnan(elev[i]) != std::isnan(gpu_elev[i])) ++nan_mismatch;
    else if (!std::isnan(elev[i])) max_dh = std::fmax(max_dh, std::fabs((double)elev[i] - gpu_elev[i]));
  }
  std::printf("DSM: max |dh| = %.3g m, NaN-pattern mismatches = %zu\n", max_dh, nan_mismatch);
  std::vector<float> elev_in(gpu_elev, gpu_elev + cells);
  rc = amo_ortho_backward_process(&g, &oc, tgb.data(), tcb, ptrs.data(), steps.data(),
                                  colored ? 3 : 1, F, colored ? 1 : 0, 1, 0, elev_in.data(),
                                  angle.data(), idx.data(), nobs.data(), ortho_l.data(), col.data());
  if (rc) return 4;
  bool ok = nan_mismatch == 0 && max_dh <= 1e-4;
  std::printf("ortho (%s):\n", colored ? "colored" : "gray");
  ok &= same_bits((*m)["elevation_angle"].data(), angle.data(), cells, "elevation_angle");
  ok &= same_bits((*m)["observation_index"].data(), idx.data(), cells, "observation_index");
  ok &= same_bits((*m)["num_observations"].data(), nobs.data(), cells, "num_observations");
  ok &= same_bits((*m)["ortho"].data(), ortho_l.data(), cells, "ortho");
  ok &= same_bits((*m)["colored_ortho"].data(), col.data(), cells, "colored_ortho");
  size_t covered = 0;
  for (size_t i = 0; i < cells; ++i) covered += !std::isnan(idx[i]);
  std::printf("coverage %.3f\n", (double)covered / cells);
  if (covered < cells / 10) ok = false;
  std::printf(ok ? "PARITY OK\n" : "PARITY FAILED\n");
  return ok ? 0 : 1;
}
