// dsm::Dsm over the C ABI (see include/aerial-mapper-dsm/dsm.h).
#include "aerial-mapper-dsm/dsm.h"

#include <cstdio>
#include <cstdlib>

#include "shim_common.h"

namespace dsm {

static Precision default_precision() {
  return amhip_default_dsm_precision() == AMHIP_DSM_FAST ? Precision::kFast : Precision::kReferenceIdentical;
}

Dsm::Dsm(const Settings& settings, grid_map::GridMap* map)
    : settings_(settings), precision_(default_precision()), session_(nullptr) {
  if (!map) amhip_shim::fatal("Dsm::Dsm", "CHECK(map)");
  printParams();
  // The reference builds a sample->cell-index table for every cell here
  // (dsm.cc:24-33); the GPU path needs the geometry only.
  ensureSession(*map);
}

Dsm::~Dsm() { amhip_shim::release_session(session_); }

void Dsm::setPrecision(Precision precision) { precision_ = precision; }

void Dsm::ensureSession(const grid_map::GridMap& map) {
  session_ = amhip_shim::acquire_session(map, session_, "Dsm");
}

void Dsm::process(const AlignedType<std::vector, Eigen::Vector3d>::type& point_cloud,
                  grid_map::GridMap* map) {
  if (point_cloud.empty()) {
    std::fprintf(stderr, "[aerial_mapper_hip] WARNING Passed empty point cloud to DSM module\n");
    return;
  }
  if (!map) amhip_shim::fatal("Dsm::process", "CHECK(map)");
  ensureSession(*map);
  static_assert(sizeof(Eigen::Vector3d) == 3 * sizeof(double),
                "point cloud must be contiguous x,y,z doubles");
  const double* xyz = reinterpret_cast<const double*>(point_cloud.data());
  grid_map::Matrix& elevation = (*map)["elevation"];
  // (the session is shared with every object working on this map: the mode is this object's)
  amhip_shim::check_status(
      amhip_session_set_dsm_precision(session_, precision_ == Precision::kFast ? AMHIP_DSM_FAST
                                                                                : AMHIP_DSM_EXACT),
      "Dsm::process");
  amhip_shim::check_status(
      amhip_session_dsm_process(session_, xyz, point_cloud.size(), settings_.interpolation_radius,
                                settings_.center_easting, settings_.center_northing,
                                elevation.data()),
      "Dsm::process");
}

void Dsm::printParams() {
  std::fprintf(stderr,
               "**************************************************\n"
               "DSM parameters (MI355X / HIP):\n"
               "  Interp. radius   %d\n  Adaptive interp. %d\n"
               "  Center easting   %f\n  Center northing  %f\n"
               "**************************************************\n",
               settings_.interpolation_radius, (int)settings_.adaptive_interpolation,
               settings_.center_easting, settings_.center_northing);
}

}  // namespace dsm
