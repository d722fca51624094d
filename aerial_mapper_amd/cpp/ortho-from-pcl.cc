// ortho::OrthoFromPcl over the C ABI
// (see include/aerial-mapper-ortho/ortho-from-pcl.h).
#include "aerial-mapper-ortho/ortho-from-pcl.h"

#include <cstdio>

#include "shim_common.h"

namespace ortho {

OrthoFromPcl::OrthoFromPcl(const Settings& settings) : settings_(settings) { printParams(); }

void OrthoFromPcl::process(const AlignedType<std::vector, Eigen::Vector3d>::type& pointcloud,
                           const std::vector<int>& intensities, grid_map::GridMap* map) const {
  if (pointcloud.empty()) amhip_shim::fatal("OrthoFromPcl::process", "CHECK(!pointcloud.empty())");
  if (!map) amhip_shim::fatal("OrthoFromPcl::process", "CHECK(map)");
  if (intensities.size() < pointcloud.size())
    amhip_shim::fatal("OrthoFromPcl::process", "CHECK(i < intensities.size())");
  std::fprintf(stderr, "[aerial_mapper_hip] Number of points: %zu\n", pointcloud.size());
  // the reference keeps no state between calls either; the map's session is shared with the
  // other drop-in objects working on this GridMap (created for the call if there is none)
  amhip_session* session = amhip_shim::acquire_session(*map, nullptr, "OrthoFromPcl");
  static_assert(sizeof(Eigen::Vector3d) == 3 * sizeof(double),
                "point cloud must be contiguous x,y,z doubles");
  static_assert(sizeof(int) == sizeof(int32_t), "intensities are 32-bit ints");
  const int status = amhip_session_ortho_from_pcl_process(
      session, reinterpret_cast<const double*>(pointcloud.data()),
      reinterpret_cast<const int32_t*>(intensities.data()), pointcloud.size(),
      settings_.interpolation_radius, settings_.use_adaptive_interpolation ? 1 : 0,
      (*map)["ortho"].data());
  if (status != AMHIP_OK) amhip_shim::fatal("OrthoFromPcl::process", amhip_last_error());
  amhip_shim::release_session(session);
}

void OrthoFromPcl::printParams() const {
  std::fprintf(stderr,
               "**************************************************\n"
               "Ortho-From-Pcl parameters (MI355X / HIP):\n"
               "  Interp. radius    %d\n  Adaptive interp.  %d\n"
               "**************************************************\n",
               settings_.interpolation_radius, (int)settings_.use_adaptive_interpolation);
}

}  // namespace ortho
