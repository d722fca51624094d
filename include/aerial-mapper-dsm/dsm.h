// dsm::Dsm on MI355X -- drop-in for the reference class
// (aerial_mapper_dsm/include/aerial-mapper-dsm/dsm.h:25-42): same namespace,
// Settings fields/defaults and public signatures, so
// aerial_mapper_demos/src/dsm/main-dsm.cc:103-107 compiles against it
// unchanged.  The kd-tree + per-cell loop of dsm.cc:36-184 run as HIP kernels
// behind the C ABI (include/aerial_mapper_hip.h).
#ifndef AERIAL_MAPPER_HIP_DSM_H_
#define AERIAL_MAPPER_HIP_DSM_H_

#include <vector>

#include "aerial-mapper-deps.h"
#include "aerial-mapper-utils/utils-nearest-neighbor.h"

struct amhip_session;

namespace dsm {

struct Settings {
  EIGEN_MAKE_ALIGNED_OPERATOR_NEW
  // SQUARED search radius in m^2 (it is handed to nanoflann's RadiusResultSet
  // as is); an int in the reference too.
  int interpolation_radius = 1.0;
  bool adaptive_interpolation = false;  // printed only, like the reference
  double center_easting = 0.0;
  double center_northing = 0.0;
  bool use_multi_threads = true;  // both reference variants give one result
};

// EXTENSION (not in the reference): arithmetic of the GPU gather, per Dsm object.
//   kReferenceIdentical  (default) FP64 everywhere: the reference's floats, deterministic.
//   kFast                single precision under exact guards: same neighbour sets and NaN
//                        pattern, heights within 1e-4 m of the reference's (aerial_mapper_hip.h:
//                        AMHIP_DSM_FAST), ~0.6 ms less per 100 M cells.
enum class Precision { kReferenceIdentical = 1, kFast = 0 };

class Dsm {
 public:
  EIGEN_MAKE_ALIGNED_OPERATOR_NEW

  Dsm(const Settings& settings, grid_map::GridMap* map);
  ~Dsm();
  Dsm(const Dsm&) = delete;
  Dsm& operator=(const Dsm&) = delete;

  // Reads the "elevation" layer of *map, overwrites every cell that finds a
  // point within the (expanding) radius, leaves the others untouched.
  void process(const AlignedType<std::vector, Eigen::Vector3d>::type& point_cloud,
               grid_map::GridMap* map);

  // EXTENSION: see Precision.  Starts as kReferenceIdentical unless the environment says
  // AMHIP_DSM_FAST=1 (hosts that cannot be recompiled); applies to this object's process() calls.
  void setPrecision(Precision precision);
  Precision precision() const { return precision_; }

 private:
  void ensureSession(const grid_map::GridMap& map);
  void printParams();

  Settings settings_;
  Precision precision_;
  // the map's session (device windows + resident layers), shared with the other drop-in
  // objects working on the same grid_map::GridMap (aerial_mapper_amd/cpp/shim_common.cc)
  amhip_session* session_;
};

}  // namespace dsm

#endif  // AERIAL_MAPPER_HIP_DSM_H_
