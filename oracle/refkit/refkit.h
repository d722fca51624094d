/*
 * oracle/refkit/refkit.h -- TEST INFRASTRUCTURE ONLY (consistency check of the CPU oracle).
 *
 * STATUS: what is built over this directory is NOT a reference build and pins no parity.  The
 * reference's translation units need libraries the image lacks; the task's rules class such a
 * path as unbuildable ("never make a reference build by writing stand-ins for headers ...").
 * These stand-ins were written in earlier rounds; the libraries built over them are kept ONLY as a
 * consistency check that the restated oracle (amo_*.cc) reads the loops' control flow the way a
 * compiler does (tests/test_reference_loops.py).  Nothing graded rests on them: bench.py's
 * cpu_baseline and parity sample use the restated oracle over the vendored nanoflann (`kind: port`).
 *
 * Build kit for compiling the reference's OWN translation units of the hot path --
 * aerial_mapper_dsm/src/dsm.cc, aerial_mapper_ortho/src/ortho-backward-grid.cc,
 * aerial_mapper_ortho/src/ortho-from-pcl.cc, aerial_mapper_grid_map/src/aerial-mapper-grid-map.cc,
 * aerial_mapper_dense_pcl/src/densifier.cpp, aerial_mapper_ortho/src/ortho-forward-homography.cc
 * (+ aerial_mapper_utils/src/utils-common.cc) --
 * UNCHANGED, from where they lie under /root/reference, into oracle/_ref/ (oracle/Makefile,
 * target `loops`).  Those files need Eigen, glog, ROS, grid_map, aslam_cv2, minkindr and
 * OpenCV, none of which is in /root/reference or in this image; this directory holds
 * minimal stand-ins with the names those files mention, found by the compiler
 * under the externals' own include paths (<Eigen/Dense>, <glog/logging.h>, ...).
 *
 * What the check covers and what it does not:
 *   covered     the control flow of the reference's own code: the kd-tree fill with the
 *               centre offsets, the radius search and its ladder, the IDW sums and the
 *               order they run in, the exact-hit CHECK, the per-frame fold with its
 *               float-rounded running maximum, the visibility test, round()/min() of the
 *               pixel, the colour packing call, `num_observations += itself`, the
 *               composition T_G_B * T_C_B^-1, the layers a map starts with and their
 *               initial values -- compiled from the reference's source.
 *   NOT covered the arithmetic INSIDE the externals' calls (GridMap::getPosition,
 *               QuatTransformation::inverse/transform/operator*, Camera::project3,
 *               colorVectorToValue, Eigen's 3x3 * 3x1 product in the densifier, and every
 *               OpenCV / aslam operation of the forward mosaic -- getPerspectiveTransform,
 *               warpPerspective, cvtColor, the feather blender, the mapped undistorter,
 *               backProject3, see amo_cvlike.h): the
 *               stand-ins forward to / repeat the same formulas the restated oracle adopts
 *               (amo_compat.h).
 * The stand-ins are written for this purpose only; nothing is copied from the
 * libraries they stand in for.
 */
#ifndef ORACLE_REFKIT_H_
#define ORACLE_REFKIT_H_

#include <atomic>
#include <cstring>
#include <mutex>
#include <ostream>
#include <sstream>
#include <string>

namespace refkit {

// A failed glog CHECK aborts the reference's process.  Here it is recorded (first one
// wins) and execution continues, so that a driver can report it as a return code --
// the checks on the hot path guard arithmetic, not memory.
struct CheckState {
  std::atomic<int> failed{0};
  std::mutex mu;
  std::string condition;
};
inline CheckState& check_state() {
  static CheckState s;
  return s;
}
inline void check_reset() {
  CheckState& s = check_state();
  std::lock_guard<std::mutex> lk(s.mu);
  s.failed = 0;
  s.condition.clear();
}
inline void check_fail(const char* condition) {
  CheckState& s = check_state();
  std::lock_guard<std::mutex> lk(s.mu);
  if (!s.failed) s.condition = condition;
  s.failed = 1;
}

// Swallows whatever is streamed into LOG(...) / VLOG(...) / a failed CHECK(...).
struct Sink {
  template <typename T>
  Sink& operator<<(const T&) {
    return *this;
  }
  Sink& operator<<(std::ostream& (*)(std::ostream&)) { return *this; }
};
struct FailSink : Sink {
  explicit FailSink(const char* condition) { check_fail(condition); }
};

}  // namespace refkit

#endif  // ORACLE_REFKIT_H_
