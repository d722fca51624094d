/*
 * oracle/ref_loops_common.h -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 * Shared by the three drivers of the reference's own loops (ref_loops_*.cc, refkit/).
 */
#ifndef ORACLE_REF_LOOPS_COMMON_H_
#define ORACLE_REF_LOOPS_COMMON_H_

#include <algorithm>
#include <chrono>
#include <cstring>
#include <string>
#include <vector>

#include <grid_map_core/GridMap.hpp>

#include "amo_types.h"
#include "refkit/refkit.h"

namespace ref_loops {

inline double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

inline void set_geometry(const amo_grid& g, grid_map::GridMap* map) {
  // aerial-mapper-grid-map.cc:30-33: setGeometry(Length(delta_easting, delta_northing),
  // resolution, Position(center_easting, center_northing))
  map->setGeometry(grid_map::Length(g.length_x, g.length_y), g.resolution,
                   grid_map::Position(g.pos_x, g.pos_y));
}

inline bool same_geometry(const amo_grid& a, const amo_grid& b) {
  return a.rows == b.rows && a.cols == b.cols && a.resolution == b.resolution &&
         a.length_x == b.length_x && a.length_y == b.length_y && a.pos_x == b.pos_x &&
         a.pos_y == b.pos_y;
}

inline void layer_in(const float* src, grid_map::Matrix* m) {
  if (src) std::memcpy(m->data(), src, sizeof(float) * static_cast<size_t>(m->rows() * m->cols()));
}
inline void layer_out(const grid_map::Matrix& m, float* dst) {
  if (dst) std::memcpy(dst, m.data(), sizeof(float) * static_cast<size_t>(m.rows() * m.cols()));
}

// What a failed CHECK of the reference means to the caller of the restated oracle.
inline int check_result() {
  refkit::CheckState& s = refkit::check_state();
  if (!s.failed) return AMO_OK;
  std::lock_guard<std::mutex> lk(s.mu);
  if (s.condition.find("distances[i] > 0.0") != std::string::npos) return AMO_ERR_EXACT_HIT;
  if (s.condition.find("alpha > 0.0") != std::string::npos) return AMO_ERR_ALPHA_NONPOS;
  return AMO_ERR_ARG;
}

}  // namespace ref_loops

#endif  // ORACLE_REF_LOOPS_COMMON_H_
