// Helpers shared by the two drop-in classes: geometry extraction from a
// grid_map::GridMap and the glog-CHECK-like failure path.
#ifndef AERIAL_MAPPER_HIP_SHIM_COMMON_H_
#define AERIAL_MAPPER_HIP_SHIM_COMMON_H_

#include <cstdio>
#include <cstdlib>

#include "aerial-mapper-deps.h"
#include "aerial_mapper_hip.h"

namespace amhip_shim {

// The reference aborts through glog CHECK / LOG(FATAL); so does the shim.
[[noreturn]] inline void fatal(const char* where, const char* what) {
  std::fprintf(stderr, "[aerial_mapper_hip] FATAL %s: %s\n", where, what);
  std::abort();
}

inline void check_status(int status, const char* where) {
  if (status != AMHIP_OK) fatal(where, amhip_last_error());
}

inline amhip_grid_desc describe(const grid_map::GridMap& map) {
  amhip_grid_desc g;
  g.rows = map.getSize()(0);
  g.cols = map.getSize()(1);
  g.resolution = map.getResolution();
  g.length_x = map.getLength()(0);
  g.length_y = map.getLength()(1);
  g.pos_x = map.getPosition()(0);
  g.pos_y = map.getPosition()(1);
  return g;
}

// (Re)create the context when the map's geometry is not the one it was made for.
inline void ensure_context(amhip_ctx** ctx, int* rows, int* cols, double* geom,
                           const grid_map::GridMap& map, const char* where) {
  const amhip_grid_desc g = describe(map);
  if (*ctx && *rows == g.rows && *cols == g.cols && geom[0] == g.resolution &&
      geom[1] == g.pos_x && geom[2] == g.pos_y && geom[3] == g.length_x)
    return;
  if (*ctx) amhip_ctx_destroy(*ctx);
  *ctx = nullptr;
  int device = 0;
  if (const char* env = std::getenv("AERIAL_MAPPER_HIP_DEVICE")) device = std::atoi(env);
  check_status(amhip_ctx_create(&g, device, ctx), where);
  *rows = g.rows;
  *cols = g.cols;
  geom[0] = g.resolution;
  geom[1] = g.pos_x;
  geom[2] = g.pos_y;
  geom[3] = g.length_x;
}

}  // namespace amhip_shim

#endif  // AERIAL_MAPPER_HIP_SHIM_COMMON_H_
