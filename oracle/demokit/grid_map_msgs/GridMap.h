// oracle/demokit: the grid map message = "write the layers" (see ../demokit.h).
#ifndef ORACLE_DEMOKIT_GRID_MAP_MSGS_GRIDMAP_H_
#define ORACLE_DEMOKIT_GRID_MAP_MSGS_GRIDMAP_H_
#include <cstdlib>
#include <functional>
namespace grid_map_msgs {
struct GridMap {
  std::function<void()> dump;
};
inline void published(const GridMap& m) {
  if (m.dump) m.dump();
  std::exit(0);  // publishUntilShutdown() never returns: the first message ends the demo
}
}  // namespace grid_map_msgs
#endif  // ORACLE_DEMOKIT_GRID_MAP_MSGS_GRIDMAP_H_
