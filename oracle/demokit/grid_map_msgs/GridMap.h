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
  // publishUntilShutdown() never returns: the first message ends the batch demos.  The
  // incremental main publishes once per stereo pair and returns on its own
  // (main-ortho-backward-grid-incremental.cc:160,168): AMHIP_DEMO_KEEP_RUNNING=1, the last
  // message's layers are what stays on disk.
  if (!std::getenv("AMHIP_DEMO_KEEP_RUNNING")) std::exit(0);
}
}  // namespace grid_map_msgs
#endif  // ORACLE_DEMOKIT_GRID_MAP_MSGS_GRIDMAP_H_
