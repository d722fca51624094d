// oracle/refkit: stand-in for <grid_map_msgs/GridMap.h> (see ../refkit.h): a message type that
// is only ever handed to a publisher stub.  TEST INFRASTRUCTURE ONLY.
#ifndef ORACLE_REFKIT_GRID_MAP_MSGS_GRIDMAP_H_
#define ORACLE_REFKIT_GRID_MAP_MSGS_GRIDMAP_H_
namespace grid_map_msgs {
struct GridMap {};
}  // namespace grid_map_msgs
#endif  // ORACLE_REFKIT_GRID_MAP_MSGS_GRIDMAP_H_
