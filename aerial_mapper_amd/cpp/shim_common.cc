// The per-map session registry of the drop-in classes (see shim_common.h).
#include "shim_common.h"

#include <cstring>
#include <mutex>
#include <vector>

namespace amhip_shim {

// AERIAL_MAPPER_HIP_DEVICE: the device the drop-in classes create their contexts on (default 0)
int default_device() {
  const char* env = std::getenv("AERIAL_MAPPER_HIP_DEVICE");
  return env ? std::atoi(env) : 0;
}

namespace {

struct Entry {
  const grid_map::GridMap* map;
  amhip_grid_desc geom;
  amhip_session* session;
  int refs;
};

std::mutex g_mutex;
std::vector<Entry> g_entries;

bool same_geometry(const amhip_grid_desc& a, const amhip_grid_desc& b) {
  return a.rows == b.rows && a.cols == b.cols && a.resolution == b.resolution &&
         a.length_x == b.length_x && a.length_y == b.length_y && a.pos_x == b.pos_x &&
         a.pos_y == b.pos_y;
}

// AERIAL_MAPPER_HIP_DEVICES=0,1,2,...: one window per entry (an entry may repeat); otherwise
// one window on AERIAL_MAPPER_HIP_DEVICE (default 0)
std::vector<int32_t> device_list() {
  std::vector<int32_t> d;
  if (const char* env = std::getenv("AERIAL_MAPPER_HIP_DEVICES")) {
    const char* p = env;
    while (*p) {
      char* end = nullptr;
      const long v = std::strtol(p, &end, 10);
      if (end == p) break;
      d.push_back(static_cast<int32_t>(v));
      p = end;
      while (*p == ',' || *p == ' ') ++p;
    }
  }
  if (d.empty()) {
    d.push_back(default_device());
  }
  return d;
}

// near-square cut, more windows along the longer axis
void layout_for(int rows, int cols, int world, int* ti, int* tj) {
  *ti = 1;
  *tj = world;
  double best = -1.0;
  for (int a = 1; a <= world; ++a) {
    if (world % a) continue;
    const int b = world / a;
    const double diff = std::abs((double)rows / a - (double)cols / b);
    if (best < 0.0 || diff < best) {
      best = diff;
      *ti = a;
      *tj = b;
    }
  }
}

void drop(amhip_session* s) {  // g_mutex held
  for (size_t k = 0; k < g_entries.size(); ++k)
    if (g_entries[k].session == s) {
      if (--g_entries[k].refs == 0) {
        amhip_session_destroy(s);
        g_entries.erase(g_entries.begin() + k);
      }
      return;
    }
}

}  // namespace

amhip_session* acquire_session(const grid_map::GridMap& map, amhip_session* held, const char* where) {
  const amhip_grid_desc g = describe(map);
  std::lock_guard<std::mutex> lock(g_mutex);
  for (Entry& e : g_entries)
    if (e.map == &map && same_geometry(e.geom, g)) {
      if (e.session == held) return held;
      ++e.refs;
      if (held) drop(held);
      return e.session;
    }
  // (a map object that changed its geometry: its old session dies with its last user)
  const std::vector<int32_t> devs = device_list();
  int ti, tj;
  layout_for(g.rows, g.cols, static_cast<int>(devs.size()), &ti, &tj);
  amhip_session* s = nullptr;
  check_status(amhip_session_create(&g, ti, tj, devs.data(), &s), where);
  Entry e = {&map, g, s, 1};
  g_entries.push_back(e);
  if (held) drop(held);
  return s;
}

void release_session(amhip_session* held) {
  if (!held) return;
  std::lock_guard<std::mutex> lock(g_mutex);
  drop(held);
}

}  // namespace amhip_shim
