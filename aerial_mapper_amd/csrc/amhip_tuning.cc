// amhip_tuning.cc -- see amhip_tuning.h.  Host-only translation unit.
#include "amhip_tuning.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>
#include <string>

namespace amhip {

namespace {

// every key the library looks up (a typo in AMHIP_TUNING or amhip_set_tuning is an error, not a no-op)
const char* const kKeys[] = {
    // sort
    "sort_one_level", "p3_min_points", "p3_target", "p3_cap", "p3_rounds_cap", "p3_rounds_reread",
    "no_launch_skips",
    // DSM gather
    "dsm_canon_all", "knn_global_bins", "dsm_no_rough_switch", "dsm_no_subwindow", "eager_reset",
    // mosaic
    "ortho_exact_fold", "ortho_no_prune", "no_coarse_cull", "ortho_no_tile_list",
    "no_distorted_cull", "no_distorted_prune", "distorted_square_cull",
    // session
    "session_always_copy", "session_threads", "session_scalar_sums", "session_no_partial",
    "session_verify_partial", "session_trace", "session_serial_sums",
    // lab builds (-DAMHIP_TIMING_PROBES) only
    "gather_tj", "gather_nt", "gather_class_cap0", "gather_class_cap1", "gather_class_cap2", "f32_variant",
    "fx_theta"};

constexpr int kNumKeys = (int)(sizeof(kKeys) / sizeof(kKeys[0]));

int key_index(const char* k, size_t len) {
  for (int i = 0; i < kNumKeys; ++i)
    if (std::strlen(kKeys[i]) == len && std::memcmp(kKeys[i], k, len) == 0) return i;
  return -1;
}

// One atomic slot per key (NaN = not set): a look-up is a string compare over ~30 short keys and one
// relaxed load -- no lock, no allocation on the per-call host path; the environment is
// read once.
struct Store {
  std::atomic<double> values[kNumKeys];
  std::once_flag env_once;
  Store() {
    for (auto& v : values) v.store(std::nan(""), std::memory_order_relaxed);
  }
  void read_env() {
    const char* e = std::getenv("AMHIP_TUNING");
    if (!e) return;
    std::string s(e);
    size_t pos = 0;
    while (pos <= s.size()) {
      size_t end = s.find(',', pos);
      if (end == std::string::npos) end = s.size();
      std::string item = s.substr(pos, end - pos);
      pos = end + 1;
      while (!item.empty() && item[0] == ' ') item.erase(0, 1);
      if (item.empty()) continue;
      const size_t eq = item.find('=');
      const std::string key = item.substr(0, eq);
      const double v = eq == std::string::npos ? 1.0 : std::atof(item.c_str() + eq + 1);
      const int k = key_index(key.data(), key.size());
      if (k < 0) {
        std::fprintf(stderr, "libaerial_mapper_hip: AMHIP_TUNING names an unknown key '%s' (ignored)\n", key.c_str());
        continue;
      }
      values[k].store(v, std::memory_order_relaxed);
    }
  }
  void ensure_env() { std::call_once(env_once, [this] { read_env(); }); }
};

Store& store() {
  static Store s;
  return s;
}

}  // namespace

double tuning(const char* key, double dflt) {
  Store& s = store();
  s.ensure_env();
  const int k = key ? key_index(key, std::strlen(key)) : -1;
  if (k < 0) return dflt;
  const double v = s.values[k].load(std::memory_order_relaxed);
  return std::isnan(v) ? dflt : v;
}

bool tuning_set(const char* key, double value) {
  const int k = key ? key_index(key, std::strlen(key)) : -1;
  if (k < 0) return false;
  Store& s = store();
  s.ensure_env();
  s.values[k].store(value, std::memory_order_relaxed);   // (NaN clears the key)
  return true;
}

}  // namespace amhip
