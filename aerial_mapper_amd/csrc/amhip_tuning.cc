// amhip_tuning.cc -- see amhip_tuning.h.  Host-only translation unit.
#include "amhip_tuning.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>

namespace amhip {

namespace {

// every key the library looks up (a typo in AMHIP_TUNING or amhip_set_tuning is an error, not a no-op)
const char* const kKeys[] = {
    // sort
    "sort_one_level", "p3_min_points", "p3_target", "p3_cap", "p3_rounds_cap", "p3_rounds_reread",
    "sort_no_speculation", "sort_spec_max_points", "sort_spec_margin_shift", "no_launch_skips",
    // DSM gather
    "dsm_canon_all", "dsm_no_rough_switch", "dsm_no_subwindow", "eager_reset",
    // mosaic
    "ortho_exact_fold", "ortho_no_prune", "ortho_fast_waves", "no_coarse_cull", "ortho_no_tile_list",
    "no_distorted_cull", "no_distorted_prune", "distorted_square_cull",
    // session
    "session_always_copy", "session_threads", "session_scalar_sums", "session_no_partial",
    "session_verify_partial", "session_trace",
    // lab builds (-DAMHIP_TIMING_PROBES) only
    "gather_tj", "gather_nt", "gather_class_cap0", "gather_class_cap1", "gather_class_cap2", "f32_variant",
    "fx_theta"};

bool known(const std::string& k) {
  for (const char* s : kKeys)
    if (k == s) return true;
  return false;
}

struct Store {
  std::mutex mu;
  std::map<std::string, double> values;
  bool env_read = false;
  void read_env() {  // (mu held)
    if (env_read) return;
    env_read = true;
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
      if (!known(key)) {
        std::fprintf(stderr, "libaerial_mapper_hip: AMHIP_TUNING names an unknown key '%s' (ignored)\n", key.c_str());
        continue;
      }
      values[key] = v;
    }
  }
};

Store& store() {
  static Store s;
  return s;
}

}  // namespace

double tuning(const char* key, double dflt) {
  Store& s = store();
  std::lock_guard<std::mutex> lock(s.mu);
  s.read_env();
  const auto it = s.values.find(key);
  return it == s.values.end() ? dflt : it->second;
}

bool tuning_set(const char* key, double value) {
  if (!key || !known(key)) return false;
  Store& s = store();
  std::lock_guard<std::mutex> lock(s.mu);
  s.read_env();
  if (std::isnan(value)) s.values.erase(key);
  else s.values[key] = value;
  return true;
}

}  // namespace amhip
