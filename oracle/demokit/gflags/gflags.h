// oracle/demokit: stand-in for <gflags/gflags.h> (see ../demokit.h).  TEST INFRASTRUCTURE ONLY.
#ifndef ORACLE_DEMOKIT_GFLAGS_H_
#define ORACLE_DEMOKIT_GFLAGS_H_
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>

namespace demokit {
struct FlagRef {
  char type;  // s d i b
  void* ptr;
};
inline std::map<std::string, FlagRef>& flag_registry() {
  static std::map<std::string, FlagRef> r;
  return r;
}
struct FlagRegistrar {
  FlagRegistrar(const char* name, char type, void* ptr) { flag_registry()[name] = FlagRef{type, ptr}; }
};
inline void set_flag(const std::string& name, const std::string& value) {
  auto it = flag_registry().find(name);
  if (it == flag_registry().end()) return;  // (unknown flags are ignored like the .ff files' stale keys)
  switch (it->second.type) {
    case 's': *static_cast<std::string*>(it->second.ptr) = value; break;
    case 'd': *static_cast<double*>(it->second.ptr) = std::atof(value.c_str()); break;
    case 'i': *static_cast<int32_t*>(it->second.ptr) = std::atoi(value.c_str()); break;
    case 'b': *static_cast<bool*>(it->second.ptr) = (value.empty() || value == "true" || value == "1"); break;
  }
}
}  // namespace demokit

#define DEFINE_string(name, val, txt) \
  std::string FLAGS_##name = val;     \
  static ::demokit::FlagRegistrar flagreg_##name(#name, 's', &FLAGS_##name)
#define DEFINE_double(name, val, txt) \
  double FLAGS_##name = val;          \
  static ::demokit::FlagRegistrar flagreg_##name(#name, 'd', &FLAGS_##name)
#define DEFINE_int32(name, val, txt) \
  int32_t FLAGS_##name = val;        \
  static ::demokit::FlagRegistrar flagreg_##name(#name, 'i', &FLAGS_##name)
#define DEFINE_bool(name, val, txt) \
  bool FLAGS_##name = val;          \
  static ::demokit::FlagRegistrar flagreg_##name(#name, 'b', &FLAGS_##name)

namespace google {
inline void ParseCommandLineFlags(int* argc, char*** argv, bool) {
  for (int k = 1; k < *argc; ++k) {
    const char* a = (*argv)[k];
    if (std::strncmp(a, "--", 2) != 0) continue;
    const char* eq = std::strchr(a, '=');
    if (eq) ::demokit::set_flag(std::string(a + 2, eq), std::string(eq + 1));
    else ::demokit::set_flag(std::string(a + 2), "true");
  }
}
}  // namespace google
#endif  // ORACLE_DEMOKIT_GFLAGS_H_
