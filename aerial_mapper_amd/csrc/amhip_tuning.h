// amhip_tuning.h -- ONE door for the knobs that select among correct implementations (tests force
// each path through them, A-B timing flips them): amhip_set_tuning(key, value) of the C ABI, or
// AMHIP_TUNING="key=value,key,..." in the environment of a host that cannot be recompiled (read
// once, at the first look-up; a bare key means 1).  The keys are listed in
// include/aerial_mapper_hip.h next to amhip_set_tuning.  None is needed in normal use.
#ifndef AMHIP_TUNING_H_
#define AMHIP_TUNING_H_

namespace amhip {

// value of `key`, or dflt when it is not set
double tuning(const char* key, double dflt);
inline bool tuning_on(const char* key) { return tuning(key, 0.0) != 0.0; }
// (amhip_set_tuning: value NaN clears the key; false: unknown key)
bool tuning_set(const char* key, double value);

}  // namespace amhip

#endif  // AMHIP_TUNING_H_
