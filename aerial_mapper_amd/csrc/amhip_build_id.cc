// amhip_build_id.cc -- the library's build id: the first 16 hex digits of the SHA-256 over every
// source and header of the library and the compiler flags (aerial_mapper_amd/build.py passes it).
// bench.py refuses rocprofv3 evidence under profiles/ that was collected from another build.
#include "aerial_mapper_hip.h"

#ifndef AMHIP_BUILD_ID
#define AMHIP_BUILD_ID "unknown"
#endif

extern "C" const char* amhip_build_id(void) { return AMHIP_BUILD_ID; }
