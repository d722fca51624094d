// oracle/demokit/dropin: this include path of an external -> the stand-ins of include/aerial-mapper-compat
#include "aerial-mapper-deps.h"
