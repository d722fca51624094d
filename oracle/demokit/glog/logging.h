// oracle/demokit: <glog/logging.h> = refkit's LOG / CHECK + the two calls the mains make.
#ifndef ORACLE_DEMOKIT_GLOG_LOGGING_H_
#define ORACLE_DEMOKIT_GLOG_LOGGING_H_
#include "../../refkit/glog/logging.h"
namespace google {
inline void InitGoogleLogging(const char*) {}
inline void InstallFailureSignalHandler() {}
}  // namespace google
#endif  // ORACLE_DEMOKIT_GLOG_LOGGING_H_
