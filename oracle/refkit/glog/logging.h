// oracle/refkit: stand-in for <glog/logging.h> (see ../refkit.h).  TEST INFRASTRUCTURE ONLY.
#ifndef ORACLE_REFKIT_GLOG_LOGGING_H_
#define ORACLE_REFKIT_GLOG_LOGGING_H_

#include <vector>

#include "../refkit.h"

#define CHECK(condition) \
  if (condition) {       \
  } else                 \
    ::refkit::FailSink(#condition)
#define CHECK_EQ(a, b) CHECK((a) == (b))
#define CHECK_NE(a, b) CHECK((a) != (b))
#define CHECK_GT(a, b) CHECK((a) > (b))
#define CHECK_GE(a, b) CHECK((a) >= (b))
#define CHECK_LT(a, b) CHECK((a) < (b))
#define CHECK_LE(a, b) CHECK((a) <= (b))
#define CHECK_NOTNULL(p) (p)
#define LOG(severity) ::refkit::Sink()
#define VLOG(level) ::refkit::Sink()

#endif  // ORACLE_REFKIT_GLOG_LOGGING_H_
