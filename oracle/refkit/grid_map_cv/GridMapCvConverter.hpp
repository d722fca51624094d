// oracle/refkit: the reference files include this header but use nothing from it on the hot
// path (see ../refkit.h).  TEST INFRASTRUCTURE ONLY.
