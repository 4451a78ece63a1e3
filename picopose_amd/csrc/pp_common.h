// Shared host-side helpers of libpicopose_hip.so (not part of the public ABI).
#ifndef PP_COMMON_H
#define PP_COMMON_H
#include <hip/hip_runtime.h>
#include "../../include/picopose_hip.h"

#define PP_CHECK_HIP(expr)                         \
    do {                                           \
        if ((expr) != hipSuccess) return PP_ELAUNCH; \
    } while (0)

static inline int pp_last_launch() { return hipGetLastError() == hipSuccess ? PP_OK : PP_ELAUNCH; }

#endif
