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


// Timing hooks: when enabled (pp_prof_enable) the entry point that launches a roofline
// kernel brackets exactly that launch with two hipEvents on the caller's stream.
struct PpProf {
    hipEvent_t* ev = nullptr;
    int capacity = 0;
    int count = 0;
};
PpProf* pp_prof_state();

struct PpProfScope {
    PpProf* p;
    hipStream_t s;
    bool on;
    PpProfScope(hipStream_t stream) : p(pp_prof_state()), s(stream) {
        on = p->capacity > 0 && p->count < p->capacity;
        if (on) (void)hipEventRecord(p->ev[2 * p->count], s);
    }
    ~PpProfScope() {
        if (on) {
            (void)hipEventRecord(p->ev[2 * p->count + 1], s);
            p->count++;
        }
    }
};

// Same idea for the contraction engine (pp_gemm): one event pair + the launch's flop count per record;
// kind 0 = gemm_f16x3s_kernel (both operands pre-split), 1 = the other GEMM kernels.
struct PpGemmProf {
    hipEvent_t* ev = nullptr;
    double* flops = nullptr;
    int* kind = nullptr;
    int (*shape)[5] = nullptr;  // M, N, K, conv kernel size, chosen configuration (PP_GEMM_TRACE dump)
    int capacity = 0;
    int count = 0;
};
PpGemmProf* pp_gemm_prof_state();

#endif
