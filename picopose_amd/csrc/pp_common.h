// Shared host-side helpers of libpicopose_hip.so (not part of the public ABI).
#ifndef PP_COMMON_H
#define PP_COMMON_H
#include <hip/hip_runtime.h>
#include "../../include/picopose_hip.h"

#define PP_CHECK_HIP(expr)                         \
    do {                                           \
        if ((expr) != hipSuccess) return PP_ELAUNCH; \
    } while (0)

// f16x3 activation operand: v -> hi = f16(4 v), lo = f16(4 v - hi) (saturated); the same split in every producer
#define PP_A_SCALE 4.f
// (precision study builds only — tests/precision_study.py: -DPP_STUDY_ACT_LO_ZERO drops the lo term of every ACTIVATION
// operand, -DPP_STUDY_W_LO_ZERO that of every weight: the engine then evaluates the 2-term "weights split only" and the
// 1-term plain-fp16 products bit for bit, at the 3-term kernels' speed.  Never defined in the product build.)
__device__ __forceinline__ void pp_split_f16(float v, _Float16& hi, _Float16& lo) {
    const float x = v * PP_A_SCALE;
    hi = (_Float16)fminf(fmaxf(x, -65504.f), 65504.f);
#ifdef PP_STUDY_ACT_LO_ZERO
    lo = (_Float16)0.f;
#else
    lo = (_Float16)fminf(fmaxf(x - (float)hi, -65504.f), 65504.f);
#endif
}

// plain-fp16 operand ("h" format, PP_PREC_F16): v -> f16(4 v), saturated like the hi term above
__device__ __forceinline__ _Float16 pp_to_f16(float v) { return (_Float16)fminf(fmaxf(v * PP_A_SCALE, -65504.f), 65504.f); }

// Sticky saturation word (include/picopose_hip.h pp_set_saturation_word).  A clamped operand term is finite but WRONG; the kernels that
// WRITE operand buffers (the producers of the inference path: split passes, LayerNorm, attention output, GEMM epilogues, resize, warp,
// Winograd transforms) report it by OR-ing bit 0 into a device word — an atomic only from a lane that actually saw a clamped term.
// Every translation unit holds its own copy of the registered pointer (no relocatable device code in this build);
// pp_set_saturation_word updates all of them (PP_SAT_SETTER below).  nullptr: reporting off.
static __device__ unsigned* pp_sat_dev_word = nullptr;
__device__ __forceinline__ void pp_sat_flag(bool bad) {
    if (bad) {
        unsigned* w = pp_sat_dev_word;
        if (w) atomicOr(w, 1u);
    }
}
// the same splits as above for a PRODUCER of an operand buffer: the value beyond the fp16 range is reported
__device__ __forceinline__ void pp_split_f16_chk(float v, _Float16& hi, _Float16& lo) {
    pp_sat_flag(!(fabsf(v) * PP_A_SCALE < 65504.f));
    pp_split_f16(v, hi, lo);
}
__device__ __forceinline__ _Float16 pp_to_f16_chk(float v) {
    pp_sat_flag(!(fabsf(v) * PP_A_SCALE < 65504.f));
    return pp_to_f16(v);
}
#define PP_SAT_SETTER(name)                                                                                                    \
    int name(unsigned* w) {                                                                                                    \
        return hipMemcpyToSymbol(HIP_SYMBOL(pp_sat_dev_word), &w, sizeof(w), 0, hipMemcpyHostToDevice) == hipSuccess ? PP_OK : PP_ELAUNCH; \
    }

// "hl" operand format (include/picopose_hip.h): half index of element (k, term p) inside a row
__device__ __forceinline__ int pp_hl_col(int k, int p) { return ((k >> 3) << 4) + (p << 3) + (k & 7); }

// One-time per-DEVICE initialisation (kernel attributes such as the > 64 KB dynamic-LDS opt-in are per device, and a
// process may drive several GPUs): `state[pp_cur_device()]` is 0 until initialised, then 1 (ok) or -1 (failed).
// Re-running an initialiser is harmless, so no lock is needed.
#define PP_MAX_DEVICES 64
static inline int pp_cur_device() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= PP_MAX_DEVICES) d = 0;
    return d;
}
static inline int pp_cu_count() {
    static int cus[PP_MAX_DEVICES];
    const int d = pp_cur_device();
    if (cus[d] == 0) {
        int n = 256;
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d);
        cus[d] = n > 0 ? n : 256;
    }
    return cus[d];
}

// device word the operand producers OR bit 0 into when a term hit the fp16 clamp (pp_set_saturation_word; NULL: off)
unsigned* pp_saturation_word();

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
    double* bytes = nullptr;    // algorithmic bytes of the launch: A + B + C (+ residuals) in the formats actually used
    int* kind = nullptr;
    int (*shape)[6] = nullptr;  // M, N, K, conv kernel size, chosen configuration (PP_GEMM_TRACE dump), A-delivery mode (0 dense, 1 / 2 conv)
    int capacity = 0;
    int count = 0;
};
PpGemmProf* pp_gemm_prof_state();

#endif
