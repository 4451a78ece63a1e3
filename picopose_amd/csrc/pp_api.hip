// Library-level entry points of the C ABI (include/picopose_hip.h).
#include <cstdio>
#include <cstdlib>
#include "pp_common.h"

extern "C" {

const char* pp_strerror(int code) {
    switch (code) {
        case PP_OK: return "ok";
        case PP_EINVAL: return "invalid argument (null pointer, unsupported shape or mode)";
        case PP_EWORKSPACE: return "workspace too small or not 256-byte aligned";
        case PP_ELAUNCH: return "HIP launch or runtime call failed";
        default: return "unknown error code";
    }
}

int pp_version(void) { return 100; }

}  // extern "C"
int pp_sat_set_gemm(unsigned*);
int pp_sat_set_gemm_u1(unsigned*);
int pp_sat_set_gemm_u2(unsigned*);
int pp_sat_set_gemm_uh(unsigned*);
int pp_sat_set_attn(unsigned*);
int pp_sat_set_sample(unsigned*);
static unsigned* g_sat_word[PP_MAX_DEVICES];
unsigned* pp_saturation_word() { return g_sat_word[pp_cur_device()]; }
extern "C" {

int pp_set_saturation_word(unsigned int* word) {
    // every translation unit with a producer kernel holds its own device-side copy of the pointer (pp_common.h)
    int (*const setters[])(unsigned*) = {pp_sat_set_gemm, pp_sat_set_gemm_u1, pp_sat_set_gemm_u2, pp_sat_set_gemm_uh, pp_sat_set_attn, pp_sat_set_sample};
    for (auto f : setters) {
        const int rc = f(word);
        if (rc != PP_OK) return rc;
    }
    g_sat_word[pp_cur_device()] = word;      // (the Winograd transforms take it as a kernel argument)
    return PP_OK;
}

// ---- optional in-library timing of the dominant kernel (bench.py's roofline leg) ----------
}  // extern "C"
static PpProf g_prof;
PpProf* pp_prof_state() { return &g_prof; }  // internal (C++ linkage, hidden from the header)
static PpGemmProf g_gemm_prof;
PpGemmProf* pp_gemm_prof_state() { return &g_gemm_prof; }
extern "C" {

int pp_prof_enable(int max_records) {
    PpProf& p = g_prof;
    for (int i = 0; i < p.capacity * 2; ++i) (void)hipEventDestroy(p.ev[i]);
    delete[] p.ev;
    p.ev = nullptr;
    p.capacity = p.count = 0;
    if (max_records <= 0) return PP_OK;
    p.ev = new hipEvent_t[2 * max_records];
    for (int i = 0; i < 2 * max_records; ++i)
        if (hipEventCreate(&p.ev[i]) != hipSuccess) return PP_ELAUNCH;
    p.capacity = max_records;
    return PP_OK;
}

int pp_prof_collect(float* out_ms, int max_out, int* count) {
    PpProf& p = g_prof;
    if (!out_ms || !count) return PP_EINVAL;
    int n = p.count < max_out ? p.count : max_out;
    for (int i = 0; i < n; ++i) {
        if (hipEventSynchronize(p.ev[2 * i + 1]) != hipSuccess) return PP_ELAUNCH;
        if (hipEventElapsedTime(&out_ms[i], p.ev[2 * i], p.ev[2 * i + 1]) != hipSuccess)
            return PP_ELAUNCH;
    }
    *count = n;
    p.count = 0;
    return PP_OK;
}

int pp_prof_gemm_enable(int max_records) {
    PpGemmProf& p = g_gemm_prof;
    for (int i = 0; i < p.capacity * 2; ++i) (void)hipEventDestroy(p.ev[i]);
    delete[] p.ev;
    delete[] p.flops;
    delete[] p.bytes;
    delete[] p.kind;
    delete[] p.shape;
    p.shape = nullptr;
    p.ev = nullptr;
    p.flops = nullptr;
    p.bytes = nullptr;
    p.kind = nullptr;
    p.capacity = p.count = 0;
    if (max_records <= 0) return PP_OK;
    p.ev = new hipEvent_t[2 * max_records];
    p.flops = new double[max_records];
    p.bytes = new double[max_records];
    p.kind = new int[max_records];
    p.shape = new int[max_records][6];
    for (int i = 0; i < 2 * max_records; ++i)
        if (hipEventCreate(&p.ev[i]) != hipSuccess) return PP_ELAUNCH;
    p.capacity = max_records;
    return PP_OK;
}

int pp_prof_gemm_collect(double* ms, double* flops, int* launches) {
    PpGemmProf& p = g_gemm_prof;
    if (!ms || !flops || !launches) return PP_EINVAL;
    for (int k = 0; k < 2; ++k) {
        ms[k] = flops[k] = 0.0;
        launches[k] = 0;
    }
    const bool trace = getenv("PP_GEMM_TRACE") != nullptr;
    for (int i = 0; i < p.count; ++i) {
        float t = 0.f;
        if (hipEventSynchronize(p.ev[2 * i + 1]) != hipSuccess) return PP_ELAUNCH;
        if (hipEventElapsedTime(&t, p.ev[2 * i], p.ev[2 * i + 1]) != hipSuccess) return PP_ELAUNCH;
        const int k = p.kind[i];
        if (trace)
            fprintf(stderr, "[pp_gemm] kind=%d M=%d N=%d K=%d k=%d cfg=%d %.4f ms %.1f TFLOP/s\n", k, p.shape[i][0], p.shape[i][1],
                    p.shape[i][2], p.shape[i][3], p.shape[i][4], t, p.flops[i] / (t * 1e-3) / 1e12);
        ms[k] += t;
        flops[k] += p.flops[i];
        launches[k]++;
    }
    p.count = 0;
    return PP_OK;
}

int pp_prof_gemm_records2(int max_records, int* shape, float* ms, double* flops, double* bytes, int* count);
int pp_prof_gemm_records(int max_records, int* shape, float* ms, double* flops, int* count) {
    return pp_prof_gemm_records2(max_records, shape, ms, flops, nullptr, count);
}

int pp_prof_gemm_records2(int max_records, int* shape, float* ms, double* flops, double* bytes, int* count) {
    PpGemmProf& p = g_gemm_prof;
    if (!shape || !ms || !flops || !count || max_records < 0) return PP_EINVAL;
    const int n = p.count < max_records ? p.count : max_records;
    for (int i = 0; i < n; ++i) {
        if (hipEventSynchronize(p.ev[2 * i + 1]) != hipSuccess) return PP_ELAUNCH;
        if (hipEventElapsedTime(&ms[i], p.ev[2 * i], p.ev[2 * i + 1]) != hipSuccess) return PP_ELAUNCH;
        const int st = bytes ? 8 : 6;      // records2: {M, N, K, conv kernel size, cfg, kind, A-delivery mode, 0}
        for (int k = 0; k < 5; ++k) shape[st * i + k] = p.shape[i][k];
        shape[st * i + 5] = p.kind[i];
        flops[i] = p.flops[i];
        if (bytes) {
            bytes[i] = p.bytes[i];
            shape[st * i + 6] = p.shape[i][5];
            shape[st * i + 7] = 0;
        }
    }
    *count = n;
    return PP_OK;
}

}  // extern "C"
