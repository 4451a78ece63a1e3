// What the fp32 matrix pipe sustains: v_mfma_f32_32x32x2_f32 back to back from W waves per CU (4 = one per SIMD, 8 = two) with NACC
// independent accumulators per wave and VALU extra v_max instructions per MFMA — the ceiling the fp32 engine's K loop is priced against.
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f32_probe.hip -o tools/mfma_f32_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int VALU>
__global__ __launch_bounds__(512, 1) void probe(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float a = a0 + threadIdx.x, b = b0, x = a0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
                for (int v = 0; v < VALU; ++v) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(b));
            }
    }
    float s = x;
    for (int i = 0; i < NACC; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == -12345.f) out[threadIdx.x] = s;
}

template <int NACC, int VALU>
void run(int waves, float* out) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    probe<NACC, VALU><<<256, waves * 64>>>(out, 100, 1.f, 2.f);
    hipEventRecord(e0);
    probe<NACC, VALU><<<256, waves * 64>>>(out, iters, 1.f, 2.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double fl = 256.0 * waves * iters * 4.0 * NACC * 4096.0;
    printf("waves/CU %d  accumulators %d  v_max per MFMA %d: %.3f ms  %.1f TFLOP/s\n", waves, NACC, VALU, ms, fl / ms / 1e9);
}

int main() {
    float* out;
    hipMalloc(&out, 4096);
    for (int rep = 0; rep < 2; ++rep) {
        run<4, 0>(4, out);
        run<8, 0>(4, out);
        run<4, 0>(8, out);
        run<8, 0>(8, out);
        run<8, 1>(8, out);
        run<8, 2>(8, out);
        run<8, 4>(8, out);
        run<8, 2>(4, out);
    }
    return 0;
}
