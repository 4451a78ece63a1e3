// Measurement aid (not part of the library): read-only HBM streaming rates on this GPU for
// (a) a linear float4 grid-stride read and (b) the stage-1 bank access pattern (512-byte row
// segments, one half-template per workgroup), to calibrate the stage-1 roofline.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void read_linear(const f4* __restrict__ p, size_t n4, float* out) {
    f4 acc = {0, 0, 0, 0};
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        f4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
        acc += a + b + c + d;
    }
    for (; i < n4; i += stride) acc += p[i];
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = 1.f;
}
// one workgroup per (template, half): rows of 256 floats, this WG reads 128 of them per row
__global__ __launch_bounds__(256, 2) void read_pattern(const float* __restrict__ bank, int C, float* out) {
    const int item = blockIdx.x, half = item & 1;
    const size_t bn = item >> 1;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, lh = lane >> 5;
    const float* xp = bank + bn * (size_t)C * 256 + half * 128 + (size_t)(2 * w + lh) * 256 + 4 * l31;
    f4 acc = {0, 0, 0, 0};
    for (int c = 0; c < C; c += 64) {
        f4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = *(const f4*)(xp + (size_t)(c + 8 * j) * 256);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[j];
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = 1.f;
}
// one workgroup per template: full 1 KB rows
__global__ __launch_bounds__(256, 2) void read_rows(const float* __restrict__ bank, int C, float* out) {
    const size_t bn = blockIdx.x;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float* xp = bank + bn * (size_t)C * 256 + (size_t)w * 256 + 4 * lane;
    f4 acc = {0, 0, 0, 0};
    for (int c = 0; c < C; c += 32) {
        f4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = *(const f4*)(xp + (size_t)(c + 4 * j) * 256);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[j];
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = 1.f;
}
int main() {
    const int BN = 32 * 162, C = 768;
    const size_t n = (size_t)BN * C * 256;
    float *bank, *out;
    hipMalloc(&bank, n * 4);
    hipMalloc(&out, 4);
    hipMemset(bank, 0x11, n * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto time = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipDeviceSynchronize();
        float best = 1e9, tot = 0;
        for (int r = 0; r < 10; ++r) {
            hipEventRecord(e0);
            launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
            tot += ms;
        }
        printf("%-28s avg %.3f ms  best %.3f ms  -> %.0f GB/s (best %.0f)\n", name, tot / 10, best,
               n * 4 / (tot / 10) / 1e6, n * 4 / best / 1e6);
    };
    for (int g : {2048, 4096, 8192})
        time(g == 2048 ? "linear grid 2048" : g == 4096 ? "linear grid 4096" : "linear grid 8192",
             [&] { hipLaunchKernelGGL(read_linear, dim3(g), dim3(256), 0, 0, (const f4*)bank, n / 4, out); });
    time("pattern half-template WGs", [&] { hipLaunchKernelGGL(read_pattern, dim3(BN * 2), dim3(256), 0, 0, bank, C, out); });
    time("pattern whole-template WGs", [&] { hipLaunchKernelGGL(read_rows, dim3(BN), dim3(256), 0, 0, bank, C, out); });
    return 0;
}
