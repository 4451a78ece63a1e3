// write probe 2: is the epilogue's store cost a per-CU limit or a chip-wide (HBM) one, and does the access pattern matter?
//   pattern 0: the GEMM epilogue's (two 16-B stores per lane covering 32 contiguous bytes, 4 lanes = one 128-B line of a row, 16 rows per instruction)
//   pattern 1: wave-contiguous (lane l stores 16 B at +16 l: every instruction writes 8 whole 128-B lines)
// for grids of 256 / 64 / 16 / 4 workgroups (one per CU), each storing `tiles` tiles of 256 KB.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int PAT>
__global__ __launch_bounds__(512) void wr(float* out, int tiles, int ld) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int t = 0; t < tiles; ++t) {
        f4 v = {1.f * t, 2.f, 3.f, 4.f};
        if (PAT == 0) {
            float* base = out + ((size_t)(t * gridDim.x + blockIdx.x) * 256) * ld + (w & 3) * 64 + (size_t)(w >> 2) * 128 * ld;
            for (int mi = 0; mi < 8; ++mi)
                for (int jp = 0; jp < 2; ++jp) {
                    float* p = base + (size_t)(mi * 16 + (lane & 15)) * ld + jp * 32 + (lane >> 4) * 8;
                    *(f4*)p = v; *(f4*)(p + 4) = v;
                }
        } else {
            float* base = out + ((size_t)(t * gridDim.x + blockIdx.x) * 256) * 256 + (size_t)w * 32 * 256;   // 32 KB per wave, contiguous
            for (int i = 0; i < 32; ++i) *(f4*)(base + i * 256 + lane * 4) = v;
        }
    }
}
int main() {
    const int ld = 2304, tiles = 24;
    float* out; size_t n = (size_t)tiles * 256 * 256 * ld;
    hipMalloc(&out, n * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pat = 0; pat < 2; ++pat)
        for (int g : {256, 64, 16, 4}) {
            float best = 1e30f;
            for (int r = 0; r < 4; ++r) {
                hipEventRecord(e0);
                for (int k = 0; k < 3; ++k) { if (pat == 0) wr<0><<<g, 512>>>(out, tiles, ld); else wr<1><<<g, 512>>>(out, tiles, ld); }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3; best = ms < best ? ms : best;
            }
            const double bytes = (double)tiles * g * 256.0 * 256 * 4;
            printf("pattern %d, %3d workgroups: %.1f MB in %.3f ms = %.2f TB/s, %.1f B/clk per workgroup's CU at 2.1 GHz\n", pat, g, bytes / 1e6, best, bytes / best / 1e9,
                   bytes / g / (best * 1e-3 * 2.1e9));
        }
    return 0;
}
