// write-burst probe: every CU's 512-thread workgroup stores `tiles` tiles of 256 KB (dwordx4 per lane, rows of 128 B per 4 lanes like the GEMM epilogue)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void wr(float* out, int tiles, int ld) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int t = 0; t < tiles; ++t) {
        float* base = out + ((size_t)(t * gridDim.x + blockIdx.x) * 256) * ld + (w & 3) * 64 + (size_t)(w >> 2) * 128 * ld;
        f4 v = {1.f * t, 2.f, 3.f, 4.f};
        for (int mi = 0; mi < 8; ++mi)
            for (int jp = 0; jp < 2; ++jp) {
                float* p = base + (size_t)(mi * 16 + (lane & 15)) * ld + jp * 32 + (lane >> 4) * 8;
                *(f4*)p = v; *(f4*)(p + 4) = v;
            }
    }
}
int main() {
    const int ld = 2304, tiles = 6, cus = 256;
    float* out; size_t n = (size_t)tiles * cus * 256 * ld;
    hipMalloc(&out, n * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0); for (int k = 0; k < 5; ++k) wr<<<cus, 512>>>(out, tiles, ld); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("write burst: %.1f MB in %.3f ms = %.2f TB/s (%.1f B/cycle/CU at 2.1 GHz)\n", tiles * cus * 256.0 * 256 * 4 / 1e6, ms, tiles * cus * 256.0 * 256 * 4 / ms / 1e9, tiles * 256.0 * 256 * 4 / (ms * 1e-3 * 2.1e9));
    }
    return 0;
}
