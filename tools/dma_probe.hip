// Measurement aid (not part of the library): L2 -> LDS delivery rate of the GEMM operand tiles by LDS-DMA
// (buffer_load_dwordx4 ... lds), for two global layouts of the same bytes:
//   SEG=64 : separate hi / lo planes, a K tile of 32 halfs = one 64-byte segment per row per plane
//   SEG=128: interleaved planes, a K tile = one 128-byte segment per row
// The access/reuse pattern mimics the 256x128-tile GEMM on a dense [M][K] operand: 2048 workgroups, each
// streaming 384 rows (256 of A, 128 of B) along K, B shared by all, A shared by `NCOL` column workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr int STAGE = 49152;  // bytes
template <int SEG>
__global__ __launch_bounds__(512, 1) void probe(const char* A, const char* B, long long a_bytes, long long b_bytes, int K,
                                                int ncol, int nk, int depth, float* out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const __amdgpu_buffer_rsrc_t Ar = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t Br = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)b_bytes, 0x00020000);
    const int orig = blockIdx.x, nwg = gridDim.x;
    const int xcd = orig & 7, q8 = nwg >> 3;
    const int wg = xcd * q8 + (orig >> 3);
    const int rt = wg / ncol, ct = wg % ncol;
    // per wave-instruction 1 KB: SEG=64 -> 16 rows x 64 B of one plane; SEG=128 -> 8 rows x 128 B (both planes)
    // bytes per row of the operand (both planes): 4*K.  plane-separated: plane p at p*rows*2K
    unsigned offA[4], offB[2];
    const long long rowsA = a_bytes / (4LL * K), rowsB = b_bytes / (4LL * K);
    if (SEG == 64) {
        const int r = lane >> 2, c = lane & 3;
        for (int j = 0; j < 2; ++j) {
            const long long row = (long long)rt * 256 + (j * 8 + w) * 16 + r;
            offA[2 * j] = (unsigned)((row % rowsA) * 2 * K + c * 16);
            offA[2 * j + 1] = (unsigned)(rowsA * 2 * K + (row % rowsA) * 2 * K + c * 16);
        }
        const long long row = (long long)ct * 128 + w * 16 + r;
        offB[0] = (unsigned)((row % rowsB) * 2 * K + c * 16);
        offB[1] = (unsigned)(rowsB * 2 * K + (row % rowsB) * 2 * K + c * 16);
    } else {
        const int r = lane >> 3, c = lane & 7;
        for (int j = 0; j < 4; ++j) {
            const long long row = (long long)rt * 256 + (j * 8 + w) * 8 + r;
            offA[j] = (unsigned)((row % rowsA) * 4 * K + c * 16);
        }
        for (int j = 0; j < 2; ++j) {
            const long long row = (long long)ct * 128 + (j * 8 + w) * 8 + r;
            offB[j] = (unsigned)((row % rowsB) * 4 * K + c * 16);
        }
    }
    auto fetch = [&](int stage, int kt) __attribute__((always_inline)) {
        char* st = lds + stage * STAGE;
        const unsigned ko = (unsigned)(kt * SEG);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(Ar, (lds_ptr_t)(st + (j * 8 + w) * 1024), 16, offA[j] + ko, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(Br, (lds_ptr_t)(st + 32768 + (j * 8 + w) * 1024), 16, offB[j] + ko, 0, 0, 0);
    };
    float acc = 0.f;
    if (depth == 2) {
        fetch(0, 0);
        fetch(1, 1);
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            fetch((kt + 2) % 3, (kt + 2) % nk);
            acc += *(const float*)(lds + (kt % 3) * STAGE + tid * 4);
        }
    } else {
        fetch(0, 0);
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            fetch((kt + 1) % 3, (kt + 1) % nk);
            acc += *(const float*)(lds + (kt % 3) * STAGE + tid * 4);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 123.456f) out[0] = acc;
}
int main(int argc, char** argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 768;           // halfs per row per plane
    const int M = 41120, N = argc > 2 ? atoi(argv[2]) : 3072;
    const int ncol = N / 128, nrow = (M + 255) / 256;
    const long long a_bytes = (long long)nrow * 256 * K * 4, b_bytes = (long long)N * K * 4;
    char *A, *B;
    float* out;
    (void)hipMalloc(&A, a_bytes);
    (void)hipMalloc(&B, b_bytes);
    (void)hipMalloc(&out, 4);
    (void)hipMemset(A, 0x11, a_bytes);
    (void)hipMemset(B, 0x11, b_bytes);
    (void)hipFuncSetAttribute((const void*)probe<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * STAGE);
    (void)hipFuncSetAttribute((const void*)probe<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * STAGE);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int nwg = nrow * ncol / 8 * 8, nk = K / 32;
    for (int depth = 1; depth <= 2; ++depth)
        for (int seg = 64; seg <= 128; seg += 64) {
            auto launch = [&]() {
                if (seg == 64) hipLaunchKernelGGL(probe<64>, dim3(nwg), dim3(512), 3 * STAGE, 0, A, B, a_bytes, b_bytes, K, ncol, nk, depth, out);
                else hipLaunchKernelGGL(probe<128>, dim3(nwg), dim3(512), 3 * STAGE, 0, A, B, a_bytes, b_bytes, K, ncol, nk, depth, out);
            };
            for (int i = 0; i < 2; ++i) launch();
            (void)hipEventRecord(e0, 0);
            for (int i = 0; i < 5; ++i) launch();
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            ms /= 5;
            const double bytes = (double)nwg * nk * STAGE;
            printf("K=%d N=%d seg=%3d B in-flight tiles=%d: %.3f ms  %.2f TB/s L2->LDS  (%.2f us per K tile per WG round)\n", K, N, seg, depth, ms,
                   bytes / ms / 1e9, ms * 1e3 / ((double)nwg / 256 * nk));
        }
    return 0;
}
