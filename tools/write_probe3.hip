// write probe 3: how long does ONE tile-epilogue burst (256 KB per CU) take to drain when bursts recur every ~50 us (a GEMM's rhythm)?
//   sync = 1: every workgroup bursts at the same moment (64 MB chip-wide per burst);  sync = 0: four phase groups, a quarter period apart
//   pattern 0: the epilogue's half-line pieces; pattern 1: whole 128-byte lines per instruction
// in-kernel timing: s_memrealtime (100 MHz) around [32 stores per wave ; s_waitcnt vmcnt(0)], mean / max over workgroups and bursts.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int PAT>
__global__ __launch_bounds__(512) void wr(float* out, int tiles, int ld, int period_ticks, int sync, unsigned* times) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned long long start = __builtin_amdgcn_s_memrealtime();
    const unsigned long long phase = sync ? 0ull : (unsigned long long)(((blockIdx.x >> 3) & 3) * (period_ticks / 4));
    for (int t = 0; t < tiles; ++t) {
        while (__builtin_amdgcn_s_memrealtime() - start < phase + (unsigned long long)t * period_ticks) __builtin_amdgcn_s_sleep(4);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        f4 v = {1.f * t, 2.f, 3.f, 4.f};
        if (PAT == 0) {
            float* base = out + ((size_t)(t * gridDim.x + blockIdx.x) * 256) * ld + (w & 3) * 64 + (size_t)(w >> 2) * 128 * ld;
            for (int mi = 0; mi < 8; ++mi)
                for (int jp = 0; jp < 2; ++jp) {
                    float* p = base + (size_t)(mi * 16 + (lane & 15)) * ld + jp * 32 + (lane >> 4) * 8;
                    *(f4*)p = v; *(f4*)(p + 4) = v;
                }
        } else {
            float* base = out + ((size_t)(t * gridDim.x + blockIdx.x) * 256) * 256 + (size_t)w * 32 * 256;
            for (int i = 0; i < 32; ++i) *(f4*)(base + i * 256 + lane * 4) = v;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        if (threadIdx.x == 0) times[blockIdx.x * tiles + t] = (unsigned)(t1 - t0);
    }
}
int main() {
    const int ld = 2304, tiles = 40, cus = 256, period = 5000;   // 50 us between bursts
    float* out; size_t n = (size_t)tiles * cus * 256 * ld;
    hipMalloc(&out, n * 4);
    unsigned* times; hipMalloc(&times, cus * tiles * 4);
    std::vector<unsigned> h(cus * tiles);
    for (int pat = 0; pat < 2; ++pat)
        for (int sync = 1; sync >= 0; --sync) {
            for (int r = 0; r < 2; ++r) {
                if (pat == 0) wr<0><<<cus, 512>>>(out, tiles, ld, period, sync, times); else wr<1><<<cus, 512>>>(out, tiles, ld, period, sync, times);
                hipDeviceSynchronize();
            }
            hipMemcpy(h.data(), times, h.size() * 4, hipMemcpyDeviceToHost);
            double sum = 0; unsigned mx = 0; std::vector<unsigned> v;
            for (int b = 0; b < cus; ++b) for (int t = 5; t < tiles; ++t) { v.push_back(h[b * tiles + t]); }
            std::sort(v.begin(), v.end());
            for (auto x : v) sum += x;
            printf("pattern %d, %s bursts of 256 KB per CU every 50 us: drain (issue -> vmcnt(0)) mean %.2f us, median %.2f, p95 %.2f, max %.2f us\n", pat,
                   sync ? "SYNCHRONISED (64 MB chip-wide)" : "4 phase groups (16 MB at a time)", sum / v.size() / 100, v[v.size() / 2] / 100.0, v[v.size() * 95 / 100] / 100.0, v.back() / 100.0);
        }
    return 0;
}
