// Stage-1 template matching for MI355X (gfx950): one fused pass over the
// template bank that never materialises the (B,N,256,256) similarity volume.
//
// Replaces utils/matching.py:29-69 (matching_templates) of the reference.
// Per (crop b, template n) the reference computes
//     sim[t,s]   = <q_hat[:,t], x_hat[:,s]> * m[t]               (matching.py:47-48)
//     score[t]   = max_s sim[t,s],  i1[t] = argmax_s sim[t,s]    (:50)
//     i2[s]      = argmax_t sim[t,s]                              (:51)
//     mask_all   = m * (i2 != 0) * (i1 != 0)                      (:56-60)
//     sim_avg    = sum(score * mask_all) / 256  (0 if mask_all is empty)  (:63-67)
// Only `score`, "is the arg-max patch 0?" and sim_avg leave the function, so
// the kernel keeps the 256x256 tile in MFMA accumulators and reduces it in
// registers/LDS:  i1[t] != 0  <=>  max_s sim[t,s] > sim[t,0]   (first max wins
// ties in torch.max, so a tie with column 0 yields index 0), same for i2.
//
// Work decomposition: a work item is one template against all 256 query patches; a
// 512-thread workgroup (8 waves: 2 template halves x 2 x 2) processes it with a 128x64
// fp32 tile per wave in 128 accumulator registers.  Workgroups are persistent (one per
// CU) and walk the items in bank order.  The bank is read exactly once (coalesced
// 16 B/lane, 512 B row segments, three K-steps ahead in registers, bounded buffer loads,
// nt policy); the pre-normalised, pre-masked query operand comes from L2 by LDS-DMA into
// a 3-slot ring shared by both halves (the copy instructions, not their bytes, are what
// costs: with 4-wave workgroups on template halves — PP_S1_WAVES=4, two per CU — the
// kernel issues twice as many and runs 9 % slower).  The tile stream rolls from one item
// into the next, so the next item's first tiles are in flight during the epilogue.
//
// Two arithmetic modes share the skeleton:
//   EXACT  v_mfma_f32_32x32x2_f32 — bit-for-bit an fp32 fma chain over c.
//   FAST   v_mfma_f32_32x32x16_f16 on fp16-rounded operands (fp32 accumulate),
//          HBM-bound; every row/column whose "index 0" decision lies within
//          eps of a tie is re-evaluated by pp_s1_fixup in exact fp32, so the
//          discrete outputs agree with EXACT mode.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "../../include/picopose_hip.h"
#include "pp_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));

// cache policy of the bank loads: 2 = nt (streamed once: keeps the query operand resident in L2;
// measured HBM fetch 4.25 GB per launch against 4.45 GB with the default policy, 4.10 GB algorithmic)
#ifndef PP_S1_XAUX
#define PP_S1_XAUX 2
#endif
constexpr int P = 256;           // patches per image (16x16), fixed by the reference
constexpr int XROW_F16 = 320;    // bytes per k-row of the fp16 X tile (256 + 64 pad:
                                 // the 4 rows of a ds_read_b64_tr_b16 block land on
                                 // disjoint 16-dword bank groups)
constexpr int TROW = 12;         // floats per row of the epilogue transpose tile (8 used)

struct S1Ws {
    _Float16* qh;   // (B, C/32, 2, 8, 64, 8) fp16 A-fragment order, normalised*mask
    float* qf;      // (B, C, 256) fp32 normalised*mask
    float* m16;     // (B, 256) sampled mask
    float* denom;   // (B, 256) max(||q_t||, 1e-12)
    float4* rowrec; // (B*N, 2, 256) per-half row records over s > 0: {best, second, arg bits, -}
    float* simt0;   // (B*N, 256) sim[t,0]
    float* colmax;  // (B*N, 256) column maxima over t > 0
    float* sim0s;   // (B*N, 256) sim[0,s]
    int* counters;  // (B) resolve workgroups of a crop that have written their sim_avg (fused resolve + top-k)
    size_t total;
};

__host__ inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

__host__ S1Ws carve(void* base, int B, int N, int C) {
    S1Ws w;
    char* p = (char*)base;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* r = p + off; off += align256(bytes); return r; };
    size_t BN = (size_t)B * N;
    w.qh = (_Float16*)take((size_t)B * C * P * 2);
    w.qf = (float*)take((size_t)B * C * P * 4);
    w.m16 = (float*)take((size_t)B * P * 4);
    w.denom = (float*)take((size_t)B * P * 4);
    w.rowrec = (float4*)take(BN * 2 * P * 16);
    w.simt0 = (float*)take(BN * P * 4);
    w.colmax = (float*)take(BN * P * 4);
    w.sim0s = (float*)take(BN * P * 4);
    w.counters = (int*)take((size_t)B * 4);
    w.total = off;
    return w;
}

// ---------------------------------------------------------------------------
// Query pre-pack: F.normalize(tar_feat, dim=1) (matching.py:40), the nearest
// 16x16 resample of the mask (matching.py:38-39) and the row mask multiply
// (matching.py:48) folded into the A operand.
//   s1_qnorm : grid (B,4)   — ||q_t|| for 64 patches per workgroup, fixed-order reduction
//   s1_qpack : grid (B,C/32) — 32 channels x 256 patches per workgroup
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void s1_qnorm(const float* __restrict__ query,
                                                 const float* __restrict__ mask, int mh, int mw,
                                                 int C, float* __restrict__ denom,
                                                 float* __restrict__ m16) {
    __shared__ float part[16][64];
    const int b = blockIdx.x, tq = blockIdx.y;
    const int tl = threadIdx.x & 63, cs = threadIdx.x >> 6;
    const int t = tq * 64 + tl;
    const float* q = query + (size_t)b * C * P + t;
    float ss = 0.f;
    for (int c0 = cs; c0 < C; c0 += 64) {  // this thread: channels == cs (mod 16); C % 64 == 0
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = q[(size_t)(c0 + 16 * j) * P];
#pragma unroll
        for (int j = 0; j < 4; ++j) ss = fmaf(v[j], v[j], ss);
    }
    part[cs][tl] = ss;
    __syncthreads();
    if (cs == 0) {
        float tot = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) tot += part[j][tl];
        denom[b * P + t] = fmaxf(sqrtf(tot), 1e-12f);
        // nearest: src = min(floor(dst * (float)in/out), in-1)   (ATen nearest_idx)
        const int py = t >> 4, px = t & 15;
        const float sy = (float)mh / 16.0f, sx = (float)mw / 16.0f;
        int iy = (int)floorf((float)py * sy);
        int ix = (int)floorf((float)px * sx);
        iy = iy < mh - 1 ? iy : mh - 1;
        ix = ix < mw - 1 ? ix : mw - 1;
        m16[b * P + t] = mask[((size_t)b * mh + iy) * mw + ix];
    }
}

__global__ __launch_bounds__(256) void s1_qpack(const float* __restrict__ query,
                                                const float* __restrict__ denom,
                                                const float* __restrict__ m16, int C,
                                                _Float16* __restrict__ qh, float* __restrict__ qf) {
    const int b = blockIdx.x, ks = blockIdx.y, t = threadIdx.x;
    const int KT = C >> 5;
    const float d = denom[b * P + t], m = m16[b * P + t];
    const float* q = query + ((size_t)b * C + ks * 32) * P + t;
    float* qo = qf + ((size_t)b * C + ks * 32) * P + t;
    const int tb = t >> 5, tl = t & 31;
    float v[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) v[j] = q[(size_t)j * P];
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        v[j] = (v[j] / d) * m;
        qo[(size_t)j * P] = v[j];
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {  // 8 consecutive channels -> one 16-byte A-fragment piece
        h8 pk;
#pragma unroll
        for (int j = 0; j < 8; ++j) pk[j] = (_Float16)v[g * 8 + j];
        const int kh = g >> 1, hh = g & 1;
        const size_t off = ((((size_t)b * KT + ks) * 2 + kh) * 8 + tb) * 512 + (tl + 32 * hh) * 8;
        *(h8*)(qh + off) = pk;
    }
}

// s1_qprep = s1_qnorm + s1_qpack in ONE launch for small problems (BASELINE configs[1]: 8 crops, C = 384 — there the two
// launches and the gap between them were 12 of the call's 70 us; tried in round 4, measured slower, kept behind PP_S1_QPREP=1): every (crop, 32-channel slice) workgroup recomputes the
// crop's 256 patch norms itself (the crop's C x 256 floats come from L2; B C^2 32 bytes in total, so only for small B C^2)
// in EXACTLY s1_qnorm's summation order — 16 partial sums over the channels of one residue class mod 16, ascending, then
// their sum in class order — so `denom` and everything packed from it keep their bits.  Block (0, 0) also zeroes the
// per-crop arrival counters of the fused resolve + top-k (pp_stage1_match).
__global__ __launch_bounds__(256) void s1_qprep(const float* __restrict__ query, const float* __restrict__ mask, int mh, int mw, int C,
                                                float* __restrict__ denom, float* __restrict__ m16g, _Float16* __restrict__ qh,
                                                float* __restrict__ qf, int* __restrict__ counters, int ncounters) {
    const int b = blockIdx.x, ks = blockIdx.y, t = threadIdx.x;
    if (b == 0 && ks == 0 && counters)
        for (int j = t; j < ncounters; j += 256) counters[j] = 0;
    const float* qc = query + (size_t)b * C * P + t;
    float ss[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) ss[j] = 0.f;
    for (int c0 = 0; c0 < C; c0 += 16) {      // channel c0 + j belongs to class j: ascending within every class
        float v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = qc[(size_t)(c0 + j) * P];
#pragma unroll
        for (int j = 0; j < 16; ++j) ss[j] = fmaf(v[j], v[j], ss[j]);
    }
    float tot = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) tot += ss[j];
    const float d = fmaxf(sqrtf(tot), 1e-12f);
    const int py = t >> 4, px = t & 15;
    const float sy = (float)mh / 16.0f, sx = (float)mw / 16.0f;
    int iy = (int)floorf((float)py * sy), ix = (int)floorf((float)px * sx);
    iy = iy < mh - 1 ? iy : mh - 1;
    ix = ix < mw - 1 ? ix : mw - 1;
    const float m = mask[((size_t)b * mh + iy) * mw + ix];
    if (ks == 0) {
        denom[b * P + t] = d;
        m16g[b * P + t] = m;
    }
    // the pack of s1_qpack
    const int KT = C >> 5;
    const float* q = query + ((size_t)b * C + ks * 32) * P + t;
    float* qo = qf + ((size_t)b * C + ks * 32) * P + t;
    const int tb = t >> 5, tl = t & 31;
    float v[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) v[j] = q[(size_t)j * P];
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        v[j] = (v[j] / d) * m;
        qo[(size_t)j * P] = v[j];
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        h8 pk;
#pragma unroll
        for (int j = 0; j < 8; ++j) pk[j] = (_Float16)v[g * 8 + j];
        const int kh = g >> 1, hh = g & 1;
        const size_t off = ((((size_t)b * KT + ks) * 2 + kh) * 8 + tb) * 512 + (tl + 32 * hh) * 8;
        *(h8*)(qh + off) = pk;
    }
}

// ---------------------------------------------------------------------------
// Main kernel
// ---------------------------------------------------------------------------
template <int MODE>
struct Cfg;
template <>
struct Cfg<PP_MATCH_EXACT> {
    static constexpr int KS = 16;                 // channels per K-step
    static constexpr int XL = 2;                  // 16-byte X loads per thread per step
    static constexpr int XS_BYTES = 16 * 128 * 4; // [16][128] fp32
};
template <>
struct Cfg<PP_MATCH_FAST> {
    static constexpr int KS = 32;
    static constexpr int XL = 4;
    static constexpr int XS_BYTES = 32 * XROW_F16;  // [32][160 halfs] (128 used)
};

// LDS: [query ring: 3 slots of 16 KB][X tile buffer 1][X tile buffer 0][small epilogue arrays].
// The epilogue's transpose tile T lives in ring slot 2 + X buffer 1: slots 0/1 receive the NEXT item's first
// query tiles while the epilogue runs, and X buffer 0 (written by the next item's step 0) stays untouched, so
// the step's own barrier is the only one needed between an epilogue and the following K loop.
constexpr int QS = 16384;                         // one query tile: 16 fragment chunks of 1 KB (fast) / [16][256] fp32
constexpr int Q_RING = 3;
constexpr int XS_MAX = Cfg<PP_MATCH_FAST>::XS_BYTES;
// NW = waves per workgroup: 4 (a work item is one half of a template, two workgroups per CU) or 8 (a work item
// is a whole template — both halves share every query tile, which halves the LDS-DMA copies — one per CU).
template <int NW>
struct Lay {
    static constexpr int HV = NW / 4;                    // template halves per workgroup
    static constexpr int XS1_OFF = Q_RING * QS;          // X buffers of odd K-steps, one per half
    static constexpr int T_BYTES = NW * 128 * TROW * 4;  // float[NW waves][128 rows][TROW]
    static constexpr int T_PAD = T_BYTES > QS + HV * XS_MAX ? T_BYTES - QS - HV * XS_MAX : 0;
    static constexpr int XS0_OFF = XS1_OFF + HV * XS_MAX + T_PAD;  // X buffers of even K-steps
    static constexpr int EPI_T = 2 * QS;                 // ring slot 2 + X buffers 1 (+ pad)
    static constexpr int EPI_RED = XS0_OFF + HV * XS_MAX;          // float[HV][8][128]
    static constexpr int EPI_COLP = EPI_RED + HV * 8 * 128 * 4;    // float[HV][2][128]
    static constexpr int EPI_SIM0 = EPI_COLP + HV * 2 * 128 * 4;   // float[HV][128]
    static constexpr int EPI_ST0 = EPI_SIM0 + HV * 128 * 4;        // float[256]
    // results of up to PARK items wait here and leave in one burst (NW == 8: LDS to spare; NW == 4: registers)
    static constexpr int PARK = NW == 8 ? 4 : 0;
    static constexpr int PARK_OFF = EPI_ST0 + 256 * 4;   // per item: float4[HV][256], float[HV][128] x 2, float[256]
    static constexpr int PARK_ITEM = HV * 256 * 16 + 2 * HV * 128 * 4 + 256 * 4;
    static constexpr int SMEM_BYTES = PARK_OFF + PARK * PARK_ITEM;
    static_assert(EPI_T + T_BYTES <= XS0_OFF, "epilogue transpose tile overlaps X buffer 0");
    static_assert((NW == 4 ? 2 : 1) * SMEM_BYTES <= 160 * 1024, "workgroups per CU");
};
static_assert(Cfg<PP_MATCH_EXACT>::XS_BYTES <= XS_MAX, "");
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// Epilogue arithmetic as single instructions.  fmaxf() through hipcc costs up to three (it canonicalises both
// inputs with v_max x,x first), and update_dpp + fmaxf costs four (zero the destination, v_mov_dpp, canonicalise,
// max): the epilogue is VALU-bound, so these are inline asm.  NaN handling is v_max_f32's (IEEE mode: a quiet NaN
// input loses), as before.
__device__ __forceinline__ float vmax(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmin(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmax3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// max(own, partner's `src`) with the partner = lane^1 / lane^2 of the quad (DPP, no LDS traffic).  The s_nop
// covers the 2 wait states a DPP read needs after a VALU write of the same register.
__device__ __forceinline__ float vmax_xor1(float own, float src) {
    float r;
    asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(src), "v"(own));
    return r;
}
__device__ __forceinline__ float vmax_xor2(float own, float src) {
    float r;
    asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(src), "v"(own));
    return r;
}

// Work item v -> (crop, template, half), in bank order: the workgroups resident at any time stream one
// contiguous window of the bank (measured 4 % faster than keeping each crop on one XCD; the query operand is
// then cached by every XCD's L2, 393 KB per crop and XCD against 127 MB of bank per crop).
template <int NW>
__device__ __forceinline__ void s1_item(int v, int N, int& b, int& n, int& half) {
    if (NW == 8) {  // whole templates
        b = v / N;
        n = v % N;
        half = 0;
        return;
    }
    const int per_crop = 2 * N;
    b = v / per_crop;
    const int r = v % per_crop;
    n = r >> 1;
    half = r & 1;
}

// Persistent workgroups of NW waves.  NW = 8 (default): a work item = (crop b, template n), waves 0-3 own
// template patches 0..127 and waves 4-7 patches 128..255, one workgroup per CU.  NW = 4: a work item =
// (crop, template, half), two workgroups per CU.  Within a half the 4 waves are 2 x 2 over (128 query patches,
// 64 template patches): each holds a 128x64 fp32 tile in 128 accumulator registers.  The K loop runs over a tile stream that rolls from one item into the next, so the next
// item's first tiles are in flight while the epilogue of the current one runs.
// XT = element type of the bank in HBM: float (the reference's layout) or _Float16 (a bank stored in half precision —
// BASELINE configs[4]: half the bytes per template; the values ARE the fp16-rounded features, all arithmetic as before).
template <int MODE, int NW, typename XT = float>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void s1_main(const XT* __restrict__ bank,
                                                  const _Float16* __restrict__ qh,
                                                  const float* __restrict__ qf, int N, int C, int total,
                                                  float4* __restrict__ rowrec,
                                                  float* __restrict__ simt0,
                                                  float* __restrict__ colmax,
                                                  float* __restrict__ sim0s) {
    using K = Cfg<MODE>;
    using L = Lay<NW>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int QELT = MODE == PP_MATCH_FAST ? 2 : 4;
    constexpr int ES = (int)sizeof(XT);      // bytes per bank element
    typedef typename std::conditional<ES == 4, f4, u2>::type xreg_t;   // 4 bank elements of one thread

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hw = NW == 8 ? w8 >> 2 : 0;  // NW == 8: waves 0-3 own template half 0, waves 4-7 half 1
    const int w = w8 & 3;                  // wave within its half: 2 x 2 over (128 query rows, 64 columns)
    const int wr = w >> 1, wc = w & 1;
    const int th = tid & 255;              // thread within its half
    const int l31 = lane & 31, lh = lane >> 5;
    const int KT = C / K::KS;
    const int KT3 = (KT + 2) / 3 * 3;  // K-steps per item; steps past KT multiply zero tiles
    const int G = gridDim.x;

    // ---- first item of this workgroup, and the one after it
    int b, n, half;
    int v_cur = blockIdx.x;
    if (v_cur >= total) return;
    s1_item<NW>(v_cur, N, b, n, half);
    if (NW == 8) half = hw;
    int v_nxt = v_cur + G;

    char* Qring = smem;

    // Tile stream.  Bank tiles: buffer loads bounded to the item's slice (base = its first row at this half's
    // column offset), query tiles: linear 16 KB copies global -> LDS by LDS-DMA (1 KB per wave instruction).
    // Tiles past the end of a slice read zeros without memory traffic, so every step issues the same loads and
    // the counted waits below are exact.  After an item's last tile the descriptors switch to the next item's
    // (an empty slice when there is none).
#define X_DESC(b_, n_, half_, ok_)                                                                         \
    __builtin_amdgcn_make_buffer_rsrc((void*)(bank + ((size_t)(b_) * N + (n_)) * (size_t)C * P + (half_) * 128), 0, \
                                      (ok_) ? C * P * ES - (half_) * 128 * ES : 0, 0x00020000)
#define Q_DESC(b_, ok_)                                                                                     \
    __builtin_amdgcn_make_buffer_rsrc(MODE == PP_MATCH_FAST ? (void*)(qh + (size_t)(b_) * C * P)           \
                                                            : (void*)(qf + (size_t)(b_) * C * P),          \
                                      0, (ok_) ? C * P * QELT : 0, 0x00020000)
    __amdgpu_buffer_rsrc_t Xd = X_DESC(b, n, NW == 8 ? 0 : half, true), Qd = Q_DESC(b, true);
    __amdgpu_buffer_rsrc_t XdN, QdN;
    {
        int b2, n2, h2;
        const bool ok = v_nxt < total;
        s1_item<NW>(ok ? v_nxt : v_cur, N, b2, n2, h2);
        XdN = X_DESC(b2, n2, h2, ok);
        QdN = Q_DESC(b2, ok);
    }
    int xt = 0, qt = -1;  // next tile of the stream; the very first query copy is a dummy (out of bounds: zeros)
    const unsigned xvoff = ((NW == 8 ? hw * 128 : 0) + (2 * w + lh) * P + 4 * l31) * ES;
    const unsigned qvoff = tid * 16;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // FAST: sum of squares of this thread's 4 columns; EXACT: ssq0/ssq1 = column blocks sb 0/1
    float ssq0 = 0.f, ssq1 = 0.f, ssq2 = 0.f, ssq3 = 0.f;
    xreg_t x0[K::XL], x1[K::XL], x2[K::XL];  // X tiles of the next three K-steps, in flight

#define LOAD_XB(x_)                                                                               \
    do {                                                                                          \
        _Pragma("unroll") for (int j = 0; j < K::XL; ++j) {                                       \
            if constexpr (ES == 4)                                                                \
                x_[j] = __builtin_bit_cast(xreg_t, __builtin_amdgcn_raw_buffer_load_b128(Xd, xvoff, (xt * K::KS + 8 * j) * P * 4, PP_S1_XAUX)); \
            else                                                                                  \
                x_[j] = __builtin_bit_cast(xreg_t, __builtin_amdgcn_raw_buffer_load_b64(Xd, xvoff, (xt * K::KS + 8 * j) * P * 2, PP_S1_XAUX)); \
        }                                                                                         \
        if (++xt == KT3) {                                                                        \
            xt = 0;                                                                               \
            Xd = XdN;                                                                             \
        }                                                                                         \
    } while (0)
#define DMA_Q(slot_)                                                                              \
    do {                                                                                          \
        const int qtile = qt < 0 ? KT3 : qt; /* the dummy: a tile past the end */                 \
        _Pragma("unroll") for (int j = 0; j < 16 / NW; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds( \
            Qd, (lds_ptr_t)(Qring + (slot_) * QS + (j * NW + w8) * 1024), 16, qvoff,              \
            (qtile * 1024 + j * NW * 64) * 16, 0, 0);                                             \
        if (++qt == KT3) {                                                                        \
            qt = 0;                                                                               \
            Qd = QdN;                                                                             \
        }                                                                                         \
    } while (0)
#define STORE_X(Xs_, x_)                                                              \
    do {                                                                              \
        if (MODE == PP_MATCH_FAST) {                                                  \
            _Pragma("unroll") for (int j = 0; j < K::XL; ++j) {                       \
                f4 v;                                                                 \
                h4 hv;                                                                \
                if constexpr (ES == 4) {                                              \
                    v = __builtin_bit_cast(f4, x_[j]);                                \
                    hv[0] = (_Float16)v.x;                                            \
                    hv[1] = (_Float16)v.y;                                            \
                    hv[2] = (_Float16)v.z;                                            \
                    hv[3] = (_Float16)v.w;                                            \
                } else {   /* the bank already is fp16: the MFMA operand as stored */ \
                    hv = __builtin_bit_cast(h4, x_[j]);                               \
                    v = f4{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};   \
                }                                                                     \
                ssq0 = fmaf(v.x, v.x, ssq0);                                          \
                ssq1 = fmaf(v.y, v.y, ssq1);                                          \
                ssq2 = fmaf(v.z, v.z, ssq2);                                          \
                ssq3 = fmaf(v.w, v.w, ssq3);                                          \
                *(h4*)((Xs_) + (8 * j + 2 * w + lh) * XROW_F16 + 8 * l31) = hv;       \
            }                                                                         \
        } else {                                                                      \
            _Pragma("unroll") for (int j = 0; j < K::XL; ++j) {                       \
                f4 v;                                                                 \
                if constexpr (ES == 4) v = __builtin_bit_cast(f4, x_[j]);             \
                else {                                                                \
                    const h4 hv = __builtin_bit_cast(h4, x_[j]);                      \
                    v = f4{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};   \
                }                                                                     \
                *(f4*)((Xs_) + ((8 * j + 2 * w + lh) * 128 + 4 * l31) * 4) = v;       \
            }                                                                         \
        }                                                                             \
    } while (0)
#define LDS_BARRIER()                                          \
    do {                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
        __builtin_amdgcn_s_barrier();                          \
    } while (0)
// One K-step (ct = K-step of the current item, < 0 while the stream warms up).  VMEM issue order per step is
// Q(ct+2) X(ct+3), so when step ct starts X(ct+1), Q(ct+1), X(ct+2) may still be in flight: a counted vmcnt
// retires this wave's share of Q(ct) (and X(ct)); the barrier then publishes Q(ct) of all waves together
// with the X tile.  Each tile register has ONE defining load (no copies of in-flight registers at the back
// edge), which is why the warm-up runs through the same code with the compute parts switched off.
#define STEP(ct_, x_, qcur_, qnxt_)                                                   \
    do {                                                                              \
        char* Xs_ = smem + (((ct_)&1) ? L::XS1_OFF : L::XS0_OFF) + hw * XS_MAX;       \
        __builtin_amdgcn_sched_barrier(0);                                            \
        if ((ct_) >= 0) {                                                             \
            /* 2 X tiles + 1 query tile of this wave may stay in flight */            \
            if (MODE == PP_MATCH_FAST && NW == 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); \
            else if (MODE == PP_MATCH_FAST) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); \
            else if (NW == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");        \
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                     \
            STORE_X(Xs_, x_);                                                         \
            __builtin_amdgcn_sched_barrier(0); /* x_ consumed before it is reloaded */ \
            LDS_BARRIER();                                                            \
        }                                                                             \
        DMA_Q(qnxt_);                                                                 \
        LOAD_XB(x_);                                                                  \
        if ((ct_) >= 0) mfma_step(Xs_, Qring + (qcur_) * QS);                         \
    } while (0)

    auto mfma_step = [&](const char* Xs, const char* Qs) __attribute__((always_inline)) {
        if (MODE == PP_MATCH_FAST) {
            // The transposing reads are inline asm: as builtins hipcc orders them after every pending LDS-DMA
            // (s_waitcnt vmcnt(0) in the K loop).  Their results are retired by the explicit lgkmcnt(0).
            h8 a[2][4];
            fp16x4_t lo[2][2], hi[2][2];
            const int col = wc * 64 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
            const int row = 8 * lh + ((lane & 15) >> 2);
            const unsigned xaddr =
                (unsigned)(size_t)(__attribute__((address_space(3))) const char*)(Xs + row * XROW_F16 + col * 2);
#define S1_READ(kh_)                                                                          \
    do {                                                                                      \
        _Pragma("unroll") for (int tb = 0; tb < 4; ++tb) a[kh_][tb] =                         \
            *(const h8*)(Qs + ((kh_) * 8 + wr * 4 + tb) * 1024 + lane * 16);                  \
        asm volatile("ds_read_b64_tr_b16 %0, %4 offset:%5\n\tds_read_b64_tr_b16 %1, %4 offset:%6\n\t" \
                     "ds_read_b64_tr_b16 %2, %4 offset:%7\n\tds_read_b64_tr_b16 %3, %4 offset:%8"     \
                     : "=&v"(lo[kh_][0]), "=&v"(hi[kh_][0]), "=&v"(lo[kh_][1]), "=&v"(hi[kh_][1])      \
                     : "v"(xaddr), "n"((kh_) * 16 * XROW_F16), "n"((kh_) * 16 * XROW_F16 + 4 * XROW_F16), \
                       "n"((kh_) * 16 * XROW_F16 + 64), "n"((kh_) * 16 * XROW_F16 + 4 * XROW_F16 + 64)  \
                     : "memory");                                                             \
    } while (0)
#define S1_MMA(kh_)                                                                           \
    do {                                                                                      \
        h8 bf[2];                                                                             \
        _Pragma("unroll") for (int sb = 0; sb < 2; ++sb)                                      \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                   \
                bf[sb][e] = (_Float16)lo[kh_][sb][e];                                         \
                bf[sb][4 + e] = (_Float16)hi[kh_][sb][e];                                     \
            }                                                                                 \
        _Pragma("unroll") for (int tb = 0; tb < 4; ++tb)                                      \
            _Pragma("unroll") for (int sb = 0; sb < 2; ++sb) acc[tb][sb] =                    \
                __builtin_amdgcn_mfma_f32_32x32x16_f16(a[kh_][tb], bf[sb], acc[tb][sb], 0, 0, 0); \
    } while (0)
            S1_READ(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            S1_READ(1);
            S1_MMA(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            S1_MMA(1);
#undef S1_READ
#undef S1_MMA
        } else {
            const float* Xf = (const float*)Xs;
            const float* Qf = (const float*)Qs;
#pragma unroll
            for (int p = 0; p < K::KS / 2; ++p) {
                float a[4], bv[2];
#pragma unroll
                for (int tb = 0; tb < 4; ++tb) a[tb] = Qf[(2 * p + lh) * 256 + wr * 128 + tb * 32 + l31];
#pragma unroll
                for (int sb = 0; sb < 2; ++sb)
                    bv[sb] = Xf[(2 * p + lh) * 128 + wc * 64 + sb * 32 + l31];
                ssq0 = fmaf(bv[0], bv[0], ssq0);
                ssq1 = fmaf(bv[1], bv[1], ssq1);
#pragma unroll
                for (int tb = 0; tb < 4; ++tb)
#pragma unroll
                    for (int sb = 0; sb < 2; ++sb)
                        acc[tb][sb] =
                            __builtin_amdgcn_mfma_f32_32x32x2f32(a[tb], bv[sb], acc[tb][sb], 0, 0, 0);
            }
        }
    };

    // Results leave at the START of a later epilogue, not at the end of their own: vmcnt retires in order, a store
    // is acknowledged only after microseconds under the read stream, and the K loop's counted waits sit behind it.
    // Issued at the start they retire during the epilogue's arithmetic.  With 8 waves the results of 4 items are
    // parked in LDS (each thread re-reads only what it wrote) and leave in one burst — the stall is per episode,
    // not per byte: 791 -> 737 us — with 4 waves (no LDS to spare) one item waits in registers.
    float4 p_rec = make_float4(0.f, 0.f, 0.f, 0.f);
    float p_cm = 0.f, p_s0 = 0.f, p_st0 = 0.f;
    int p_idx = -1;   // NW == 4: (crop * N + template) * 2 + half of the pending results, -1 = none
    int p_cnt = 0;    // NW == 8: parked items
    int p_bn0 = 0, p_bn1 = 0, p_bn2 = 0, p_bn3 = 0;  // their crop * N + template
    char* park = smem + L::PARK_OFF;
#define PARK_REC(k_) ((float4*)(park + (k_) * L::PARK_ITEM) + hw * 256 + th)
#define PARK_CM(k_) ((float*)(park + (k_) * L::PARK_ITEM + L::HV * 4096) + hw * 128 + th)
#define PARK_S0(k_) ((float*)(park + (k_) * L::PARK_ITEM + L::HV * 4096 + L::HV * 512) + hw * 128 + th)
#define PARK_ST0(k_) ((float*)(park + (k_) * L::PARK_ITEM + L::HV * 4096 + L::HV * 1024) + th)
#define STORE_ONE(bn_, half_, rec_, cm_, s0_, st0_)                                           \
    do {                                                                                       \
        /* non-temporal: measured 1.6 % faster than write-back stores under the read stream */   \
        const float4 r4_ = (rec_);                                                             \
        __builtin_nontemporal_store(f4{r4_.x, r4_.y, r4_.z, r4_.w},                            \
                                    (f4*)&rowrec[((size_t)(bn_) * 2 + (half_)) * P + th]);     \
        if (th < 128) {                                                                        \
            __builtin_nontemporal_store((cm_), &colmax[(size_t)(bn_) * P + (half_) * 128 + th]); \
            __builtin_nontemporal_store((s0_), &sim0s[(size_t)(bn_) * P + (half_) * 128 + th]); \
        }                                                                                      \
        if ((half_) == 0) __builtin_nontemporal_store((st0_), &simt0[(size_t)(bn_) * P + th]); \
    } while (0)
#define STORE_PENDING(all_)                                                                    \
    do {                                                                                       \
        if (NW == 8) {                                                                         \
            if (p_cnt == L::PARK || ((all_) && p_cnt > 0)) {                                   \
                _Pragma("unroll") for (int k = 0; k < L::PARK; ++k) if (k < p_cnt) {           \
                    const int pbn = k == 0 ? p_bn0 : k == 1 ? p_bn1 : k == 2 ? p_bn2 : p_bn3;  \
                    STORE_ONE(pbn, hw, *PARK_REC(k), *PARK_CM(k), *PARK_S0(k), *PARK_ST0(k));  \
                }                                                                              \
                p_cnt = 0;                                                                     \
            }                                                                                  \
        } else if (p_idx >= 0) {                                                               \
            STORE_ONE(p_idx >> 1, p_idx & 1, p_rec, p_cm, p_s0, p_st0);                        \
        }                                                                                      \
    } while (0)

    int ct = -3;
    asm volatile("" : "+s"(ct));  // opaque: hipcc must not peel the load-only iteration
#pragma clang loop unroll(disable)
    for (;;) {
        if (ct == KT3) {
            // ------------------------------------------------------------ epilogue of item (b, n, half)
            const size_t bn = (size_t)b * N + n;
            float* T = (float*)(smem + L::EPI_T);
            float* red = (float*)(smem + L::EPI_RED) + hw * 8 * 128;
            float* colp = (float*)(smem + L::EPI_COLP) + hw * 2 * 128;
            float* sim0 = (float*)(smem + L::EPI_SIM0) + hw * 128;
            float* st0 = (float*)(smem + L::EPI_ST0);
            STORE_PENDING(false);

            // 1. column norms -> 1/max(||x_s||, 1e-12)   (F.normalize, matching.py:43): partial sums through
            //    LDS (the barrier also tells that every wave is past its last tile reads), then every thread
            //    adds up the partials of its own two columns in a fixed order
            if (MODE == PP_MATCH_FAST) {
                *(f4*)(red + (2 * w + lh) * 128 + 4 * l31) = f4{ssq0, ssq1, ssq2, ssq3};
            } else {
                const float s0 = ssq0 + __shfl_xor(ssq0, 32);
                const float s1 = ssq1 + __shfl_xor(ssq1, 32);
                if (wr == 0 && lh == 0) {
                    red[wc * 64 + l31] = s0;
                    red[wc * 64 + 32 + l31] = s1;
                }
            }
            LDS_BARRIER();

            // 2. scale columns, 3. column maxima over this wave's 128 rows, row t = 0 excluded
            //    (i2[s] != 0  <=>  max_{t>0} sim[t,s] > sim[0,s]); row 0 is tb 0, register 0, lanes 0..31
            float cm[2];
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) {
                const int col = wc * 64 + sb * 32 + l31;
                float ss;
                if (MODE == PP_MATCH_FAST) {
                    ss = 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) ss += red[i * 128 + col];
                } else {
                    ss = red[col];
                }
                const float r = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
                for (int tb = 0; tb < 4; ++tb)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[tb][sb][e] *= r;
                float m = acc[0][sb][1];  // elements (tb, e) != (0, 0), two per v_max3
#pragma unroll
                for (int k = 2; k < 64; k += 2) m = vmax3(m, acc[k >> 4][sb][k & 15], acc[(k + 1) >> 4][sb][(k + 1) & 15]);
                if (!(wr == 0 && lh == 0)) m = vmax(m, acc[0][sb][0]);
                cm[sb] = vmax(m, __shfl_xor(m, 32));
            }
            if (lh == 0) {
                colp[wr * 128 + wc * 64 + l31] = cm[0];
                colp[wr * 128 + wc * 64 + 32 + l31] = cm[1];
                if (wr == 0) {
                    sim0[wc * 64 + l31] = acc[0][0][0];
                    sim0[wc * 64 + 32 + l31] = acc[0][1][0];
                }
            }

            // 4. row maxima over this workgroup's 128 columns, column s = 0 excluded
            //    (i1[t] != 0  <=>  max_{s>0} sim[t,s] > sim[t,0]).
            //    In registers: max over the two column blocks, then a two-step reduce-scatter over
            //    the lane quad (DPP): afterwards lane q of a quad owns rows 4i+q and each value is the
            //    max of 8 columns {c..c+3, c+32..c+35}, c = 4*(lane column / 4).  The 16 "entries" of
            //    a row (8 per wave column) are transposed through LDS and scanned by one thread.
            //    The waves holding template patch 0 (half 0, wave column 0: lanes 0 and 32, sb 0) first save
            //    that column (sim[t,0]) and then take it out of the maxima.
            if (half == 0 && wc == 0) {
                if (l31 == 0) {
#pragma unroll
                    for (int tb = 0; tb < 4; ++tb)
#pragma unroll
                        for (int e4 = 0; e4 < 4; ++e4)  // rows tb*32 + 8*e4 + 4*lh + (0..3)
                            *(f4*)(st0 + wr * 128 + tb * 32 + 8 * e4 + 4 * lh) =
                                f4{acc[tb][0][4 * e4], acc[tb][0][4 * e4 + 1], acc[tb][0][4 * e4 + 2],
                                   acc[tb][0][4 * e4 + 3]};
                }
#pragma unroll
                for (int tb = 0; tb < 4; ++tb)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[tb][0][e] = l31 == 0 ? -INFINITY : acc[tb][0][e];
            }
            {
                const bool odd = lane & 1, bit1 = (lane >> 1) & 1;
                const int q = lane & 3;
#pragma unroll
                for (int i = 0; i < 16; ++i) {  // rows r = 4i..4i+3  (r = tb*16 + e)
                    float v[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int r = 4 * i + k, tb = r >> 4, e = r & 15;
                        v[k] = vmax(acc[tb][0][e], acc[tb][1][e]);
                    }
                    const float u0 = vmax_xor1(odd ? v[1] : v[0], odd ? v[0] : v[1]);  // row 4i + odd
                    const float u1 = vmax_xor1(odd ? v[3] : v[2], odd ? v[2] : v[3]);  // row 4i+2+odd
                    const float wv = vmax_xor2(bit1 ? u1 : u0, bit1 ? u0 : u1);        // row 4i + q
                    // row r = 4i+q: tb = i>>2, e = 4*(i&3)+q -> t_local = tb*32 + (e&3) + 8*(e>>2) + 4*lh
                    const int tl = (i >> 2) * 32 + q + 8 * (i & 3) + 4 * lh;
                    T[(w8 * 128 + tl) * TROW + (l31 >> 2)] = wv;
                }
            }
            LDS_BARRIER();
            {
                // Row record over this half's 16 entries:
                //   .x best entry   .y second best entry   .z first column of the best entry (int bits)
                const int wrr = th >> 7, tl = th & 127;
                const float* r0 = T + ((4 * hw + 2 * wrr) * 128 + tl) * TROW;
                const float* r1 = T + ((4 * hw + 2 * wrr + 1) * 128 + tl) * TROW;
                float a1 = -INFINITY, a2 = -INFINITY;
                int p1 = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f4 u = *(const f4*)((i < 2 ? r0 : r1) + 4 * (i & 1));
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float val = u[e];
                        a2 = vmax(a2, vmin(a1, val));
                        p1 = val > a1 ? 4 * i + e : p1;
                        a1 = vmax(a1, val);
                    }
                }
                const int scol = half * 128 + (p1 >> 3) * 64 + 4 * (p1 & 7);
                p_rec = make_float4(a1, a2, __int_as_float(scol), 0.f);
            }
            if (th < 128) {
                p_cm = vmax(colp[th], colp[128 + th]);
                p_s0 = sim0[th];
            }
            if (half == 0) p_st0 = st0[th];
            if (NW == 8) {  // park (thread-private slots: no barrier between these writes and the burst's reads)
#pragma unroll
                for (int k = 0; k < L::PARK; ++k)
                    if (k == p_cnt) {
                        *PARK_REC(k) = p_rec;
                        if (th < 128) {
                            *PARK_CM(k) = p_cm;
                            *PARK_S0(k) = p_s0;
                        }
                        if (half == 0) *PARK_ST0(k) = p_st0;
                    }
                p_bn0 = p_cnt == 0 ? (int)bn : p_bn0;
                p_bn1 = p_cnt == 1 ? (int)bn : p_bn1;
                p_bn2 = p_cnt == 2 ? (int)bn : p_bn2;
                p_bn3 = p_cnt == 3 ? (int)bn : p_bn3;
                ++p_cnt;
            } else {
                p_idx = (int)(bn * 2 + half);
            }

            // ---- next item (its first tiles are already in flight)
            v_cur = v_nxt;
            if (v_cur >= total) break;
            s1_item<NW>(v_cur, N, b, n, half);
            if (NW == 8) half = hw;
            v_nxt = v_cur + G;
            {
                int b2, n2, h2;
                const bool ok = v_nxt < total;
                s1_item<NW>(ok ? v_nxt : v_cur, N, b2, n2, h2);
                XdN = X_DESC(b2, n2, h2, ok);
                QdN = Q_DESC(b2, ok);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            ssq0 = ssq1 = ssq2 = ssq3 = 0.f;
            ct = 0;
            // Stores and loads retire in no fixed order relative to each other, so the counted waits of the
            // K loop need the epilogue's stores out of the way (the builtin keeps hipcc's own count in step).
            // No barrier here: the next step writes X buffer 0 only, and its own barrier comes before anything
            // touches ring slot 2 / X buffer 1 (the transpose tile) or the small arrays again.
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        }
        STEP(ct, x0, 0, 2);
        STEP(ct + 1, x1, 1, 0);
        STEP(ct + 2, x2, 2, 1);
        ct += 3;
    }
    STORE_PENDING(true);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no DMA may still target this workgroup's LDS at exit
#undef STORE_PENDING
#undef STORE_ONE
#undef PARK_REC
#undef PARK_CM
#undef PARK_S0
#undef PARK_ST0
#undef STEP
#undef LDS_BARRIER
#undef STORE_X
#undef DMA_Q
#undef LOAD_XB
#undef X_DESC
#undef Q_DESC
}

// torch.topk(sim_avg, k, dim=1) (matching.py:68): descending score, NaN sorts above every
// number (as torch does), ties -> lower template id.
__device__ __forceinline__ bool topk_better(float v, int j, float bv, int bi) {
    if (bi < 0) return true;
    const bool vn = v != v, bn_ = bv != bv;
    if (vn != bn_) return vn;
    if (!vn && v != bv) return v > bv;
    return j < bi;
}

// one crop's row by one 256-thread workgroup; sc: N floats followed by N taken flags (LDS).  COHERENT: the row was written by
// other workgroups of THIS launch (the fused resolve + top-k): read it with agent-scope atomic loads
template <bool COHERENT>
__device__ __forceinline__ void topk_block(const float* __restrict__ row, int N, int k, float* __restrict__ out_score,
                                           int64_t* __restrict__ out_index, float* sc, float* wv, int* wi) {
    unsigned char* taken = (unsigned char*)(sc + N);
    const int tid = threadIdx.x;
    for (int j = tid; j < N; j += 256) {
        sc[j] = COHERENT ? __hip_atomic_load(row + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : row[j];
        taken[j] = 0;
    }
    __syncthreads();
    for (int it = 0; it < k; ++it) {
        float bv = 0.f;
        int bi = -1;
        for (int j = tid; j < N; j += 256)
            if (!taken[j] && topk_better(sc[j], j, bv, bi)) {
                bv = sc[j];
                bi = j;
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o);
            const int oi = __shfl_xor(bi, o);
            if (oi >= 0 && topk_better(ov, oi, bv, bi)) {
                bv = ov;
                bi = oi;
            }
        }
        if ((tid & 63) == 0) {
            wv[tid >> 6] = bv;
            wi[tid >> 6] = bi;
        }
        __syncthreads();
        if (tid == 0) {
            for (int q = 1; q < 4; ++q)
                if (wi[q] >= 0 && topk_better(wv[q], wi[q], bv, bi)) {
                    bv = wv[q];
                    bi = wi[q];
                }
            out_score[it] = bv;
            out_index[it] = bi;
            taken[bi] = 1;
        }
        __syncthreads();
    }
}

// N <= 1024: one WAVE per crop — every lane keeps its N / 64 scores in registers, an iteration is one local scan + one wave
// reduction, no barrier (the 256-thread form above spends 5.8 us on 8 rows of 42: five rounds of barriers on one workgroup)
__global__ __launch_bounds__(256) void topk_rows_small(const float* __restrict__ scores, int B, int N, int k,
                                                       float* __restrict__ out_score, int64_t* __restrict__ out_index) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    float v[16];
    unsigned taken = 0u;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int j = lane + 64 * q;
        v[q] = j < N ? scores[(size_t)b * N + j] : 0.f;
        if (j >= N) taken |= 1u << q;
    }
    for (int it = 0; it < k; ++it) {
        float bv = 0.f;
        int bi = -1;
#pragma unroll
        for (int q = 0; q < 16; ++q)
            if (!((taken >> q) & 1u) && topk_better(v[q], lane + 64 * q, bv, bi)) {
                bv = v[q];
                bi = lane + 64 * q;
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o);
            const int oi = __shfl_xor(bi, o);
            if (oi >= 0 && topk_better(ov, oi, bv, bi)) {
                bv = ov;
                bi = oi;
            }
        }
        if (lane == 0) {
            out_score[(size_t)b * k + it] = bv;
            out_index[(size_t)b * k + it] = bi;
        }
        if (bi >= 0 && (bi & 63) == lane) taken |= 1u << (bi >> 6);
    }
}

__global__ __launch_bounds__(256) void topk_rows(const float* __restrict__ scores, int N, int k,
                                                 float* __restrict__ out_score,
                                                 int64_t* __restrict__ out_index) {
    extern __shared__ float sc[];
    __shared__ float wv[4];
    __shared__ int wi[4];
    const int b = blockIdx.x;
    topk_block<false>(scores + (size_t)b * N, N, k, out_score + (size_t)b * k, out_index + (size_t)b * k, sc, wv, wi);
}


// ---------------------------------------------------------------------------
// s1_resolve: per (b,n) — FAST mode's exact re-evaluation of near-tie decisions, then
// sim_avg (matching.py:53-66).
//
// A decision "arg-max is patch 0" is near a tie when |best_other - sim_0| <= eps.  Rows:
//   kind 0: the best entry is more than 2*eps above every other entry -> the exact max over
//           s > 0 is one of its 8 columns {c..c+3, c+32..c+35}; re-evaluate those and column 0
//   kind 1: several entries are close -> re-evaluate the whole row
// Columns (kind 2): re-evaluate the whole column.
// The arithmetic repeats EXACT mode's: the dot product is the c-ordered fp32 fma chain of
// v_mfma_f32_32x32x2_f32, the column norm is the sum of an even-channel and an odd-channel
// fma chain, sim = dot * (1 / max(sqrt(ss), 1e-12)) — so the decisions equal EXACT mode's.
// ---------------------------------------------------------------------------
template <typename XT>
__global__ __launch_bounds__(256) void s1_resolve(const XT* __restrict__ bank,
                                                  const float* __restrict__ qf,
                                                  const float* __restrict__ m16, int N, int C,
                                                  int fast, float eps,
                                                  const float4* __restrict__ rowrec,
                                                  const float* __restrict__ simt0,
                                                  const float* __restrict__ colmax,
                                                  const float* __restrict__ sim0s,
                                                  float* __restrict__ sim_avg,
                                                  int32_t* __restrict__ stats, int topk_k, int* __restrict__ counters,
                                                  float* __restrict__ out_score, int64_t* __restrict__ out_index) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* L = (float*)smem;  // [10][C]: q[:,t], x[:,0], x[:,c..c+3], x[:,c+32..c+35]
    __shared__ int nent;
    __shared__ unsigned ent[512];
    __shared__ float res[12];
    __shared__ float wred[8];
    const size_t bn = blockIdx.x;
    const int b = (int)(bn / N), i = threadIdx.x;
    const int lane = i & 63;
    const float m = m16[b * P + i];
    const float4 r0 = rowrec[(bn * 2) * P + i], r1 = rowrec[(bn * 2 + 1) * P + i];
    float rme = fmaxf(r0.x, r1.x);  // best over s > 0 (row i)
    float st = simt0[bn * P + i];   // sim[i, 0]
    float cme = colmax[bn * P + i]; // best over t > 0 (column i)
    float s0 = sim0s[bn * P + i];   // sim[0, i]

    if (fast) {
        if (i == 0) nent = 0;
        __syncthreads();
        if (m != 0.f) {  // mask_all[i] = 0 whatever the decisions are when m[i] = 0
            if (fabsf(rme - st) <= eps) {
                const bool w0 = r0.x >= r1.x;
                const float second = fmaxf(w0 ? r1.x : r0.x, w0 ? r0.y : r1.y);
                const unsigned scol = (unsigned)__float_as_int(w0 ? r0.z : r1.z);
                const unsigned kind = second < rme - 2.f * eps ? 0u : 1u;
                ent[atomicAdd(&nent, 1)] = (kind << 16) | ((unsigned)i << 8) | scol;
            }
            if (fabsf(cme - s0) <= eps) ent[atomicAdd(&nent, 1)] = (2u << 16) | ((unsigned)i << 8);
        }
        __syncthreads();
        const int ne = nent;
        const XT* X = bank + bn * (size_t)C * P;
        const float* Q = qf + (size_t)b * C * P;
        for (int e = 0; e < ne; ++e) {
            const unsigned f = ent[e];
            const int kind = (int)(f >> 16), idx = (f >> 8) & 255, sc = f & 255;
            if (kind == 0) {
                // stage the 10 channel vectors with every load in flight at once
                for (int c0 = 0; c0 < C; c0 += 1024) {
                    float v[4][10];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int c = min(c0 + 256 * j + i, C - 1);
                        v[j][0] = Q[(size_t)c * P + idx];
                        v[j][1] = X[(size_t)c * P];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            v[j][2 + k] = X[(size_t)c * P + sc + k];
                            v[j][6 + k] = X[(size_t)c * P + sc + 32 + k];
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int c = c0 + 256 * j + i;
                        if (c < C) {
#pragma unroll
                            for (int k = 0; k < 10; ++k) L[k * C + c] = v[j][k];
                        }
                    }
                }
                __syncthreads();
                if (i < 9) {
                    const float* xv = L + (i + 1) * C;
                    float dot = 0.f, se = 0.f, so = 0.f;
#pragma unroll 8
                    for (int c = 0; c < C; c += 2) {
                        const float x0 = xv[c], x1 = xv[c + 1];
                        dot = fmaf(L[c], x0, dot);
                        dot = fmaf(L[c + 1], x1, dot);
                        se = fmaf(x0, x0, se);
                        so = fmaf(x1, x1, so);
                    }
                    res[i] = dot * (1.0f / fmaxf(sqrtf(se + so), 1e-12f));
                }
                __syncthreads();
                if (i == idx) {
                    float mx = -INFINITY;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int col = sc + (k & 3) + 32 * (k >> 2);
                        if (col != 0) mx = fmaxf(mx, res[1 + k]);  // column 0 is the other side
                    }
                    rme = mx;
                    st = res[0];
                }
                __syncthreads();
            } else {
                // thread o = i: row entry: a = q[:,idx] (uniform), x = X[:,o];
                //               column entry: a = q[:,o], x = X[:,idx]
                const float* ap = Q + (kind == 1 ? idx : i);
                const XT* xp = X + (kind == 1 ? i : idx);
                float dot = 0.f, se = 0.f, so = 0.f;
                for (int c0 = 0; c0 < C; c0 += 32) {
                    float a[32], x[32];
#pragma unroll
                    for (int j = 0; j < 32; ++j) {
                        a[j] = ap[(size_t)(c0 + j) * P];
                        x[j] = xp[(size_t)(c0 + j) * P];
                    }
#pragma unroll
                    for (int j = 0; j < 32; j += 2) {
                        dot = fmaf(a[j], x[j], dot);
                        dot = fmaf(a[j + 1], x[j + 1], dot);
                        se = fmaf(x[j], x[j], se);
                        so = fmaf(x[j + 1], x[j + 1], so);
                    }
                }
                const float sim = dot * (1.0f / fmaxf(sqrtf(se + so), 1e-12f));
                float mx = i == 0 ? -INFINITY : sim;  // index 0 is the other side
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
                if (lane == 0) wred[i >> 6] = mx;
                if (i == 0) res[0] = sim;
                __syncthreads();
                if (i == idx) {
                    const float mxa = fmaxf(fmaxf(wred[0], wred[1]), fmaxf(wred[2], wred[3]));
                    if (kind == 1) {
                        rme = mxa;
                        st = res[0];
                    } else {
                        cme = mxa;
                        s0 = res[0];
                    }
                }
                __syncthreads();
            }
            if (stats && i == 0) atomicAdd(&stats[kind], 1);
        }
    }

    const float rnz = rme > st ? 1.f : 0.f;   // idx_tar2src != 0 (first max wins ties)
    const float cnz = cme > s0 ? 1.f : 0.f;   // idx_src2tar != 0
    const float rm = fmaxf(rme, st);          // score_tar2src
    const float mall = m * cnz * rnz;
    float s = rm * mall, ms = mall;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o);
        ms += __shfl_xor(ms, o);
    }
    if (lane == 0) {
        wred[i >> 6] = s;
        wred[4 + (i >> 6)] = ms;
    }
    __syncthreads();
    __shared__ int last_of_crop;
    if (i == 0) {
        const float tot = (wred[0] + wred[1]) + (wred[2] + wred[3]);
        const float mt = (wred[4] + wred[5]) + (wred[6] + wred[7]);
        const float v = mt > 0.f ? tot / 256.0f : 0.f;
        if (topk_k > 0) {
            // fused top-k (pp_stage1_match): the workgroup that arrives LAST for its crop ranks the crop's N scores.  The score is
            // published with an agent-scope store, the arrival is an agent-scope acq_rel add (MI355X_MICROARCH.md, "Valid forms":
            // agent atomics on both sides), the last arriver reads the row with agent-scope loads — the per-XCD L2s are not
            // coherent with each other for plain accesses.  The counter returns to zero for the next call.
            __hip_atomic_store(sim_avg + bn, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int ticket = __hip_atomic_fetch_add(counters + b, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            last_of_crop = ticket == N - 1 ? 1 : 0;
            if (ticket == N - 1) __hip_atomic_store(counters + b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            sim_avg[bn] = v;
        }
    }
    if (topk_k > 0) {
        __syncthreads();
        if (last_of_crop) {
            __shared__ float twv[4];
            __shared__ int twi[4];
            topk_block<true>(sim_avg + (size_t)b * N, N, topk_k, out_score + (size_t)b * topk_k, out_index + (size_t)b * topk_k, L, twv, twi);
        }
    }
}

}  // namespace

// ---------------------------------------------------------------------------
extern "C" {

int pp_stage1_workspace_bytes(int B, int N, int C, size_t* bytes) {
    if (!bytes || B <= 0 || N <= 0 || C <= 0) return PP_EINVAL;
    S1Ws w = carve(nullptr, B, N, C);
    *bytes = w.total;
    return PP_OK;
}

// topk_k > 0: the fused form of pp_stage1_match — the last resolve workgroup of a crop ranks the crop's scores (out_score / out_index)
static int stage1_run(const void* bank_, int bank_dtype, const float* query, const float* mask, int mask_h,
                      int mask_w, int B, int N, int C, int mode, float eps, void* workspace,
                      size_t workspace_bytes, float* sim_avg, int32_t* stats, int topk_k, float* out_score, int64_t* out_index,
                      void* stream_) {
    const float* bank = (const float*)bank_;
    const _Float16* bank16 = (const _Float16*)bank_;
    const bool f16 = bank_dtype == PP_BANK_F16;
    if (bank_dtype != PP_BANK_F32 && bank_dtype != PP_BANK_F16) return PP_EINVAL;
    if (!bank || !query || !mask || !sim_avg || !workspace) return PP_EINVAL;
    if (B <= 0 || N <= 0 || C <= 0 || mask_h <= 0 || mask_w <= 0) return PP_EINVAL;
    if (mode != PP_MATCH_EXACT && mode != PP_MATCH_FAST) return PP_EINVAL;
    if (C % 64 != 0 || C > 2048) return PP_EINVAL;
    if ((size_t)B * N >= (1u << 23)) return PP_EINVAL;
    if (((uintptr_t)workspace & 255) != 0) return PP_EWORKSPACE;
    if (((uintptr_t)bank & 15) != 0) return PP_EINVAL;   // (fp16: 8-byte loads of 16-byte aligned rows)
    S1Ws w = carve(workspace, B, N, C);
    if (workspace_bytes < w.total) return PP_EWORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    // default band: 8 sigma of the fp16 rounding error of a score difference,
    // sigma ~ sqrt(2) * 2.8e-4 / sqrt(C) for unit vectors with spread-out energy
    if (eps <= 0.f) eps = 3.2e-3f / sqrtf((float)C);

    const int cus = pp_cu_count();
    // small problems (B C^2 32 bytes of L2 re-reads <= 64 MB: BASELINE configs[1]): the query pre-pack as ONE launch
    // (measured SLOWER than the two launches at configs[1] — 71.0 vs 68.7 us per call, A/B on one box, profiles/r04/stage1_small.txt:
    // the norms recomputed per workgroup cost more than the launch they save — so off unless PP_S1_QPREP=1)
    static const bool qprep_on = [] { const char* e = getenv("PP_S1_QPREP"); return e && e[0] == '1'; }();
    if (qprep_on && (long long)B * C * C * 32 <= (64LL << 20)) {
        hipLaunchKernelGGL(s1_qprep, dim3(B, C / 32), dim3(256), 0, stream, query, mask, mask_h, mask_w, C, w.denom, w.m16, w.qh,
                           w.qf, topk_k > 0 ? w.counters : nullptr, B);
    } else {
        hipLaunchKernelGGL(s1_qnorm, dim3(B, 4), dim3(1024), 0, stream, query, mask, mask_h, mask_w, C,
                           w.denom, w.m16);
        hipLaunchKernelGGL(s1_qpack, dim3(B, C / 32), dim3(256), 0, stream, query, w.denom, w.m16, C,
                           w.qh, w.qf);
        if (topk_k > 0) PP_CHECK_HIP(hipMemsetAsync(w.counters, 0, (size_t)B * sizeof(int), stream));
    }
    // persistent workgroups: 8 waves / whole templates / one per CU, or 4 waves / template halves / two per CU.  The 4-wave
    // shape issues twice the query copies (9 % slower per byte when the chip is full) but halves the work item: with fewer than
    // two items per CU (configs[1]: 336 items on 256 CUs) the tail round is shorter (36.4 vs 39.3 us).  PP_S1_WAVES=4|8 pins one.
    constexpr int S1_F16_DEFAULT_WAVES = 8;      // (fp16-stored bank: see profiles/r06/README.md for the 4-wave measurement)
    const char* nw_env = getenv("PP_S1_WAVES");  // read per call: the tests run both shapes in one process
    const int nw_auto = B * N < 2 * cus ? 4 : 8;
    const int nw = nw_env ? (atoi(nw_env) == 4 ? 4 : 8) : (f16 ? S1_F16_DEFAULT_WAVES : nw_auto);
    static signed char lds_state[PP_MAX_DEVICES];   // > 64 KB of dynamic LDS needs the opt-in, per device
    signed char& lds_ok = lds_state[pp_cur_device()];
    if (lds_ok == 0) {
        auto set = [](const void* f, int bytes) {
            return hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
        };
        lds_ok = set((const void*)s1_main<PP_MATCH_FAST, 4>, Lay<4>::SMEM_BYTES) &&
                 set((const void*)s1_main<PP_MATCH_EXACT, 4>, Lay<4>::SMEM_BYTES) &&
                 set((const void*)s1_main<PP_MATCH_FAST, 8>, Lay<8>::SMEM_BYTES) &&
                 set((const void*)s1_main<PP_MATCH_EXACT, 8>, Lay<8>::SMEM_BYTES) &&
                 set((const void*)s1_main<PP_MATCH_FAST, 8, _Float16>, Lay<8>::SMEM_BYTES) &&
                 set((const void*)s1_main<PP_MATCH_EXACT, 8, _Float16>, Lay<8>::SMEM_BYTES) &&
                 set((const void*)s1_main<PP_MATCH_FAST, 4, _Float16>, Lay<4>::SMEM_BYTES) &&
                 set((const void*)s1_main<PP_MATCH_EXACT, 4, _Float16>, Lay<4>::SMEM_BYTES) &&
                 set((const void*)s1_resolve<_Float16>, 10 * 2048 * (int)sizeof(float)) &&
                 set((const void*)s1_resolve<float>, 10 * 2048 * (int)sizeof(float))   // 10 * C floats, C <= 2048
                     ? 1 : -1;
    }
    if (lds_ok < 0) return PP_ELAUNCH;
    {
        PpProfScope prof(stream);  // roofline kernel of stage 1 (bench.py)
        const int total = nw == 8 ? B * N : B * 2 * N, slots = nw == 8 ? cus : 2 * cus;
        const int grid = total < slots ? total : slots;
#define S1_LAUNCH(MODE_, NW_)                                                                                  \
    hipLaunchKernelGGL((s1_main<MODE_, NW_>), dim3(grid), dim3(64 * NW_), Lay<NW_>::SMEM_BYTES, stream, bank, w.qh, \
                       w.qf, N, C, total, w.rowrec, w.simt0, w.colmax, w.sim0s)
#define S1_LAUNCH16(MODE_, NW_)                                                                                \
    hipLaunchKernelGGL((s1_main<MODE_, NW_, _Float16>), dim3(grid), dim3(64 * NW_), Lay<NW_>::SMEM_BYTES, stream, bank16, w.qh, \
                       w.qf, N, C, total, w.rowrec, w.simt0, w.colmax, w.sim0s)
        if (f16) {
            if (mode == PP_MATCH_FAST) {
                if (nw == 8) S1_LAUNCH16(PP_MATCH_FAST, 8);
                else S1_LAUNCH16(PP_MATCH_FAST, 4);
            } else {
                if (nw == 8) S1_LAUNCH16(PP_MATCH_EXACT, 8);
                else S1_LAUNCH16(PP_MATCH_EXACT, 4);
            }
        } else if (mode == PP_MATCH_FAST) {
            if (nw == 8) S1_LAUNCH(PP_MATCH_FAST, 8);
            else S1_LAUNCH(PP_MATCH_FAST, 4);
        } else {
            if (nw == 8) S1_LAUNCH(PP_MATCH_EXACT, 8);
            else S1_LAUNCH(PP_MATCH_EXACT, 4);
        }
#undef S1_LAUNCH
#undef S1_LAUNCH16
    }
    if (stats) PP_CHECK_HIP(hipMemsetAsync(stats, 0, 4 * sizeof(int32_t), stream));
    if (f16)
        hipLaunchKernelGGL(s1_resolve<_Float16>, dim3(B * N), dim3(256), (size_t)10 * C * sizeof(float), stream,
                           bank16, w.qf, w.m16, N, C, mode == PP_MATCH_FAST ? 1 : 0, eps, w.rowrec,
                           w.simt0, w.colmax, w.sim0s, sim_avg, stats, topk_k, w.counters, out_score, out_index);
    else
        hipLaunchKernelGGL(s1_resolve<float>, dim3(B * N), dim3(256), (size_t)10 * C * sizeof(float), stream,
                           bank, w.qf, w.m16, N, C, mode == PP_MATCH_FAST ? 1 : 0, eps, w.rowrec,
                           w.simt0, w.colmax, w.sim0s, sim_avg, stats, topk_k, w.counters, out_score, out_index);
    return pp_last_launch();
}

int pp_stage1_scores_ex(const void* bank, int bank_dtype, const float* query, const float* mask, int mask_h,
                        int mask_w, int B, int N, int C, int mode, float eps, void* workspace,
                        size_t workspace_bytes, float* sim_avg, int32_t* stats, void* stream) {
    return stage1_run(bank, bank_dtype, query, mask, mask_h, mask_w, B, N, C, mode, eps, workspace, workspace_bytes, sim_avg, stats, 0,
                      nullptr, nullptr, stream);
}

int pp_stage1_match_ex(const void* bank, int bank_dtype, const float* query, const float* mask, int mask_h,
                       int mask_w, int B, int N, int C, int k, int mode, float eps, void* workspace,
                       size_t workspace_bytes, float* sim_avg, float* out_score, int64_t* out_index,
                       int32_t* stats, void* stream) {
    if (!out_score || !out_index || k <= 0 || k > N || N > 12288) return PP_EINVAL;
    // The last resolve workgroup of a crop CAN rank its scores itself (s1_resolve's topk_k argument, 5 N <= 40 C bytes of LDS):
    // measured slower than a second launch at BASELINE configs[1] — the agent-scope release / acquire of every resolve workgroup
    // (buffer_wbl2 + buffer_inv, ~3.5 us each, MI355X_MICROARCH.md) and the serial ranking at the end of the kernel cost 12 us
    // against the 7.7 us of a top-k launch and its gap (profiles/r04/stage1_small.txt).  PP_S1_FUSE_TOPK=1 switches it on.
    static const bool fuse_env = [] { const char* e = getenv("PP_S1_FUSE_TOPK"); return e && e[0] == '1'; }();
    const bool fused = fuse_env && (long long)N * 5 <= 40LL * C;
    int rc = stage1_run(bank, bank_dtype, query, mask, mask_h, mask_w, B, N, C, mode, eps, workspace, workspace_bytes, sim_avg, stats,
                        fused ? k : 0, out_score, out_index, stream);
    if (rc != PP_OK || fused) return rc;
    return pp_topk(sim_avg, B, N, k, out_score, out_index, stream);
}

int pp_stage1_scores(const float* bank, const float* query, const float* mask, int mask_h,
                     int mask_w, int B, int N, int C, int mode, float eps, void* workspace,
                     size_t workspace_bytes, float* sim_avg, int32_t* stats, void* stream) {
    return pp_stage1_scores_ex(bank, PP_BANK_F32, query, mask, mask_h, mask_w, B, N, C, mode, eps, workspace,
                               workspace_bytes, sim_avg, stats, stream);
}

int pp_topk(const float* scores, int B, int N, int k, float* out_score, int64_t* out_index,
            void* stream_) {
    if (!scores || !out_score || !out_index) return PP_EINVAL;
    if (B <= 0 || N <= 0 || k <= 0 || k > N || N > 12288) return PP_EINVAL;
    // (one wave per crop: measured 0.8 us SLOWER per call than the 256-thread form at configs[1] — both sit on the ~6 us floor
    // of a dependent tiny launch; off unless PP_S1_TOPK_SMALL=1)
    static const bool small_on = [] { const char* e = getenv("PP_S1_TOPK_SMALL"); return e && e[0] == '1'; }();
    if (small_on && N <= 1024)
        hipLaunchKernelGGL(topk_rows_small, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream_, scores, B, N, k, out_score, out_index);
    else
        hipLaunchKernelGGL(topk_rows, dim3(B), dim3(256), N * 5, (hipStream_t)stream_,
                           scores, N, k, out_score, out_index);
    return pp_last_launch();
}

int pp_stage1_match(const float* bank, const float* query, const float* mask, int mask_h,
                    int mask_w, int B, int N, int C, int k, int mode, float eps, void* workspace,
                    size_t workspace_bytes, float* sim_avg, float* out_score, int64_t* out_index,
                    int32_t* stats, void* stream) {
    return pp_stage1_match_ex(bank, PP_BANK_F32, query, mask, mask_h, mask_w, B, N, C, k, mode, eps, workspace, workspace_bytes, sim_avg,
                              out_score, out_index, stats, stream);
}

}  // extern "C"
