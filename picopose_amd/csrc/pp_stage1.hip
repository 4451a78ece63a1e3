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
// Work decomposition: one 256-thread workgroup (4 waves, 2x2) owns all 256
// query patches x 128 template patches (one half) of one template; each wave
// holds a 128x64 fp32 tile in 128 accumulator registers.  The bank is read
// exactly once (coalesced 16 B/lane, 512 B row segments); the pre-normalised,
// pre-masked query operand is re-read from L2 (workgroups of one crop are
// placed on one XCD).  Two workgroups per CU overlap one's epilogue with the
// other's stream.
//
// Two arithmetic modes share the skeleton:
//   EXACT  v_mfma_f32_32x32x2_f32 — bit-for-bit an fp32 fma chain over c.
//   FAST   v_mfma_f32_32x32x16_f16 on fp16-rounded operands (fp32 accumulate),
//          HBM-bound; every row/column whose "index 0" decision lies within
//          eps of a tie is re-evaluated by pp_s1_fixup in exact fp32, so the
//          discrete outputs agree with EXACT mode.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/picopose_hip.h"
#include "pp_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));

constexpr int P = 256;           // patches per image (16x16), fixed by the reference
constexpr int XROW_F16 = 320;    // bytes per k-row of the fp16 X tile (256 + 64 pad:
                                 // the 4 rows of a ds_read_b64_tr_b16 block land on
                                 // disjoint 16-dword bank groups)
constexpr int TROW = 36;         // floats per row of the epilogue transpose tile

struct S1Ws {
    _Float16* qh;   // (B, C/32, 2, 8, 64, 8) fp16 A-fragment order, normalised*mask
    float* qf;      // (B, C, 256) fp32 normalised*mask
    float* m16;     // (B, 256) sampled mask
    float* rowmax;  // (B*N, 2, 256) per-half row maxima
    float* simt0;   // (B*N, 256) sim[t,0]
    float* colmax;  // (B*N, 256) column maxima
    float* sim0s;   // (B*N, 256) sim[0,s]
    int32_t* counter;  // [0]=#entries [1]=#rows [2]=#cols
    uint32_t* flags;   // fix-up entries: bn<<9 | kind<<8 | index
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
    w.rowmax = (float*)take(BN * 2 * P * 4);
    w.simt0 = (float*)take(BN * P * 4);
    w.colmax = (float*)take(BN * P * 4);
    w.sim0s = (float*)take(BN * P * 4);
    w.counter = (int32_t*)take(64);
    w.flags = (uint32_t*)take(BN * 2 * P * 4);
    w.total = off;
    return w;
}

// ---------------------------------------------------------------------------
// Query pre-pack: F.normalize(tar_feat, dim=1) (matching.py:40), the nearest
// 16x16 resample of the mask (matching.py:38-39) and the row mask multiply
// (matching.py:48) folded into the A operand.  One workgroup per crop.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void s1_prepack(const float* __restrict__ query,
                                                  const float* __restrict__ mask, int mh, int mw,
                                                  int C, _Float16* __restrict__ qh,
                                                  float* __restrict__ qf, float* __restrict__ m16) {
    const int b = blockIdx.x, t = threadIdx.x;
    // nearest: src = min(floor(dst * (float)in/out), in-1)   (ATen nearest_idx)
    const int py = t >> 4, px = t & 15;
    const float sy = (float)mh / 16.0f, sx = (float)mw / 16.0f;
    int iy = (int)floorf((float)py * sy);
    int ix = (int)floorf((float)px * sx);
    iy = iy < mh - 1 ? iy : mh - 1;
    ix = ix < mw - 1 ? ix : mw - 1;
    const float m = mask[((size_t)b * mh + iy) * mw + ix];
    m16[b * P + t] = m;

    const float* q = query + (size_t)b * C * P + t;
    float ss = 0.f;
    for (int c = 0; c < C; ++c) {
        float v = q[(size_t)c * P];
        ss = fmaf(v, v, ss);
    }
    const float denom = fmaxf(sqrtf(ss), 1e-12f);
    const int KT = C >> 5;
    const int tb = t >> 5, tl = t & 31;
    for (int c8 = 0; c8 < C; c8 += 8) {
        h8 pk;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = (q[(size_t)(c8 + j) * P] / denom) * m;
            qf[((size_t)b * C + c8 + j) * P + t] = v;
            pk[j] = (_Float16)v;
        }
        const int ks = c8 >> 5, kh = (c8 >> 4) & 1, hh = (c8 >> 3) & 1;
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
    static constexpr int XL = 2;                  // float4 X loads per thread per step
    static constexpr int XS_BYTES = 16 * 128 * 4; // [16][128] fp32
    static constexpr int QS_BYTES = 16 * 256 * 4; // [16][256] fp32
};
template <>
struct Cfg<PP_MATCH_FAST> {
    static constexpr int KS = 32;
    static constexpr int XL = 4;
    static constexpr int XS_BYTES = 32 * XROW_F16;  // [32][160 halfs] (128 used)
    static constexpr int QS_BYTES = 16 * 1024;      // 16 fragment chunks of 1 KB
};

constexpr int EPI_T_BYTES = 4 * 64 * TROW * 4;  // 36864
constexpr int EPI_RED = EPI_T_BYTES;            // float[8][128]
constexpr int EPI_RS = EPI_RED + 8 * 128 * 4;   // float[128]
constexpr int EPI_COLP = EPI_RS + 128 * 4;      // float[2][128]
constexpr int EPI_SIM0 = EPI_COLP + 2 * 128 * 4;  // float[128]
constexpr int EPI_ST0 = EPI_SIM0 + 128 * 4;     // float[256]
constexpr int EPI_BYTES = EPI_ST0 + 256 * 4;
constexpr int SMEM_BYTES = 53248;
static_assert(EPI_BYTES <= SMEM_BYTES, "epilogue LDS overflow");
static_assert(2 * Cfg<PP_MATCH_FAST>::XS_BYTES + 2 * Cfg<PP_MATCH_FAST>::QS_BYTES <= SMEM_BYTES, "");
static_assert(2 * Cfg<PP_MATCH_EXACT>::XS_BYTES + 2 * Cfg<PP_MATCH_EXACT>::QS_BYTES <= SMEM_BYTES, "");

template <int MODE>
__global__ __launch_bounds__(256, 2) void s1_main(const float* __restrict__ bank,
                                                  const _Float16* __restrict__ qh,
                                                  const float* __restrict__ qf, int B, int N, int C,
                                                  float* __restrict__ rowmax,
                                                  float* __restrict__ simt0,
                                                  float* __restrict__ colmax,
                                                  float* __restrict__ sim0s) {
    using K = Cfg<MODE>;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    // ---- workgroup -> (crop, template, half); crops of one XCD label stay together
    int b, n, half;
    {
        const int bid = blockIdx.x;
        const int per_crop = 2 * N;
        if (B >= 8) {
            const int x = bid & 7, j = bid >> 3;
            b = x + 8 * (j / per_crop);
            const int r = j % per_crop;
            n = r >> 1;
            half = r & 1;
            if (b >= B) return;
        } else {
            b = bid / per_crop;
            const int r = bid % per_crop;
            n = r >> 1;
            half = r & 1;
        }
    }
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 1, wc = w & 1;
    const int l31 = lane & 31, lh = lane >> 5;

    const size_t bn = (size_t)b * N + n;
    const float* Xg = bank + bn * (size_t)C * P + half * 128;
    const int KT = C / K::KS;

    char* Xs0 = smem;
    char* Xs1 = smem + K::XS_BYTES;
    char* Qs0 = smem + 2 * K::XS_BYTES;
    char* Qs1 = Qs0 + K::QS_BYTES;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // FAST: sum of squares of this thread's 4 columns; EXACT: ssq0/ssq1 = column blocks sb 0/1
    float ssq0 = 0.f, ssq1 = 0.f, ssq2 = 0.f, ssq3 = 0.f;

    // per-thread global pointers of the K-step loads
    const float* xptr = Xg + (size_t)(2 * w + lh) * P + 4 * l31;          // + (ks*KS + 8j)*P
    const u4* qptr;
    if (MODE == PP_MATCH_FAST)
        qptr = (const u4*)(qh + (size_t)b * C * P) + tid;              // + ks*1024 + j*256
    else
        qptr = (const u4*)(qf + (size_t)b * C * P) + tid;              // + ks*1024 + j*256

    f4 xa[K::XL], xb[K::XL];
    u4 qa[4], qb[4];

#define LOAD_STEP(ks_, x_, q_)                                                        \
    do {                                                                              \
        _Pragma("unroll") for (int j = 0; j < K::XL; ++j) x_[j] =                     \
            *(const f4*)(xptr + (size_t)((ks_) * K::KS + 8 * j) * P);             \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) q_[j] =                         \
            qptr[(size_t)(ks_) * 1024 + j * 256];                                     \
    } while (0)

#define STORE_STEP(Xs_, Qs_, x_, q_)                                                  \
    do {                                                                              \
        if (MODE == PP_MATCH_FAST) {                                                  \
            _Pragma("unroll") for (int j = 0; j < K::XL; ++j) {                       \
                const f4 v = x_[j];                                               \
                ssq0 = fmaf(v.x, v.x, ssq0);                                          \
                ssq1 = fmaf(v.y, v.y, ssq1);                                          \
                ssq2 = fmaf(v.z, v.z, ssq2);                                          \
                ssq3 = fmaf(v.w, v.w, ssq3);                                          \
                h4 hv;                                                                \
                hv[0] = (_Float16)v.x;                                                \
                hv[1] = (_Float16)v.y;                                                \
                hv[2] = (_Float16)v.z;                                                \
                hv[3] = (_Float16)v.w;                                                \
                *(h4*)((Xs_) + (8 * j + 2 * w + lh) * XROW_F16 + 8 * l31) = hv;       \
            }                                                                         \
        } else {                                                                      \
            _Pragma("unroll") for (int j = 0; j < K::XL; ++j)                         \
                *(f4*)((Xs_) + ((8 * j + 2 * w + lh) * 128 + 4 * l31) * 4) = x_[j]; \
        }                                                                             \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                 \
            *(u4*)((Qs_) + (j * 256 + tid) * 16) = q_[j];                          \
    } while (0)

    auto mfma_step = [&](const char* Xs, const char* Qs) __attribute__((always_inline)) {
        if (MODE == PP_MATCH_FAST) {
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                h8 a[4], bf[2];
#pragma unroll
                for (int tb = 0; tb < 4; ++tb)
                    a[tb] = *(const h8*)(Qs + (kh * 8 + wr * 4 + tb) * 1024 + lane * 16);
#pragma unroll
                for (int sb = 0; sb < 2; ++sb) {
                    const int col = wc * 64 + sb * 32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
                    const int row = kh * 16 + 8 * lh + ((lane & 15) >> 2);
                    const char* p = Xs + row * XROW_F16 + col * 2;
                    fp16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                        (__attribute__((address_space(3))) fp16x4_t*)(p));
                    fp16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                        (__attribute__((address_space(3))) fp16x4_t*)(p + 4 * XROW_F16));
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        bf[sb][e] = (_Float16)lo[e];
                        bf[sb][4 + e] = (_Float16)hi[e];
                    }
                }
#pragma unroll
                for (int tb = 0; tb < 4; ++tb)
#pragma unroll
                    for (int sb = 0; sb < 2; ++sb)
                        acc[tb][sb] =
                            __builtin_amdgcn_mfma_f32_32x32x16_f16(a[tb], bf[sb], acc[tb][sb], 0, 0, 0);
            }
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

    // ---- main loop: register prefetch two K-steps ahead, LDS double buffer, one barrier/step
    LOAD_STEP(0, xa, qa);
    LOAD_STEP(1, xb, qb);
    for (int ks = 0; ks < KT; ks += 2) {
        STORE_STEP(Xs0, Qs0, xa, qa);
        if (ks + 2 < KT) LOAD_STEP(ks + 2, xa, qa);
        __syncthreads();
        mfma_step(Xs0, Qs0);
        STORE_STEP(Xs1, Qs1, xb, qb);
        if (ks + 3 < KT) LOAD_STEP(ks + 3, xb, qb);
        __syncthreads();
        mfma_step(Xs1, Qs1);
    }
#undef LOAD_STEP
#undef STORE_STEP
    __syncthreads();

    // ---------------------------------------------------------------- epilogue
    float* T = (float*)smem;
    float* red = (float*)(smem + EPI_RED);
    float* rs = (float*)(smem + EPI_RS);
    float* colp = (float*)(smem + EPI_COLP);
    float* sim0 = (float*)(smem + EPI_SIM0);
    float* st0 = (float*)(smem + EPI_ST0);

    // 1. column norms -> 1/max(||x_s||, 1e-12)   (F.normalize, matching.py:43)
    if (MODE == PP_MATCH_FAST) {
        *(float4*)(red + (2 * w + lh) * 128 + 4 * l31) = make_float4(ssq0, ssq1, ssq2, ssq3);
    } else {
        const float s0 = ssq0 + __shfl_xor(ssq0, 32);
        const float s1 = ssq1 + __shfl_xor(ssq1, 32);
        if (wr == 0 && lh == 0) {
            red[wc * 64 + l31] = s0;
            red[wc * 64 + 32 + l31] = s1;
        }
    }
    __syncthreads();
    if (tid < 128) {
        float s;
        if (MODE == PP_MATCH_FAST) {
            s = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) s += red[i * 128 + tid];
        } else {
            s = red[tid];
        }
        rs[tid] = 1.0f / fmaxf(sqrtf(s), 1e-12f);
    }
    __syncthreads();

    // 2. scale columns, 3. column maxima over this wave's 128 rows
    float cm[2];
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
        const float r = rs[wc * 64 + sb * 32 + l31];
        float m = -INFINITY;
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                acc[tb][sb][e] *= r;
                m = fmaxf(m, acc[tb][sb][e]);
            }
        cm[sb] = fmaxf(m, __shfl_xor(m, 32));
    }
    if (lh == 0) {
        colp[wr * 128 + wc * 64 + l31] = cm[0];
        colp[wr * 128 + wc * 64 + 32 + l31] = cm[1];
        if (wr == 0) {  // row t = 0 lives in tb 0, register 0, lanes 0..31
            sim0[wc * 64 + l31] = acc[0][0][0];
            sim0[wc * 64 + 32 + l31] = acc[0][1][0];
        }
    }

    // 4. row maxima over this workgroup's 128 columns: transpose through LDS, two rounds
#pragma unroll
    for (int round = 0; round < 2; ++round) {
#pragma unroll
        for (int tbb = 0; tbb < 2; ++tbb) {
            const int tb = round * 2 + tbb;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rl = tbb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                T[(w * 64 + rl) * TROW + l31] = fmaxf(acc[tb][0][e], acc[tb][1][e]);
                if (wc == 0 && l31 == 0) st0[wr * 128 + round * 64 + rl] = acc[tb][0][e];
            }
        }
        __syncthreads();
        if (tid < 128) {
            const int wrr = tid >> 6, rl = tid & 63;
            const float4* r0 = (const float4*)(T + ((2 * wrr) * 64 + rl) * TROW);
            const float4* r1 = (const float4*)(T + ((2 * wrr + 1) * 64 + rl) * TROW);
            float m = -INFINITY;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float4 u = r0[i], v = r1[i];
                m = fmaxf(m, fmaxf(fmaxf(u.x, u.y), fmaxf(u.z, u.w)));
                m = fmaxf(m, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
            }
            rowmax[(bn * 2 + half) * P + wrr * 128 + round * 64 + rl] = m;
        }
        __syncthreads();
    }
    if (tid < 128) {
        colmax[bn * P + half * 128 + tid] = fmaxf(colp[tid], colp[128 + tid]);
        sim0s[bn * P + half * 128 + tid] = sim0[tid];
    }
    if (half == 0) simt0[bn * P + tid] = st0[tid];
}

// ---------------------------------------------------------------------------
// FAST mode: find the rows/columns whose "arg-max is patch 0" decision is
// within eps of a tie and queue them for exact re-evaluation.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void s1_detect(const float* __restrict__ m16, int N, float eps,
                                                 const float* __restrict__ rowmax,
                                                 const float* __restrict__ simt0,
                                                 const float* __restrict__ colmax,
                                                 const float* __restrict__ sim0s,
                                                 int32_t* __restrict__ counter,
                                                 uint32_t* __restrict__ flags) {
    const size_t bn = blockIdx.x;
    const int b = (int)(bn / N), i = threadIdx.x;
    const float m = m16[b * P + i];
    if (m == 0.f) return;  // mask_all[i] = 0 whatever the decisions are
    const float rm = fmaxf(rowmax[(bn * 2) * P + i], rowmax[(bn * 2 + 1) * P + i]);
    if (fabsf(rm - simt0[bn * P + i]) <= eps) {
        const int k = atomicAdd(&counter[0], 1);
        flags[k] = ((uint32_t)bn << 9) | (uint32_t)i;
        atomicAdd(&counter[1], 1);
    }
    if (fabsf(colmax[bn * P + i] - sim0s[bn * P + i]) <= eps) {
        const int k = atomicAdd(&counter[0], 1);
        flags[k] = ((uint32_t)bn << 9) | 256u | (uint32_t)i;
        atomicAdd(&counter[2], 1);
    }
}

// Exact fp32 re-evaluation of one row (all s for query patch i) or one column
// (all t for template patch i): the same c-ordered fma chain as EXACT mode.
__global__ __launch_bounds__(256) void s1_fixup(const float* __restrict__ bank,
                                                const float* __restrict__ qf, int N, int C,
                                                const int32_t* __restrict__ counter,
                                                const uint32_t* __restrict__ flags,
                                                float* __restrict__ rowmax,
                                                float* __restrict__ simt0,
                                                float* __restrict__ colmax,
                                                float* __restrict__ sim0s) {
    __shared__ float wmax[4];
    const int cnt = counter[0];
    const int tid = threadIdx.x;
    for (int e = blockIdx.x; e < cnt; e += gridDim.x) {
        const uint32_t f = flags[e];
        const size_t bn = f >> 9;
        const int kind = (f >> 8) & 1, i = f & 255;
        const int b = (int)(bn / N);
        const float* X = bank + bn * (size_t)C * P;
        const float* Q = qf + (size_t)b * C * P;
        float dot = 0.f, ss = 0.f;
        if (kind == 0) {  // row i: thread = template patch s
            for (int c = 0; c < C; ++c) {
                const float x = X[(size_t)c * P + tid];
                dot = fmaf(Q[(size_t)c * P + i], x, dot);
                ss = fmaf(x, x, ss);
            }
        } else {  // column i: thread = query patch t
            for (int c = 0; c < C; ++c) {
                const float x = X[(size_t)c * P + i];
                dot = fmaf(Q[(size_t)c * P + tid], x, dot);
                ss = fmaf(x, x, ss);
            }
        }
        const float sim = dot * (1.0f / fmaxf(sqrtf(ss), 1e-12f));
        float m = sim;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        __syncthreads();
        if ((tid & 63) == 0) wmax[tid >> 6] = m;
        __syncthreads();
        if (tid == 0) {
            const float mx = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
            if (kind == 0) {
                rowmax[(bn * 2) * P + i] = mx;
                rowmax[(bn * 2 + 1) * P + i] = -INFINITY;
                simt0[bn * P + i] = sim;
            } else {
                colmax[bn * P + i] = mx;
                sim0s[bn * P + i] = sim;
            }
        }
    }
}

// sim_avg (matching.py:53-66).  One workgroup per (b,n).
__global__ __launch_bounds__(256) void s1_finalize(const float* __restrict__ m16, int N,
                                                   const float* __restrict__ rowmax,
                                                   const float* __restrict__ simt0,
                                                   const float* __restrict__ colmax,
                                                   const float* __restrict__ sim0s,
                                                   float* __restrict__ sim_avg) {
    __shared__ float ps[4], pm[4];
    const size_t bn = blockIdx.x;
    const int b = (int)(bn / N), i = threadIdx.x;
    const float m = m16[b * P + i];
    const float rm = fmaxf(rowmax[(bn * 2) * P + i], rowmax[(bn * 2 + 1) * P + i]);
    const float rnz = rm > simt0[bn * P + i] ? 1.f : 0.f;
    const float cnz = colmax[bn * P + i] > sim0s[bn * P + i] ? 1.f : 0.f;
    const float mall = m * cnz * rnz;
    float s = rm * mall, ms = mall;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o);
        ms += __shfl_xor(ms, o);
    }
    if ((i & 63) == 0) {
        ps[i >> 6] = s;
        pm[i >> 6] = ms;
    }
    __syncthreads();
    if (i == 0) {
        const float tot = (ps[0] + ps[1]) + (ps[2] + ps[3]);
        const float mt = (pm[0] + pm[1]) + (pm[2] + pm[3]);
        sim_avg[bn] = mt > 0.f ? tot / 256.0f : 0.f;
    }
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

__global__ __launch_bounds__(256) void topk_rows(const float* __restrict__ scores, int N, int k,
                                                 float* __restrict__ out_score,
                                                 int64_t* __restrict__ out_index) {
    extern __shared__ float sc[];  // N floats followed by N taken flags
    unsigned char* taken = (unsigned char*)(sc + N);
    __shared__ float wv[4];
    __shared__ int wi[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int j = tid; j < N; j += 256) {
        sc[j] = scores[(size_t)b * N + j];
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
            out_score[(size_t)b * k + it] = bv;
            out_index[(size_t)b * k + it] = bi;
            taken[bi] = 1;
        }
        __syncthreads();
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

int pp_stage1_scores(const float* bank, const float* query, const float* mask, int mask_h,
                     int mask_w, int B, int N, int C, int mode, float eps, void* workspace,
                     size_t workspace_bytes, float* sim_avg, int32_t* stats, void* stream_) {
    if (!bank || !query || !mask || !sim_avg || !workspace) return PP_EINVAL;
    if (B <= 0 || N <= 0 || C <= 0 || mask_h <= 0 || mask_w <= 0) return PP_EINVAL;
    if (mode != PP_MATCH_EXACT && mode != PP_MATCH_FAST) return PP_EINVAL;
    if (C % 64 != 0) return PP_EINVAL;
    if ((size_t)B * N >= (1u << 23)) return PP_EINVAL;
    if (((uintptr_t)workspace & 255) != 0) return PP_EWORKSPACE;
    if (((uintptr_t)bank & 15) != 0) return PP_EINVAL;
    S1Ws w = carve(workspace, B, N, C);
    if (workspace_bytes < w.total) return PP_EWORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    if (eps <= 0.f) eps = 2e-4f;

    hipLaunchKernelGGL(s1_prepack, dim3(B), dim3(256), 0, stream, query, mask, mask_h, mask_w, C,
                       w.qh, w.qf, w.m16);
    const int grid = (B >= 8) ? 8 * ((B + 7) / 8) * 2 * N : B * 2 * N;
    if (mode == PP_MATCH_FAST) {
        hipLaunchKernelGGL(s1_main<PP_MATCH_FAST>, dim3(grid), dim3(256), SMEM_BYTES, stream, bank,
                           w.qh, w.qf, B, N, C, w.rowmax, w.simt0, w.colmax, w.sim0s);
        PP_CHECK_HIP(hipMemsetAsync(w.counter, 0, 64, stream));
        hipLaunchKernelGGL(s1_detect, dim3(B * N), dim3(256), 0, stream, w.m16, N, eps, w.rowmax,
                           w.simt0, w.colmax, w.sim0s, w.counter, w.flags);
        hipLaunchKernelGGL(s1_fixup, dim3(2048), dim3(256), 0, stream, bank, w.qf, N, C, w.counter,
                           w.flags, w.rowmax, w.simt0, w.colmax, w.sim0s);
        if (stats)
            PP_CHECK_HIP(hipMemcpyAsync(stats, w.counter + 1, 2 * sizeof(int32_t),
                                        hipMemcpyDeviceToDevice, stream));
    } else {
        hipLaunchKernelGGL(s1_main<PP_MATCH_EXACT>, dim3(grid), dim3(256), SMEM_BYTES, stream, bank,
                           w.qh, w.qf, B, N, C, w.rowmax, w.simt0, w.colmax, w.sim0s);
        if (stats) PP_CHECK_HIP(hipMemsetAsync(stats, 0, 2 * sizeof(int32_t), stream));
    }
    hipLaunchKernelGGL(s1_finalize, dim3(B * N), dim3(256), 0, stream, w.m16, N, w.rowmax, w.simt0,
                       w.colmax, w.sim0s, sim_avg);
    return pp_last_launch();
}

int pp_topk(const float* scores, int B, int N, int k, float* out_score, int64_t* out_index,
            void* stream_) {
    if (!scores || !out_score || !out_index) return PP_EINVAL;
    if (B <= 0 || N <= 0 || k <= 0 || k > N || N > 12288) return PP_EINVAL;
    hipLaunchKernelGGL(topk_rows, dim3(B), dim3(256), N * 5, (hipStream_t)stream_,
                       scores, N, k, out_score, out_index);
    return pp_last_launch();
}

int pp_stage1_match(const float* bank, const float* query, const float* mask, int mask_h,
                    int mask_w, int B, int N, int C, int k, int mode, float eps, void* workspace,
                    size_t workspace_bytes, float* sim_avg, float* out_score, int64_t* out_index,
                    int32_t* stats, void* stream) {
    int rc = pp_stage1_scores(bank, query, mask, mask_h, mask_w, B, N, C, mode, eps, workspace,
                              workspace_bytes, sim_avg, stats, stream);
    if (rc != PP_OK) return rc;
    return pp_topk(sim_avg, B, N, k, out_score, out_index, stream);
}

}  // extern "C"
