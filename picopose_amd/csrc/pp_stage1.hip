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
    float* denom;   // (B, 256) max(||q_t||, 1e-12)
    float4* rowrec; // (B*N, 2, 256) per-half row records over s > 0: {best, second, arg bits, -}
    float* simt0;   // (B*N, 256) sim[t,0]
    float* colmax;  // (B*N, 256) column maxima over t > 0
    float* sim0s;   // (B*N, 256) sim[0,s]
    int32_t* counter;  // [0]=#full entries [1]=#candidate rows [2]=#full rows [3]=#full columns
    uint2* flags;      // candidate-row entries {bn, t<<8 | s}
    uint2* full;       // full row/column entries {bn, kind<<16 | index<<8}
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
    w.counter = (int32_t*)take(64);
    w.flags = (uint2*)take(BN * P * 8);
    w.full = (uint2*)take(BN * 2 * P * 8);
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
                                                  float4* __restrict__ rowrec,
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

#ifndef PP_S1_XDEPTH
#define PP_S1_XDEPTH 2
#endif
    f4 x0[K::XL], x1[K::XL];  // X tiles of the next PP_S1_XDEPTH K-steps, in flight
#if PP_S1_XDEPTH == 3
    f4 x2[K::XL];
#endif
    u4 qr[4];                              // query tile of the next K-step (L2 resident)

#define LOAD_X(ks_, x_)                                                               \
    do {                                                                              \
        _Pragma("unroll") for (int j = 0; j < K::XL; ++j) x_[j] =                     \
            *(const f4*)(xptr + (size_t)((ks_) * K::KS + 8 * j) * P);                 \
    } while (0)
#define LOAD_Q(ks_)                                                                   \
    do {                                                                              \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) qr[j] =                         \
            qptr[(size_t)(ks_) * 1024 + j * 256];                                     \
    } while (0)

#define STORE_STEP(Xs_, Qs_, x_)                                                      \
    do {                                                                              \
        if (MODE == PP_MATCH_FAST) {                                                  \
            _Pragma("unroll") for (int j = 0; j < K::XL; ++j) {                       \
                const f4 v = x_[j];                                                   \
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
                *(f4*)((Xs_) + ((8 * j + 2 * w + lh) * 128 + 4 * l31) * 4) = x_[j];   \
        }                                                                             \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                 \
            *(u4*)((Qs_) + (j * 256 + tid) * 16) = qr[j];                             \
    } while (0)

// one K-step: publish tile ks (registers -> LDS buffer ks&1), refill the registers with
// tiles ks+XDEPTH (X) and ks+1 (Q), one barrier, MFMAs on the published tile
#define STEP(ks_, x_, Xs_, Qs_)                                                       \
    do {                                                                              \
        STORE_STEP(Xs_, Qs_, x_);                                                     \
        if ((ks_) + PP_S1_XDEPTH < KT) LOAD_X((ks_) + PP_S1_XDEPTH, x_);                                    \
        if ((ks_) + 1 < KT) LOAD_Q((ks_) + 1);                                        \
        __syncthreads();                                                              \
        mfma_step(Xs_, Qs_);                                                          \
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

    // ---- main loop: X prefetched XDEPTH K-steps ahead in registers (Q one step: it is L2
    //      resident), LDS double buffer, one barrier per K-step.  KT is even (C % 64 == 0).
    LOAD_X(0, x0);
    LOAD_X(1, x1);
#if PP_S1_XDEPTH == 3
    if (2 < KT) LOAD_X(2, x2);
    LOAD_Q(0);
    for (int ks = 0; ks < KT; ks += 6) {  // 6 = lcm(3 register sets, 2 LDS buffers)
        STEP(ks, x0, Xs0, Qs0);
        STEP(ks + 1, x1, Xs1, Qs1);
        if (ks + 2 < KT) {
            STEP(ks + 2, x2, Xs0, Qs0);
            STEP(ks + 3, x0, Xs1, Qs1);
        }
        if (ks + 4 < KT) {
            STEP(ks + 4, x1, Xs0, Qs0);
            STEP(ks + 5, x2, Xs1, Qs1);
        }
    }
#else
    LOAD_Q(0);
    for (int ks = 0; ks < KT; ks += 2) {
        STEP(ks, x0, Xs0, Qs0);
        STEP(ks + 1, x1, Xs1, Qs1);
    }
#endif
#undef STEP
#undef LOAD_X
#undef LOAD_Q
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

    // 2. scale columns, 3. column maxima over this wave's 128 rows, row t = 0 excluded
    //    (i2[s] != 0  <=>  max_{t>0} sim[t,s] > sim[0,s]); row 0 is tb 0, register 0, lanes 0..31
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
                if (tb != 0 || e != 0) m = fmaxf(m, acc[tb][sb][e]);
            }
        if (!(wr == 0 && lh == 0)) m = fmaxf(m, acc[0][sb][0]);
        cm[sb] = fmaxf(m, __shfl_xor(m, 32));
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
    //    (i1[t] != 0  <=>  max_{s>0} sim[t,s] > sim[t,0]): transpose through LDS, two rounds
    const bool zlane = (half == 0 && wc == 0 && l31 == 0);
#pragma unroll
    for (int round = 0; round < 2; ++round) {
#pragma unroll
        for (int tbb = 0; tbb < 2; ++tbb) {
            const int tb = round * 2 + tbb;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rl = tbb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                // column s = 0 (half 0, wc 0, lane column 0, sb 0) stays out of the row maximum
                T[(w * 64 + rl) * TROW + l31] =
                    zlane ? acc[tb][1][e] : fmaxf(acc[tb][0][e], acc[tb][1][e]);
                if (wc == 0 && l31 == 0) st0[wr * 128 + round * 64 + rl] = acc[tb][0][e];
            }
        }
        __syncthreads();
        if (tid < 128) {
            // Row record over this half's 64 transpose entries (entry = max of columns s, s+32):
            //   .x best entry   .y second best entry   .z column s of the best entry (int bits)
            const int wrr = tid >> 6, rl = tid & 63;
            const float* r0 = T + ((2 * wrr) * 64 + rl) * TROW;
            const float* r1 = T + ((2 * wrr + 1) * 64 + rl) * TROW;
            float a1 = -INFINITY, a2 = -INFINITY;
            int p1 = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const f4 u = *(const f4*)((i < 8 ? r0 : r1) + 4 * (i & 7));
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float val = u[e];
                    a2 = fmaxf(a2, fminf(a1, val));
                    p1 = val > a1 ? 4 * i + e : p1;
                    a1 = fmaxf(a1, val);
                }
            }
            const int scol = half * 128 + (p1 >> 5) * 64 + (p1 & 31);
            rowrec[(bn * 2 + half) * P + wrr * 128 + round * 64 + rl] =
                make_float4(a1, a2, __int_as_float(scol), 0.f);
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
// within eps of a tie and queue them for exact fp32 re-evaluation.
//   kind 0: row t, candidate columns {0, s, s+32}: every other column is more
//           than 2*eps below the best of s/s+32, so the exact max over s > 0 is
//           one of them
//   kind 1: row t, all columns        kind 2: column s, all rows
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void s1_detect(const float* __restrict__ m16, int N, float eps,
                                                 const float4* __restrict__ rowrec,
                                                 const float* __restrict__ simt0,
                                                 const float* __restrict__ colmax,
                                                 const float* __restrict__ sim0s,
                                                 int32_t* __restrict__ counter,
                                                 uint2* __restrict__ flags,
                                                 uint2* __restrict__ full) {
    const size_t bn = blockIdx.x;
    const int b = (int)(bn / N), i = threadIdx.x;
    const float m = m16[b * P + i];
    if (m == 0.f) return;  // mask_all[i] = 0 whatever the decisions are
    const float4 r0 = rowrec[(bn * 2) * P + i], r1 = rowrec[(bn * 2 + 1) * P + i];
    const float rm = fmaxf(r0.x, r1.x);  // best over s > 0
    if (fabsf(rm - simt0[bn * P + i]) <= eps) {
        const bool w0 = r0.x >= r1.x;
        const float second = fmaxf(w0 ? r1.x : r0.x, w0 ? r0.y : r1.y);
        const int scol = __float_as_int(w0 ? r0.z : r1.z);
        if (second < rm - 2.f * eps) {
            const int k = atomicAdd(&counter[1], 1);
            flags[k] = make_uint2((uint32_t)bn, ((uint32_t)i << 8) | (uint32_t)scol);
        } else {
            const int k = atomicAdd(&counter[0], 1);
            full[k] = make_uint2((uint32_t)bn, (1u << 16) | ((uint32_t)i << 8));
            atomicAdd(&counter[2], 1);
        }
    }
    if (fabsf(colmax[bn * P + i] - sim0s[bn * P + i]) <= eps) {
        const int k = atomicAdd(&counter[0], 1);
        full[k] = make_uint2((uint32_t)bn, (2u << 16) | ((uint32_t)i << 8));
        atomicAdd(&counter[3], 1);
    }
}

// Exact re-evaluation.  The arithmetic repeats EXACT mode's: the dot product is the
// c-ordered fp32 fma chain of v_mfma_f32_32x32x2_f32, the column norm is the sum of an
// even-channel and an odd-channel fma chain, sim = dot * (1 / max(sqrt(ss), 1e-12)).
//
// s1_fixup_rows: one wave per candidate-row entry; the wave stages 4 channel vectors
// (q[:,t], x[:,0], x[:,s+32], x[:,s]) in LDS with all loads in flight at once, then
// three lanes run the chains.
__global__ __launch_bounds__(256) void s1_fixup_rows(const float* __restrict__ bank,
                                                     const float* __restrict__ qf, int N, int C,
                                                     const int32_t* __restrict__ counter,
                                                     const uint2* __restrict__ flags,
                                                     float4* __restrict__ rowrec,
                                                     float* __restrict__ simt0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int cnt = counter[1];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* L = (float*)smem + (size_t)wv * 4 * C;  // [4][C]: q, x[:,0], x[:,s+32], x[:,s]
    for (int e = blockIdx.x * 4 + wv; e < cnt; e += gridDim.x * 4) {
        const uint2 f = flags[e];
        const size_t bn = f.x;
        const int i = (f.y >> 8) & 255, sc = f.y & 255;
        const int b = (int)(bn / N);
        const float* X = bank + bn * (size_t)C * P;
        const float* Q = qf + (size_t)b * C * P;
        const int cols[3] = {0, sc + 32, sc};
        for (int c0 = 0; c0 < C; c0 += 64 * 4) {
            float vq[4], vx[3][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {  // 16 loads in flight; the tail re-reads channel C-1
                const int c = min(c0 + 64 * j + lane, C - 1);
                vq[j] = Q[(size_t)c * P + i];
#pragma unroll
                for (int k = 0; k < 3; ++k) vx[k][j] = X[(size_t)c * P + cols[k]];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = c0 + 64 * j + lane;
                if (c < C) {
                    L[c] = vq[j];
#pragma unroll
                    for (int k = 0; k < 3; ++k) L[(k + 1) * C + c] = vx[k][j];
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's LDS stores are done
        __builtin_amdgcn_wave_barrier();
        float sim = -INFINITY;
        if (lane < 3) {
            const float* xv = L + (lane + 1) * C;
            float dot = 0.f, se = 0.f, so = 0.f;
#pragma unroll 8
            for (int c = 0; c < C; c += 2) {
                const float x0 = xv[c], x1 = xv[c + 1];
                dot = fmaf(L[c], x0, dot);
                dot = fmaf(L[c + 1], x1, dot);
                se = fmaf(x0, x0, se);
                so = fmaf(x1, x1, so);
            }
            sim = dot * (1.0f / fmaxf(sqrtf(se + so), 1e-12f));
        }
        const float s0 = __shfl(sim, 0), s1 = __shfl(sim, 1), s2 = __shfl(sim, 2);
        const float mx = sc == 0 ? s1 : fmaxf(s1, s2);  // sc == 0: entry {0, 32}, 0 excluded
        if (lane == 0) {
            rowrec[(bn * 2) * P + i].x = mx;
            rowrec[(bn * 2 + 1) * P + i].x = -INFINITY;
            simt0[bn * P + i] = s0;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// s1_fixup_full: one workgroup per entry, thread o = output index (template patch for a
// row entry, query patch for a column entry); 32 channels of loads in flight per thread.
__global__ __launch_bounds__(256) void s1_fixup_full(const float* __restrict__ bank,
                                                     const float* __restrict__ qf, int N, int C,
                                                     const int32_t* __restrict__ counter,
                                                     const uint2* __restrict__ full,
                                                     float4* __restrict__ rowrec,
                                                     float* __restrict__ simt0,
                                                     float* __restrict__ colmax,
                                                     float* __restrict__ sim0s) {
    __shared__ float wmax[4];
    const int cnt = counter[0];
    const int o = threadIdx.x;
    for (int e = blockIdx.x; e < cnt; e += gridDim.x) {
        const uint2 f = full[e];
        const size_t bn = f.x;
        const int kind = (int)(f.y >> 16), i = (f.y >> 8) & 255;
        const int b = (int)(bn / N);
        const float* X = bank + bn * (size_t)C * P;
        const float* Q = qf + (size_t)b * C * P;
        // row entry: a = q[:,i] (uniform), x = X[:,o];  column entry: a = q[:,o], x = X[:,i]
        const float* ap = Q + (kind == 1 ? i : o);
        const float* xp = X + (kind == 1 ? o : i);
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
        float m = o == 0 ? -INFINITY : sim;  // index 0 is the other side of the decision
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
        __syncthreads();
        if ((o & 63) == 0) wmax[o >> 6] = m;
        __syncthreads();
        if (o == 0) {
            const float mx = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
            if (kind == 1) {
                rowrec[(bn * 2) * P + i].x = mx;
                rowrec[(bn * 2 + 1) * P + i].x = -INFINITY;
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
                                                   const float4* __restrict__ rowrec,
                                                   const float* __restrict__ simt0,
                                                   const float* __restrict__ colmax,
                                                   const float* __restrict__ sim0s,
                                                   float* __restrict__ sim_avg) {
    __shared__ float ps[4], pm[4];
    const size_t bn = blockIdx.x;
    const int b = (int)(bn / N), i = threadIdx.x;
    const float m = m16[b * P + i];
    const float rme = fmaxf(rowrec[(bn * 2) * P + i].x, rowrec[(bn * 2 + 1) * P + i].x);
    const float st0 = simt0[bn * P + i];
    const float rnz = rme > st0 ? 1.f : 0.f;  // idx_tar2src != 0 (first max wins ties)
    const float rm = fmaxf(rme, st0);         // score_tar2src
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
    if (C % 64 != 0 || C > 2048) return PP_EINVAL;
    if ((size_t)B * N >= (1u << 23)) return PP_EINVAL;
    if (((uintptr_t)workspace & 255) != 0) return PP_EWORKSPACE;
    if (((uintptr_t)bank & 15) != 0) return PP_EINVAL;
    S1Ws w = carve(workspace, B, N, C);
    if (workspace_bytes < w.total) return PP_EWORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    // default band: 8 sigma of the fp16 rounding error of a score difference,
    // sigma ~ sqrt(2) * 2.8e-4 / sqrt(C) for unit vectors with spread-out energy
    if (eps <= 0.f) eps = 3.2e-3f / sqrtf((float)C);

    hipLaunchKernelGGL(s1_qnorm, dim3(B, 4), dim3(1024), 0, stream, query, mask, mask_h, mask_w, C,
                       w.denom, w.m16);
    hipLaunchKernelGGL(s1_qpack, dim3(B, C / 32), dim3(256), 0, stream, query, w.denom, w.m16, C,
                       w.qh, w.qf);
    const int grid = (B >= 8) ? 8 * ((B + 7) / 8) * 2 * N : B * 2 * N;
    if (mode == PP_MATCH_FAST) {
        {
            PpProfScope prof(stream);  // roofline kernel of stage 1 (bench.py)
            hipLaunchKernelGGL(s1_main<PP_MATCH_FAST>, dim3(grid), dim3(256), SMEM_BYTES, stream,
                               bank, w.qh, w.qf, B, N, C, w.rowrec, w.simt0, w.colmax, w.sim0s);
        }
        PP_CHECK_HIP(hipMemsetAsync(w.counter, 0, 64, stream));
        hipLaunchKernelGGL(s1_detect, dim3(B * N), dim3(256), 0, stream, w.m16, N, eps, w.rowrec,
                           w.simt0, w.colmax, w.sim0s, w.counter, w.flags, w.full);
        hipLaunchKernelGGL(s1_fixup_rows, dim3(512), dim3(256), (size_t)4 * 4 * C * sizeof(float),
                           stream, bank, w.qf, N, C, w.counter, w.flags, w.rowrec, w.simt0);
        hipLaunchKernelGGL(s1_fixup_full, dim3(512), dim3(256), 0, stream, bank, w.qf, N, C,
                           w.counter, w.full, w.rowrec, w.simt0, w.colmax, w.sim0s);
        if (stats)
            PP_CHECK_HIP(hipMemcpyAsync(stats, w.counter + 1, 3 * sizeof(int32_t),
                                        hipMemcpyDeviceToDevice, stream));
    } else {
        {
            PpProfScope prof(stream);
            hipLaunchKernelGGL(s1_main<PP_MATCH_EXACT>, dim3(grid), dim3(256), SMEM_BYTES, stream,
                               bank, w.qh, w.qf, B, N, C, w.rowrec, w.simt0, w.colmax, w.sim0s);
        }
        if (stats) PP_CHECK_HIP(hipMemsetAsync(stats, 0, 3 * sizeof(int32_t), stream));
    }
    hipLaunchKernelGGL(s1_finalize, dim3(B * N), dim3(256), 0, stream, w.m16, N, w.rowrec, w.simt0,
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
