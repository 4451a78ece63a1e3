// Adjoint of the fused self-attention (model/stage1/layers/attention.py:49-62) for the training step: the T x T probabilities
// are never stored.  The forward (pp_attention_train, pp_attn.hip) keeps, per (image, head, query), the base-2 log-sum-exp of its
// scaled scores; here the scores are recomputed tile by tile from q and k, the probabilities follow as exp2(s - lse), and
//     D  = rowsum(dO . O)            dP = dO V^T            dS = P . (dP - D)
//     dV = P^T dO                    dQ = scale dS K        dK = scale dS^T Q
// in the engine's f16x3 arithmetic (two fp16 terms per operand, 3 x v_mfma_f32_32x32x16_f16 per product, fp32 accumulation).
// Two kernels, no atomics (bit-reproducible): attn_bwd_dq_kernel walks the keys for 32 queries per wave, attn_bwd_dkv_kernel walks
// the queries for 32 keys per wave; both recompute S and dP (7 products instead of 5 — the price of having no exchange).
// Both keep the forward kernel's TRANSPOSED tiles: a lane owns one column (a query, resp. a key), so the tile in the MFMA accumulator
// layout IS the B operand of the next product, and the A operand of that product is read from LDS with ds_read_b64_tr_b16.
//
// Ranges.  dO arrives with a device scalar g = 2^k that brings max|dO| into [512, 1024) (autograd._pow2_scale): its operand is
// split(g dO).  dS spans many orders of magnitude (P down to 1e-30), so every 32 x 32 tile of dS is split with ITS OWN power of two
// per column (the column's largest |dS| lands in [512, 1024)), multiplied into a zeroed accumulator and added to the running one
// with the inverse power — exact, and the small tiles keep their 22 bits.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/picopose_hip.h"
#include "pp_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));

constexpr int HD = 64, KC = 32;
constexpr int RLD = 72;   // halfs per row of a row-major operand tile (144-byte stride: conflict-free ds_read_b128)
constexpr int TLD = 96;   // halfs per row of a slot-major tile read with ds_read_b64_tr_b16 (192-byte stride, as pp_attn.hip)
constexpr int OLD = 68;   // floats per staged output row
constexpr float P_SCALE = 1024.f;
constexpr int ROW_TILE = KC * RLD, SLOT_TILE = KC * TLD;   // halfs per plane

// row (0..31) of a chunk -> k-slot order of accumulator registers (pp_attn.hip vt_slot)
__device__ __forceinline__ int vt_slot(int r32) {
    const int s = r32 >> 4, r = r32 & 15;
    return s * 16 + ((r >> 2) & 1) * 8 + (r & 3) + 4 * (r >> 3);
}

// 32 rows x 64 floats (row r0 + i of a matrix with row stride ld; rows >= T read as zero), times mul, split, into LDS:
// row-major planes R (hi at R, lo at R + ROW_TILE) and / or slot-major planes S (hi at S, lo at S + SLOT_TILE).
template <bool ROWMAJ, bool SLOTMAJ>
__device__ __forceinline__ void stage32(const float* __restrict__ src, size_t ld, int r0, int T, float mul, _Float16* R, _Float16* S,
                                        int tid, int nthr) {
    for (int idx = tid; idx < KC * (HD / 4); idx += nthr) {
        const int row = idx >> 4, c4 = (idx & 15) * 4;
        f4 v = {0.f, 0.f, 0.f, 0.f};
        if (r0 + row < T) v = *(const f4*)(src + (size_t)(r0 + row) * ld + c4);
        h4 hh, ll;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            _Float16 a, c;
            pp_split_f16(v[i] * mul, a, c);
            hh[i] = a;
            ll[i] = c;
        }
        if (ROWMAJ) {
            *(h4*)(R + row * RLD + c4) = hh;
            *(h4*)(R + ROW_TILE + row * RLD + c4) = ll;
        }
        if (SLOTMAJ) {
            const int slot = vt_slot(row);
            *(h4*)(S + slot * TLD + c4) = hh;
            *(h4*)(S + SLOT_TILE + slot * TLD + c4) = ll;
        }
    }
}

// B-operand fragments of one row of 64 floats (times mul): step s holds d = 16 s + 8 lh .. + 7
__device__ __forceinline__ void row_frags(const float* __restrict__ p, int lh, float mul, h8 (&fh)[4], h8 (&fl)[4]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const f4 a = *(const f4*)(p + 16 * s + 8 * lh), c = *(const f4*)(p + 16 * s + 8 * lh + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            _Float16 hh, ll;
            pp_split_f16(a[i] * mul, hh, ll);
            fh[s][i] = hh;
            fl[s][i] = ll;
            pp_split_f16(c[i] * mul, hh, ll);
            fh[s][4 + i] = hh;
            fl[s][4 + i] = ll;
        }
    }
}

// A fragment (lane: d = l31 + 32 dhalf, k-slots 16 s + 8 lh .. + 7) by transposing reads from a slot-major plane
__device__ __forceinline__ h8 tfrag(const _Float16* V, int s, int dhalf, int lane) {
    const int lh = lane >> 5;
    const int col = 32 * dhalf + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    const int row = 16 * s + 8 * lh + ((lane & 15) >> 2);
    const _Float16* p = V + row * TLD + col;
    const fp16x4_t lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)p);
    const fp16x4_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(p + 4 * TLD));
    h8 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        r[e] = (_Float16)lo4[e];
        r[4 + e] = (_Float16)hi4[e];
    }
    return r;
}

// rows x columns product of a row-major LDS tile (A: lane = row l31) with register fragments (B): 3 MFMAs per k step
__device__ __forceinline__ f32x16 tile_nt(const _Float16* R, int l31, int lh, const h8 (&bh)[4], const h8 (&bl)[4]) {
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const h8 ah = *(const h8*)(R + l31 * RLD + 16 * s + 8 * lh);
        const h8 al = *(const h8*)(R + ROW_TILE + l31 * RLD + 16 * s + 8 * lh);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[s], acc, 0, 0, 0);
    }
    return acc;
}

// (d x columns) += (slot-major tile)^T x (tile in accumulator layout, split as xh / xl): both d halves
__device__ __forceinline__ void tile_tn(const _Float16* S, int lane, const h8 (&xh)[2], const h8 (&xl)[2], f32x16& a0, f32x16& a1) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const h8 t0h = tfrag(S, s, 0, lane), t1h = tfrag(S, s, 1, lane);
        const h8 t0l = tfrag(S + SLOT_TILE, s, 0, lane), t1l = tfrag(S + SLOT_TILE, s, 1, lane);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(t0l, xh[s], a0, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(t0h, xl[s], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(t1l, xh[s], a1, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(t1h, xl[s], a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(t0h, xh[s], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(t1h, xh[s], a1, 0, 0, 0);
    }
}

// split a tile in accumulator layout with ONE power of two per column (lane pair l31 / l31 + 32): returns 2^-k
__device__ __forceinline__ float split_ranged(const f32x16& x, h8 (&xh)[2], h8 (&xl)[2]) {
    float mx = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) mx = fmaxf(mx, fabsf(x[e]));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    uint32_t e8 = __float_as_uint(mx) >> 23;            // biased exponent (mx >= 0); NaN / inf columns saturate below
    e8 = e8 < 40u ? 40u : (e8 > 250u ? 250u : e8);
    const float sc = __uint_as_float((263u - e8) << 23);     // 2^(9 - (e8 - 127)): the column's maximum lands in [512, 1024)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const float v = fminf(fmaxf(x[e] * sc, -65504.f), 65504.f);
        const _Float16 hh = (_Float16)v;
        xh[e >> 3][e & 7] = hh;
        xl[e >> 3][e & 7] = (_Float16)(v - (float)hh);
    }
    return __uint_as_float((e8 - 9u) << 23);
}

// (d x column) accumulators -> rows of 64 floats per column, through the wave's LDS region, written as full lines
__device__ __forceinline__ void write_rows(float* Os, const f32x16& a0, const f32x16& a1, float mul, float* __restrict__ dst, size_t ld,
                                           int row0, int T, int lane) {
    const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f4 a, c;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[i] = a0[4 * g + i] * mul;
            c[i] = a1[4 * g + i] * mul;
        }
        *(f4*)(Os + l31 * OLD + 8 * g + 4 * lh) = a;
        *(f4*)(Os + l31 * OLD + 32 + 8 * g + 4 * lh) = c;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave reads only what it wrote itself
    const int r = lane >> 1, half = lane & 1;
    if (row0 + r < T) {
        float* d = dst + (size_t)(row0 + r) * ld + 32 * half;
        const float* s = Os + r * OLD + 32 * half;
#pragma unroll
        for (int j = 0; j < 8; ++j) *(f4*)(d + 4 * j) = *(const f4*)(s + 4 * j);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
}

// Dg[(b heads + h) T + q] = g sum_d dO[b, q, h, d] O[b, q, h, d]: 16 lanes per (row, head)
__global__ __launch_bounds__(256) void attn_bwd_d_kernel(const float* __restrict__ out, const float* __restrict__ dout,
                                                         const float* __restrict__ gpair, int B, int T, int heads, float* __restrict__ Dg) {
    const long item = (long)blockIdx.x * 16 + (threadIdx.x >> 4);    // (b T + q) heads + h
    const int sub = threadIdx.x & 15;
    const long items = (long)B * T * heads;
    float acc = 0.f;
    if (item < items) {
        const f4 a = *(const f4*)(out + item * HD + 4 * sub), c = *(const f4*)(dout + item * HD + 4 * sub);
        acc = a.x * c.x + a.y * c.y + a.z * c.z + a.w * c.w;
    }
    acc += __shfl_xor(acc, 8);
    acc += __shfl_xor(acc, 4);
    acc += __shfl_xor(acc, 2);
    acc += __shfl_xor(acc, 1);
    if (item < items && sub == 0) {
        const long row = item / heads;
        const int h = (int)(item - row * heads);
        const long b = row / T;
        const int q = (int)(row - b * T);
        Dg[((size_t)b * heads + h) * T + q] = acc * gpair[0];
    }
}

// ---- dQ: a wave = 32 queries (lane = query column), the workgroup's waves share the staged K / V chunks -------------------------------
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void attn_bwd_dq_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                          const float* __restrict__ lse2, const float* __restrict__ Dg,
                                                          const float* __restrict__ gpair, int T, int heads, float scale,
                                                          float* __restrict__ dqkv, int ntx, int npairs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16* Kr = (_Float16*)smem;               // K chunk, row-major planes   [2][KC][RLD]
    _Float16* Kt = Kr + 2 * ROW_TILE;             // K chunk, slot-major planes  [2][KC][TLD]
    _Float16* Vr = Kt + 2 * SLOT_TILE;            // V chunk, row-major planes
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    // XCD-aware 1-D grid as in pp_attn.hip: the workgroups of one (image, head) share an XCD (and its L2 copy of K / V)
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3, bx = jj % ntx, pair = (jj / ntx) * 8 + xcd;
    if (pair >= npairs) return;
    const int b = pair / heads, h = pair % heads;
    const int C1 = heads * HD, C3 = 3 * C1;
    const float* base = qkv + (size_t)b * T * C3 + h * HD;
    const int q0 = (bx * (nthr >> 6) + w) * 32, q = q0 + l31, qc = q < T ? q : T - 1;
    const float g = gpair[0], ginv = gpair[1];

    h8 qh[4], ql[4], doh[4], dol[4];
    row_frags(base + (size_t)qc * C3 + 0, lh, 1.f, qh, ql);
    row_frags(dout + ((size_t)b * T + qc) * C1 + h * HD, lh, g, doh, dol);
    const float lse_q = lse2[((size_t)b * heads + h) * T + qc], D_q = Dg[((size_t)b * heads + h) * T + qc];
    const float S_DESCALE = scale * 1.44269504088896340736f / (PP_A_SCALE * PP_A_SCALE);
    constexpr float DP_DESCALE = 1.0f / (PP_A_SCALE * PP_A_SCALE);

    f32x16 dq0, dq1;
#pragma unroll
    for (int e = 0; e < 16; ++e) dq0[e] = dq1[e] = 0.f;
    for (int k0 = 0; k0 < T; k0 += KC) {
        __syncthreads();
        stage32<true, true>(base + C1, C3, k0, T, 1.f, Kr, Kt, tid, nthr);
        stage32<true, false>(base + 2 * C1, C3, k0, T, 1.f, Vr, nullptr, tid, nthr);
        __syncthreads();
        const f32x16 sacc = tile_nt(Kr, l31, lh, qh, ql);       // S^T  (keys x queries), units 16 q.k
        const f32x16 pacc = tile_nt(Vr, l31, lh, doh, dol);     // dP^T (keys x queries), units 16 g dO.v
        f32x16 ds;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int key = k0 + (e & 3) + 8 * (e >> 2) + 4 * lh;
            const float p = key < T ? __builtin_amdgcn_exp2f(sacc[e] * S_DESCALE - lse_q) : 0.f;
            ds[e] = p * (pacc[e] * DP_DESCALE - D_q);
        }
        h8 xh[2], xl[2];
        const float inv = split_ranged(ds, xh, xl);
        f32x16 t0, t1;
#pragma unroll
        for (int e = 0; e < 16; ++e) t0[e] = t1[e] = 0.f;
        tile_tn(Kt, lane, xh, xl, t0, t1);                      // K^T dS^T (d x queries)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            dq0[e] = fmaf(t0[e], inv, dq0[e]);
            dq1[e] = fmaf(t1[e], inv, dq1[e]);
        }
    }
    __syncthreads();
    float* Os = (float*)smem + w * 32 * OLD;
    write_rows(Os, dq0, dq1, scale * ginv / PP_A_SCALE, dqkv + (size_t)b * T * C3 + h * HD, C3, q0, T, lane);
}

// ---- dK, dV: a wave = 32 keys (lane = key column), the workgroup's waves share the staged Q / dO chunks --------------------------------
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_bwd_dkv_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                           const float* __restrict__ lse2, const float* __restrict__ Dg,
                                                           const float* __restrict__ gpair, int T, int heads, float scale,
                                                           float* __restrict__ dqkv, int ntx, int npairs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16* Qr = (_Float16*)smem;               // Q chunk row-major / slot-major, dO chunk row-major / slot-major
    _Float16* Qt = Qr + 2 * ROW_TILE;
    _Float16* Gr = Qt + 2 * SLOT_TILE;
    _Float16* Gt = Gr + 2 * ROW_TILE;
    float* Ls = (float*)(Gt + 2 * SLOT_TILE);     // [KC] lse of the chunk's queries (+inf past T), then [KC] their D
    float* Ds = Ls + KC;
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3, bx = jj % ntx, pair = (jj / ntx) * 8 + xcd;
    if (pair >= npairs) return;
    const int b = pair / heads, h = pair % heads;
    const int C1 = heads * HD, C3 = 3 * C1;
    const float* base = qkv + (size_t)b * T * C3 + h * HD;
    const float* dbase = dout + (size_t)b * T * C1 + h * HD;
    const int key0 = (bx * (nthr >> 6) + w) * 32, key = key0 + l31, kc = key < T ? key : T - 1;
    const float g = gpair[0], ginv = gpair[1];

    h8 kh[4], kl[4], vh[4], vl[4];
    row_frags(base + (size_t)kc * C3 + C1, lh, 1.f, kh, kl);
    row_frags(base + (size_t)kc * C3 + 2 * C1, lh, 1.f, vh, vl);
    const float S_DESCALE = scale * 1.44269504088896340736f / (PP_A_SCALE * PP_A_SCALE);
    constexpr float DP_DESCALE = 1.0f / (PP_A_SCALE * PP_A_SCALE);

    f32x16 dk0, dk1, dv0, dv1;
#pragma unroll
    for (int e = 0; e < 16; ++e) dk0[e] = dk1[e] = dv0[e] = dv1[e] = 0.f;
    for (int q0 = 0; q0 < T; q0 += KC) {
        __syncthreads();
        stage32<true, true>(base, C3, q0, T, 1.f, Qr, Qt, tid, nthr);
        stage32<true, true>(dbase, C1, q0, T, g, Gr, Gt, tid, nthr);
        if (tid < KC) {
            const bool in = q0 + tid < T;
            Ls[tid] = in ? lse2[((size_t)b * heads + h) * T + q0 + tid] : INFINITY;
            Ds[tid] = in ? Dg[((size_t)b * heads + h) * T + q0 + tid] : 0.f;
        }
        __syncthreads();
        const f32x16 sacc = tile_nt(Qr, l31, lh, kh, kl);       // S  (queries x keys)
        const f32x16 pacc = tile_nt(Gr, l31, lh, vh, vl);       // dP (queries x keys)
        f32x16 ds;
        h8 ph[2], pl[2];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int qi = (e & 3) + 8 * (e >> 2) + 4 * lh;
            const float p = __builtin_amdgcn_exp2f(sacc[e] * S_DESCALE - Ls[qi]);   // 0 for the queries past T (lse = +inf)
            ds[e] = p * (pacc[e] * DP_DESCALE - Ds[qi]);
            const float x = fminf(p, 2.f) * P_SCALE;
            const _Float16 hh = (_Float16)x;
            ph[e >> 3][e & 7] = hh;
            pl[e >> 3][e & 7] = (_Float16)(x - (float)hh);
        }
        tile_tn(Gt, lane, ph, pl, dv0, dv1);                    // dO^T P (d x keys), units 4 g 1024
        h8 xh[2], xl[2];
        const float inv = split_ranged(ds, xh, xl);
        f32x16 t0, t1;
#pragma unroll
        for (int e = 0; e < 16; ++e) t0[e] = t1[e] = 0.f;
        tile_tn(Qt, lane, xh, xl, t0, t1);                      // Q^T dS (d x keys)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            dk0[e] = fmaf(t0[e], inv, dk0[e]);
            dk1[e] = fmaf(t1[e], inv, dk1[e]);
        }
    }
    __syncthreads();
    float* Os = (float*)smem + w * 32 * OLD;
    float* drow = dqkv + (size_t)b * T * C3 + h * HD;
    write_rows(Os, dk0, dk1, scale * ginv / PP_A_SCALE, drow + C1, C3, key0, T, lane);
    write_rows(Os, dv0, dv1, ginv / (PP_A_SCALE * P_SCALE), drow + 2 * C1, C3, key0, T, lane);
}

}  // namespace

extern "C" int pp_attention_backward(const float* qkv, const float* out, const float* dout, const float* lse2, const float* gpair, int B,
                                     int T, int heads, int head_dim, float scale, float* Dws, float* dqkv, void* stream) {
    if (!qkv || !out || !dout || !lse2 || !gpair || !Dws || !dqkv || B <= 0 || T <= 0 || heads <= 0 || head_dim != HD) return PP_EINVAL;
    if (((uintptr_t)qkv | (uintptr_t)out | (uintptr_t)dout | (uintptr_t)dqkv) % 16 != 0) return PP_EINVAL;
    if ((long)B * heads * ((T + 63) / 64) > 0x7FFFFFF0L) return PP_EINVAL;
    const long items = (long)B * T * heads;
    hipLaunchKernelGGL(attn_bwd_d_kernel, dim3((unsigned)((items + 15) / 16)), dim3(256), 0, (hipStream_t)stream, out, dout, gpair, B, T,
                       heads, Dws);
    const int tiles = (T + 31) / 32;
    int wpb = 4, best = 1 << 30;
    for (int c = 4; c >= 2; --c) {   // the split of the tiles that wastes the fewest wave slots (pp_attn.hip)
        const int waste = (tiles + c - 1) / c * c - tiles;
        if (waste < best) {
            best = waste;
            wpb = c;
        }
    }
    const size_t os = (size_t)wpb * 32 * OLD * sizeof(float);
    const size_t s_dq = (size_t)(4 * ROW_TILE + 2 * SLOT_TILE) * sizeof(_Float16);
    const size_t s_kv = (size_t)(4 * ROW_TILE + 4 * SLOT_TILE) * sizeof(_Float16) + 2 * KC * sizeof(float);
    const int ntx = (tiles + wpb - 1) / wpb, npairs = B * heads;
    const dim3 grid((unsigned)(ntx * ((npairs + 7) / 8) * 8));
    hipLaunchKernelGGL(attn_bwd_dq_kernel, grid, dim3(64 * wpb), s_dq > os ? s_dq : os, (hipStream_t)stream, qkv, dout, lse2, Dws, gpair, T,
                       heads, scale, dqkv, ntx, npairs);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, grid, dim3(64 * wpb), s_kv > os ? s_kv : os, (hipStream_t)stream, qkv, dout, lse2, Dws, gpair, T,
                       heads, scale, dqkv, ntx, npairs);
    return pp_last_launch();
}
