// Device-side pieces shared by the contraction kernels (pp_gemm.hip: the fp32-operand kernels and the host dispatch;
// pp_gemm_u.hip: the unified pre-split kernel): vector types, the activation functions, the fused epilogue
//   out = residual + residual2 + gamma * act(descale * acc + bias)
// and the LDS tile image every pre-split kernel uses.
#ifndef PP_GEMM_DEV_H
#define PP_GEMM_DEV_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/picopose_hip.h"
#include "pp_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

// erf(z) = z P(z^2) / Q(z^2) on |z| <= 3.925 (clamped beyond: erf = +-1 to fp32 precision), a least-squares
// rational fit (coefficients derived and checked against scipy.special.erf: max |error| 4.2e-7, i.e. GELU within
// 1.5e-6 absolute over |x| <= 10).  13 FMAs + v_rcp_f32 instead of libm's branchy erff (~50 instructions, 15 % of the
// fc1 GEMM of a ViT block).
__device__ __forceinline__ float erf_rational(float z) {
    const float zc = fminf(fmaxf(z, -3.925f), 3.925f), t = zc * zc;
    float p = 2.086927816e-06f, q = 3.855828442e-05f;
    p = fmaf(p, t, 2.864863205e-04f);
    p = fmaf(p, t, 3.736014319e-03f);
    p = fmaf(p, t, 5.266064834e-02f);
    p = fmaf(p, t, 1.894152597e-01f);
    p = fmaf(p, t, 1.128379076e+00f);
    q = fmaf(q, t, 1.159680598e-03f);
    q = fmaf(q, t, 1.490643815e-02f);
    q = fmaf(q, t, 1.137392213e-01f);
    q = fmaf(q, t, 5.011971411e-01f);
    q = fmaf(q, t, 1.0f);
    return zc * p * __builtin_amdgcn_rcpf(q);
}

__device__ __forceinline__ float act_apply(float v, int act) {
    switch (act) {
        case PP_ACT_RELU: return v > 0.f ? v : 0.f;
        case PP_ACT_GELU: return 0.5f * v * (1.0f + erf_rational(v * 0.70710678118654752440f));
        case PP_ACT_LEAKY01: return v > 0.f ? v : 0.1f * v;
        case PP_ACT_TANH: return tanhf(v);
        default: return v;
    }
}

// LDS image of an operand tile in every pre-split kernel: rows of 128 bytes (one K tile), the 16-byte chunk c of row r at
// chunk position c ^ pp_swz_key(r).  With this key the ds_read_b128 of a 16x16x32 MFMA fragment (lane l: row r0 + (l & 15),
// chunk 2 (l >> 4) + term for the hl format, (l >> 4) + 4 step for the h format) is bank-conflict free for EVERY row offset
// r0 (searched exhaustively over the GF(2)-linear keys against the lane groups of ds_read_b128, MI355X_MICROARCH.md "LDS").
__device__ __forceinline__ int pp_swz_key(int r) { return (((r >> 2) & 1) * 3) | (((r >> 1) & 1) << 2); }

// Order in which the persistent kernels walk the output tiles: bands of 4 tile rows, inside a band column groups of
// <= 8 tile columns, inside a group row-major.  The 32 workgroups of an XCD work on 32 consecutive tiles, i.e. on
// ~4 tile rows x 8 tile columns: each A row slice and each B column slice missed in L2 serves 8 resp. 4 tiles (row-major
// order over a wide N would be 1.3 rows x 24 columns: the B operand streams from the Infinity Cache all the time —
// 29 % L2 misses on the fc1 GEMM).  For gx <= 8 this is plain row-major.  A bijection of [0, gx*gy).
__device__ __forceinline__ void pp_tile_rc(int t, int gx, int gy, int& r, int& c) {
    const int ncg = (gx + 7) >> 3, band = 4 * gx;
    const int rg = t / band;
    int u = t - rg * band;
    int br = gy - 4 * rg;
    br = br > 4 ? 4 : br;  // (the last band may be short; the bands before it are full, so rg is right)
    const int wq = gx / ncg, wrem = gx - wq * ncg;  // the first wrem groups have wq + 1 columns
    int c0 = 0;
    r = c = 0;
    for (int g = 0; g < ncg; ++g) {
        const int wg = wq + (g < wrem ? 1 : 0), cnt = br * wg;
        if (u < cnt) {
            r = 4 * rg + u / wg;
            c = c0 + u % wg;
            return;
        }
        u -= cnt;
        c0 += wg;
    }
}

// Operand formats of the pre-split kernels (PpGemmDesc.prec):
//   PP_PREC_F16X3 "hl": fp16 [rows][ld/8][2][8] — per 8 consecutive k the 8 hi terms then the 8 lo terms of 4 x (32 bytes);
//   PP_PREC_F16   "h" : fp16 [rows][ld] = f16(4 x) (16 bytes per 8 k).
// TERMS = 2 / 1.  A 128-byte row segment is one K tile: 32 k (hl) or 64 k (h).

// Store the 8 consecutive output columns n .. n + 7 of output row m held in v (already descaled / biased / activated):
// fp32 (C, with the residuals added) and / or operand form (C_hl, in the engine's current operand format).
template <int TERMS>
__device__ __forceinline__ void epilogue_emit8(const PpGemmDesc& d, f4 (&v)[2], int m, int n) {
    float* C = d.C;
    const float* R = d.residual;
    const float* R2 = d.residual2;
    size_t off, orow = (size_t)m;     // output row / first column of this lane's 8 values (fp32 and operand alike)
    int ocol = n;
    if (d.shuffle_r == 0) {
        off = (size_t)m * d.ldc + n;
    } else {  // ConvTranspose2d(kernel = stride = r): columns n .. n + 7 = channels co .. co + 7 of sub-pixel (dy, dx)
        const int rr_ = d.shuffle_r, cout = d.N / (rr_ * rr_);
        const int sub = n / cout, co = n - sub * cout, dy = sub / rr_, dx = sub - dy * rr_;
        const int per = d.shuffle_h * d.shuffle_w;
        const int b = m / per, rem = m - b * per, y = rem / d.shuffle_w, x = rem - y * d.shuffle_w;
        orow = ((size_t)b * d.shuffle_h * rr_ + y * rr_ + dy) * (d.shuffle_w * rr_) + x * rr_ + dx;
        ocol = co;
        off = orow * d.ldc + co;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (R) v[h] += *(const f4*)(R + off + 4 * h);
        if (R2) v[h] += *(const f4*)(R2 + off + 4 * h);
        if (C) *(f4*)(C + off + 4 * h) = v[h];
    }
    if (d.C_hl) {
        if (TERMS == 2) {
            h8 hh, ll;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float x = v[c >> 2][c & 3];
                _Float16 a, b;
                pp_split_f16(d.c_relu ? fmaxf(x, 0.f) : x, a, b);
                hh[c] = a;
                ll[c] = b;
            }
            _Float16* hp = (_Float16*)d.C_hl + orow * 2 * d.ldc_h + 2 * ocol;
            *(h8*)hp = hh;
            *(h8*)(hp + 8) = ll;
        } else {
            h8 hh;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float x = v[c >> 2][c & 3];
                hh[c] = pp_to_f16(d.c_relu ? fmaxf(x, 0.f) : x);
            }
            *(h8*)((_Float16*)d.C_hl + orow * d.ldc_h + ocol) = hh;
        }
    }
}

// Epilogue of a wave's (16 MI) x (16 NJ) block held as 16x16 MFMA tiles (v_mfma_f32_16x16x32: lane l holds column l & 15,
// rows 4 (l >> 4) + r of a tile in register r).  The block leaves through a wave-private 2 KB LDS patch, one 16-row x
// 32-column slab at a time, so that a lane owns 8 consecutive columns of a row: 32-byte fp32 stores / residual loads and
// one 32-byte (16-byte) group of the operand output, 4 lanes = one 128-byte line per row.  Requires the vector conditions
// (N % 8 == 0, aligned rows); the caller falls back to epilogue_scalar16 otherwise.  Os: 512 floats, private to the wave.
template <int MI, int NJ, int TERMS>
__device__ __forceinline__ void epilogue_wave16(const PpGemmDesc& d, float descale, f32x4 (&acc)[MI][NJ], float* Os, int mw, int nw,
                                                int lane) {
    static_assert(NJ % 2 == 0, "slabs are two 16-column tiles wide");
    const int l15 = lane & 15, lq = lane >> 4;
    const int rr = lane >> 2, c8 = (lane & 3) * 8;             // read side: row of the slab, first of 8 columns
    // patch image: element (row, col) at row * 32 + (((col >> 3) ^ ((row >> 1) & 3)) << 3) + (col & 7): the 16-byte reads of
    // a lane group then spread over all banks
    const int rd = rr * 32 + (((c8 >> 3) ^ ((rr >> 1) & 3)) << 3);
#pragma unroll
    for (int jp = 0; jp < NJ / 2; ++jp) {
        const int n = nw + jp * 32 + c8;
        const bool ncol_ok = n < d.N;                        // N % 8 == 0: a group of 8 is in or out as a whole
        f4 bias[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, gam[2] = {{1.f, 1.f, 1.f, 1.f}, {1.f, 1.f, 1.f, 1.f}};
        if (ncol_ok) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (d.bias) bias[h] = *(const f4*)(d.bias + n + 4 * h);
                if (d.gamma) gam[h] = *(const f4*)(d.gamma + n + 4 * h);
            }
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 4 * lq + r, col = jj * 16 + l15;
                    Os[row * 32 + (((col >> 3) ^ ((row >> 1) & 3)) << 3) + (col & 7)] = acc[mi][2 * jp + jj][r];
                }
            f4 v[2];
            v[0] = *(const f4*)(Os + rd);
            v[1] = *(const f4*)(Os + rd + 4);
            const int m = mw + mi * 16 + rr;
            if (m < d.M && ncol_ok) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[h][c] = act_apply(v[h][c] * descale + bias[h][c], d.act) * gam[h][c];
                epilogue_emit8<TERMS>(d, v, m, n);
            }
        }
    }
}

// element-wise form of the same epilogue (N % 8 != 0, unaligned rows, pixel shuffle with odd channel counts)
template <int MI, int NJ, int TERMS>
__device__ __forceinline__ void epilogue_scalar16(const PpGemmDesc& d, float descale, f32x4 (&acc)[MI][NJ], int mw, int nw, int lane) {
    const int l15 = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = nw + j * 16 + l15;
        if (n >= d.N) continue;
        const float bias = d.bias ? d.bias[n] : 0.f;
        const float gamma = d.gamma ? d.gamma[n] : 1.f;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mw + i * 16 + 4 * lq + r;
                if (m >= d.M) continue;
                float v = act_apply(acc[i][j][r] * descale + bias, d.act) * gamma;
                size_t off, orow = (size_t)m;
                int ocol = n;
                if (d.shuffle_r == 0) {
                    off = (size_t)m * d.ldc + n;
                } else {
                    const int rs = d.shuffle_r, cout = d.N / (rs * rs);
                    const int sub = n / cout, co = n - sub * cout, dy = sub / rs, dx = sub - dy * rs;
                    const int per = d.shuffle_h * d.shuffle_w;
                    const int b = m / per, rem = m - b * per, y = rem / d.shuffle_w, x = rem - y * d.shuffle_w;
                    orow = ((size_t)b * d.shuffle_h * rs + y * rs + dy) * (d.shuffle_w * rs) + x * rs + dx;
                    ocol = co;
                    off = orow * d.ldc + co;
                }
                if (d.residual) v += d.residual[off];
                if (d.residual2) v += d.residual2[off];
                if (d.C) d.C[off] = v;
                if (d.C_hl) {
                    const float x = d.c_relu ? fmaxf(v, 0.f) : v;
                    if (TERMS == 2) {
                        _Float16 h, l;
                        pp_split_f16(x, h, l);
                        _Float16* hp = (_Float16*)d.C_hl + orow * 2 * d.ldc_h + pp_hl_col(ocol, 0);
                        hp[0] = h;
                        hp[8] = l;
                    } else {
                        ((_Float16*)d.C_hl)[orow * d.ldc_h + ocol] = pp_to_f16(x);
                    }
                }
            }
    }
}

__device__ __forceinline__ bool epilogue_vector_ok(const PpGemmDesc& d) {
    const bool shuffle_vec = d.shuffle_r == 0 || ((d.N / (d.shuffle_r * d.shuffle_r)) & 7) == 0;
    return shuffle_vec && (d.N & 7) == 0 && (d.ldc & 3) == 0 && ((uintptr_t)d.C & 15) == 0 && (!d.residual || ((uintptr_t)d.residual & 15) == 0) &&
           (!d.residual2 || ((uintptr_t)d.residual2 & 15) == 0) && (!d.bias || ((uintptr_t)d.bias & 15) == 0) &&
           (!d.gamma || ((uintptr_t)d.gamma & 15) == 0);
}

#endif
