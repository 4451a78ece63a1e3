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
#ifdef PP_STUDY_EXACT_ERF   // (accuracy study builds only: libm's erff in every GELU epilogue — profiles/r05/grad_f64.txt)
    return erff(z);
#endif
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

// alpha of a launch: the host scalar times the optional device scalars (the inverse range scales of backward operands)
__device__ __forceinline__ float pp_alpha(const PpGemmDesc& d) {
    float a = d.alpha;
    if (d.alpha_dev) a *= d.alpha_dev[0];
    if (d.alpha_dev2) a *= d.alpha_dev2[0];
    return a;
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
// (gw tile columns per group, br tile rows per band: 8 / 4 for the pre-split kernels; the fp32 engine passes its own, pp_gemm_f.hip)
__device__ __forceinline__ void pp_tile_rc_g(int t, int gx, int gy, int gw, int br_, int& r, int& c) {
    const int ncg = (gx + gw - 1) / gw, band = br_ * gx;
    const int rg = t / band;
    int u = t - rg * band;
    int br = gy - br_ * rg;
    br = br > br_ ? br_ : br;  // (the last band may be short; the bands before it are full, so rg is right)
    const int wq = gx / ncg, wrem = gx - wq * ncg;  // the first wrem groups have wq + 1 columns
    int c0 = 0;
    r = c = 0;
    for (int g = 0; g < ncg; ++g) {
        const int wg = wq + (g < wrem ? 1 : 0), cnt = br * wg;
        if (u < cnt) {
            r = br_ * rg + u / wg;
            c = c0 + u % wg;
            return;
        }
        u -= cnt;
        c0 += wg;
    }
}
__device__ __forceinline__ void pp_tile_rc(int t, int gx, int gy, int& r, int& c) { pp_tile_rc_g(t, gx, gy, 8, 4, r, c); }

// Operand formats of the pre-split kernels (PpGemmDesc.prec):
//   PP_PREC_F16X3 "hl": fp16 [rows][ld/8][2][8] — per 8 consecutive k the 8 hi terms then the 8 lo terms of 4 x (32 bytes);
//   PP_PREC_F16   "h" : fp16 [rows][ld] = f16(4 x) (16 bytes per 8 k).
// TERMS = 2 / 1.  A 128-byte row segment is one K tile: 32 k (hl) or 64 k (h).

// ---- the vector epilogue --------------------------------------------------------------------------------------------
// Accumulator layout of every pre-split kernel.  The MFMAs are issued TRANSPOSED — weights as the instruction's A matrix,
// activations as its B matrix (v_mfma_f32_16x16x32: D[i][j] with i from A, j from B; lane l holds column j = l & 15 and rows
// i = 4 (l >> 4) + r in register r) — so a lane holds ONE output row m = l & 15 of a 16-row block and, per 16-column weight
// tile, the four weight-tile rows 4 q + r (q = l >> 4).  The weight tile is laid into LDS with its rows permuted inside each
// block of 32 (pp_wperm: LDS row rho <- output column n_off(rho)), so that the registers of the tile PAIR (2 jp, 2 jp + 1)
// are the 8 CONSECUTIVE output columns 32 jp + 8 q + {0 .. 7}: the accumulators leave straight from the registers as 32-byte
// fp32 stores / one operand group per lane, 4 lanes = one 128-byte line of a row — no LDS staging, no transposition pass.
// LDS weight row rho (low five bits t q1 q0 r1 r0) holds output column (rho & ~31) + 8 q + 4 t + r.
__host__ __device__ __forceinline__ int pp_wperm(int rho) {
    return (rho & ~31) | (((rho >> 2) & 3) << 3) | (((rho >> 4) & 1) << 2) | (rho & 3);
}

// All global accesses of the epilogue are raw buffer accesses with 32-bit byte offsets (the host checks the extents,
// pp_gemm_u_vec_ok): a row m >= M or a column group n >= N gets the offset 0xFFFFFFFF — out of range: the store is dropped,
// a load returns zeros — so the row loop has no branches and no 64-bit address arithmetic (measured on the K = 768 ViT linears:
// the epilogue was 24 % of the kernel, almost all of it VALU / scalar-branch issue, 4 % the stores themselves).
// (the pointer goes through v_readfirstlane: left to itself hipcc parked the descriptor of a kernel argument in VGPRs when SGPRs
// ran short and then wrapped EVERY store in a waterfall loop — readfirstlane x 4, compare, saveexec, branch)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pp_rsrc(const void* p) {
    const uint64_t a = (uint64_t)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    const unsigned n = __builtin_amdgcn_readfirstlane(p ? 0xFFFFFFFFu : 0u);
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, (int)n, 0x00020000);
}
__device__ __forceinline__ void pp_bstore(__amdgpu_buffer_rsrc_t r, f4 v, unsigned off) {
#ifdef PP_STUDY_NOSTORE   // (timing study builds only: the epilogue's arithmetic without its stores)
    if (v[0] != -12345.f) return;
#endif
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), r, off, 0, 0);
}
__device__ __forceinline__ f4 pp_bload(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}

// split 8 values (already in the operand's 4x scale, any input ReLU applied) into the operand group(s) and store them:
// hl: [8 hi | 8 lo] at byte offset `off`; h: 8 halfs at `off`
template <int TERMS>
__device__ __forceinline__ void pp_store_operand8(__amdgpu_buffer_rsrc_t H, const f4 (&x4)[2], unsigned off, unsigned long long& sbad) {
    h8 hh, ll;
    // saturation report (pp_common.h): the largest of the 8 magnitudes (v_max3 chain, a short-lived register) against the fp16 range; the
    // verdict is a wave mask in SCALAR registers — a per-lane running maximum kept across the epilogue cost the 256x256 kernels ~90
    // spilled registers (they sit at the 256-VGPR limit)
    {
        float t = fmaxf(fmaxf(fabsf(x4[0][0]), fabsf(x4[0][1])), fabsf(x4[0][2]));
        t = fmaxf(fmaxf(t, fabsf(x4[0][3])), fabsf(x4[1][0]));
        t = fmaxf(fmaxf(t, fabsf(x4[1][1])), fabsf(x4[1][2]));
        t = fmaxf(t, fabsf(x4[1][3]));
        sbad |= __builtin_amdgcn_ballot_w64(!(t < 65504.f));
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float x = x4[c >> 2][c & 3];
        const _Float16 h = (_Float16)fminf(fmaxf(x, -65504.f), 65504.f);
        hh[c] = h;
        if (TERMS == 2) ll[c] = (_Float16)fminf(fmaxf(x - (float)h, -65504.f), 65504.f);
#ifdef PP_STUDY_ACT_LO_ZERO
        if (TERMS == 2) ll[c] = (_Float16)0.f;
#endif
    }
    pp_bstore(H, __builtin_bit_cast(f4, hh), off);
    if (TERMS == 2) pp_bstore(H, __builtin_bit_cast(f4, ll), off == 0xFFFFFFFFu ? off : off + 16);
}

// Epilogue of a wave's (16 MI) x (16 NJ) block: out = residual + residual2 + gamma * act(descale * acc + bias) for the 8
// columns a lane holds per tile pair.  Requires the vector conditions (N % 8 == 0, aligned rows, extents below 4 GB:
// pp_gemm_u_vec_ok on the host); the element-wise form below covers the rest.  Two code paths, chosen once per tile:
//   * operand-only outputs without residual / LayerScale / pixel shuffle (the qkv and fc1 linears, the convolution chains):
//     everything happens in the operand's 4x scale — one fma (descale and bias pre-multiplied), the activation as
//     max(x, slope x) for the positively homogeneous ones (none / ReLU / LeakyReLU: slope 1 / 0 / 0.1) or the rational-erf
//     GELU, the consumer's input ReLU as a max with 0 or -inf, the split, two 16-byte stores;
//   * the general form (fp32 output, residuals, LayerScale, pixel-shuffle stores, optional operand output as well), with the
//     residual rows of the next 16-row block loaded while the current one is processed.
template <int MI, int NJ, int TERMS>
__device__ __forceinline__ void epilogue_wave16(const PpGemmDesc& d, float descale, f32x4 (&acc)[MI][NJ], int mw, int nw, int lane) {
    static_assert(NJ % 2 == 0, "a lane's 8 columns come from a pair of 16-column tiles");
    const int l15 = lane & 15, lq = lane >> 4;
    const __amdgpu_buffer_rsrc_t Hr = pp_rsrc(d.C_hl);
    const unsigned hrow = (unsigned)d.ldc_h * (2u * TERMS);       // operand bytes per output row
    const float slope = d.act == PP_ACT_RELU ? 0.f : (d.act == PP_ACT_LEAKY01 ? 0.1f : 1.f);
    const float hfloor = d.c_relu ? 0.f : -INFINITY;              // the consumer's input ReLU folded into the operand
    const bool lin_act = d.act != PP_ACT_GELU && d.act != PP_ACT_TANH;
    const int m0 = mw + l15;
    // lanes that turned a magnitude beyond the fp16 range into operand terms (rows past M / columns past N are products of zero-filled
    // operand tiles — bias and activation only — and cannot be what saturates)
    unsigned long long sbad = 0ull;
    if (!d.C && !d.residual && !d.residual2 && !d.gamma && d.shuffle_r == 0 && d.act != PP_ACT_TANH) {
        const float ds4 = descale * PP_A_SCALE;
#pragma unroll
        for (int jp = 0; jp < NJ / 2; ++jp) {
            const int n = nw + jp * 32 + 8 * lq;
            const bool ncol = n < d.N;                        // N % 8 == 0: a group of 8 is in or out as a whole
            f4 bias[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            if (d.bias && ncol) {
                bias[0] = *(const f4*)(d.bias + n);
                bias[1] = *(const f4*)(d.bias + n + 4);
            }
            const f4 bias4[2] = {bias[0] * PP_A_SCALE, bias[1] * PP_A_SCALE};
            unsigned off = (unsigned)m0 * hrow + (unsigned)n * (2u * TERMS);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                f4 x[2] = {acc[mi][2 * jp], acc[mi][2 * jp + 1]};
                if (lin_act) {
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            float v = fmaf(x[h][c], ds4, bias4[h][c]);
                            v = fmaxf(v, v * slope);
                            x[h][c] = fmaxf(v, hfloor);
                        }
                } else {
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const float v = fmaf(x[h][c], descale, bias[h][c]);
                            const float g = 0.5f * v * (1.0f + erf_rational(v * 0.70710678118654752440f));
                            x[h][c] = fmaxf(g * PP_A_SCALE, hfloor);
                        }
                }
                const bool ok = ncol && m0 + mi * 16 < d.M;
                pp_store_operand8<TERMS>(Hr, x, ok ? off : 0xFFFFFFFFu, sbad);
                off += 16u * hrow;
            }
        }
        pp_sat_flag(sbad != 0ull && lane == 0);
        return;
    }
    const __amdgpu_buffer_rsrc_t Cr = pp_rsrc(d.C), Rr = pp_rsrc(d.residual), R2r = pp_rsrc(d.residual2);
    const bool hasR = d.residual != nullptr, hasR2 = d.residual2 != nullptr;
    const unsigned crow = (unsigned)d.ldc * 4u;
    // byte offsets of the 8 columns n .. n + 7 of output row m: fp32 image and operand image (pixel-shuffle stores move both)
    auto offsets = [&](int m, int n, unsigned& coff, unsigned& hoff) __attribute__((always_inline)) {
        if (d.shuffle_r == 0) {
            coff = (unsigned)m * crow + (unsigned)n * 4u;
            hoff = (unsigned)m * hrow + (unsigned)n * (2u * TERMS);
        } else {  // ConvTranspose2d(kernel = stride = r): columns n .. n + 7 = channels co .. co + 7 of sub-pixel (dy, dx)
            const int rr_ = d.shuffle_r, cout = d.N / (rr_ * rr_);
            const int sub = n / cout, co = n - sub * cout, dy = sub / rr_, dx = sub - dy * rr_;
            const int per = d.shuffle_h * d.shuffle_w;
            const int b = m / per, rem = m - b * per, y = rem / d.shuffle_w, x = rem - y * d.shuffle_w;
            const unsigned orow = ((unsigned)b * d.shuffle_h * rr_ + y * rr_ + dy) * (d.shuffle_w * rr_) + x * rr_ + dx;
            coff = orow * crow + (unsigned)co * 4u;
            hoff = orow * hrow + (unsigned)co * (2u * TERMS);
        }
    };
#pragma unroll
    for (int jp = 0; jp < NJ / 2; ++jp) {
        const int n = nw + jp * 32 + 8 * lq;
        const bool ncol = n < d.N;
        f4 bias[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, gam[2] = {{1.f, 1.f, 1.f, 1.f}, {1.f, 1.f, 1.f, 1.f}};
        if (ncol) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (d.bias) bias[h] = *(const f4*)(d.bias + n + 4 * h);
                if (d.gamma) gam[h] = *(const f4*)(d.gamma + n + 4 * h);
            }
        }
        unsigned coff, hoff;
        offsets(m0, n, coff, hoff);
        bool ok = ncol && m0 < d.M;
        f4 r[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, r2[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        if (hasR) {
            r[0] = pp_bload(Rr, ok ? coff : 0xFFFFFFFFu);
            r[1] = pp_bload(Rr, ok ? coff + 16 : 0xFFFFFFFFu);
        }
        if (hasR2) {
            r2[0] = pp_bload(R2r, ok ? coff : 0xFFFFFFFFu);
            r2[1] = pp_bload(R2r, ok ? coff + 16 : 0xFFFFFFFFu);
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            // the next block's residual rows are on their way while this one is processed
            unsigned ncoff = 0xFFFFFFFFu, nhoff = 0xFFFFFFFFu;
            bool nok = false;
            f4 rn[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, r2n[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            if (mi + 1 < MI) {
                offsets(m0 + (mi + 1) * 16, n, ncoff, nhoff);
                nok = ncol && m0 + (mi + 1) * 16 < d.M;
                if (hasR) {
                    rn[0] = pp_bload(Rr, nok ? ncoff : 0xFFFFFFFFu);
                    rn[1] = pp_bload(Rr, nok ? ncoff + 16 : 0xFFFFFFFFu);
                }
                if (hasR2) {
                    r2n[0] = pp_bload(R2r, nok ? ncoff : 0xFFFFFFFFu);
                    r2n[1] = pp_bload(R2r, nok ? ncoff + 16 : 0xFFFFFFFFu);
                }
            }
            f4 v[2] = {acc[mi][2 * jp], acc[mi][2 * jp + 1]};
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float t = fmaf(v[h][c], descale, bias[h][c]);
                    t = lin_act ? fmaxf(t, t * slope) : act_apply(t, d.act);
                    v[h][c] = fmaf(t, gam[h][c], r[h][c] + r2[h][c]);
                }
            if (d.C) {
                pp_bstore(Cr, v[0], ok ? coff : 0xFFFFFFFFu);
                pp_bstore(Cr, v[1], ok ? coff + 16 : 0xFFFFFFFFu);
            }
            if (d.C_hl) {
                f4 x[2];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int c = 0; c < 4; ++c) x[h][c] = fmaxf(v[h][c] * PP_A_SCALE, hfloor);
                pp_store_operand8<TERMS>(Hr, x, ok ? hoff : 0xFFFFFFFFu, sbad);
            }
            coff = ncoff;
            hoff = nhoff;
            ok = nok;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                r[h] = rn[h];
                r2[h] = r2n[h];
            }
        }
    }
    if (d.C_hl) pp_sat_flag(sbad != 0ull && lane == 0);
}

// element-wise form of the same epilogue (N % 8 != 0, unaligned rows, pixel shuffle with odd channel counts)
template <int MI, int NJ, int TERMS>
__device__ __forceinline__ void epilogue_scalar16(const PpGemmDesc& d, float descale, f32x4 (&acc)[MI][NJ], int mw, int nw, int lane) {
    const int l15 = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = nw + (j >> 1) * 32 + 8 * lq + 4 * (j & 1) + r;
            if (n >= d.N) continue;
            const float bias = d.bias ? d.bias[n] : 0.f;
            const float gamma = d.gamma ? d.gamma[n] : 1.f;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int m = mw + i * 16 + l15;
                if (m >= d.M) continue;
                float v = act_apply(fmaf(acc[i][j][r], descale, bias), d.act) * gamma;
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
                        pp_split_f16_chk(x, h, l);
                        _Float16* hp = (_Float16*)d.C_hl + orow * 2 * d.ldc_h + pp_hl_col(ocol, 0);
                        hp[0] = h;
                        hp[8] = l;
                    } else {
                        ((_Float16*)d.C_hl)[orow * d.ldc_h + ocol] = pp_to_f16_chk(x);
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
