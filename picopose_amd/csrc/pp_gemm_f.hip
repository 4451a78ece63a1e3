// The fp32 contraction kernel of the network engine (PpGemmDesc.prec == PP_PREC_F32, bench.py --mode exact): the reference's own
// arithmetic — every product and every accumulation in fp32 (v_mfma_f32_32x32x2_f32 is bit for bit a k-ordered fmaf chain) — on
// the same structure as the pre-split kernels of pp_gemm_u_kernel.h:
//
//   C[m, n] = residual + residual2 + gamma * act( alpha * sum_k A(m, k) * B(n, k) + bias ),
//   A: dense fp32 rows or an NHWC fp32 image through an implicit im2col;  B: fp32 [N][ldb]
//
//   * operand tiles go global -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`): a K tile is 32 k = ONE 128-byte segment per row,
//     no staging registers, no ds_write; padded taps and the M / N / K tails are out-of-range buffer offsets (zeros, no traffic);
//   * LDS image: rows of 128 bytes; the 16-byte chunk c of row r sits at position c ^ key(r), key(r) = (r >> 3 & 3) << 1 |
//     (r >> 1 & 1) — conflict-free ds_read_b128 for the 32x32x2 fragment pattern (lane l: row l & 31, chunk 4 (l >> 5) + q), the
//     lane groups of ds_read_b128 taken from MI355X_MICROARCH.md "LDS"; with 4 or 8 waves a lane's key is the same for every
//     DMA piece, so it is applied once to the per-lane SOURCE address;
//   * ring of two stages, one `s_waitcnt vmcnt(0)` + s_barrier per K tile, placed before the last quarter of the tile's MFMAs;
//     the DMA pieces of K tile kt + 2 and the first fragments of K tile kt + 1 are issued between those MFMAs;
//   * a K tile is walked in four "quads" of 8 k: lanes 0-31 hold k = 4 q + e, lanes 32-63 k = 16 + 4 q + e (e = 0 .. 3 of one
//     ds_read_b128), the pair one MFMA consumes — the accumulation order of the round-1 gemm_kernel, which this kernel
//     replaces on every aligned shape; fragments of quad q + 1 load while quad q multiplies;
//   * fp32 MFMA runs at 1/16 of the fp16 rate: a 256x256 K tile is 16 384 matrix cycles per CU for 64 KB of operands
//     (4 B/clk/CU), so the K loop is bound by the matrix pipe alone and the tile shape only decides the L2 traffic
//     (256x256: 64 flop per operand byte);
//   * persistent: a workgroup walks XCD-contiguous chunks of the tile list (pp_tile_rc), the DMA stream runs ahead across
//     tile boundaries;
//   * the accumulators leave in the MFMA's natural layout: lane = output column, so every store instruction writes two full
//     128-byte lines; residual rows of the next 32-row block load while the current one is processed.
// EVERY tile configuration accumulates an output element in the same order (K tiles in K order — channel-slice-major for
// convolutions with Cin % 32 == 0, natural otherwise — and quads / pairs as above), so the value of an output element does
// not depend on the configuration the autotuner picks (tests/test_engine_gpu.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "pp_gemm_dev.h"
#include "pp_gemm_u.h"

typedef __attribute__((address_space(3))) void* lds_ptr_f;
// cache policy of the result stores (buffer instruction aux bits: 2 = nt).  Measured with nt: FETCH_SIZE of the ViT linears -2 %, time
// unchanged (gpurun_out/nt_study.txt) — a band's 4 MB of results are not what evicts the shared operand lines; left at the default.
#ifndef PP_F_STORE_AUX
#define PP_F_STORE_AUX 0
#endif

template <int BM_, int BN_, int WM_, int WN_, int OCC_>
struct FTile {
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, OCC = OCC_;
    static constexpr int NW = WM_ * WN_;
    static constexpr int TM = BM_ / WM_, TN = BN_ / WN_;     // wave block
    static constexpr int MI = TM / 32, NJ = TN / 32;         // 32x32 MFMA tiles per wave block
    static constexpr int PA = BM_ / 8 / NW, PB = BN_ / 8 / NW;   // LDS-DMA pieces (8 rows each) per wave and K tile
    static constexpr int A_F = BM_ * 32, B_F = BN_ * 32;     // floats per operand per stage (128-byte rows)
    static constexpr int STAGE = A_F + B_F;
    static constexpr int LDS_BYTES = 2 * STAGE * 4;
    // the walk over the output tiles (pp_tile_rc_g): bands of BR tile rows, inside a band groups of GW tile columns.  An XCD has
    // 32 CUs x OCC tiles in flight = BR x GW; the weights (7 - 13 MB per ViT linear in fp32) do not fit its 4 MB L2 and are
    // streamed from the Infinity Cache once per BAND, so bands are 1024 rows whatever the tile (128-row tiles with 4-row bands
    // re-read them 80 times over M = 41 120: 2.3 x the algorithmic bytes of those launches, profiles/r05/exact)
    static constexpr int BR = 1024 / BM_, GW = 32 * OCC_ / BR;
    static_assert(NW % 4 == 0, "the swizzle key of a DMA lane must not depend on the piece");
    static_assert(BM_ % (8 * NW) == 0 && BN_ % (8 * NW) == 0, "DMA pieces are 8 rows per wave instruction");
    static_assert(PA + PB <= 16, "pieces are spread over the 4 MI NJ MFMAs of one quad");
};

__device__ __forceinline__ int pp_fkey(int r) { return (((r >> 3) & 3) << 1) | ((r >> 1) & 1); }

// ---- epilogue of a wave's (32 MI) x (32 NJ) block in the natural accumulator layout: lane l holds output column l & 31 and, in
// register e, row (e & 3) + 8 (e >> 2) + 4 (l >> 5) of each 32-row block.  All global accesses are raw buffer accesses with
// 32-bit byte offsets (extents checked on the host, pp_gemm_f_ok); an element outside the matrix gets offset 0xFFFFFFFF.
// SHUF: pixel-shuffle store of a ConvTranspose2d(kernel = stride); GELU: the erf GELU, otherwise none / ReLU / LeakyReLU(0.1) as
// max(v, slope v) (tanh layers stay on the round-1 kernel: pp_gemm_f_ok).  The variants are chosen once per tile by uniform
// branches — one body with every case inside is 30 000 instructions per kernel, far beyond the instruction cache.
// A buffer descriptor over all of memory at p, rebuilt where it is used: its four words are pinned to SGPRs by an opaque asm, so
// hipcc can neither hoist it out of the unrolled row loop nor park it in VGPRs (which wraps every access in a waterfall loop:
// 1 166 of them in the first build of this epilogue).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pp_rsrc_pin(const void* p) {
    const uint64_t a = (uint64_t)p;
    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    unsigned n = __builtin_amdgcn_readfirstlane(p ? 0xFFFFFFFFu : 0u);
    asm volatile("" : "+s"(lo), "+s"(hi), "+s"(n));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, (int)n, 0x00020000);
}

template <int MI, int NJ, bool SHUF, bool GELU>
__device__ __forceinline__ void epilogue_f32_impl(const PpGemmDesc& d, float alpha, f32x16 (&acc)[MI][NJ], int mw, int nw, int lane) {
    const int l31 = lane & 31, lh = lane >> 5;
    const bool hasR = d.residual != nullptr, hasR2 = d.residual2 != nullptr;
    const unsigned crow = (unsigned)d.ldc * 4u;
    float bias[NJ], gam[NJ];
    unsigned colb[NJ];   // byte offset of the lane's column inside an output row (SHUF: + the sub-pixel's row offset)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = nw + j * 32 + l31;
        const bool ncol = n < d.N;
        bias[j] = (d.bias && ncol) ? d.bias[n] : 0.f;
        gam[j] = (d.gamma && ncol) ? d.gamma[n] : 1.f;
        if (!SHUF) colb[j] = ncol ? (unsigned)n * 4u : 0xFFFFFFFFu;
        else {   // ConvTranspose2d(kernel = stride = r): column n = (dy r + dx) Cout + co -> sub-pixel (dy, dx), channel co
            const int r = d.shuffle_r, cout = d.N / (r * r);
            const int sub = n / cout, co = n - sub * cout, dy = sub / r, dx = sub - dy * r;
            colb[j] = ncol ? (unsigned)(dy * d.shuffle_w * r + dx) * crow + (unsigned)co * 4u : 0xFFFFFFFFu;
        }
    }
    const int mlane = mw + 4 * lh;     // the lane's first row; its row "off" of a 32-row block is mlane + 32 i + off
    // byte offset of output row m (SHUF: of the r x r block's first pixel); out of range past M
    auto rowb = [&](int m) __attribute__((always_inline)) -> unsigned {
        unsigned v;
        if (!SHUF) v = (unsigned)m * crow;
        else {
            const int r = d.shuffle_r, per = d.shuffle_h * d.shuffle_w;
            const int b = m / per, rem = m - b * per, y = rem / d.shuffle_w, x = rem - y * d.shuffle_w;
            v = (((unsigned)b * d.shuffle_h * r + y * r) * (unsigned)(d.shuffle_w * r) + x * r) * crow;
        }
        return m < d.M ? v : 0xFFFFFFFFu;
    };
    auto off_of = [&](unsigned rb, unsigned cb) __attribute__((always_inline)) -> unsigned {
        return ((rb & cb) == 0xFFFFFFFFu || rb == 0xFFFFFFFFu || cb == 0xFFFFFFFFu) ? 0xFFFFFFFFu : rb + cb;
    };
    const float slope = d.act == PP_ACT_RELU ? 0.f : (d.act == PP_ACT_LEAKY01 ? 0.1f : 1.f);
    // residual rows of the next half block (8 rows per lane) are on their way while the current one is processed
    constexpr int NH = 2 * MI;
    float res[8][NJ];
    auto load_res = [&](int h) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t Rr = pp_rsrc_pin(d.residual);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const unsigned rb = rowb(mlane + (h >> 1) * 32 + (e & 3) + 8 * (2 * (h & 1) + (e >> 2)));
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                res[e][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(Rr, off_of(rb, colb[j]), 0, 0));
        }
    };
    if (hasR) load_res(0);
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        const int i = h >> 1;
        float rc[8][NJ];
        if (hasR) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
#pragma unroll
                for (int j = 0; j < NJ; ++j) rc[e][j] = res[e][j];
            if (h + 1 < NH) load_res(h + 1);
        }
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
            const int g = 2 * (h & 1) + gg;
            const __amdgpu_buffer_rsrc_t Cr = pp_rsrc_pin(d.C), R2r = pp_rsrc_pin(d.residual2);
            unsigned off[4][NJ];
            float r2[4][NJ];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const unsigned rb = rowb(mlane + i * 32 + t + 8 * g);
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    off[t][j] = off_of(rb, colb[j]);
                    r2[t][j] = 0.f;
                }
            }
            if (hasR2) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) r2[t][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(R2r, off[t][j], 0, 0));
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    float v = fmaf(acc[i][j][4 * g + t], alpha, bias[j]);
                    if (GELU) v = 0.5f * v * (1.0f + erf_rational(v * 0.70710678118654752440f));
                    else v = fmaxf(v, v * slope);
                    v = hasR ? fmaf(v, gam[j], rc[4 * gg + t][j]) : v * gam[j];
                    v += r2[t][j];
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), Cr, off[t][j], 0, 0);
                }
        }
    }
}

// The epilogue of every layer without a pixel-shuffle store.  fp32 MFMA and the VALU share their issue slots (tools/mfma_f32_probe:
// one extra VALU instruction per MFMA costs 8 % of the matrix rate), so nothing overlaps an epilogue and its length is VALU count.
// All addressing is SCALAR here: a store / load instruction serves ONE output row per lane half (rows r and r + 4 of a 32-row
// block, 2 x 128-byte lines); its descriptor is rebuilt per row — base = C + row * pitch, num_records = the bytes of the (at most
// 8) valid rows from there on — so rows past M are out of range by the hardware's check, and a lane's offset is the same for all
// of its rows: 4 rows of pitch for the upper lane half + the column (columns past N: + 2^31, out of every range).  Per element
// 3 - 5 vector instructions (fma, activation, LayerScale / residual) + the store.
template <int MI, int NJ, bool GELU>
__device__ __forceinline__ void epilogue_f32_rows(const PpGemmDesc& d, float alpha, f32x16 (&acc)[MI][NJ], int mw, int nw, int lane) {
    const int l31 = lane & 31, lh = lane >> 5;
    const bool hasR = d.residual != nullptr, hasR2 = d.residual2 != nullptr;
    const unsigned crow = (unsigned)d.ldc * 4u;
    float bias[NJ], gam[NJ];
    unsigned colv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = nw + j * 32 + l31;
        const bool ncol = n < d.N;
        bias[j] = (d.bias && ncol) ? d.bias[n] : 0.f;
        gam[j] = (d.gamma && ncol) ? d.gamma[n] : 1.f;
        colv[j] = (ncol ? (unsigned)n * 4u : 0x80000000u) + (unsigned)lh * 4u * crow;
    }
    const float slope = d.act == PP_ACT_RELU ? 0.f : (d.act == PP_ACT_LEAKY01 ? 0.1f : 1.f);
    const int rows_left = d.M - mw;           // wave-uniform
    auto row_rsrc = [&](const float* p, int r) __attribute__((always_inline)) -> __amdgpu_buffer_rsrc_t {
        int nr = rows_left - r;
        nr = nr < 0 ? 0 : (nr > 8 ? 8 : nr);
        const char* base = (const char*)p + (size_t)(unsigned)(mw + r) * crow;
        return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, p ? (int)((unsigned)nr * crow) : 0, 0x00020000);
    };
    // residual rows of the next half block (8 row instructions) are on their way while the current one is processed
    constexpr int NH = 2 * MI;
    float res[8][NJ];
    auto load_res = [&](int h) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const __amdgpu_buffer_rsrc_t Rr = row_rsrc(d.residual, (h >> 1) * 32 + (e & 3) + 8 * (2 * (h & 1) + (e >> 2)));
#pragma unroll
            for (int j = 0; j < NJ; ++j) res[e][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(Rr, colv[j], 0, 0));
        }
    };
    if (hasR) load_res(0);
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        const int i = h >> 1;
        float rc[8][NJ];     // (zeros without a residual: fma(v, gamma, 0) and + 0 are exact — no selects per element)
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int j = 0; j < NJ; ++j) rc[e][j] = hasR ? res[e][j] : 0.f;
        if (hasR && h + 1 < NH) load_res(h + 1);
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
            const int g = 2 * (h & 1) + gg;
            float r2[4][NJ];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < NJ; ++j) r2[t][j] = 0.f;
            if (hasR2) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const __amdgpu_buffer_rsrc_t R2r = row_rsrc(d.residual2, i * 32 + t + 8 * g);
#pragma unroll
                    for (int j = 0; j < NJ; ++j) r2[t][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(R2r, colv[j], 0, 0));
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const __amdgpu_buffer_rsrc_t Cr = row_rsrc(d.C, i * 32 + t + 8 * g);
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    float v = fmaf(acc[i][j][4 * g + t], alpha, bias[j]);
                    if (GELU) v = 0.5f * v * (1.0f + erf_rational(v * 0.70710678118654752440f));
                    else v = fmaxf(v, v * slope);
                    v = fmaf(v, gam[j], rc[4 * gg + t][j]) + r2[t][j];
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), Cr, colv[j], 0, PP_F_STORE_AUX);
                }
            }
        }
    }
}

template <int MI, int NJ>
__device__ __forceinline__ void epilogue_f32(const PpGemmDesc& d, float alpha, f32x16 (&acc)[MI][NJ], int mw, int nw, int lane) {
    if (d.shuffle_r != 0) epilogue_f32_impl<MI, NJ, true, false>(d, alpha, acc, mw, nw, lane);
    else if (d.act == PP_ACT_GELU) epilogue_f32_rows<MI, NJ, true>(d, alpha, acc, mw, nw, lane);
    else epilogue_f32_rows<MI, NJ, false>(d, alpha, acc, mw, nw, lane);
}

// MODE 0: dense A; 1: convolution, channel-slice-major K order (Cin % 32 == 0, <= 32 taps); 2: convolution, natural K order
// (any Cin % 4 == 0: the 4 k of a lane's chunk share a tap)
template <class T, int MODE>
__global__ __launch_bounds__(T::NW * 64, T::OCC) void pp_gemm_f_kernel(const PpGemmDesc d, int gx, int gy, int walk) {
#if defined(__HIP_DEVICE_COMPILE__)  // the host pass only needs the launch stub (it cannot instantiate the LDS-DMA builtins)
    constexpr bool DENSE = MODE == 0;
    constexpr int NW = T::NW, PA = T::PA, PB = T::PB, MI = T::MI, NJ = T::NJ, NP = PA + PB;
    constexpr int KT = 32, EB = 4;
    constexpr int STAGE = T::STAGE, A_F = T::A_F;
    extern __shared__ __attribute__((aligned(16))) float flds[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w / T::WN, wc = w % T::WN, l31 = lane & 31, lh = lane >> 5;
    // tiles of this workgroup: XCD x = id % 8 owns a contiguous chunk of the tile list; its workgroups interleave over it
    const int ntiles = gx * gy, nxw = (int)gridDim.x >> 3;  // gridDim.x is a multiple of 8
    const int xcd = blockIdx.x & 7, q8 = ntiles >> 3, r8 = ntiles & 7;
    const int chunk0 = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int chunk1 = chunk0 + (xcd < r8 ? q8 + 1 : q8);
    const int first = chunk0 + (int)(blockIdx.x >> 3);
    if (first >= chunk1) return;
    const __amdgpu_buffer_rsrc_t Ar = __builtin_amdgcn_make_buffer_rsrc((void*)d.A, 0, (int)d.a_hl_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t Br = __builtin_amdgcn_make_buffer_rsrc((void*)d.B, 0, (int)d.b_hl_bytes, 0x00020000);
    // DMA slot of this lane: piece q of wave w fills LDS rows (q NW + w) 8 + (lane >> 3); LDS chunk position lane & 7 of such a
    // row holds source chunk sc (the key reads bits 3, 4 of the row = w & 3 for every piece, and bit 1 = bit 1 of lane >> 3)
    const int lr = lane >> 3;
    const int sc = (lane & 7) ^ (((w & 3) << 1) | ((lr >> 1) & 1));
    const int kch = sc * 4;              // first k, inside a K tile, of this lane's chunk
    const unsigned cbyte = (unsigned)sc * 16u;
    const int ntaps = d.conv_kh * d.conv_kw;
    const int nk = MODE == 1 ? ntaps * (d.conv_cin / KT) : (d.K + KT - 1) / KT;
    const long long abias = MODE == 1 ? ((long long)d.conv_pad * d.conv_w + d.conv_pad) * d.lda : 0;   // elements
    const bool ktail = DENSE && d.K % KT != 0;
    unsigned tmask = 0u;   // MODE 0, K % 32 != 0: all-ones in the lanes whose chunk of the current K tile lies past K

    // ---- fetch side: addressing state of the tile the DMA stream is in (the scheme of pp_gemm_u_kernel.h with 4-byte elements)
    unsigned abyte[PA], amask[PA], bbyte[PB];
    int aoy[MODE == 2 ? PA : 1], aox[MODE == 2 ? PA : 1];
    int ftile = first, fkt = 0;
    int ctap = 0, cky = 0, ckx = 0, cci = 0;     // MODE 1 (wave-uniform): tap / channel slice of the next K tile
    int tky = 0, tkx = 0, tci = 0;               // MODE 2 (per lane): tap / channel of k = fkt KT + kch
#define PP_F_SETUP(TILE)                                                                                             \
    {                                                                                                                \
        int tr_, tc_;                                                                                                \
        pp_tile_rc_g((TILE), gx, gy, walk & 255, walk >> 8, tr_, tc_);                                                                        \
        const int m0_ = tr_ * T::BM, n0_ = tc_ * T::BN;                                                              \
        _Pragma("unroll") for (int j = 0; j < PA; ++j) {                                                             \
            const int m = m0_ + (j * NW + w) * 8 + lr;                                                               \
            const bool ok = m < d.M;                                                                                 \
            long long base = ok ? (long long)m * d.lda : 0;                                                          \
            unsigned mask = ok ? 1u : 0u;                                                                            \
            if (!DENSE) {                                                                                            \
                mask = 0u;                                                                                           \
                int oy = 0, ox = 0;                                                                                  \
                if (ok) {                                                                                            \
                    const int per = d.conv_ho * d.conv_wo;                                                           \
                    const int bi = m / per, r = m - bi * per;                                                        \
                    oy = (r / d.conv_wo) * d.conv_stride - d.conv_pad;                                               \
                    ox = (r % d.conv_wo) * d.conv_stride - d.conv_pad;                                               \
                    base = (long long)bi * d.conv_bstride + ((long long)oy * d.conv_w + ox) * d.lda;                 \
                    if (MODE == 1) {                                                                                 \
                        for (int t = 0; t < ntaps; ++t) {                                                            \
                            const int iy = oy + t / d.conv_kw, ix = ox + t % d.conv_kw;                              \
                            if (iy >= 0 && iy < d.conv_h && ix >= 0 && ix < d.conv_w) mask |= 1u << t;               \
                        }                                                                                            \
                    } else {                                                                                         \
                        mask = 1u;                                                                                   \
                    }                                                                                                \
                }                                                                                                    \
                if (MODE == 2) {                                                                                     \
                    aoy[MODE == 2 ? j : 0] = oy;                                                                     \
                    aox[MODE == 2 ? j : 0] = ox;                                                                     \
                }                                                                                                    \
            }                                                                                                        \
            abyte[j] = DENSE ? (ok ? (unsigned)(base * EB) + cbyte : 0xFFFFFFFFu)                                    \
                             : (unsigned)((base + (MODE == 1 ? abias : 0)) * EB) + cbyte;                            \
            amask[j] = MODE == 1 ? ~mask : mask;                                                                     \
        }                                                                                                            \
        /* grouped launch (pp_gemm: a batch as extra row tiles): this row tile reads the weights of ITS group */     \
        const unsigned gofs_ = (DENSE && d.grp_rows != 0) ? (unsigned)(m0_ / d.grp_rows) * (unsigned)d.grp_b_bytes : 0u;   \
        _Pragma("unroll") for (int j = 0; j < PB; ++j) {                                                             \
            const int nb = n0_ + (j * NW + w) * 8 + lr;                                                              \
            bbyte[j] = nb < d.N ? (unsigned)((long long)nb * d.ldb * EB) + cbyte + gofs_ : 0xFFFFFFFFu;              \
        }                                                                                                            \
        fkt = 0;                                                                                                     \
        ctap = cky = ckx = cci = 0;                                                                                  \
        if (MODE == 2) {                                                                                             \
            const int tap = kch / d.conv_cin;                                                                        \
            tci = kch - tap * d.conv_cin;                                                                            \
            tky = tap / d.conv_kw;                                                                                   \
            tkx = tap - tky * d.conv_kw;                                                                             \
        }                                                                                                            \
    }
#define PP_F_NEXT_TILE_IF_DONE()                                \
    if (fkt == nk && ftile < chunk1) {                          \
        ftile += nxw;                                           \
        if (ftile < chunk1) PP_F_SETUP(ftile) else fkt = 0;     \
    }
    auto off_a2 = [&](int j) __attribute__((always_inline)) -> unsigned {   // MODE 2: everything per lane
        const unsigned live = ftile < chunk1 ? 1u : 0u;
        const int iy = aoy[MODE == 2 ? j : 0] + tky, ix = aox[MODE == 2 ? j : 0] + tkx;
        const unsigned ok = amask[j] & live & (fkt * KT + kch < d.K ? 1u : 0u) & (iy >= 0 ? 1u : 0u) & (iy < d.conv_h ? 1u : 0u) &
                            (ix >= 0 ? 1u : 0u) & (ix < d.conv_w ? 1u : 0u);
        return (abyte[j] + (unsigned)(((tky * d.conv_w + tkx) * d.lda + tci) * EB) - (unsigned)(kch * EB)) | (ok - 1u);
    };
    auto off_b2 = [&](int j) __attribute__((always_inline)) -> unsigned {
        const unsigned live = (ftile < chunk1 ? 1u : 0u) & (bbyte[j] != 0xFFFFFFFFu ? 1u : 0u);
        return (bbyte[j] + (unsigned)(fkt * 128)) | ((live & (fkt * KT + kch < d.K ? 1u : 0u)) - 1u);
    };
    auto dma_a = [&](int stage, int j) __attribute__((always_inline)) {
#ifdef PP_STUDY_F_NODMA   // (timing study builds only: the K loop without its operand traffic)
        if (d.M > 0) return;
#endif
        const lds_ptr_f dst = (lds_ptr_f)(flds + stage * STAGE + ((j * NW + w) * 8) * 32);
        if (MODE == 2) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(Ar, dst, 16, off_a2(j), 0, 0, 0);
        } else {
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
                (void*)((const char*)d.A - abias * EB), 0, ftile < chunk1 ? (int)(d.a_hl_bytes + abias * EB) : 0, 0x00020000);
            const unsigned v = DENSE ? abyte[j] | tmask : abyte[j] | (unsigned)__builtin_amdgcn_sbfe((int)amask[j], (unsigned)ctap, 1u);
            const int so = DENSE ? fkt * 128 : ((cky * d.conv_w + ckx) * d.lda + cci) * EB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 16, v, so, 0, 0);
        }
    };
    auto dma_b = [&](int stage, int j) __attribute__((always_inline)) {
#ifdef PP_STUDY_F_NODMA
        if (d.M > 0) return;
#endif
        const lds_ptr_f dst = (lds_ptr_f)(flds + stage * STAGE + A_F + ((j * NW + w) * 8) * 32);
        if (MODE == 2) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(Br, dst, 16, off_b2(j), 0, 0, 0);
        } else {
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)d.B, 0, ftile < chunk1 ? (int)d.b_hl_bytes : 0, 0x00020000);
            const unsigned v = DENSE ? bbyte[j] | tmask : bbyte[j];
            const int so = DENSE ? fkt * 128 : (ctap * d.conv_cin + cci) * EB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 16, v, so, 0, 0);
        }
    };
    auto piece = [&](int stage, int q) __attribute__((always_inline)) {
        if (q < PA) dma_a(stage, q);
        else dma_b(stage, q - PA);
    };
#define PP_F_KTILE()                                                               \
    {                                                                              \
        PP_F_NEXT_TILE_IF_DONE()                                                   \
        if (ktail) tmask = fkt * KT + kch < d.K ? 0u : 0xFFFFFFFFu;                \
    }
#define PP_F_ADVANCE() /* after the pieces of a K tile */                                           \
    {                                                                                              \
        if (MODE == 1) {                                                                           \
            const bool row_end = ckx + 1 == d.conv_kw, tap_end = row_end && cky + 1 == d.conv_kh;  \
            ckx = row_end ? 0 : ckx + 1;                                                           \
            cky = tap_end ? 0 : (row_end ? cky + 1 : cky);                                         \
            ctap = tap_end ? 0 : ctap + 1;                                                         \
            cci = tap_end ? cci + KT : cci;                                                        \
        }                                                                                          \
        if (MODE == 2) {                                                                           \
            tci += KT;                                                                             \
            while (tci >= d.conv_cin) {                                                            \
                tci -= d.conv_cin;                                                                 \
                if (++tkx == d.conv_kw) {                                                          \
                    tkx = 0;                                                                       \
                    ++tky;                                                                         \
                }                                                                                  \
            }                                                                                      \
        }                                                                                          \
        ++fkt;                                                                                     \
    }
#define PP_F_FETCH(STAGE_)                                               \
    {                                                                    \
        PP_F_KTILE()                                                     \
        _Pragma("unroll") for (int j = 0; j < PA; ++j) dma_a(STAGE_, j); \
        _Pragma("unroll") for (int j = 0; j < PB; ++j) dma_b(STAGE_, j); \
        PP_F_ADVANCE()                                                   \
    }

    // ---- compute side
    // fragment of a 32-row block for quad q: lane l reads row l & 31, chunk 4 (l >> 5) + q (4 consecutive k)
    const int fkey = pp_fkey(l31);
    int fo[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) fo[q] = l31 * 32 + (((lh << 2) | q) ^ fkey) * 4;
    const bool relu_in = d.relu_in != 0;   // ReLU on the A operand (ResidualConvUnit, dpt.py:82-86), applied to the fragments
    struct FS {
        f4 a[MI], b[NJ];
    };
    f32x16 acc[MI][NJ];
    auto load_q = [&](FS& f, int stage, int q) __attribute__((always_inline)) {
#ifdef PP_STUDY_F_NOLDS   // (timing study builds only: the K loop without its fragment reads)
        if (d.M > 0) return;
#endif
        const float* sa = flds + stage * STAGE + (wr * T::TM) * 32 + fo[q];
        const float* sb = flds + stage * STAGE + A_F + (wc * T::TN) * 32 + fo[q];
#pragma unroll
        for (int i = 0; i < MI; ++i) f.a[i] = *(const f4*)(sa + i * 32 * 32);
#pragma unroll
        for (int j = 0; j < NJ; ++j) f.b[j] = *(const f4*)(sb + j * 32 * 32);
    };
    // (a uniform branch and ONE v_max per element: vector instructions take their issue cycles from the fp32 MFMAs)
    auto relu_a = [&](FS& f) __attribute__((always_inline)) {
        if (relu_in) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) asm volatile("v_max_f32 %0, 0, %0" : "+v"(f.a[i][e]));
        }
    };
    auto mma_q = [&](const FS& f) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i][e], f.b[j][e], acc[i][j], 0, 0, 0);
    };

    // one quad's MFMAs with (issue: a wave-uniform flag — ONE path for the accumulators, small scalar branches around the pieces) the NP
    // DMA pieces of the fetch cursor's K tile spread between them, into `stage`
    auto mma_q_issue = [&](const FS& f, int stage, bool issue) __attribute__((always_inline)) {
        constexpr int NM = 4 * MI * NJ;
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int s_ = (e * MI + i) * NJ + j;
                    if ((s_ * NP) / NM != ((s_ + 1) * NP) / NM) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (issue) {
#pragma unroll
                            for (int q = (s_ * NP) / NM; q < ((s_ + 1) * NP) / NM; ++q) piece(stage, q);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i][e], f.b[j][e], acc[i][j], 0, 0, 0);
                }
    };
    // (measured, round 5: staggering the two waves of a SIMD bought the dense layers + 1.5 % on the 256x256 tile and cost the
    // convolutions 3 - 6 % — the operand streams of the dense layers are not what holds their K loop back; off unless -DPP_F_STAGGER)
#ifdef PP_F_STAGGER
    const bool late = NW == 8 && w >= 4;
#else
    constexpr bool late = false;
#endif
    bool late_pending = false;

    PP_F_SETUP(first)
    PP_F_FETCH(0)
    PP_F_FETCH(1)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
    __builtin_amdgcn_s_barrier();
    FS f0, f1;
#ifdef PP_STUDY_F_NOLDS
    {
        const f4 seed = {(float)lane, 1.f, 2.f, (float)w};
        for (int i = 0; i < MI; ++i) f0.a[i] = f1.a[i] = seed;
        for (int j = 0; j < NJ; ++j) f0.b[j] = f1.b[j] = seed;
    }
#endif
    load_q(f0, 0, 0);
    int cur = 0;
    const float alpha = pp_alpha(d);
    for (int tile = first; tile < chunk1; tile += nxw) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        for (int kt = 0; kt < nk; ++kt) {
            load_q(f1, cur, 1);
            relu_a(f0);
            // (LATE waves: the pieces of the K tile the other half issued in the previous last quad go out here — see below)
            const bool issue0 = late && late_pending;
            if (issue0) PP_F_KTILE()
            mma_q_issue(f0, cur ^ 1, issue0);
            if (issue0) PP_F_ADVANCE()
            __builtin_amdgcn_sched_barrier(0);
            load_q(f0, cur, 2);
            relu_a(f1);
            mma_q(f1);
            __builtin_amdgcn_sched_barrier(0);
            load_q(f1, cur, 3);
            relu_a(f0);
            mma_q(f0);
            __builtin_amdgcn_sched_barrier(0);
            // K tile kt + 1 has landed (this wave's pieces: the wait; every wave's: the barrier) and every fragment of K tile kt is
            // in registers: its stage takes K tile kt + 2
#ifdef PP_STUDY_F_NOBAR   // (timing study builds only)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#endif
            relu_a(f1);
            load_q(f0, cur ^ 1, 0);
            // last quad | DMA pieces of K tile kt + 2 pinned between its MFMAs (hipcc would issue them back to back).  In an 8-wave
            // workgroup the two waves of a SIMD leave the barrier together; when the operands stream from beyond the L2 (dense layers)
            // an LDS-DMA instruction waits for a slot in the memory pipeline and blocks its wave's MFMAs behind it — both waves of the
            // SIMD at once.  So only waves 0-3 issue here; waves 4-7 ("late") issue theirs one quad later, in the first quad of the
            // next K tile (the stage is free since this barrier and is not read before the next one): one wave of every SIMD always
            // has MFMAs to issue.
            if (!late) PP_F_KTILE()
            mma_q_issue(f1, cur, !late);
            if (!late) PP_F_ADVANCE()
            late_pending = late;
            __builtin_amdgcn_sched_barrier(0);
            cur ^= 1;
        }
        int tr, tc;
        pp_tile_rc_g(tile, gx, gy, walk & 255, walk >> 8, tr, tc);
#ifdef PP_STUDY_F_NOEPI   // (timing study builds only: the K loop alone; one store keeps the accumulators alive)
        {
            f32x16 keep = acc[0][0];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) keep += acc[i][j];
            float sum_ = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) sum_ += keep[e];
            if (sum_ == -12345.f) d.C[lane] = sum_;
        }
#else
        epilogue_f32<MI, NJ>(d, alpha, acc, tr * T::BM + wr * T::TM, tc * T::BN + wc * T::TN, lane);
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no DMA may still target this workgroup's LDS at exit
#undef PP_F_SETUP
#undef PP_F_NEXT_TILE_IF_DONE
#undef PP_F_KTILE
#undef PP_F_ADVANCE
#undef PP_F_FETCH
#endif
}

typedef FTile<256, 256, 2, 4, 1> F256x256;   // 8 waves, 128x64 each, 2 x 64 KB ring
typedef FTile<256, 128, 4, 2, 1> F256x128;   // 8 waves, 64x64 each, 2 x 48 KB ring
typedef FTile<128, 128, 2, 2, 2> F128x128;   // 4 waves, 64x64 each, 2 x 32 KB ring: two workgroups per CU
typedef FTile<128, 64, 2, 2, 2> F128x64;     // 4 waves, 64x32 each, 2 x 24 KB ring: two workgroups per CU
typedef FTile<256, 192, 4, 2, 1> F256x192;   // 8 waves, 64x96 each, 2 x 56 KB ring: layers whose N is a multiple of 192 (the decoder's 192-channel maps)

template <class T, int MODE>
static int pp_f_launch_one(const PpGemmDesc& d, int slots, hipStream_t st) {
    static signed char attr_state[PP_MAX_DEVICES];   // the > 64 KB dynamic-LDS opt-in is per device (and per kernel)
    signed char& ok = attr_state[pp_cur_device()];
    if (ok == 0)
        ok = hipFuncSetAttribute((const void*)pp_gemm_f_kernel<T, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES) == hipSuccess ? 1 : -1;
    if (ok < 0) return PP_ELAUNCH;
    const int gx = (d.N + T::BN - 1) / T::BN, gy = (d.M + T::BM - 1) / T::BM;
    const int nt = gx * gy;
    const int g = nt < slots ? (nt + 7) / 8 * 8 : slots / 8 * 8;
    static const int walk_env = [] { const char* e = getenv("PP_F_WALK"); return e ? atoi(e) : 0; }();   // (study: rows << 8 | columns)
    const int walk = walk_env > 0 ? walk_env : (T::BR << 8 | T::GW);
    hipLaunchKernelGGL((pp_gemm_f_kernel<T, MODE>), dim3(g), dim3(T::NW * 64), T::LDS_BYTES, st, d, gx, gy, walk);
    return PP_OK;
}

template <class T>
static int pp_f_launch_tile(const PpGemmDesc& d, int mode, int slots, hipStream_t st) {
    if (mode == 0) return pp_f_launch_one<T, 0>(d, slots, st);
    if (mode == 1) return pp_f_launch_one<T, 1>(d, slots, st);
    // (natural-order convolutions — odd channel counts, small layers — carry per-lane tap state: no 256x256 instantiation)
    if constexpr (T::BM == 256 && T::BN == 256) return pp_f_launch_one<F256x128, 2>(d, slots, st);
    else return pp_f_launch_one<T, 2>(d, slots, st);
}

int pp_gemm_f_mode(const PpGemmDesc& d) {
    if (d.conv_kh == 0) return 0;
    return (d.conv_cin % 32 == 0 && d.conv_kh * d.conv_kw <= 32) ? 1 : 2;
}

// The fp32 engine takes a launch when: one problem (no batch), B [N][K], 16-byte aligned operands with rows of a multiple of 4
// elements, K % 4 == 0 (a lane's 16-byte chunk is inside K or past it as a whole), fp32 output only, and every buffer within the
// 32-bit byte offsets of the buffer instructions.  Fills a_hl_bytes / b_hl_bytes (the operand extents) on success.
bool pp_gemm_f_ok(PpGemmDesc& d) {
    if (!d.A || !d.B || !d.C || d.A_hl || d.C_hl || d.b_kn || d.batch0 * d.batch1 != 1 || d.ksplit > 1) return false;
    if (d.act == PP_ACT_TANH || (d.shuffle_r != 0 && d.act == PP_ACT_GELU)) return false;   // (epilogue variants this engine does not carry)
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    if (!al16(d.A) || !al16(d.B) || d.lda % 4 != 0 || d.ldb % 4 != 0 || d.K % 4 != 0) return false;
    if (d.conv_kh != 0 && (d.conv_cin % 4 != 0 || d.conv_bstride % 4 != 0)) return false;
    const long long per = (long long)d.conv_ho * d.conv_wo;
    const long long a_elems = d.conv_kh != 0 ? ((d.M + per - 1) / per - 1) * d.conv_bstride + (long long)d.conv_h * d.conv_w * d.lda
                                             : (long long)(d.M - 1) * d.lda + d.K;
    const long long b_elems = (long long)(d.N - 1) * d.ldb + d.K;
    const long long abias = d.conv_kh != 0 ? ((long long)d.conv_pad * d.conv_w + d.conv_pad) * d.lda : 0;
    if ((a_elems + abias) * 4 >= 0xFFFFFF00LL || b_elems * 4 >= 0xFFFFFF00LL) return false;
    const int r2 = d.shuffle_r > 0 ? d.shuffle_r * d.shuffle_r : 1;
    const long long out_rows = (long long)d.M * r2, out_cols = d.N / r2;
    if ((out_rows - 1) * d.ldc * 4 + out_cols * 4 >= 0xFFFFFF00LL) return false;
    d.a_hl_bytes = a_elems * 4;
    d.b_hl_bytes = b_elems * 4;
    return true;
}

int pp_gemm_f_launch(const PpGemmDesc& d, int tile, int cus, hipStream_t st) {
    const int mode = pp_gemm_f_mode(d);
    switch (tile) {
        case PP_U_256x256: return pp_f_launch_tile<F256x256>(d, mode, cus, st);
        case PP_U_256x128: return pp_f_launch_tile<F256x128>(d, mode, cus, st);
        case PP_U_128x128: return pp_f_launch_tile<F128x128>(d, mode, 2 * cus, st);
        case PP_U_128x64: return pp_f_launch_tile<F128x64>(d, mode, 2 * cus, st);
        case PP_F_256x192: return mode == 2 ? pp_f_launch_tile<F256x128>(d, mode, cus, st) : pp_f_launch_tile<F256x192>(d, mode, cus, st);
        default: return PP_EINVAL;
    }
}
