#ifndef PP_GEMM_U_KERNEL_H
#define PP_GEMM_U_KERNEL_H
// (header: included by pp_gemm_u1.hip / pp_gemm_u2.hip / pp_gemm_uh.hip, which instantiate it per operand format)
// The pre-split contraction kernel of the network engine: ONE templated K loop for every GEMM / implicit-GEMM convolution
// whose operands are both in the engine's operand format (PpGemmDesc.A_hl / B_hl).
//
//   C[m, n] = epilogue( sum_k A(m, k) * B(n, k) ),   A: dense rows or an NHWC image through an implicit im2col
//
// Arithmetic (template parameter TERMS):
//   TERMS = 2  "f16x3": an operand element is hi + lo (two fp16 terms of 4 x, pp_common.h); a product is evaluated as
//              lo_a hi_b + hi_a lo_b + hi_a hi_b, three v_mfma_f32_16x16x32_f16 per 32 k, fp32 accumulation (22 operand bits);
//   TERMS = 1  "f16":   plain fp16 operands f16(4 x), one v_mfma_f32_16x16x32_f16 per 32 k, fp32 accumulation.
// EVERY instantiation accumulates an output element in the same order — K tiles in K order (channel-slice-major for the
// convolutions with Cin % 32 == 0, natural otherwise), per K tile the terms in the order above — so the value of an output
// element does not depend on the tile configuration the autotuner picks (tests: ..._agree_bitwise_across_tile_configurations).
//
// Structure (template parameter T = block tile / waves / ring depth / workgroups per CU):
//   * operand tiles go global -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`): a K tile is ONE 128-byte segment per row
//     (32 k of hl, 64 k of h), a wave instruction moves 8 full lines, no staging registers, no ds_write;
//   * LDS image: rows of 128 bytes; the 16-byte chunk c of row r sits at position c ^ key(r), key(r) = 3 (r >> 2 & 1) |
//     (r >> 1 & 1) << 2 — conflict-free ds_read_b128 for the 16x16x32 fragment pattern (lane l: row l & 15, k-group l >> 4)
//     at EVERY row offset (the row-shared convolution kernel reads shifted rows) and for both operand formats; it is
//     applied to the per-lane SOURCE address because an LDS-DMA writes lane-linearly;
//   * ring of S stages, S - 1 K tiles in flight; one counted `s_waitcnt vmcnt((S - 2) P)` + raw s_barrier per K tile
//     (P = DMA pieces per wave and K tile), placed before the LAST unit of the tile;
//   * the MFMAs are issued transposed (weights as the instruction's A matrix) and the weight tile's LDS rows are permuted, so
//     a lane's accumulators are 8 consecutive columns of one output row: the epilogue stores straight from the registers;
//   * a wave's block (16 MI x 16 NJ) is multiplied in units (pair of 16-row blocks) x (half of the column blocks):
//     A fragment pairs alternate between two register sets, the two B halves are refilled in place one unit after their
//     last use, so the fragment registers of the next unit are always loading while the current one multiplies;
//   * the DMA pieces of K tile kt + S - 1 ... are issued between the MFMAs of the last unit (an LDS-DMA costs the wave 60-180
//     issue cycles: back to back they stall the matrix pipe of both waves of a SIMD);
//   * persistent: a workgroup walks a list of output tiles (XCD-contiguous chunks, bands of 4 tile rows x <= 8 tile
//     columns), the DMA stream runs ahead ACROSS tile boundaries; a launch with one tile per workgroup is the same code;
//   * padded taps, M / N / K tails: out-of-range buffer offsets (the DMA writes zeros, no traffic, no branches).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pp_gemm_dev.h"
#include "pp_gemm_u.h"

typedef __attribute__((address_space(3))) void* lds_ptr_t;
#ifdef PP_STUDY_OFFMASK   // (timing study builds only: every operand read lands in a small cache-resident window)
#define PP_STUDY_OFF(x) ((x) & (unsigned)(PP_STUDY_OFFMASK))
#else
#define PP_STUDY_OFF(x) (x)
#endif

// tile rows per band of the persistent tile walk (pp_tile_rc_g) for the ONE-term (h format) kernels: with 2-byte operands a band of 4 tile rows
// re-reads the weight columns twice as often per operand byte as the hl kernels do (study switch; profiles/r06/README.md)
#ifndef PP_U1_BAND_ROWS
#define PP_U1_BAND_ROWS 4
#endif

template <int BM_, int BN_, int WM_, int WN_, int S_, int OCC_, int PREF_ = 0, int B3_ = 0>
struct TileCfg {
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, S = S_, OCC = OCC_;
    static constexpr bool PREF = PREF_ != 0;             // fragments of K tile kt + 1 are read during K tile kt (two register sets)
    static constexpr bool B3 = B3_ != 0;                 // S = 2 for the A operand, a ring of THREE stages for the B operand
    static constexpr int NW = WM_ * WN_;                 // waves
    static constexpr int TM = BM_ / WM_, TN = BN_ / WN_;   // wave block
    static constexpr int MI = TM / 16, NJ = TN / 16;     // 16x16 MFMA tiles per wave block
    static constexpr int PA = BM_ / 8 / NW, PB = BN_ / 8 / NW;   // LDS-DMA pieces (8 rows each) per wave and K tile
    static constexpr int A_H = BM_ * 64, B_H = BN_ * 64;   // halfs per operand per stage (128-byte rows)
    static constexpr int STAGE = A_H + B_H;
    static constexpr int LDS_BYTES = (B3_ ? 2 * A_H + 3 * B_H : S_ * STAGE) * 2;   // the ring (the epilogue leaves from the registers)
    static_assert(!B3_ || (S_ == 2 && !PREF_), "the 2 + 3 ring is a variant of the two-stage schedule");
    static_assert(PREF_ || (MI % 2 == 0 && NJ % 2 == 0 && (MI / 2) % 2 == 0), "unit schedule: pairs of row blocks, an even number of them");
    static_assert(!PREF_ || S_ >= 3, "the prefetching schedule refills the stage read one K tile earlier: a ring of three");
    static_assert(BM_ % (8 * NW) == 0 && BN_ % (8 * NW) == 0, "DMA pieces are 8 rows per wave instruction");
};

__device__ __forceinline__ f32x4 pp_mfma16(const h8 a, const h8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// MODE 0: dense A (3: dense, K slices as extra tile rows — PpGemmDesc.ksplit); 1: convolution, channel-slice-major K order (Cin a multiple of the K tile); 2: convolution, natural K order
// (any Cin % 8 == 0: the 8 k of a lane's chunk share a tap)
// VEC: the epilogue's vector conditions hold (pp_gemm_u_vec_ok, checked on the host: N % 8 == 0, aligned rows); the
// element-wise epilogue lives in its own instantiations (both in one kernel cost 100 registers and spills in the 256-wide tiles)
template <class T, int MODE, int TERMS, bool VEC>
__global__ __launch_bounds__(T::NW * 64, T::OCC) void pp_gemm_u_kernel(const PpGemmDesc d, int gx, int gy) {
#if defined(__HIP_DEVICE_COMPILE__)  // the host pass only needs the launch stub (it cannot instantiate the LDS-DMA builtins)
    constexpr bool DENSE = MODE == 0 || MODE == 3, KS = MODE == 3;   // MODE 3: dense with K slices (PpGemmDesc.ksplit)
    constexpr int NW = T::NW, PA = T::PA, PB = T::PB, S = T::S, MI = T::MI, NJ = T::NJ, NIP = MI / 2, NJH = NJ / 2;
    constexpr int KT = 64 / TERMS;       // k per K tile (one 128-byte row segment)
    constexpr int EB = 2 * TERMS;        // operand bytes per element
    constexpr int STAGE = T::STAGE, A_H = T::A_H;
    // LDS image: stages of [A | B] (A_STR = B_STR = STAGE, B0 = A_H) or, for the 2 + 3 ring, two A stages then three B stages
    constexpr int A_STR = T::B3 ? A_H : STAGE, B_STR = T::B3 ? T::B_H : STAGE, B0 = T::B3 ? 2 * A_H : A_H;
    constexpr int NSUB = TERMS == 2 ? 3 : 2;   // MFMAs per (16x16 tile, K tile)
    extern __shared__ __attribute__((aligned(16))) _Float16 glds[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w / T::WN, wc = w % T::WN, l15 = lane & 15, lq = lane >> 4;
    // tiles of this workgroup: XCD x = id % 8 owns a contiguous chunk of the tile list; its workgroups interleave over it,
    // so the tiles in flight on one XCD at any time are neighbours (shared A rows / halo / B columns in its L2)
    const int ntiles = gx * gy, nxw = (int)gridDim.x >> 3;  // gridDim.x is a multiple of 8
    const int xcd = blockIdx.x & 7, q8 = ntiles >> 3, r8 = ntiles & 7;
    const int chunk0 = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int chunk1 = chunk0 + (xcd < r8 ? q8 + 1 : q8);
    const int first = chunk0 + (int)(blockIdx.x >> 3);
    if (first >= chunk1) return;
    const __amdgpu_buffer_rsrc_t Ar = __builtin_amdgcn_make_buffer_rsrc((void*)d.A_hl, 0, (int)d.a_hl_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t Br = __builtin_amdgcn_make_buffer_rsrc((void*)d.B_hl, 0, (int)d.b_hl_bytes, 0x00020000);
    // DMA slot of this lane: piece q of wave w fills LDS rows (q NW + w) 8 + (lane >> 3); LDS chunk position lane & 7 of such
    // a row holds source chunk sc (the key only reads bits 1, 2 of the row = of lane >> 3: the same for every piece)
    const int lr = lane >> 3;
    const int sc = (lane & 7) ^ pp_swz_key(lr);
    const int kch = TERMS == 2 ? (sc >> 1) * 8 : sc * 8;   // first k, inside a K tile, of this lane's chunk
    const unsigned cbyte = (unsigned)sc * 16u;
    const int ntaps = d.conv_kh * d.conv_kw;
    const int nk = MODE == 1 ? ntaps * (d.conv_cin / KT) : (d.K + KT - 1) / KT;
    // MODE 0 / 1: a piece's address is  base + voffset (per lane, fixed for a tile) + soffset (wave-uniform, per K tile) — the K
    // loop spends no vector instruction on a dense piece's address and two on a convolution's (the tap's padding bit).  The
    // range check looks at voffset alone: MODE 1 moves the base back by the largest negative window offset (pad rows + pad
    // pixels) so that every voffset is >= 0; past the end of the workgroup's tile list the descriptor has zero records.
    const long long abias = MODE == 1 ? ((long long)d.conv_pad * d.conv_w + d.conv_pad) * d.lda : 0;   // elements
    const bool ktail = DENSE && d.K % KT != 0;
    unsigned tmask = 0u;   // MODE 0, K % KT != 0: all-ones in the lanes whose chunk of the current K tile lies past K

    // ---- fetch side: addressing state of the tile the DMA stream is in
    unsigned abyte[PA], amask[PA], bbyte[PB];   // A rows: byte offset of k = 0 (+ this lane's chunk); tap mask / row-valid bit
    int aoy[MODE == 2 ? PA : 1], aox[MODE == 2 ? PA : 1];   // MODE 2: top-left input pixel of the row's window
    int ftile = first, fkt = 0;                  // tile and K-tile index of the next DMA group (wave-uniform)
    int ctap = 0, cky = 0, ckx = 0, cci = 0;     // MODE 1 (wave-uniform): tap / channel slice of the next K tile
    int tky = 0, tkx = 0, tci = 0;               // MODE 2 (per lane): tap / channel of k = fkt KT + kch
    // (macros, not nested lambdas: state captured by reference stayed in scratch memory, and every scratch store /
    // reload counts in vmcnt — hipcc then drained the DMA pipeline with vmcnt(0) inside the K loop)
#define PP_U_SETUP(TILE)                                                                                             \
    {                                                                                                                \
        int tr_, tc_;                                                                                                \
        pp_tile_rc_g((TILE), gx, gy, 8, TERMS == 1 ? PP_U1_BAND_ROWS : 4, tr_, tc_);                                                                        \
        const int m0_ = tr_ * T::BM, n0_ = tc_ * T::BN;                                                              \
        _Pragma("unroll") for (int j = 0; j < PA; ++j) {                                                             \
            const int m = m0_ + (j * NW + w) * 8 + lr;                                                               \
            const bool ok = m < d.M;                                                                                 \
            long long base = ok ? (long long)m * d.lda : 0;                                                          \
            if (KS && ok) { /* K slices: tile rows s M .. are slice s of the real rows (d.K = its length) */ \
                const int sl = m0_ / d.ks_rows;                                                                      \
                base = (long long)(m - sl * d.ks_rows) * d.lda + (long long)sl * d.K;                                \
            }                                                                                                        \
            unsigned mask = ok ? 1u : 0u;                                                                            \
            if (!DENSE) {                                                                                         \
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
            /* MODE 0: an invalid row is an out-of-range offset; MODE 1: offsets relative to the biased base (>= 0), amask = the */ \
            /* INVALID taps (one v_bfe_i32 turns the tap's bit into the all-ones mask); MODE 2: as computed              */ \
            abyte[j] = DENSE ? (ok ? (unsigned)(base * EB) + cbyte : 0xFFFFFFFFu)                                \
                                 : (unsigned)((base + (MODE == 1 ? abias : 0)) * EB) + cbyte;                        \
            amask[j] = MODE == 1 ? ~mask : mask;                                                                     \
        }                                                                                                            \
        _Pragma("unroll") for (int j = 0; j < PB; ++j) {                                                             \
            const int nb = n0_ + pp_wperm((j * NW + w) * 8 + lr);   /* LDS weight rows are permuted: pp_gemm_dev.h */  \
            const long long koff_ = KS ? (long long)(m0_ / d.ks_rows) * d.K : 0;          \
            /* grouped launch (the frequencies of a Winograd convolution): the row tile reads the weights of ITS group */ \
            const unsigned gofs_ = (MODE == 0 && d.grp_rows != 0) ? (unsigned)(m0_ / d.grp_rows) * (unsigned)d.grp_b_bytes : 0u;   \
            bbyte[j] = nb < d.N ? (unsigned)(((long long)nb * d.ldb + koff_) * EB) + cbyte + gofs_ : 0xFFFFFFFFu;    \
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
    // the DMA stream moves to the workgroup's next tile once a tile's nk K tiles have been issued
#define PP_U_NEXT_TILE_IF_DONE()                                \
    if (fkt == nk && ftile < chunk1) {                          \
        ftile += nxw;                                           \
        if (ftile < chunk1) PP_U_SETUP(ftile) else fkt = 0;     \
    }
    // pieces of K tile fkt of tile ftile (an offset of 0xFFFFFFFF reads zeros without traffic: padding, tails)
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
#ifdef PP_STUDY_NODMA   // (timing study builds only: the K loop without its operand traffic)
        if (d.M > 0) return;
#endif
        const lds_ptr_t dst = (lds_ptr_t)(glds + stage * A_STR + ((j * NW + w) * 8) * 64);
        if (MODE == 2) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(Ar, dst, 16, PP_STUDY_OFF(off_a2(j)), 0, 0, 0);
        } else {
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
                (void*)((const char*)d.A_hl - abias * EB), 0, ftile < chunk1 ? (int)(d.a_hl_bytes + abias * EB) : 0, 0x00020000);
            const unsigned v = DENSE ? abyte[j] | tmask : abyte[j] | (unsigned)__builtin_amdgcn_sbfe((int)amask[j], (unsigned)ctap, 1u);
            const int so = DENSE ? fkt * 128 : ((cky * d.conv_w + ckx) * d.lda + cci) * EB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 16, PP_STUDY_OFF(v), so, 0, 0);
        }
    };
    auto dma_b = [&](int stage, int j) __attribute__((always_inline)) {
#ifdef PP_STUDY_NODMA
        if (d.M > 0) return;
#endif
        const lds_ptr_t dst = (lds_ptr_t)(glds + stage * B_STR + B0 + ((j * NW + w) * 8) * 64);
        if (MODE == 2) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(Br, dst, 16, PP_STUDY_OFF(off_b2(j)), 0, 0, 0);
        } else {
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)d.B_hl, 0, ftile < chunk1 ? (int)d.b_hl_bytes : 0, 0x00020000);
            const unsigned v = DENSE ? bbyte[j] | tmask : bbyte[j];
            const int so = DENSE ? fkt * 128 : (ctap * d.conv_cin + cci) * EB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 16, PP_STUDY_OFF(v), so, 0, 0);
        }
    };
    // before the pieces of a K tile: move to the next tile if this one is issued; the K-tail lane mask
#define PP_U_KTILE()                                                               \
    {                                                                              \
        PP_U_NEXT_TILE_IF_DONE()                                                   \
        if (ktail) tmask = fkt * KT + kch < d.K ? 0u : 0xFFFFFFFFu;                \
    }
#define PP_U_ADVANCE() /* after the pieces of a K tile */                                           \
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
#define PP_U_FETCH(STAGE_)                                               \
    {                                                                    \
        PP_U_KTILE()                                                     \
        _Pragma("unroll") for (int j = 0; j < PA; ++j) dma_a(STAGE_, j); \
        _Pragma("unroll") for (int j = 0; j < PB; ++j) dma_b(STAGE_, j); \
        PP_U_ADVANCE()                                                   \
    }

    // ---- compute side
    // fragment of a 16-row block: lane l reads row l & 15, the 8 k of k-group l >> 4: sub s = term (hl) or 32-k step (h)
    const int sw = pp_swz_key(l15);
    const int fo0 = l15 * 64 + ((TERMS == 2 ? 2 * lq : lq) ^ sw) * 8;            // halfs: sub 0
    const int fo1 = l15 * 64 + ((TERMS == 2 ? 2 * lq + 1 : lq + 4) ^ sw) * 8;    // sub 1
    struct FA {
        h8 x[2][2];        // [row block of the pair][sub]
    };
    struct FB {
        h8 x[NJH][2];      // [column block of the half][sub]
    };
    f32x4 acc[MI][NJ];
#ifdef PP_STUDY_LDSNODEP
    f32x4 sink = {0.f, 0.f, 0.f, 0.f};
#endif
    auto load_a = [&](FA& f, int stage, int ip) __attribute__((always_inline)) {
#ifdef PP_STUDY_NOLDS   // (timing study builds only: the K loop without its fragment reads)
        if (d.M > 0) return;
#endif
#ifdef PP_STUDY_LDSNODEP   // (timing study builds only: the fragment reads are issued, the MFMAs do not depend on them)
        {
            const unsigned a0 = (unsigned)(uintptr_t)(lds_ptr_t)(glds + stage * A_STR + (wr * T::TM + ip * 32) * 64);
            for (int i = 0; i < 2; ++i) {
                asm volatile("ds_read_b128 %0, %1" : "+v"(sink) : "v"(a0 + (i * 16 * 64 + fo0) * 2));
                asm volatile("ds_read_b128 %0, %1" : "+v"(sink) : "v"(a0 + (i * 16 * 64 + fo1) * 2));
            }
            return;
        }
#endif
        const _Float16* st = glds + stage * A_STR + (wr * T::TM + ip * 32) * 64;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f.x[i][0] = *(const h8*)(st + i * 16 * 64 + fo0);
            f.x[i][1] = *(const h8*)(st + i * 16 * 64 + fo1);
        }
    };
    auto load_b = [&](FB& f, int stage, int jh) __attribute__((always_inline)) {
#ifdef PP_STUDY_NOLDS
        if (d.M > 0) return;
#endif
#ifdef PP_STUDY_LDSNODEP
        {
            const unsigned b0 = (unsigned)(uintptr_t)(lds_ptr_t)(glds + stage * B_STR + B0 + (wc * T::TN + jh * NJH * 16) * 64);
            for (int j = 0; j < NJH; ++j) {
                asm volatile("ds_read_b128 %0, %1" : "+v"(sink) : "v"(b0 + (j * 16 * 64 + fo0) * 2));
                asm volatile("ds_read_b128 %0, %1" : "+v"(sink) : "v"(b0 + (j * 16 * 64 + fo1) * 2));
            }
            return;
        }
#endif
        const _Float16* st = glds + stage * B_STR + B0 + (wc * T::TN + jh * NJH * 16) * 64;
#pragma unroll
        for (int j = 0; j < NJH; ++j) {
            f.x[j][0] = *(const h8*)(st + j * 16 * 64 + fo0);
            f.x[j][1] = *(const h8*)(st + j * 16 * 64 + fo1);
        }
    };
    // one 16x16 accumulator tile: the MFMAs of one K tile in the engine's fixed term order
    auto mma1 = [&](const FA& a, const FB& b, int ip, int jh, int i, int j) __attribute__((always_inline)) {
        f32x4& c = acc[2 * ip + i][jh * NJH + j];
        // (transposed: the weight fragment is the instruction's A matrix — accumulator layout in pp_gemm_dev.h)
        if (TERMS == 2) {
            c = pp_mfma16(b.x[j][0], a.x[i][1], c);
            c = pp_mfma16(b.x[j][1], a.x[i][0], c);
            c = pp_mfma16(b.x[j][0], a.x[i][0], c);
        } else {
            c = pp_mfma16(b.x[j][0], a.x[i][0], c);
            c = pp_mfma16(b.x[j][1], a.x[i][1], c);
        }
    };
    auto mma_unit = [&](const FA& a, const FB& b, int ip, int jh) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJH; ++j) mma1(a, b, ip, jh, i, j);
    };
    (void)NSUB;
    // SPREAD (rings of three stages or more): the stage read in K tile kt - 1 is free from that tile's barrier on, so the pieces
    // of K tile kt + S - 1 are issued one by one over ALL units of K tile kt instead of together in its last unit.  A burst of
    // LDS-DMA instructions blocks the issuing wave for 60-180 cycles apiece, and the two waves of a SIMD — in step after the
    // barrier — burst at the same time: the matrix pipe idles (profiles/r03/engine_study.md, "K loop without DMA").
    constexpr int EPI_ST_ = MI * NJ;   // VMEM operations of a vector epilogue, at least (see EPI_ST below)
    constexpr bool SPREAD = S >= 3 && MODE != 2;
    constexpr int NU = 2 * NIP, NM = 2 * NJH, NSLOT = NU * NM, NP = PA + PB;   // units / accumulator tiles per unit / per K tile
    // B3 (A ring of two, B ring of three — all of the 160 KB of LDS for the 256x256 tile): the B stage read one K tile earlier is
    // free during the whole K tile, so the B pieces of K tile kt + 2 are spread over units 0 .. NU - 2 and only the A pieces wait
    // for the barrier (half the burst of the two-stage schedule; same prefetch distance, same arithmetic)
    constexpr bool B3 = T::B3 && MODE != 2;
    // B3E: with four row pairs (NIP == 4) every fragment of the K tile is in registers one unit earlier than the two-stage schedule
    // assumes (the last two units multiply the LAST row pair by the two column halves), so wait + barrier move in front of unit NU - 2
    // and the A pieces go out two and two over the last TWO units
    constexpr bool B3E = B3 && NIP == 4;
    constexpr int NBEFORE = SPREAD ? ((NU - 1) * NM * NP) / NSLOT : (B3E ? ((NU - 2) * PB) / (NU - 1) : (B3 ? PB : 0));  // pieces issued before the wait
    constexpr int INFLIGHT = SPREAD ? (S - 3) * NP + NBEFORE : (S - 2) * NP + NBEFORE;   // DMA pieces younger than the awaited K tile
    constexpr int RELAX = SPREAD ? S - 2 : S - 1;   // K tiles after an epilogue whose awaited pieces are OLDER than its stores
    int fill = S - 1;                               // SPREAD: the stage being refilled during the current K tile
    auto piece = [&](int stage, int q) __attribute__((always_inline)) {
        if (q < PA) dma_a(stage, q);
        else dma_b(stage, q - PA);
    };
    // unit u of a K tile: its accumulator tiles, with (SPREAD) the DMA pieces that fall into its slots pinned between them
    int bfill = 2;                                  // B3: the B stage being refilled during the current K tile ((kt + 2) % 3)
    int a_fill = 0;                                 // B3E: the A stage of the current K tile (refilled from unit NU - 2 on)
    auto unit = [&](const FA& a, const FB& b, int ip, int jh, int u) __attribute__((always_inline)) {
        if (B3 && u < NU - 1 && ((u * PB) / (NU - 1) != ((u + 1) * PB) / (NU - 1) || (B3E && u == NU - 2))) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = (u * PB) / (NU - 1); q < ((u + 1) * PB) / (NU - 1); ++q) dma_b(bfill, q);
            if (B3E && u == NU - 2) {   // (after the barrier: the first half of the A pieces of K tile kt + 2)
#pragma unroll
                for (int q = 0; q < PA / 2; ++q) dma_a(a_fill, q);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int t = 0; t < NM; ++t) {
            const int slot = u * NM + t;
            if (SPREAD && (slot * NP) / NSLOT != ((slot + 1) * NP) / NSLOT) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = (slot * NP) / NSLOT; q < ((slot + 1) * NP) / NSLOT; ++q) piece(fill, q);
                __builtin_amdgcn_sched_barrier(0);
            }
            mma1(a, b, ip, jh, t / NJH, t % NJH);
        }
    };

    if constexpr (T::PREF) {
        // ---- the prefetching schedule (wave blocks of at most 64x64: two full fragment sets fit the register budget) ----------
        // Measured on the unit schedule (profiles/r03/engine_study.md): with the LDS-DMA stream running, a fragment read issued
        // one unit (12 MFMAs) ahead of its use is late — the same loop with the MFMAs made independent of the reads runs 20 %
        // faster, while reads or DMA alone cost 0 % / 13 %.  Here EVERY fragment of K tile kt + 1 is read during the first half
        // of K tile kt into the other register set; the MFMAs of a K tile never wait for LDS.  Ring: K tile kt + 1 is being
        // read, kt + 2 .. kt + S - 1 are landed or in flight, the stage of K tile kt (in registers since the last barrier) is
        // refilled with K tile kt + S, its pieces spread over the K tile's accumulator tiles.
        constexpr int NBLK = MI + NJ;                          // 16-row fragment blocks per K tile (A then B), two reads each
        constexpr int NSL = MI * NJ;                           // accumulator tiles = slots of a K tile
#ifndef PP_PREF_RDSL_DIV
#define PP_PREF_RDSL_DIV 2
#endif
#ifndef PP_PREF_WAIT_BACK
#define PP_PREF_WAIT_BACK (NSL / 4)
#endif
        constexpr int RDSL = NSL / PP_PREF_RDSL_DIV;           // the reads go into the first slots
        constexpr int WAITSL = NSL - PP_PREF_WAIT_BACK;        // wait + barrier before this slot
        constexpr int NBEF = (WAITSL * NP) / NSL;              // pieces of the current K tile issued before the wait
        constexpr int INFL = (S - 3) * NP + NBEF;              // DMA pieces younger than the awaited K tile (kt + 2)
        static_assert(INFL + EPI_ST_ <= 63, "vmcnt is a 6-bit counter");
        struct FS {
            h8 a[MI][2], b[NJ][2];
        };
        FS f0, f1;
        auto load_blk = [&](FS& f, int stage, int blk) __attribute__((always_inline)) {
            if (blk < MI) {
                const _Float16* st = glds + stage * A_STR + (wr * T::TM + blk * 16) * 64;
                f.a[blk][0] = *(const h8*)(st + fo0);
                f.a[blk][1] = *(const h8*)(st + fo1);
            } else {
                const _Float16* st = glds + stage * B_STR + B0 + (wc * T::TN + (blk - MI) * 16) * 64;
                f.b[blk - MI][0] = *(const h8*)(st + fo0);
                f.b[blk - MI][1] = *(const h8*)(st + fo1);
            }
        };
        auto mma_p = [&](const FS& f, int i, int j) __attribute__((always_inline)) {
            f32x4& c = acc[i][j];
            if (TERMS == 2) {
                c = pp_mfma16(f.b[j][0], f.a[i][1], c);
                c = pp_mfma16(f.b[j][1], f.a[i][0], c);
                c = pp_mfma16(f.b[j][0], f.a[i][0], c);
            } else {
                c = pp_mfma16(f.b[j][0], f.a[i][0], c);
                c = pp_mfma16(f.b[j][1], f.a[i][1], c);
            }
        };
        int cur = 0, nxt = 1;
        int since_epi = S;
        const float descale = pp_alpha(d) / (PP_A_SCALE * d.b_scale);
        // one K tile: MFMAs from set FC; fragments of the next K tile (stage nxt) into FN; pieces of K tile + S into stage cur
        // (a macro: through a nested lambda the two sets stayed in scratch memory)
#define PP_U_KTILE_PREF(FC, FN)                                                                                          \
    {                                                                                                                    \
        PP_U_KTILE()                                                                                                     \
        _Pragma("unroll") for (int sl = 0; sl < NSL; ++sl) {                                                             \
            if (sl == WAITSL) {                                                                                          \
                /* K tile kt + 2 has landed (this wave's pieces: counted wait; every wave's: barrier) and every wave */  \
                /* holds K tile kt + 1 in registers: its stage is the next K tile's refill target                    */  \
                if (since_epi < S - 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(INFL + EPI_ST_) : "memory");  \
                else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(INFL) : "memory");                              \
                ++since_epi;                                                                                             \
                PP_U_BARRIER()                                                                                           \
            }                                                                                                            \
            if ((sl * NP) / NSL != ((sl + 1) * NP) / NSL) {                                                              \
                __builtin_amdgcn_sched_barrier(0);                                                                       \
                _Pragma("unroll") for (int q = (sl * NP) / NSL; q < ((sl + 1) * NP) / NSL; ++q) piece(cur, q);           \
                __builtin_amdgcn_sched_barrier(0);                                                                       \
            }                                                                                                            \
            if (sl < RDSL) {                                                                                             \
                _Pragma("unroll") for (int bk = (sl * NBLK) / RDSL; bk < ((sl + 1) * NBLK) / RDSL; ++bk)                 \
                    load_blk(FN, nxt, bk);                                                                               \
            }                                                                                                            \
            mma_p(FC, sl / NJ, sl % NJ);                                                                                 \
        }                                                                                                                \
        PP_U_ADVANCE()                                                                                                   \
        cur = nxt;                                                                                                       \
        nxt = nxt == S - 1 ? 0 : nxt + 1;                                                                                \
    }
#ifdef PP_STUDY_NOBAR
#define PP_U_BARRIER()
#else
#define PP_U_BARRIER() __builtin_amdgcn_s_barrier();
#endif
        PP_U_SETUP(first)
#pragma unroll
        for (int s = 0; s < S; ++s) PP_U_FETCH(s)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 1) * NP) : "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int bk = 0; bk < NBLK; ++bk) load_blk(f0, 0, bk);
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((S - 2) * NP) : "memory");
        __builtin_amdgcn_s_barrier();
        for (int tile = first; tile < chunk1; tile += nxw) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            // (two K tiles per trip, the sets named statically: selected by a parity branch they cost 200 spilled registers)
            for (int kt = 0; kt + 1 < nk; kt += 2) {
                PP_U_KTILE_PREF(f0, f1)
                PP_U_KTILE_PREF(f1, f0)
            }
            if (nk & 1) {
                PP_U_KTILE_PREF(f0, f1)
                f0 = f1;   // an odd K tile count: the next tile's first fragments move to the set every tile starts from
            }
            int tr, tc;
            pp_tile_rc_g(tile, gx, gy, 8, TERMS == 1 ? PP_U1_BAND_ROWS : 4, tr, tc);
            const int mw = tr * T::BM + wr * T::TM, nw = tc * T::BN + wc * T::TN;
#ifdef PP_STUDY_NOEPI
            {
                f32x4 keep = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) keep += acc[i][j];
                if (keep[0] + keep[1] + keep[2] + keep[3] == -12345.f) d.C[lane] = keep[0];
            }
#else
            if (VEC) epilogue_wave16<MI, NJ, TERMS>(d, descale, acc, mw, nw, lane);
            else epilogue_scalar16<MI, NJ, TERMS>(d, descale, acc, mw, nw, lane);
#endif
            since_epi = VEC ? 0 : S;
            if (!VEC) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        }
    } else {
#ifdef PP_STUDY_STAGGER   // (timing study builds only: four phase groups of workgroups, offset by PP_STUDY_STAGGER x 10 ns each)
    {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        const unsigned long long wait_ = (unsigned long long)((blockIdx.x >> 3) & 3) * (unsigned long long)(PP_STUDY_STAGGER);
        while (__builtin_amdgcn_s_memrealtime() - t0 < wait_) __builtin_amdgcn_s_sleep(32);
    }
#endif
    PP_U_SETUP(first)
#pragma unroll
    for (int s = 0; s < (SPREAD ? S - 1 : S); ++s) PP_U_FETCH(s)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(((SPREAD ? S - 1 : S) - 1) * (PA + PB)) : "memory");
    __builtin_amdgcn_s_barrier();
    FA fa0, fa1;
    FB fb0, fb1;
#if defined(PP_STUDY_NOLDS) || defined(PP_STUDY_LDSNODEP)
    {
        const h8 seed = {(_Float16)lane, (_Float16)1, (_Float16)2, (_Float16)3, (_Float16)w, (_Float16)5, (_Float16)6, (_Float16)7};
        for (int i = 0; i < 2; ++i)
            for (int k = 0; k < 2; ++k) fa0.x[i][k] = fa1.x[i][k] = seed;
        for (int i = 0; i < NJH; ++i)
            for (int k = 0; k < 2; ++k) fb0.x[i][k] = fb1.x[i][k] = seed;
    }
#endif
    load_a(fa0, 0, 0);
    load_b(fb0, 0, 0);
    int cur = 0, nxt = S > 1 ? 1 : 0;
    int bcur = 0, bnxt = 1;             // B3: the B ring's cursors (three stages); otherwise the B operand shares cur / nxt
    const float descale = pp_alpha(d) / (PP_A_SCALE * d.b_scale);
    // The vector epilogue issues a fixed number of VMEM operations per wave, whatever the tile (rows / columns out of range
    // are out-of-range offsets, not skipped instructions): at least one 16-byte store per 4 accumulator registers.
    constexpr int EPI_ST = MI * NJ;
    static_assert(INFLIGHT + EPI_ST <= 63, "vmcnt is a 6-bit counter");
    int since_epi = S;   // K tiles since the last epilogue (>= S - 1: nothing of it in flight)
    for (int tile = first; tile < chunk1; tile += nxw) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < nk; ++kt) {
            // unit (0, 0): fragments of the second column half and of the next row pair start loading
            // (sched_barrier: the fragment reads of a unit stay in that unit — hoisted further up they lengthen the live ranges
            // past the 256 registers of a 512-thread workgroup and hipcc spills inside the loop)
            if (SPREAD || B3) PP_U_KTILE()
            load_b(fb1, B3 ? bcur : cur, 1);
            load_a(fa1, cur, 1);
            unit(fa0, fb0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            unit(fa0, fb1, 0, 1, 1);
            __builtin_amdgcn_sched_barrier(0);
            if (NIP == 4) {
                load_a(fa0, cur, 2);
                unit(fa1, fb0, 1, 0, 2);
                __builtin_amdgcn_sched_barrier(0);
                unit(fa1, fb1, 1, 1, 3);
                __builtin_amdgcn_sched_barrier(0);
                load_a(fa1, cur, 3);
                unit(fa0, fb0, 2, 0, 4);
                __builtin_amdgcn_sched_barrier(0);
                unit(fa0, fb1, 2, 1, 5);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!B3E) {
                unit(fa1, fb0, NIP - 1, 0, NU - 2);
                __builtin_amdgcn_sched_barrier(0);
            }
            // K tile kt + 1 has landed (this wave's pieces: counted wait; every wave's: barrier); every fragment of tile kt is in registers
            // (the first S - 1 K tiles after an epilogue: its stores — at least EPI_ST per wave, all younger than the pieces waited
            // for here — may stay in flight; from then on they are older than the awaited pieces and have had S - 1 K tiles to retire)
#ifdef PP_STUDY_NOWAIT   // (timing study builds only: the DMA stream is issued but never waited for)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
            if (since_epi < RELAX) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(INFLIGHT + EPI_ST) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(INFLIGHT) : "memory");
#endif
            ++since_epi;
#ifndef PP_STUDY_NOBAR   // (timing study builds only)
            __builtin_amdgcn_s_barrier();
#endif
#ifdef PP_STUDY_SPREADST   // (timing study builds only, with PP_STUDY_NOEPI: a tile's 32 KB of stores per wave spread over its first K tiles)
            if (kt < 16 && d.C_hl) {
                // (bounded descriptor: the last tile row reaches past M — those stores are dropped by the range check)
                const __amdgpu_buffer_rsrc_t Hs = __builtin_amdgcn_make_buffer_rsrc((void*)d.C_hl, 0, (int)((long long)d.M * d.ldc_h * EB), 0x00020000);
                const unsigned so_ = (unsigned)(((tile * 8 + w) * 32 + kt * 2) * 1024 + lane * 16);
                pp_bstore(Hs, __builtin_bit_cast(f4, fa1.x[0][0]), so_);
                pp_bstore(Hs, __builtin_bit_cast(f4, fa1.x[1][0]), so_ + 1024u);
            }
#endif
            if (B3E) {
                a_fill = cur;
                unit(fa1, fb0, NIP - 1, 0, NU - 2);
                __builtin_amdgcn_sched_barrier(0);
            }
            // last unit | DMA of K tile kt + S into the stage just freed, first fragments of K tile kt + 1
            if (SPREAD) {
                load_a(fa0, nxt, 0);
#pragma unroll
                for (int t = 0; t < NM; ++t) {
                    const int slot = (NU - 1) * NM + t;
#pragma unroll
                    for (int q = (slot * NP) / NSLOT; q < ((slot + 1) * NP) / NSLOT; ++q) piece(fill, q);
                    if (t == 1) load_b(fb0, nxt, 0);
                    mma1(fa1, fb1, NIP - 1, 1, t / NJH, t % NJH);
                    __builtin_amdgcn_sched_barrier(0);
                }
                PP_U_ADVANCE()
                fill = cur;
                cur = nxt;
                nxt = nxt == S - 1 ? 0 : nxt + 1;
                continue;
            }
            if (!B3) PP_U_KTILE()
            if (MODE == 2) {
#pragma unroll
                for (int j = 0; j < PA; ++j) dma_a(cur, j);
#pragma unroll
                for (int j = 0; j < PB; ++j) dma_b(cur, j);
                PP_U_ADVANCE()
                load_a(fa0, nxt, 0);
                load_b(fb0, nxt, 0);
                mma_unit(fa1, fb1, NIP - 1, 1);
            } else {
                // pieces spread over the MFMAs of the unit (order pinned: hipcc would issue all DMA pieces first)
                load_a(fa0, nxt, 0);
#pragma unroll
                for (int t = 0; t < NM; ++t) {
                    // pieces [t NP / NM, (t + 1) NP / NM) before tile t's MFMAs (B3: the A pieces only — the B pieces went out
                    // over the earlier units)
                    constexpr int NPL = B3 ? PA : NP, Q0 = B3E ? PA / 2 : 0;   // (B3E: the first half went out in unit NU - 2)
#pragma unroll
                    for (int q = Q0 + t * (NPL - Q0) / NM; q < Q0 + (t + 1) * (NPL - Q0) / NM; ++q) {
                        if (q < PA) dma_a(cur, q);
                        else dma_b(cur, q - PA);
                    }
                    if (t == 1) load_b(fb0, B3 ? bnxt : nxt, 0);
                    mma1(fa1, fb1, NIP - 1, 1, t / NJH, t % NJH);
                    __builtin_amdgcn_sched_barrier(0);
                }
                PP_U_ADVANCE()
            }
            cur = nxt;
            nxt = nxt == S - 1 ? 0 : nxt + 1;
            if (B3) {
                bfill = bcur;
                bcur = bnxt;
                bnxt = bnxt == 2 ? 0 : bnxt + 1;
            }
        }
        // epilogue of this tile; the ring keeps receiving the next tile meanwhile (fa0 / fb0 already hold its first fragments)
        int tr, tc;
        pp_tile_rc_g(tile, gx, gy, 8, TERMS == 1 ? PP_U1_BAND_ROWS : 4, tr, tc);
        const int mw = tr * T::BM + wr * T::TM, nw = tc * T::BN + wc * T::TN;
#ifdef PP_STUDY_NOEPI   // (timing study builds only: the K loop alone; one store keeps the accumulators alive)
        {
            f32x4 keep = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) keep += acc[i][j];
#ifdef PP_STUDY_LDSNODEP
            keep += sink;
#endif
            if (keep[0] + keep[1] + keep[2] + keep[3] == -12345.f) d.C[lane] = keep[0];
        }
#else
        if (VEC) epilogue_wave16<MI, NJ, TERMS>(d, descale, acc, mw, nw, lane);
        else epilogue_scalar16<MI, NJ, TERMS>(d, descale, acc, mw, nw, lane);
#endif
        since_epi = VEC ? 0 : S;   // (the element-wise epilogue issues a data-dependent number of stores: full waits)
        if (!VEC) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no DMA may still target this workgroup's LDS at exit
#undef PP_U_SETUP
#undef PP_U_NEXT_TILE_IF_DONE
#undef PP_U_KTILE
#undef PP_U_KTILE_PREF
#undef PP_U_BARRIER
#undef PP_U_ADVANCE
#undef PP_U_FETCH
#endif
}

// ---------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 convolutions (Cin a multiple of the K tile, W a power of two in [16, 256]) on the persistent 256x256
// tile with ROW-SHARED A delivery.  The generic kernel copies the 256-pixel A tile into LDS once per TAP — nine LDS-DMA tiles
// per channel slice although the three taps of a filter row read the same pixels shifted by one.  Here the A buffer is
// filled once per (slice, filter row dy) and the three taps read it at row offsets dx: A copies / 3, all LDS-DMA
// instructions of the K loop - 30 %.  For the shift to be exact at the image's left / right edge the buffer holds the tile's
// 256 / W image rows with an explicit ZERO pixel before and after each of them (LDS row pitch W + 2 pixels): those rows are
// out-of-range DMA offsets — written as zeros without traffic, like the rows y + dy outside the image.  W divides the tile
// and tiles start at multiples of 256, so the LDS row of a tile row is the same for every tile.  Two A buffers alternate per
// filter row; the weight tiles keep their per-tap ring of two stages.  K order, term order and hence every result bit are
// those of pp_gemm_u_kernel<.., MODE 1, ..>; the swizzle key is conflict-free at the shifted rows too.
// ---------------------------------------------------------------------------
constexpr int H_BM = 256, H_BN = 256;
constexpr int H_A_ROWS = 288;                       // 256 + 2 * 256 / W rows used (W >= 16), 36 LDS-DMA instructions
constexpr int H_A_H = H_A_ROWS * 64;                // halfs per A buffer (36 KB)
constexpr int H_B_H = H_BN * 64;                    // halfs per weight stage (32 KB)
constexpr int H_LDS_BYTES = (2 * H_A_H + 2 * H_B_H) * 2;

template <int TERMS>
__global__ __launch_bounds__(512, 1) void pp_gemm_uh_kernel(const PpGemmDesc d, int gx, int gy) {
    constexpr bool VEC = true;   // (the host only picks this kernel when the vector epilogue applies)
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int KT = 64 / TERMS, EB = 2 * TERMS, NW = 8, MI = 8, NJ = 4, NIP = 4, NJH = 2;
    extern __shared__ __attribute__((aligned(16))) _Float16 glds[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 2, wc = w & 3, l15 = lane & 15, lq = lane >> 4;
    const int ntiles = gx * gy, nxw = (int)gridDim.x >> 3;
    const int xcd = blockIdx.x & 7, q8 = ntiles >> 3, r8 = ntiles & 7;
    const int chunk0 = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int chunk1 = chunk0 + (xcd < r8 ? q8 + 1 : q8);
    const int first = chunk0 + (int)(blockIdx.x >> 3);
    if (first >= chunk1) return;
    const __amdgpu_buffer_rsrc_t Ar = __builtin_amdgcn_make_buffer_rsrc((void*)d.A_hl, 0, (int)d.a_hl_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t Br = __builtin_amdgcn_make_buffer_rsrc((void*)d.B_hl, 0, (int)d.b_hl_bytes, 0x00020000);
    const int lr = lane >> 3;
    const int sc = (lane & 7) ^ pp_swz_key(lr);
    const unsigned cbyte = (unsigned)sc * 16u;
    const int nk = 9 * (d.conv_cin / KT);
    const int W = d.conv_w, WP = W + 2, nrows = (H_BM / W) * WP;
    _Float16* const Bbase = glds + 2 * H_A_H;

    // Per lane, constant over the tiles: its (up to five) A rows rho = (j * 8 + w) * 8 + lr of the padded buffer ->
    // tile pixel mu (or a zero pixel), as a byte offset relative to the tile's first pixel; its four weight rows.
    unsigned arel[5];
    auto a_pixel = [&](int j, int& mu) __attribute__((always_inline)) -> bool {   // row j is a pixel (not a pad / unused row)
        const int rho = (j * 8 + w) * 8 + lr, ir = rho / WP, c = rho - ir * WP;
        mu = ir * W + c - 1;
        return rho < nrows && c >= 1 && c <= W;
    };
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        int mu;
        const bool pix = a_pixel(j, mu);
        arel[j] = (unsigned)((long long)(pix ? mu : 0) * d.lda * EB) + cbyte;
    }
    // (LDS weight row rho = w 8 + lr + 64 j holds output column pp_wperm(rho) = pp_wperm(w 8 + lr) + 64 j: pp_gemm_dev.h)
    const int wrow = pp_wperm(w * 8 + lr);
    const unsigned brel = (unsigned)((long long)wrow * d.ldb * EB) + cbyte;
    const unsigned bstep = (unsigned)(64 * d.ldb * EB);
    unsigned amask = 0u, bmask = 0u;          // per tile: bit 3 j + dy = source row y + dy - 1 of A row j exists; bit j = column in range
    unsigned abase = 0u, bbase = 0u;          // per tile (scalar): byte offset of the tile's first pixel / first weight row
    int ftile = first, fkt = 0, fab = 0;      // fetch cursor: tile, K step, A buffer of the filter row being fetched
    int cky = 0, ckx = 0, cci = 0;
#define PP_H_SETUP(TILE)                                                                                             \
    {                                                                                                                \
        int tr_, tc_;                                                                                                \
        pp_tile_rc_g((TILE), gx, gy, 8, TERMS == 1 ? PP_U1_BAND_ROWS : 4, tr_, tc_);                                                                        \
        const int m0_ = tr_ * H_BM, n0_ = tc_ * H_BN;                                                                \
        abase = (unsigned)((long long)m0_ * d.lda * EB);                                                             \
        bbase = (unsigned)((long long)n0_ * d.ldb * EB);                                                             \
        amask = bmask = 0u;                                                                                          \
        _Pragma("unroll") for (int j = 0; j < 5; ++j) {                                                              \
            int mu_;                                                                                                 \
            const bool pix_ = a_pixel(j, mu_);                                                                       \
            const int m = m0_ + mu_;                                                                                 \
            if (pix_ && m < d.M) {                                                                                   \
                const int y = (m % (d.conv_h * W)) / W;                                                              \
                _Pragma("unroll") for (int t = 0; t < 3; ++t) if (y + t - 1 >= 0 && y + t - 1 < d.conv_h) amask |= 1u << (3 * j + t); \
            }                                                                                                        \
            if (j < 4 && n0_ + wrow + 64 * j < d.N) bmask |= 1u << j;                                                \
        }                                                                                                            \
        fkt = 0;                                                                                                     \
        cky = ckx = cci = 0;                                                                                         \
    }
#define PP_H_NEXT_TILE_IF_DONE()                                \
    if (fkt == nk && ftile < chunk1) {                          \
        ftile += nxw;                                           \
        if (ftile < chunk1) PP_H_SETUP(ftile) else fkt = 0;     \
    }
    auto off_a = [&](int j) __attribute__((always_inline)) -> unsigned {
        const unsigned tapoff = abase + (unsigned)((((cky - 1) * W) * d.lda + cci) * EB);
        const unsigned ok = (amask >> (3 * j + cky)) & (ftile < chunk1 ? 1u : 0u);
        return (arel[j] + tapoff) | ((ok & 1u) - 1u);
    };
    auto off_b = [&](int j) __attribute__((always_inline)) -> unsigned {
        const unsigned live = (ftile < chunk1 ? 1u : 0u) & (bmask >> j);
        return (brel + bbase + (unsigned)(((cky * 3 + ckx) * d.conv_cin + cci) * EB) + (unsigned)j * bstep) | ((live & 1u) - 1u);
    };
    auto dma_a = [&](int j) __attribute__((always_inline)) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(Ar, (lds_ptr_t)(glds + fab * H_A_H + ((j * 8 + w) * 8) * 64), 16, PP_STUDY_OFF(off_a(j)), 0, 0, 0);
    };
    auto dma_b = [&](int stage, int j) __attribute__((always_inline)) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(Br, (lds_ptr_t)(Bbase + stage * H_B_H + ((j * 8 + w) * 8) * 64), 16, PP_STUDY_OFF(off_b(j)), 0, 0, 0);
    };
#define PP_H_ADVANCE()                                                                   \
    {                                                                                    \
        const bool row_end = ckx == 2, tap_end = row_end && cky == 2;                    \
        fab = row_end ? fab ^ 1 : fab;    /* the next filter row goes to the other buffer */ \
        ckx = row_end ? 0 : ckx + 1;                                                     \
        cky = tap_end ? 0 : (row_end ? cky + 1 : cky);                                   \
        cci = tap_end ? cci + KT : cci;                                                  \
        ++fkt;                                                                           \
    }

    // LDS row (at dx = 0) of this lane's row in the wave's 16-row block q: rowlane + srow(q), the second term uniform
    // (tile row base_q + l15 with base_q a multiple of 16: W >= 16, so the block lies inside one image row)
    const int rowlane = l15 + 1;
    auto srow = [&](int q) __attribute__((always_inline)) -> int {
        const int bq = wr * 128 + q * 16;
        return (bq / W) * WP + (bq & (W - 1));
    };
    struct FA {
        h8 x[2][2];
    };
    struct FB {
        h8 x[NJH][2];
    };
    f32x4 acc[MI][NJ];
    // A fragments of row blocks 2 ip, 2 ip + 1 of tap column dx (a compile-time constant at every call site: the K loop is
    // unrolled over the three taps of a filter row) from A buffer ab.  The address of a fragment = row * 128 B + (chunk ^
    // key(row)) * 16 B depends on the lane AND on dx: it is recomputed at every read from ONE per-lane register — left to
    // itself hipcc hoists the distinct addresses out of the K loop and then spills them.
    int rl = rowlane;
    const int c0 = TERMS == 2 ? 2 * lq : lq;
    auto load_a = [&](FA& f, int ab, const int dx, int ip) __attribute__((always_inline)) {
        asm volatile("" : "+v"(rl));      // opaque: nothing derived from it is loop-invariant
        const char* base = (const char*)(glds + ab * H_A_H);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = rl + (srow(2 * ip + i) + dx);
            const int pos = c0 ^ pp_swz_key(r);
            f.x[i][0] = *(const h8*)(base + r * 128 + pos * 16);
            f.x[i][1] = *(const h8*)(base + r * 128 + (pos ^ (TERMS == 2 ? 1 : 4)) * 16);
        }
    };
    const int sw = pp_swz_key(l15);
    const int fo0 = l15 * 64 + (c0 ^ sw) * 8, fo1 = l15 * 64 + ((TERMS == 2 ? c0 + 1 : c0 + 4) ^ sw) * 8;
    auto load_b = [&](FB& f, int stage, int jh) __attribute__((always_inline)) {
        const _Float16* st = Bbase + stage * H_B_H + (wc * 64 + jh * 32) * 64;
#pragma unroll
        for (int j = 0; j < NJH; ++j) {
            f.x[j][0] = *(const h8*)(st + j * 16 * 64 + fo0);
            f.x[j][1] = *(const h8*)(st + j * 16 * 64 + fo1);
        }
    };
    auto mma1 = [&](const FA& a, const FB& b, int ip, int jh, int i, int j) __attribute__((always_inline)) {
        f32x4& c = acc[2 * ip + i][jh * NJH + j];
        // (transposed: the weight fragment is the instruction's A matrix — accumulator layout in pp_gemm_dev.h)
        if (TERMS == 2) {
            c = pp_mfma16(b.x[j][0], a.x[i][1], c);
            c = pp_mfma16(b.x[j][1], a.x[i][0], c);
            c = pp_mfma16(b.x[j][0], a.x[i][0], c);
        } else {
            c = pp_mfma16(b.x[j][0], a.x[i][0], c);
            c = pp_mfma16(b.x[j][1], a.x[i][1], c);
        }
    };
    auto mma_unit = [&](const FA& a, const FB& b, int ip, int jh) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJH; ++j) mma1(a, b, ip, jh, i, j);
    };
    const bool five = w < 4;    // rows 256 .. 287 of an A buffer belong to the fifth instruction of waves 0-3

    PP_H_SETUP(first)
    {   // K steps 0 (filter row 0: A buffer 0 + weights of tap 0) and 1 (weights of tap 1)
        PP_H_NEXT_TILE_IF_DONE()
#pragma unroll
        for (int j = 0; j < 4; ++j) dma_a(j);
        if (five) dma_a(4);
#pragma unroll
        for (int j = 0; j < 4; ++j) dma_b(0, j);
        PP_H_ADVANCE()
        PP_H_NEXT_TILE_IF_DONE()
#pragma unroll
        for (int j = 0; j < 4; ++j) dma_b(1, j);
        PP_H_ADVANCE()
    }
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // K step 0 has landed (the 4 weight pieces of step 1 may fly)
    __builtin_amdgcn_s_barrier();
    FA fa0, fa1;
    FB fb0, fb1;
    int cab = 0;                 // compute cursor: A buffer of the current filter row
    load_a(fa0, 0, -1, 0);
    load_b(fb0, 0, 0);
    int cur = 0;
    const float descale = pp_alpha(d) / (PP_A_SCALE * d.b_scale);
    for (int tile = first; tile < chunk1; tile += nxw) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // One K step with the tap column DX a compile-time constant; NDX / NAB: the next step's.
#define PP_H_STEP(DX, NDX, NAB)                                                                                       \
        {                                                                                                             \
            load_b(fb1, cur, 1);                                                                                      \
            load_a(fa1, cab, DX, 1);                                                                                  \
            mma_unit(fa0, fb0, 0, 0);                                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            mma_unit(fa0, fb1, 0, 1);                                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            load_a(fa0, cab, DX, 2);                                                                                  \
            mma_unit(fa1, fb0, 1, 0);                                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            mma_unit(fa1, fb1, 1, 1);                                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            load_a(fa1, cab, DX, 3);                                                                                  \
            mma_unit(fa0, fb0, 2, 0);                                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            mma_unit(fa0, fb1, 2, 1);                                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            mma_unit(fa1, fb0, 3, 0);                                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            /* the next K step has landed, every fragment read of this one is done */                                \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                               \
            __builtin_amdgcn_s_barrier();                                                                             \
            /* last unit | DMA of the step after next: its weights into the stage just freed and, when it opens a     \
               filter row (every third step: DX == 0 here), that row's pixels into the A buffer the row before last   \
               used */                                                                                                \
            PP_H_NEXT_TILE_IF_DONE()                                                                                  \
            const int nst = cur ^ 1;                                                                                  \
            if (DX == 0) dma_a(0);                                                                                    \
            load_a(fa0, NAB, NDX, 0);                                                                                 \
            mma1(fa1, fb1, 3, 1, 0, 0);                                                                               \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            if (DX == 0) { dma_a(1); dma_a(2); }                                                                      \
            load_b(fb0, nst, 0);                                                                                      \
            mma1(fa1, fb1, 3, 1, 0, 1);                                                                               \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            if (DX == 0) { dma_a(3); if (five) dma_a(4); }                                                            \
            dma_b(cur, 0);                                                                                            \
            mma1(fa1, fb1, 3, 1, 1, 0);                                                                               \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            dma_b(cur, 1);                                                                                            \
            dma_b(cur, 2);                                                                                            \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            dma_b(cur, 3);                                                                                            \
            PP_H_ADVANCE()                                                                                            \
            mma1(fa1, fb1, 3, 1, 1, 1);                                                                               \
            cur = nst;                                                                                                \
        }
        // (the fetch cursor runs two steps ahead: while tap dx = 0 of a filter row computes, the step being fetched is
        // tap dx = -1 of the NEXT filter row — the one that opens it)
        for (int kt = 0; kt < nk; kt += 3) {
            PP_H_STEP(-1, 0, cab)
            PP_H_STEP(0, 1, cab)
            PP_H_STEP(1, -1, cab ^ 1)
            cab ^= 1;
        }
#undef PP_H_STEP
        {
            int tr, tc;
            pp_tile_rc_g(tile, gx, gy, 8, TERMS == 1 ? PP_U1_BAND_ROWS : 4, tr, tc);
            const int mw = tr * H_BM + wr * 128, nw = tc * H_BN + wc * 64;
            if (VEC) epilogue_wave16<MI, NJ, TERMS>(d, descale, acc, mw, nw, lane);
            else epilogue_scalar16<MI, NJ, TERMS>(d, descale, acc, mw, nw, lane);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), visible to hipcc (see pp_gemm_u_kernel)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef PP_H_SETUP
#undef PP_H_NEXT_TILE_IF_DONE
#undef PP_H_ADVANCE
#endif
}

typedef TileCfg<256, 256, 2, 4, 2, 1, 0, 1> T256x256;   // 8 waves, 128x64 each, 2 x 64 KB ring
typedef TileCfg<256, 128, 4, 2, 3, 1, 1> T256x128;   // 8 waves, 64x64 each, 3 x 48 KB ring, prefetching schedule
typedef TileCfg<128, 128, 2, 2, 2, 2> T128x128;   // 4 waves, 64x64 each, 2 x 32 KB ring: two workgroups per CU
typedef TileCfg<128, 64, 2, 2, 3, 2> T128x64;     // 4 waves, 64x32 each, 3 x 24 KB ring: two workgroups per CU

template <class T, int MODE, int TERMS, bool VEC>
static int pp_u_launch_one(const PpGemmDesc& d, int persistent_slots, hipStream_t st) {
    static signed char attr_state[PP_MAX_DEVICES];   // the > 64 KB dynamic-LDS opt-in is per device (and per kernel)
    signed char& ok = attr_state[pp_cur_device()];
    if (ok == 0)
        ok = hipFuncSetAttribute((const void*)pp_gemm_u_kernel<T, MODE, TERMS, VEC>, hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES) == hipSuccess ? 1 : -1;
    if (ok < 0) return PP_ELAUNCH;
    const int gx = (d.N + T::BN - 1) / T::BN, gy = (d.M + T::BM - 1) / T::BM;
    const int nt = gx * gy;
    // persistent: one workgroup per slot (a multiple of 8) walks the tiles; fewer tiles than slots: one tile per workgroup
    const int g = nt < persistent_slots ? (nt + 7) / 8 * 8 : persistent_slots / 8 * 8;
    hipLaunchKernelGGL((pp_gemm_u_kernel<T, MODE, TERMS, VEC>), dim3(g), dim3(T::NW * 64), T::LDS_BYTES, st, d, gx, gy);
    return PP_OK;
}

template <class T, int TERMS, bool VEC>
static int pp_u_launch_tile(const PpGemmDesc& d, int mode, int slots, hipStream_t st) {
    if (mode == 0) return pp_u_launch_one<T, 0, TERMS, VEC>(d, slots, st);
    if (mode == 1) return pp_u_launch_one<T, 1, TERMS, VEC>(d, slots, st);
    if (mode == 3) {   // K slices: the hl format with the vector epilogue only (weight gradients of the training step)
        if constexpr (TERMS == 2 && VEC) return pp_u_launch_one<T, 3, TERMS, VEC>(d, slots, st);
        else return PP_EINVAL;
    }
    return pp_u_launch_one<T, 2, TERMS, VEC>(d, slots, st);
}

// all tiles with the vector epilogue, the two small tiles with the element-wise one
template <int TERMS>
static int pp_u_launch_terms(const PpGemmDesc& d, int tile, int mode, bool vec, int cus, hipStream_t st) {
    if (!vec) {
        if (tile == PP_U_128x64) return pp_u_launch_tile<T128x64, TERMS, false>(d, mode, 2 * cus, st);
        return pp_u_launch_tile<T128x128, TERMS, false>(d, mode, 2 * cus, st);
    }
    switch (tile) {
        case PP_U_256x256: return pp_u_launch_tile<T256x256, TERMS, true>(d, mode, cus, st);
        case PP_U_256x128: return pp_u_launch_tile<T256x128, TERMS, true>(d, mode, cus, st);
        case PP_U_128x128: return pp_u_launch_tile<T128x128, TERMS, true>(d, mode, 2 * cus, st);
        case PP_U_128x64: return pp_u_launch_tile<T128x64, TERMS, true>(d, mode, 2 * cus, st);
        default: return PP_EINVAL;
    }
}
#endif
