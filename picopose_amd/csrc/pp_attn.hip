// Fused multi-head self-attention of the DINOv2 blocks (model/stage1/layers/attention.py:49-62):
//   attn = softmax((q * d^-1/2) k^T);  out = attn v        per (image, head), head_dim 64, T = 257 tokens
// in one kernel, flash style: the T x T score matrix is never written.  Exact fp32 arithmetic on
// v_mfma_f32_32x32x2_f32 (attention is ~5 % of the ViT FLOPs; the big GEMMs run on the f16x3 engine).
//
// One wave = 32 query rows; a workgroup (4 waves) shares the K/V chunks (32 keys) staged in LDS.
// Scores are computed TRANSPOSED, S^T = K Q^T, so a lane owns one query column: the soft-max statistics
// are lane-local (16 keys per lane + one exchange with lane^32) and the probability tile in the MFMA
// accumulator layout IS the B operand of  O^T += V^T P^T  with no data movement (for MFMA step e the two
// lane halves contribute the keys of accumulator register e).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/picopose_hip.h"
#include "pp_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int HD = 64, KC = 32, KLD = 68;  // head dim, keys per chunk, floats per LDS row of the K chunk

// WPB waves of 32 queries per workgroup: T = 257 is nine query tiles — three workgroups of three waves, where four-wave workgroups run
// twelve (round 6).  KTAIL: the T % 32 keys past the last full chunk, when they are few (one for T = 257), are folded in by plain
// vector fma's per key (64 of them per lane) instead of a ninth chunk whose 64 MFMAs work on 31 masked keys.
constexpr int KTAIL_MAX = 4;
template <int WPB>
__global__ __launch_bounds__(64 * WPB) void attn_kernel(const float* __restrict__ qkv, int T, int heads, float scale,
                                                        float* __restrict__ out, _Float16* __restrict__ out_hl) {
    __shared__ __attribute__((aligned(16))) float Ks[KC * KLD];
    __shared__ __attribute__((aligned(16))) float Vs[KC * HD];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int b = blockIdx.y / heads, h = blockIdx.y % heads;
    const int C3 = 3 * heads * HD;
    const float* base = qkv + (size_t)b * T * C3 + h * HD;  // q at +0, k at +heads*HD, v at +2*heads*HD, token stride C3
    const int q = blockIdx.x * (32 * WPB) + w * 32 + l31;    // this lane's query row
    const int qc = q < T ? q : T - 1;

    // Q as the B operand of S^T = K Q^T: lane (d part lh, query l31) holds Q[q][32 lh + p] * scale, p = 0..31
    float qr[32];
    {
        const float* qp = base + (size_t)qc * C3 + 32 * lh;
#pragma unroll
        for (int p = 0; p < 32; p += 4) {
            const f4 v = *(const f4*)(qp + p);
            const float sc2 = scale * 1.44269504088896340736f;
            qr[p] = v.x * sc2; qr[p + 1] = v.y * sc2; qr[p + 2] = v.z * sc2; qr[p + 3] = v.w * sc2;
        }
    }
    f32x16 o0, o1;  // O^T: rows d (0..31 / 32..63), column = query (lane)
#pragma unroll
    for (int e = 0; e < 16; ++e) o0[e] = o1[e] = 0.f;
    float mrun = -INFINITY, lrun = 0.f;

    const float* kp = base + heads * HD;
    const float* vp = base + 2 * heads * HD;
    const int ktail = (T % KC) <= KTAIL_MAX ? T % KC : 0;   // keys left to the vector tail below
    const int kend = T - ktail;
    for (int k0 = 0; k0 < kend; k0 += KC) {
        __syncthreads();  // previous chunk fully consumed
        // stage K and V chunks: 32 keys x 64 floats each = 512 float4 per tensor
#pragma unroll
        for (int idx = tid; idx < KC * HD / 4; idx += 64 * WPB) {
            const int row = idx >> 4, c4 = (idx & 15) * 4;
            f4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
            if (k0 + row < T) {
                kv = *(const f4*)(kp + (size_t)(k0 + row) * C3 + c4);
                vv = *(const f4*)(vp + (size_t)(k0 + row) * C3 + c4);
            }
            *(f4*)(Ks + row * KLD + c4) = kv;
            *(f4*)(Vs + row * HD + c4) = vv;
        }
        __syncthreads();
        // S^T tile (keys x queries): A = K chunk (lane: key l31, d part lh), B = Q
        f32x16 s;
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
        for (int p4 = 0; p4 < 32; p4 += 4) {
            const f4 kf = *(const f4*)(Ks + l31 * KLD + 32 * lh + p4);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qr[p4], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qr[p4 + 1], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qr[p4 + 2], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qr[p4 + 3], s, 0, 0, 0);
        }
        // register e of this lane = key k0 + (e&3) + 8(e>>2) + 4 lh, query l31; mask keys past T (only the last chunk has any)
        float mx = -INFINITY;
        if (k0 + KC <= T) {
#pragma unroll
            for (int e = 0; e < 16; ++e) mx = fmaxf(mx, s[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key = k0 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                s[e] = key < T ? s[e] : -INFINITY;
                mx = fmaxf(mx, s[e]);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float mnew = fmaxf(mrun, mx);
        // the scores live in the base-2 domain (q was scaled by scale * log2 e): an exponential is ONE v_exp_f32 instead of libm's
        // ~20-instruction expf — on fp32 MFMAs every vector instruction is taken from the matrix pipe's issue slots (csrc/pp_gemm_f.hip)
        const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);  // 0 for the first chunk (mrun = -inf)
        float ls = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            s[e] = __builtin_amdgcn_exp2f(s[e] - mnew);
            ls += s[e];
        }
        ls += __shfl_xor(ls, 32);
        lrun = lrun * alpha + ls;
        mrun = mnew;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            o0[e] *= alpha;
            o1[e] *= alpha;
        }
        // O^T += V^T P^T: step e pairs key(e, lh=0) (lanes 0-31) with key(e, lh=1) (lanes 32-63);
        // A = V^T: lane (d = l31 [+32], key part lh) reads V[key(e, lh)][d]
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int key = (e & 3) + 8 * (e >> 2) + 4 * lh;
            const float v0 = Vs[key * HD + l31], v1 = Vs[key * HD + 32 + l31];
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, s[e], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, s[e], o1, 0, 0, 0);
        }
    }
    for (int kx = kend; kx < T; ++kx) {   // (wave-uniform addresses: the loads below are broadcasts)
        const float* kr = kp + (size_t)kx * C3 + 32 * lh;
        float sp = 0.f;
#pragma unroll
        for (int p = 0; p < 32; p += 4) {
            const f4 kf = *(const f4*)(kr + p);
            sp = fmaf(kf.x, qr[p], sp);
            sp = fmaf(kf.y, qr[p + 1], sp);
            sp = fmaf(kf.z, qr[p + 2], sp);
            sp = fmaf(kf.w, qr[p + 3], sp);
        }
        const float sc = sp + __shfl_xor(sp, 32);            // the score of (key kx, query l31), the same in both lane halves
        const float mnew = fmaxf(mrun, sc);
        const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);
        const float pk = __builtin_amdgcn_exp2f(sc - mnew);
        lrun = lrun * alpha + pk;
        mrun = mnew;
        const float* vr = vp + (size_t)kx * C3 + 4 * lh;
#pragma unroll
        for (int g = 0; g < 4; ++g) {                        // accumulator register 4 g + i holds d = 8 g + 4 lh + i (+ 32 for o1)
            const f4 v0 = *(const f4*)(vr + 8 * g), v1 = *(const f4*)(vr + 32 + 8 * g);
            o0[4 * g] = fmaf(v0.x, pk, o0[4 * g] * alpha);
            o0[4 * g + 1] = fmaf(v0.y, pk, o0[4 * g + 1] * alpha);
            o0[4 * g + 2] = fmaf(v0.z, pk, o0[4 * g + 2] * alpha);
            o0[4 * g + 3] = fmaf(v0.w, pk, o0[4 * g + 3] * alpha);
            o1[4 * g] = fmaf(v1.x, pk, o1[4 * g] * alpha);
            o1[4 * g + 1] = fmaf(v1.y, pk, o1[4 * g + 1] * alpha);
            o1[4 * g + 2] = fmaf(v1.z, pk, o1[4 * g + 2] * alpha);
            o1[4 * g + 3] = fmaf(v1.w, pk, o1[4 * g + 3] * alpha);
        }
    }
    if (q < T) {
        const float inv = 1.0f / lrun;
        const size_t obase = ((size_t)b * T + q) * (heads * HD) + h * HD;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int d = (e & 3) + 8 * (e >> 2) + 4 * lh;
            const float a0 = o0[e] * inv, a1 = o1[e] * inv;
            if (out) {
                out[obase + d] = a0;
                out[obase + 32 + d] = a1;
            }
            if (out_hl) {  // f16x3 operand of the output projection (obase is a multiple of 8)
                _Float16 hh, ll;
                pp_split_f16_chk(a0, hh, ll);
                out_hl[2 * obase + pp_hl_col(d, 0)] = hh;
                out_hl[2 * obase + pp_hl_col(d, 1)] = ll;
                pp_split_f16_chk(a1, hh, ll);
                out_hl[2 * obase + pp_hl_col(32 + d, 0)] = hh;
                out_hl[2 * obase + pp_hl_col(32 + d, 1)] = ll;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// f16x3 variant (the engine's default arithmetic): q, k, v and the probabilities are split into two
// fp16 terms (22 bits) and every product runs as 3 x v_mfma_f32_32x32x16_f16 with fp32 accumulation; the soft-max
// statistics stay fp32.  Same transposed-score structure as above:
//   S^T = K Q^T   A = K chunk (lane: key l31, 8 consecutive d), B = Q (registers, pre-scaled)      4 x 3 MFMA
//   O^T += V^T P^T A = V^T (lane: d l31, 8 keys), B = P in accumulator layout (registers 8s..8s+7)  4 x 3 MFMA
// The k-slot <-> key assignment of the second product is free as long as both operands agree: slot (lh, i) of
// step s is key 16 s + 4 lh + (i & 3) + 8 (i >> 2) — exactly the keys accumulator registers 8s..8s+7 of a lane
// hold — so P needs no data movement and V is staged transposed with its keys in that order.
// A workgroup = WPB waves x 32 queries sharing the staged K / V chunk; the host picks WPB so the query tiles
// divide evenly (T = 257: 9 tiles = 3 workgroups of 3 waves).  The output leaves through LDS as full rows.
// ---------------------------------------------------------------------------
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
constexpr int KHLD = 72;   // halfs per K row in LDS (64 + 8 pad: 144-byte stride, conflict-free ds_read_b128)
constexpr int VLD = 96;    // halfs per V row in LDS (64 d + 32 pad: 192-byte stride = 48 banks: the four rows a
                           // ds_read_b64_tr_b16 half-wave touches sit on disjoint bank quarters)
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
constexpr int OLD = 68;    // floats per staged output row
constexpr float P_SCALE = 1024.f;

__device__ __forceinline__ int vt_slot(int key) {  // key (0..31) of a chunk -> k-slot order of the P registers
    const int s = key >> 4, r = key & 15;
    return s * 16 + ((r >> 2) & 1) * 8 + (r & 3) + 4 * (r >> 3);
}

// HLIN: q, k, v arrive as the hl operand the qkv GEMM epilogue wrote (no split work here at all: fragments and the
// K / V chunks are 16-byte copies); otherwise fp32 qkv, split while staging.  Both give the same bits.
// TERMS = 1 (HLIN only): the h operand format (plain fp16 q, k, v, probabilities) — one MFMA per product
// (ops.PRECISION = "f16", BASELINE configs[4]); the lo-term registers / LDS planes / MFMAs compile away.
// RING = S > 0 (operand input only, WPB = waves per workgroup as a template argument): the K / V chunks go global -> LDS by LDS-DMA
// (`buffer_load_dwordx4 ... lds`) into a ring of S stages, S - 1 chunks ahead, with ONE barrier per chunk and no staging registers —
// round 4's study builds put a quarter of the kernel into the latency of the register-staged fetch (one chunk ahead, two barriers
// per chunk).  A chunk row is the 64 d of a key as the operand format stores them (hl: 16 chunks of 16 bytes, hi / lo groups
// alternating; h: 8 chunks); the LDS image has NO row padding (a DMA writes lane-linearly) — bank conflicts are removed by an XOR
// key on the chunk position instead, applied to the per-lane SOURCE address: K rows (read by ds_read_b128, lane = key) key(r) = r & 15
// (hl) / r >> 1 & 7 (h); V rows (read by ds_read_b64_tr_b16, four consecutive slots per half-wave) key(r) = (r & 1) | (r >> 1 & 1) << 3
// (hl) / (r >> 1 & 1) << 2 (h).  V rows are the key SLOTS of the second product (vt_slot), also by source addressing.
template <int TERMS>
struct RingGeom {
    static constexpr int ROWH = 64 * TERMS;                  // halfs per chunk row (128 / 256 bytes)
    static constexpr int CPR = ROWH / 8;                     // 16-byte chunks per row
    static constexpr int RPP = 64 / CPR;                     // rows per DMA piece (1 KB per wave instruction)
    static constexpr int NPT = KC / RPP;                     // pieces per tensor and chunk
    static constexpr int STAGE_H = 2 * KC * ROWH;            // halfs per stage: K rows then V rows
};

// RING = 8 with WPB = 9 ("resident"): T = 257 .. 260 — one workgroup of NINE waves per (image, head), one wave per query tile; all eight
// 32-key chunks of K and V are copied ONCE (128 KB of LDS in the hl format, every piece issued before the first product), each chunk
// has ONE barrier when it is first read and the stages are never refilled: the deepest prefetch the ring can have, a third of the
// K / V copies (three workgroups per pair staged the same chunks before), and 2 304 workgroups on 256 CUs are nine full rounds.
template <bool HLIN, int TERMS = 2, int RING = 0, int WPB = 0>
__global__ __launch_bounds__(WPB > 4 ? 64 * WPB : 256) __attribute__((amdgpu_waves_per_eu(3, 3))) void attn_f16x3_kernel(const void* __restrict__ qkv_any, int T, int heads, float scale,
                                                         float* __restrict__ out, _Float16* __restrict__ out_hl,
                                                         float* __restrict__ lse, int ntx, int npairs) {
    static_assert(RING == 0 || (HLIN && WPB >= 1 && (WPB <= 4 || (WPB == 9 && RING == 8))), "the LDS-DMA ring copies operand rows");
    constexpr bool RESIDENT = RING == 8 && WPB == 9;
    using RG = RingGeom<TERMS>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16* Kh = (_Float16*)smem;            // [KC][KHLD]
    _Float16* Kl = Kh + KC * KHLD;
    _Float16* Vh = Kl + KC * KHLD;             // [KC key slots][VLD]: row = k-slot of the key, d contiguous
    _Float16* Vl = Vh + KC * VLD;
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, w = RING ? __builtin_amdgcn_readfirstlane(tid >> 6) : tid >> 6, l31 = lane & 31, lh = lane >> 5;
    // 1-D grid, XCD-aware: workgroup ids go round-robin over the 8 XCDs (each with its own L2), so the ntx workgroups of one (image, head)
    // — which stage the SAME K / V chunks — take ids of one residue mod 8: id = xcd + 8 (ntx (pair / 8) + tile), pair = 8 (..) + xcd.
    // (On a (tile, pair) grid the three workgroups of a pair sat on three XCDs: FETCH 2.3 x the qkv bytes.)
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3, bx = jj % ntx, pair = (jj / ntx) * 8 + xcd;
    if (pair >= npairs) return;
    const int b = pair / heads, h = pair % heads;
    const int C3 = 3 * heads * HD;
    const float* base = (const float*)qkv_any + (size_t)b * T * C3 + h * HD;            // fp32 input
    const _Float16* hbase = (const _Float16*)qkv_any + ((size_t)b * T * C3 + h * HD) * TERMS;  // operand input (h*HD % 8 == 0)
    const int q = (bx * (nthr >> 6) + w) * 32 + l31;
    const int qc = q < T ? q : T - 1;

    // Q fragments: step s holds d = 16 s + 8 lh .. + 7 of query l31 (unscaled: the scores are scaled after the MFMAs)
    h8 qh[4], ql[4];
    if (HLIN) {
        const _Float16* qp = hbase + (size_t)qc * C3 * TERMS;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qh[s] = *(const h8*)(qp + TERMS * (16 * s + 8 * lh));
            if (TERMS == 2) ql[s] = *(const h8*)(qp + 2 * (16 * s + 8 * lh) + 8);
            else ql[s] = h8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    } else {
        const float* qp = base + (size_t)qc * C3;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const f4 a = *(const f4*)(qp + 16 * s + 8 * lh), c = *(const f4*)(qp + 16 * s + 8 * lh + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                _Float16 hh, ll;
                pp_split_f16_chk(a[i], hh, ll);
                qh[s][i] = hh;
                ql[s][i] = ll;
                pp_split_f16_chk(c[i], hh, ll);
                qh[s][4 + i] = hh;
                ql[s][4 + i] = ll;
            }
        }
    }
    f32x16 o0, o1;
#pragma unroll
    for (int e = 0; e < 16; ++e) o0[e] = o1[e] = 0.f;
    float mrun = -INFINITY, lrun = 0.f;
    const float* kp = base + heads * HD;
    const float* vp = base + 2 * heads * HD;
    // scores are kept in the base-2 domain (scaled by log2 e): the soft-max exponentials are single v_exp_f32 instructions
    const float S_DESCALE = scale * 1.44269504088896340736f / (PP_A_SCALE * PP_A_SCALE);
    // (lrun is kept in the units of the probability operand, 1024 p: see SOFTMAX_STEP)
    constexpr float O_RESCALE = 1.0f / PP_A_SCALE;
    // One online-soft-max step on the score tile `sacc` (raw MFMA sums, keys k0 ..): the kernel's time is this vector work (MFMA-busy 0.21),
    // so it is cut to the bone (round 5, ~210 -> ~150 vector instructions per chunk):
    //   * the maximum is taken on the RAW sums (the scale is positive) and the scale rides in the exponential's argument — one fma per
    //     score instead of a multiply and a subtract;
    //   * the probability operand is 1024 p: the factor is an addend of the same fma (exp2(x + 10)), and the running sum `lrun` is kept
    //     in those units (the final division and the log-sum-exp take the 2^10 out) — no multiply per probability;
    //   * the accumulators are rescaled only when some query's maximum moved in this chunk (wave-uniform test): after the first
    //     chunks it rarely does.
#define SOFTMAX_STEP(K0)                                                                                          \
    float mx = -INFINITY;                                                                                         \
    if ((K0) + KC > Tm) { /* (uniform) the last, partial chunk: keys past the chunked range are masked */       \
        _Pragma("unroll") for (int e = 0; e < 16; ++e) {                                                          \
            const int key = (K0) + (e & 3) + 8 * (e >> 2) + 4 * lh;                                               \
            sacc[e] = key < Tm ? sacc[e] : -INFINITY;                                                             \
        }                                                                                                         \
    }                                                                                                             \
    _Pragma("unroll") for (int e = 0; e < 16; ++e) mx = fmaxf(mx, sacc[e]);                                       \
    mx = fmaxf(mx, __shfl_xor(mx, 32));                                                                           \
    const float mnew = fmaxf(mrun, mx * S_DESCALE);                                                               \
    const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);                                                      \
    const float shift = 10.0f - mnew;                                                                             \
    float ls = 0.f;                                                                                               \
    h8 ph[2], pl[2];                                                                                              \
    _Pragma("unroll") for (int e = 0; e < 16; ++e) {                                                              \
        const float x = __builtin_amdgcn_exp2f(fmaf(sacc[e], S_DESCALE, shift)); /* 1024 p <= 1024 */             \
        ls += x;                                                                                                  \
        const _Float16 hh = (_Float16)x;                                                                          \
        ph[e >> 3][e & 7] = hh;                                                                                   \
        pl[e >> 3][e & 7] = (_Float16)(x - (float)hh);                                                            \
    }                                                                                                             \
    ls += __shfl_xor(ls, 32);                                                                                     \
    lrun = lrun * alpha + ls;                                                                                     \
    mrun = mnew;                                                                                                  \
    if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {                                                        \
        _Pragma("unroll") for (int e = 0; e < 16; ++e) {                                                          \
            o0[e] *= alpha;                                                                                       \
            o1[e] *= alpha;                                                                                       \
        }                                                                                                         \
    }

    // Keys are consumed in chunks of KC = 32 through LDS; a tail of up to TAILMAX keys (T = 257 = 8 x 32 + 1: the chunk
    // loop would spend a ninth pass, 11 % of the kernel, on ONE key) is folded in afterwards on the VALU.
    constexpr int TAILMAX = 4;
    const int ntail = (T % KC) <= TAILMAX ? T % KC : 0, Tm = T - ntail;
    // The chunk after the current one is fetched into registers while the current one is multiplied (the global -> LDS
    // staging used to sit, latency exposed, between two barriers of every chunk).
    constexpr int NIT = HLIN ? 2 : 4;            // items per thread: 256 (hi, lo) pairs / 512 float4 per tensor, >= 128 threads
    h8 pk[HLIN ? 2 * NIT : 1], pv[HLIN ? 2 * NIT : 1];
    f4 fk[HLIN ? 1 : NIT], fv[HLIN ? 1 : NIT];
    auto fetch = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * nthr;
            if (HLIN) {
                const int row = idx >> 3, g = idx & 7;
                h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                pk[2 * it] = pk[2 * it + 1] = pv[2 * it] = pv[2 * it + 1] = z;
                if (idx < KC * (HD / 8) && k0 + row < Tm) {
                    const _Float16* kp2 = hbase + ((size_t)(k0 + row) * C3 + heads * HD + 8 * g) * TERMS;
                    const _Float16* vp2 = hbase + ((size_t)(k0 + row) * C3 + 2 * heads * HD + 8 * g) * TERMS;
                    pk[2 * it] = *(const h8*)kp2;
                    pv[2 * it] = *(const h8*)vp2;
                    if (TERMS == 2) {
                        pk[2 * it + 1] = *(const h8*)(kp2 + 8);
                        pv[2 * it + 1] = *(const h8*)(vp2 + 8);
                    }
                }
            } else {
                const int row = idx >> 4, c4 = (idx & 15) * 4;
                fk[it] = fv[it] = f4{0.f, 0.f, 0.f, 0.f};
                if (idx < KC * (HD / 4) && k0 + row < Tm) {
                    fk[it] = *(const f4*)(kp + (size_t)(k0 + row) * C3 + c4);
                    fv[it] = *(const f4*)(vp + (size_t)(k0 + row) * C3 + c4);
                }
            }
        }
    };
    auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * nthr;
            if (HLIN) {
                if (idx >= KC * (HD / 8)) continue;
                const int row = idx >> 3, g = idx & 7, slot = vt_slot(row);   // K [key][d], V [slot(key)][d]
                *(h8*)(Kh + row * KHLD + 8 * g) = pk[2 * it];
                *(h8*)(Vh + slot * VLD + 8 * g) = pv[2 * it];
                if (TERMS == 2) {
                    *(h8*)(Kl + row * KHLD + 8 * g) = pk[2 * it + 1];
                    *(h8*)(Vl + slot * VLD + 8 * g) = pv[2 * it + 1];
                }
            } else {
                if (idx >= KC * (HD / 4)) continue;
                const int row = idx >> 4, c4 = (idx & 15) * 4, slot = vt_slot(row);
                h4 khh, kll, vhh, vll;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    _Float16 hh, ll;
                    pp_split_f16_chk(fk[it][i], hh, ll);
                    khh[i] = hh;
                    kll[i] = ll;
                    pp_split_f16_chk(fv[it][i], hh, ll);
                    vhh[i] = hh;
                    vll[i] = ll;
                }
                *(h4*)(Kh + row * KHLD + c4) = khh;
                *(h4*)(Kl + row * KHLD + c4) = kll;
                *(h4*)(Vh + slot * VLD + c4) = vhh;
                *(h4*)(Vl + slot * VLD + c4) = vll;
            }
        }
    };
    // ---- LDS-DMA ring (RING > 0): addressing of this lane's pieces.  Piece p of a chunk: tensor p / NPT (0 = K, 1 = V), LDS rows
    // (p % NPT) RPP .. + RPP - 1; lane L fills chunk position L % CPR of row L / CPR.  Wave w issues pieces w, w + WPB, ... — NPW per chunk,
    // wrapping past the last piece (a duplicate copy of the same bytes to the same place) so that every wave's count is the same constant.
    constexpr int NPW = RING ? (2 * RG::NPT + (WPB > 0 ? WPB : 1) - 1) / (WPB > 0 ? WPB : 1) : 1;
    unsigned poff[NPW];            // byte offset of the piece's source chunk relative to (key 0 of the chunk, this head's q column 0)
    int prow[NPW];                 // key of the chunk (0 .. 31) the lane copies
    const __amdgpu_buffer_rsrc_t Hr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)hbase, 0, RING ? (int)(((size_t)(T - 1) * C3 + 2 * heads * HD + HD) * 2 * TERMS) : 0, 0x00020000);
    if (RING) {
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            const int p = (w + (WPB > 0 ? WPB : 1) * i) % (2 * RG::NPT), isv = p / RG::NPT;
            const int rho = (p % RG::NPT) * RG::RPP + lane / RG::CPR, pos = lane % RG::CPR;
            int key, xk;
            if (isv) {     // LDS row rho = key slot: the key whose vt_slot is rho; V chunk key
                const int r16 = rho & 15;
                key = (rho & 16) | (((r16 >> 2) & 1) << 3) | ((r16 >> 3) << 2) | (r16 & 3);
                xk = TERMS == 2 ? ((rho & 1) | (((rho >> 1) & 1) << 3)) : (((rho >> 1) & 1) << 2);
            } else {
                key = rho;
                xk = TERMS == 2 ? (rho & 15) : ((rho >> 1) & 7);
            }
            prow[i] = key;
            poff[i] = (unsigned)(((size_t)key * C3 + (size_t)(1 + isv) * heads * HD) * 2 * TERMS + (unsigned)((pos ^ xk) * 16));
        }
    }
    // pieces [i0, i1) of chunk k0 into `stage` (the chunk loop spreads a chunk's pieces over its six MFMA groups: an LDS-DMA instruction
    // costs the issuing wave 60-180 cycles, and the waves of a workgroup leave the barrier together)
    auto ring_pieces = [&](int k0, int stage, int i0, int i1) __attribute__((always_inline)) {
        _Float16* st = (_Float16*)smem + stage * RG::STAGE_H;
        const unsigned kbytes = (unsigned)((size_t)k0 * C3 * 2 * TERMS);
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            if (i < i0 || i >= i1) continue;
            const int p = (w + (WPB > 0 ? WPB : 1) * i) % (2 * RG::NPT);
            const unsigned off = (k0 + prow[i] < Tm) ? poff[i] + kbytes : 0xFFFFFFFFu;      // keys past the chunked range: zeros, no traffic
            __builtin_amdgcn_raw_ptr_buffer_load_lds(Hr, (__attribute__((address_space(3))) void*)(st + p * 512), 16, off, 0, 0, 0);
        }
    };
    auto ring_issue = [&](int k0, int stage) __attribute__((always_inline)) { ring_pieces(k0, stage, 0, NPW); };
    // slot g of 6 (four S-product steps, two P V steps) carries pieces [g NPW / 6, (g + 1) NPW / 6)
    auto ring_slot = [&](int k0, int stage, int g) __attribute__((always_inline)) {
        if ((g * NPW) / 6 != ((g + 1) * NPW) / 6) {
            __builtin_amdgcn_sched_barrier(0);
            ring_pieces(k0, stage, (g * NPW) / 6, ((g + 1) * NPW) / 6);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if constexpr (RING > 0) {
        const int nch = (Tm + KC - 1) / KC;
#pragma unroll
        for (int c = 0; c < (RESIDENT ? RING : RING - 1); ++c) ring_issue(c * KC, c);
        int stage = 0, fill = RING - 1;
        // K fragment / V fragment addressing (halfs, inside a stage)
        const int kx = TERMS == 2 ? (l31 & 15) : ((l31 >> 1) & 7);
        for (int c = 0; c < nch; ++c) {
            const int k0 = c * KC;
            // chunk c has landed: this wave's pieces (counted wait: the pieces of the S - 2 younger chunks may fly), every wave's
            // (barrier) — and every wave is done with chunk c - 1, whose stage now takes chunk c + S - 1
            if constexpr (RESIDENT) {      // chunk c of 8 issued up front: the 7 - c younger chunks' pieces may fly
                switch (c) {
                    case 0: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(7 * NPW) : "memory"); break;
                    case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * NPW) : "memory"); break;
                    case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * NPW) : "memory"); break;
                    case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * NPW) : "memory"); break;
                    case 4: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NPW) : "memory"); break;
                    case 5: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPW) : "memory"); break;
                    case 6: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 * NPW) : "memory"); break;
                    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                }
            } else {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - 2) * NPW) : "memory");
            }
            __syncthreads();
            // the pieces of chunk c + S - 1, back to back: spread over the chunk's six MFMA groups (ring_slot, kept for A/B under
            // PP_ATTN_SPREAD builds) they cost MORE — 264 against 236 us at 192 images: the sched_barriers that pin them also pin the
            // fragment reads behind the MFMAs they feed
#ifndef PP_ATTN_SPREAD
            if constexpr (!RESIDENT) ring_issue(k0 + (RING - 1) * KC, fill);
#endif
            const int kn = k0 + (RING - 1) * KC;
            (void)kn;
            const _Float16* Kst = (const _Float16*)smem + stage * RG::STAGE_H;
            const _Float16* Vst = Kst + KC * RG::ROWH;
            f32x16 sacc;
#pragma unroll
            for (int e = 0; e < 16; ++e) sacc[e] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int ch = TERMS == 2 ? 2 * (2 * s + lh) : 2 * s + lh;
                const h8 kh = *(const h8*)(Kst + l31 * RG::ROWH + ((ch ^ kx) * 8));
                if (TERMS == 2) {
                    const h8 kl = *(const h8*)(Kst + l31 * RG::ROWH + (((ch + 1) ^ kx) * 8));
                    sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[s], sacc, 0, 0, 0);
                    sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[s], sacc, 0, 0, 0);
                }
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[s], sacc, 0, 0, 0);
#ifdef PP_ATTN_SPREAD
                ring_slot(kn, fill, s);
#endif
            }
            SOFTMAX_STEP(k0)
            // V^T fragments by transposing reads from the [slot][d chunks] image: lane (d = l31 [+32], lh) gets slots 16 s + 8 lh .. + 7
            auto vfrag = [&](int term, int s, int dhalf) __attribute__((always_inline)) -> h8 {
                const int col = 32 * dhalf + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
                const int row = 16 * s + 8 * lh + ((lane & 15) >> 2);
                const int ch = TERMS == 2 ? 2 * (col >> 3) + term : (col >> 3);
                auto at = [&](int r) __attribute__((always_inline)) -> const _Float16* {
                    const int xv = TERMS == 2 ? ((r & 1) | (((r >> 1) & 1) << 3)) : (((r >> 1) & 1) << 2);
                    return Vst + r * RG::ROWH + ((ch ^ xv) * 8) + (col & 7);
                };
                const fp16x4_t lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)at(row));
                const fp16x4_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)at(row + 4));
                h8 r;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    r[e] = (_Float16)lo4[e];
                    r[4 + e] = (_Float16)hi4[e];
                }
                return r;
            };
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const h8 v0h = vfrag(0, s, 0), v1h = vfrag(0, s, 1);
                if (TERMS == 2) {
                    const h8 v0l = vfrag(1, s, 0), v1l = vfrag(1, s, 1);
                    o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0l, ph[s], o0, 0, 0, 0);
                    o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0h, pl[s], o0, 0, 0, 0);
                    o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1l, ph[s], o1, 0, 0, 0);
                    o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1h, pl[s], o1, 0, 0, 0);
                }
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0h, ph[s], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1h, ph[s], o1, 0, 0, 0);
#ifdef PP_ATTN_SPREAD
                ring_slot(kn, fill, 4 + s);
#endif
            }
            fill = stage;
            stage = stage + 1 == RING ? 0 : stage + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the zero-filling pieces of the chunks past the end: no DMA may target LDS below)
    } else {
    if (Tm > 0) fetch(0);
    for (int k0 = 0; k0 < Tm; k0 += KC) {
        __syncthreads();                       // the previous chunk has been read by every wave
        commit();
        __syncthreads();
        if (k0 + KC < Tm) fetch(k0 + KC);      // in flight under this chunk's MFMAs and soft-max
        f32x16 sacc;
#pragma unroll
        for (int e = 0; e < 16; ++e) sacc[e] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const h8 kh = *(const h8*)(Kh + l31 * KHLD + 16 * s + 8 * lh);
            if (TERMS == 2) {
                const h8 kl = *(const h8*)(Kl + l31 * KHLD + 16 * s + 8 * lh);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[s], sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[s], sacc, 0, 0, 0);
            }
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[s], sacc, 0, 0, 0);
        }
        SOFTMAX_STEP(k0)
        // V^T fragments by transposing reads from the [slot][d] tile: lane (d = l31 [+32], lh) gets slots 16 s + 8 lh .. + 7
        auto vfrag = [&](const _Float16* V, int s, int dhalf) __attribute__((always_inline)) -> h8 {
            const int col = 32 * dhalf + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
            const int row = 16 * s + 8 * lh + ((lane & 15) >> 2);
            const _Float16* p = V + row * VLD + col;
            const fp16x4_t lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)p);
            const fp16x4_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(p + 4 * VLD));
            h8 r;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                r[e] = (_Float16)lo4[e];
                r[4 + e] = (_Float16)hi4[e];
            }
            return r;
        };
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const h8 v0h = vfrag(Vh, s, 0), v1h = vfrag(Vh, s, 1);
            if (TERMS == 2) {
                const h8 v0l = vfrag(Vl, s, 0), v1l = vfrag(Vl, s, 1);
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0l, ph[s], o0, 0, 0, 0);
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0h, pl[s], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1l, ph[s], o1, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1h, pl[s], o1, 0, 0, 0);
            }
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0h, ph[s], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1h, ph[s], o1, 0, 0, 0);
        }
    }
    }   // (RING == 0)
    // ---- tail keys on the VALU: one online-soft-max step per key with exact fp32 products of the operands' values
    // (hi + lo is exact in fp32).  Lane (l31, lh) holds dims 16 s + 8 lh + i of its query; its accumulator register e
    // holds O^T[d = (e & 3) + 8 (e >> 2) + 4 lh (+ 32)][query l31].
    if (ntail > 0) {
        __syncthreads();                       // every wave is done with the chunk buffers
        float* tk = (float*)smem;              // [TAILMAX][64] values of 4 k, then [TAILMAX][64] of 4 v
        float* tv = tk + TAILMAX * HD;
        for (int idx = tid; idx < ntail * HD; idx += nthr) {
            const int r = idx / HD, dd = idx - r * HD;
            float kf, vf;
            if (HLIN && TERMS == 1) {
                kf = (float)hbase[(size_t)(Tm + r) * C3 + heads * HD + dd];
                vf = (float)hbase[(size_t)(Tm + r) * C3 + 2 * heads * HD + dd];
            } else if (HLIN) {
                const _Float16* kp2 = hbase + ((size_t)(Tm + r) * C3 + heads * HD) * 2 + pp_hl_col(dd, 0);
                const _Float16* vp2 = hbase + ((size_t)(Tm + r) * C3 + 2 * heads * HD) * 2 + pp_hl_col(dd, 0);
                kf = (float)kp2[0] + (float)kp2[8];
                vf = (float)vp2[0] + (float)vp2[8];
            } else {
                _Float16 hh, ll;
                pp_split_f16_chk(kp[(size_t)(Tm + r) * C3 + dd], hh, ll);
                kf = (float)hh + (float)ll;
                pp_split_f16_chk(vp[(size_t)(Tm + r) * C3 + dd], hh, ll);
                vf = (float)hh + (float)ll;
            }
            tk[idx] = kf;
            tv[idx] = vf;
        }
        __syncthreads();
        for (int r = 0; r < ntail; ++r) {
            float dot = 0.f;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    dot = fmaf((float)qh[s4][i] + (float)ql[s4][i], tk[r * HD + 16 * s4 + 8 * lh + i], dot);
            dot += __shfl_xor(dot, 32);
            const float sv = dot * S_DESCALE;
            const float mnew = fmaxf(mrun, sv);
            const float alpha = __builtin_amdgcn_exp2f(mrun - mnew), pw = __builtin_amdgcn_exp2f(sv - mnew + 10.0f);   // 1024 p: the MFMA path's units
            lrun = lrun * alpha + pw;
            mrun = mnew;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int dd = (e & 3) + 8 * (e >> 2) + 4 * lh;
                o0[e] = fmaf(tv[r * HD + dd], pw, o0[e] * alpha);
                o1[e] = fmaf(tv[r * HD + 32 + dd], pw, o1[e] * alpha);
            }
        }
    }
    // (training) the base-2 log-sum-exp of the query's scaled scores: all the adjoint needs to recompute its probabilities
    if (lse && q < T && lh == 0) lse[((size_t)b * heads + h) * T + q] = mrun + __builtin_amdgcn_logf(lrun) - 10.0f;   // (lrun counts 1024 p)
    // output through LDS: O^T registers (lane = query) -> rows of 64 floats per query, written as full lines
    __syncthreads();
    float* Os = (float*)smem + w * 32 * OLD;
    const float inv = O_RESCALE / lrun;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f4 a, c;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[i] = o0[4 * g + i] * inv;
            c[i] = o1[4 * g + i] * inv;
        }
        *(f4*)(Os + l31 * OLD + 8 * g + 4 * lh) = a;
        *(f4*)(Os + l31 * OLD + 32 + 8 * g + 4 * lh) = c;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave reads only what it wrote itself
    const int qr = lane >> 1, half = lane & 1;
    const int qo = (bx * (nthr >> 6) + w) * 32 + qr;
    if (qo < T) {
        const size_t obase = ((size_t)b * T + qo) * (heads * HD) + h * HD + 32 * half;
        const float* src = Os + qr * OLD + 32 * half;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f4 v = *(const f4*)(src + 4 * j);
            if (out) *(f4*)(out + obase + 4 * j) = v;
            if (out_hl) {  // columns obase + 4 j .. + 3 (obase % 32 == 0): half a group of 8
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                h4 hh, ll;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    _Float16 a, c;
                    pp_split_f16_chk(v[i], a, c);
                    hh[i] = a;
                    ll[i] = c;
                }
                if (TERMS == 2) {
                    _Float16* hp = out_hl + 2 * obase + pp_hl_col(4 * j, 0);
                    *(h4*)hp = hh;
                    *(h4*)(hp + 8) = ll;
                } else {
                    *(h4*)(out_hl + obase + 4 * j) = hh;
                }
            }
        }
    }
}

}  // namespace

PP_SAT_SETTER(pp_sat_set_attn)

extern "C" {

static int attention_launch(const void* qkv, bool hl_in, int B, int T, int heads, int head_dim, float scale, int prec, float* out,
                            void* out_hl, void* stream, int terms = 2, float* lse = nullptr) {
    if (!qkv || (!out && !out_hl) || B <= 0 || T <= 0 || heads <= 0 || (hl_in && prec != PP_PREC_F16X3)) return PP_EINVAL;
    if (terms == 1 && !hl_in) return PP_EINVAL;
    if (head_dim != HD || ((uintptr_t)qkv % 16) != 0) return PP_EINVAL;
    if (prec == PP_PREC_F16X3) {
        // waves per workgroup: the split of the ceil(T/32) query tiles that wastes the fewest wave slots
        const int tiles = (T + 31) / 32;
        int wpb = 4, best = 1 << 30;
        for (int c = 4; c >= 2; --c) {
            const int waste = (tiles + c - 1) / c * c - tiles;
            if (waste < best) {
                best = waste;
                wpb = c;
            }
        }
        const size_t kv = (size_t)(2 * KC * KHLD + 2 * KC * VLD) * sizeof(_Float16), os = (size_t)wpb * 32 * OLD * sizeof(float);
        const int ntx = (tiles + wpb - 1) / wpb, npairs = B * heads;
        const dim3 grid((unsigned)(ntx * ((npairs + 7) / 8) * 8));
        // operand input: the K / V chunks through an LDS-DMA ring (PP_ATTN_RING = 0: the register-staged kernel of rounds 2-4, 2 | 3: stages)
        const char* ring_s = getenv("PP_ATTN_RING");       // (read per call: the tests compare the variants in one process)
        const int ring_env = ring_s ? atoi(ring_s) : 2;      // (two stages: three cost a workgroup per CU — 258 vs 236 vs 252 us staged)
        const size_t img_bytes = ((size_t)(T - 1) * 3 * heads * HD + 3 * heads * HD) * 2 * terms;      // (32-bit offsets inside an image's rows)
        const int tail_n = (T % KC) <= 4 ? T % KC : 0;
        if (hl_in && ring_env > 0 && ring_env != 1 && getenv("PP_ATTN_RESIDENT") != nullptr && lse == nullptr && img_bytes < 0x7FFFFFFFull && tiles == 9 &&
            T - tail_n == 8 * KC) {
            // T = 257 .. 260: all of K / V of an (image, head) resident in LDS, nine waves per workgroup (see the kernel).  MEASURED SLOWER —
            // 334 us against 252 (staged) / 243-248 (two-stage ring) at 192 images: nine waves per CU instead of twelve on a kernel whose
            // time is the soft-max's vector work (MFMA-busy 0.21), and nothing overlaps a workgroup's load phase — opt-in (PP_ATTN_RESIDENT=1)
            const dim3 grid9((unsigned)(((npairs + 7) / 8) * 8));
#define PP_ATTN_RES_LAUNCH(TERMS_)                                                                                                  \
    {                                                                                                                              \
        constexpr int lds_ = 8 * RingGeom<TERMS_>::STAGE_H * (int)sizeof(_Float16) > 9 * 32 * OLD * (int)sizeof(float)             \
                                 ? 8 * RingGeom<TERMS_>::STAGE_H * (int)sizeof(_Float16) : 9 * 32 * OLD * (int)sizeof(float);      \
        static signed char st_[PP_MAX_DEVICES];                                                                                    \
        signed char& ok_ = st_[pp_cur_device()];                                                                                   \
        if (ok_ == 0)                                                                                                              \
            ok_ = hipFuncSetAttribute((const void*)attn_f16x3_kernel<true, TERMS_, 8, 9>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_) == hipSuccess ? 1 : -1; \
        if (ok_ < 0) return PP_ELAUNCH;                                                                                            \
        hipLaunchKernelGGL((attn_f16x3_kernel<true, TERMS_, 8, 9>), grid9, dim3(64 * 9), lds_, (hipStream_t)stream, qkv, T, heads, scale, out, \
                           (_Float16*)out_hl, lse, 1, npairs);                                                                     \
    }
            if (terms == 1) PP_ATTN_RES_LAUNCH(1) else PP_ATTN_RES_LAUNCH(2)
#undef PP_ATTN_RES_LAUNCH
            return pp_last_launch();
        }
        if (hl_in && (ring_env == 2 || ring_env == 3) && lse == nullptr && img_bytes < 0x7FFFFFFFull) {
            const size_t ring = (size_t)ring_env * RingGeom<2>::STAGE_H * sizeof(_Float16) * terms / 2;
            const size_t lds = ring > os ? ring : os;
#define PP_ATTN_RING_LAUNCH(TERMS_, S_, W_)                                                                                        \
    {                                                                                                                              \
        static signed char st_[PP_MAX_DEVICES];                                                                                    \
        signed char& ok_ = st_[pp_cur_device()];                                                                                   \
        if (ok_ == 0)                                                                                                              \
            ok_ = hipFuncSetAttribute((const void*)attn_f16x3_kernel<true, TERMS_, S_, W_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)(S_ * RingGeom<TERMS_>::STAGE_H * sizeof(_Float16) > 4 * 32 * OLD * sizeof(float)               \
                                                ? S_ * RingGeom<TERMS_>::STAGE_H * sizeof(_Float16)                                  \
                                                : 4 * 32 * OLD * sizeof(float))) == hipSuccess ? 1 : -1;                            \
        if (ok_ < 0) return PP_ELAUNCH;                                                                                            \
        hipLaunchKernelGGL((attn_f16x3_kernel<true, TERMS_, S_, W_>), grid, dim3(64 * W_), lds, (hipStream_t)stream, qkv, T, heads, scale, out, \
                           (_Float16*)out_hl, lse, ntx, npairs);                                                                   \
    }
#define PP_ATTN_RING_W(TERMS_, S_)                                  \
    switch (wpb) {                                                  \
        case 2: PP_ATTN_RING_LAUNCH(TERMS_, S_, 2) break;           \
        case 3: PP_ATTN_RING_LAUNCH(TERMS_, S_, 3) break;           \
        default: PP_ATTN_RING_LAUNCH(TERMS_, S_, 4) break;          \
    }
            if (terms == 1) {
                if (ring_env == 2) PP_ATTN_RING_W(1, 2) else PP_ATTN_RING_W(1, 3)
            } else {
                if (ring_env == 2) PP_ATTN_RING_W(2, 2) else PP_ATTN_RING_W(2, 3)
            }
#undef PP_ATTN_RING_W
#undef PP_ATTN_RING_LAUNCH
            return pp_last_launch();
        }
        if (hl_in && terms == 1)
            hipLaunchKernelGGL((attn_f16x3_kernel<true, 1>), grid, dim3(64 * wpb), kv > os ? kv : os, (hipStream_t)stream, qkv, T, heads, scale, out,
                               (_Float16*)out_hl, lse, ntx, npairs);
        else if (hl_in)
            hipLaunchKernelGGL(attn_f16x3_kernel<true>, grid, dim3(64 * wpb), kv > os ? kv : os, (hipStream_t)stream, qkv, T, heads, scale, out,
                               (_Float16*)out_hl, lse, ntx, npairs);
        else
            hipLaunchKernelGGL(attn_f16x3_kernel<false>, grid, dim3(64 * wpb), kv > os ? kv : os, (hipStream_t)stream, qkv, T, heads, scale, out,
                               (_Float16*)out_hl, lse, ntx, npairs);
    } else {
        // three-wave workgroups when they run fewer waves in all (T = 257: 9 against 12)
        if (((T + 95) / 96) * 3 < ((T + 127) / 128) * 4)
            hipLaunchKernelGGL(attn_kernel<3>, dim3((T + 95) / 96, B * heads), dim3(192), 0, (hipStream_t)stream, (const float*)qkv, T, heads,
                               scale, out, (_Float16*)out_hl);
        else
            hipLaunchKernelGGL(attn_kernel<4>, dim3((T + 127) / 128, B * heads), dim3(256), 0, (hipStream_t)stream, (const float*)qkv, T, heads,
                               scale, out, (_Float16*)out_hl);
    }
    return pp_last_launch();
}

int pp_attention(const float* qkv, int B, int T, int heads, int head_dim, float scale, float* out, void* stream) {
    if (!out) return PP_EINVAL;
    return attention_launch(qkv, false, B, T, heads, head_dim, scale, PP_PREC_F32, out, nullptr, stream);
}

int pp_attention_split(const float* qkv, int B, int T, int heads, int head_dim, float scale, float* out, void* out_hl,
                       void* stream) {
    if (!out_hl) return PP_EINVAL;
    return attention_launch(qkv, false, B, T, heads, head_dim, scale, PP_PREC_F32, out, out_hl, stream);
}

int pp_attention_ex(const float* qkv, int B, int T, int heads, int head_dim, float scale, int prec, float* out, void* out_hl,
                    void* stream) {
    if (prec != PP_PREC_F32 && prec != PP_PREC_F16X3) return PP_EINVAL;
    return attention_launch(qkv, false, B, T, heads, head_dim, scale, prec, out, out_hl, stream);
}

int pp_attention_hl(const void* qkv_hl, int B, int T, int heads, int head_dim, float scale, float* out, void* out_hl,
                    void* stream) {
    if (((uintptr_t)qkv_hl % 16) != 0) return PP_EINVAL;
    return attention_launch(qkv_hl, true, B, T, heads, head_dim, scale, PP_PREC_F16X3, out, out_hl, stream);
}

int pp_attention_t(const void* qkv_operand, int terms, int B, int T, int heads, int head_dim, float scale, float* out, void* out_operand,
                   void* stream) {
    if (((uintptr_t)qkv_operand % 16) != 0 || (terms != 1 && terms != 2)) return PP_EINVAL;
    return attention_launch(qkv_operand, true, B, T, heads, head_dim, scale, PP_PREC_F16X3, out, out_operand, stream, terms);
}

int pp_attention_train(const float* qkv, int B, int T, int heads, int head_dim, float scale, float* out, float* lse2, void* stream) {
    if (!out || !lse2) return PP_EINVAL;
    return attention_launch(qkv, false, B, T, heads, head_dim, scale, PP_PREC_F16X3, out, nullptr, stream, 2, lse2);
}

}  // extern "C"
