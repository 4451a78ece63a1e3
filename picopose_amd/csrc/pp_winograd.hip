// Winograd F(2x2, 3x3) for the stride-1 3x3 convolutions of the strict-fp32 mode (ops.PRECISION = "f32", bench.py --mode exact;
// the flow decoder's and the DPT head's convolutions: raft_decoder.py:251-289, dpt.py:72-95).
//
// fp32 MFMA is 16x slower per flop than fp16 MFMA, so in this mode the 3x3 convolutions are matrix-bound at 0.91 of the
// fp32-MFMA peak (csrc/pp_gemm_f.hip) and the lever left is fewer multiplications: F(2x2, 3x3) produces a 2x2 output tile
// from a 4x4 input tile with 16 products per (input channel, output channel) instead of 36 — 2.25x fewer —
//      Y = A^T [ (G g G^T) .* (B^T d B) ] A ,
// i.e. SIXTEEN dense GEMMs  Y_xi[P, Cout] = U_xi[P, Cin] V_xi[Cout, Cin]^T  over the P = B H W / 4 tiles, one per frequency
// xi = 4 a + b.  The GEMMs run on the fp32 engine as they are (pp_gemm, dense MODE 0); this file holds the three transforms
// around them, all HBM-bound element-wise passes in fp32:
//   pp_winograd_input_f32    x (B,H,W,C) NHWC        -> U (16, P, C)      B^T d B per 4x4 tile (zero padding; optional ReLU first)
//   pp_winograd_weight_f32   w (Cout, 9 Cin) k-order -> V (16, Cout, Cin) G g G^T          (once per weight version)
//   pp_winograd_output_f32   Y (16, P, Cout)         -> out (B,H,W,ldc)   A^T Y A, + bias, activation, residuals
// Every value is fp32 and every sum is an fp32 add; the result differs from the direct convolution by the rounding of the
// transforms (measured against float64 in tests/test_engine_gpu.py).  The f16x3 mode does not use this: there the convolutions run
// 5x faster and the transform passes (20 GB per 640 -> 512 layer at 64 x 64 x 160) would cost what the saved MFMAs return.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/picopose_hip.h"
#include "pp_common.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

// B^T d B for one 4x4 tile of 4-channel vectors: d[r][c] -> u[a][b]
__device__ __forceinline__ void wino_bt_d_b(const f4 (&d)[4][4], f4 (&u)[4][4]) {
    f4 t[4][4];   // t = B^T d : rows  d0 - d2, d1 + d2, d2 - d1, d1 - d3
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        t[0][c] = d[0][c] - d[2][c];
        t[1][c] = d[1][c] + d[2][c];
        t[2][c] = d[2][c] - d[1][c];
        t[3][c] = d[1][c] - d[3][c];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {   // u = t B : columns  t0 - t2, t1 + t2, t2 - t1, t1 - t3
        u[a][0] = t[a][0] - t[a][2];
        u[a][1] = t[a][1] + t[a][2];
        u[a][2] = t[a][2] - t[a][1];
        u[a][3] = t[a][1] - t[a][3];
    }
}

// one thread = one tile x 4 consecutive channels; tiles of an image row-major over (H/2, W/2)
__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ x, int ld_x, long long bstride, int B, int H, int W, int C,
                                                         int relu, float* __restrict__ U) {
    const int c4n = C >> 2;
    const long long P = (long long)B * (H >> 1) * (W >> 1), total = P * c4n;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long p = i / c4n;
        const int c = (int)(i - p * c4n) * 4;
        const int tw = W >> 1, th = H >> 1;
        const int b = (int)(p / ((long long)th * tw)), r = (int)(p - (long long)b * th * tw), ty = r / tw, tx = r - ty * tw;
        const int y0 = 2 * ty - 1, x0 = 2 * tx - 1;
        f4 d[4][4], u[4][4];
#pragma unroll
        for (int dy = 0; dy < 4; ++dy)
#pragma unroll
            for (int dx = 0; dx < 4; ++dx) {
                const int iy = y0 + dy, ix = x0 + dx;
                f4 v = {0.f, 0.f, 0.f, 0.f};
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = *(const f4*)(x + (long long)b * bstride + ((long long)iy * W + ix) * ld_x + c);
                if (relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                }
                d[dy][dx] = v;
            }
        wino_bt_d_b(d, u);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) *(f4*)(U + ((long long)(4 * a + bb) * P + p) * C + c) = u[a][bb];
    }
}

// V_xi[co][ci] = (G g G^T)[a][b], g[ky][kx] = w[co][(ky 3 + kx) Cin + ci];  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
__global__ __launch_bounds__(256) void wino_weight_kernel(const float* __restrict__ w, int Cout, int Cin, int ldw, float* __restrict__ V) {
    const long long total = (long long)Cout * Cin;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int co = (int)(i / Cin), ci = (int)(i - (long long)co * Cin);
        float g[3][3], t[4][3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) g[ky][kx] = w[(long long)co * ldw + (ky * 3 + kx) * Cin + ci];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            t[0][kx] = g[0][kx];
            t[1][kx] = 0.5f * (g[0][kx] + g[1][kx] + g[2][kx]);
            t[2][kx] = 0.5f * (g[0][kx] - g[1][kx] + g[2][kx]);
            t[3][kx] = g[2][kx];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float v0 = t[a][0], v1 = 0.5f * (t[a][0] + t[a][1] + t[a][2]), v2 = 0.5f * (t[a][0] - t[a][1] + t[a][2]), v3 = t[a][2];
            V[((long long)(4 * a + 0) * Cout + co) * Cin + ci] = v0;
            V[((long long)(4 * a + 1) * Cout + co) * Cin + ci] = v1;
            V[((long long)(4 * a + 2) * Cout + co) * Cin + ci] = v2;
            V[((long long)(4 * a + 3) * Cout + co) * Cin + ci] = v3;
        }
    }
}

__device__ __forceinline__ float wino_act(float v, int act) {
    switch (act) {
        case PP_ACT_RELU: return v > 0.f ? v : 0.f;
        case PP_ACT_LEAKY01: return v > 0.f ? v : 0.1f * v;
        default: return v;
    }
}

// out(2x2) = A^T Y A, A^T = [[1,1,1,0],[0,1,-1,-1]];  then + bias, activation, + residual + residual2 (laid out like out)
// VEC: Cout % 4 == 0 (a thread = one tile x 4 channels, 16-byte accesses); otherwise one channel per thread
template <bool VEC>
__global__ __launch_bounds__(256) void wino_output_kernel(const float* __restrict__ Y, int B, int H, int W, int Cout, const float* __restrict__ bias,
                                                          int act, const float* __restrict__ residual, const float* __restrict__ residual2,
                                                          float* __restrict__ out, int ldc) {
    constexpr int V = VEC ? 4 : 1;
    const int cn = VEC ? Cout >> 2 : Cout;
    const long long P = (long long)B * (H >> 1) * (W >> 1), total = P * cn;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long p = i / cn;
        const int c = (int)(i - p * cn) * V;
        const int tw = W >> 1, th = H >> 1;
        const int b = (int)(p / ((long long)th * tw)), r = (int)(p - (long long)b * th * tw), ty = r / tw, tx = r - ty * tw;
        float y[4][4][V];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                const float* src = Y + ((long long)(4 * a + bb) * P + p) * Cout + c;
                if (VEC) {
                    const f4 v = *(const f4*)src;
#pragma unroll
                    for (int e = 0; e < V; ++e) y[a][bb][e] = v[e];
                } else {
                    y[a][bb][0] = src[0];
                }
            }
        float bv[V];
#pragma unroll
        for (int e = 0; e < V; ++e) bv[e] = bias ? bias[c + e] : 0.f;
#pragma unroll
        for (int oy = 0; oy < 2; ++oy)
#pragma unroll
            for (int ox = 0; ox < 2; ++ox) {
                float o[V];
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    // rows: oy = 0: y0 + y1 + y2 ; oy = 1: y1 - y2 - y3 ; the same combination over the columns
                    float col[4];
#pragma unroll
                    for (int bb = 0; bb < 4; ++bb) col[bb] = oy == 0 ? (y[0][bb][e] + y[1][bb][e]) + y[2][bb][e] : (y[1][bb][e] - y[2][bb][e]) - y[3][bb][e];
                    o[e] = ox == 0 ? (col[0] + col[1]) + col[2] : (col[1] - col[2]) - col[3];
                }
                const long long off = ((long long)b * H * W + (long long)(2 * ty + oy) * W + 2 * tx + ox) * ldc + c;
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    float v = wino_act(o[e] + bv[e], act);
                    if (residual) v += residual[off + e];
                    if (residual2) v += residual2[off + e];
                    o[e] = v;
                }
                if (VEC) {
                    *(f4*)(out + off) = f4{o[0], o[V > 1 ? 1 : 0], o[V > 2 ? 2 : 0], o[V > 3 ? 3 : 0]};
                } else {
                    out[off] = o[0];
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Winograd F(4x4, 3x3) on the f16x3 engine (round 6).  The pre-split engine runs the 3x3 convolutions at ~490 useful TFLOP/s
// (three fp16 MFMAs per product); what is left is fewer products.  F(4x4, 3x3) computes a 4x4 output tile from a 6x6 input tile
// with 36 products per (input channel, output channel) instead of 144 — FOUR times fewer — and its transformed operands are only
// 36 / 16 = 2.25 x the map (F(2x2, 3x3): 2.25 x fewer products, operands 4 x the map: on this engine it breaks even):
//      Y = A^T [ (G g G^T) .* (B^T d B) ] A ,     36 dense products  Y_xi (P, Cout) = U_xi (P, Cin) V_xi (Cout, Cin)^T ,  xi = 6 a + b,
// over the P = B H W / 16 tiles, run by pp_gemm as grouped launches of pp_gemm_u_kernel (PpGemmDesc.grp_rows).  Transforms:
//   pp_winograd4_input_hl    x: hl operand image (B, H, W) -> U (36, P, C) hl operand of (B^T d B) / 16  (operand scale 1/16 instead
//                            of the activations' 4: |B^T d B| <= 100 |d|, so the operand saturates at |d| >= 10 480)
//   pp_winograd4_weight_f32  w (Cout, 9 Cin) k-order -> V (36, Cout, Cin) fp32 = G g G^T  (then split once like any weight)
//   pp_winograd4_output      Y (36, P, Cout) fp32 -> A^T Y A + bias, activation; as fp32 map and / or as hl operand (next layer's input)
// The transforms are fp32 sums; their rounding is what F(4x4) costs: ~1e-5 of the map's maximum per layer against ~4e-7 direct
// (tools/wino_precision_study.py) — inside every parity bar of the f16x3 mode (2e-4 / 5e-4 of the maximum).
typedef _Float16 hf4 __attribute__((ext_vector_type(4)));
// cache policy of the streamed-once accesses of the transforms (Y read once, U written once): 0 = default, 2 = nt (study: profiles/r06/README.md)
#ifndef PP_WINO_LD_AUX
#define PP_WINO_LD_AUX 0
#endif
#ifndef PP_WINO_ST_AUX
#define PP_WINO_ST_AUX 0
#endif

// XCD-contiguous block order: hardware block b runs on XCD b % 8; logical blocks of one XCD are consecutive, so neighbouring tiles
// (which share input columns / rows) meet in the same L2
__device__ __forceinline__ long long wino_logical_block() {
    const unsigned per = gridDim.x >> 3;      // gridDim.x is a multiple of 8
    return (long long)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
}

// one 1-D input transform B^T (6 -> 6)
#define PP_W4_BT(d0, d1, d2, d3, d4, d5, t0, t1, t2, t3, t4, t5)          \
    {                                                                     \
        const f4 a_ = d4 - 4.f * d2, b_ = d3 - 4.f * d1;                  \
        const f4 c_ = d4 - d2, e_ = 2.f * (d3 - d1);                      \
        t0 = 4.f * d0 - 5.f * d2 + d4;                                    \
        t1 = a_ + b_;                                                     \
        t2 = a_ - b_;                                                     \
        t3 = c_ + e_;                                                     \
        t4 = c_ - e_;                                                     \
        t5 = 4.f * d1 - 5.f * d3 + d5;                                    \
    }

// split 4 values (already in operand scale) into their hi / lo fp16 terms; `top` keeps the running maximum of |v| — ONE v_max3 per two
// elements instead of a clamp and a compare per element: a value beyond the fp16 range is converted to +-inf (wrong either way) and is
// reported through the saturation word by the caller, which tests `top` once
__device__ __forceinline__ void wino_split4(f4 v, hf4& hi, hf4& lo, float& top) {
    top = fmaxf(fmaxf(top, fabsf(v[0])), fabsf(v[1]));
    top = fmaxf(fmaxf(top, fabsf(v[2])), fabsf(v[3]));
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const _Float16 h = (_Float16)v[e];
        hi[e] = h;
        lo[e] = (_Float16)(v[e] - (float)h);        // |v - h| <= half an ulp of h
    }
}

// sticky saturation word (include/picopose_hip.h pp_set_saturation_word): one atomic per wave that saw a clamped term
__device__ __forceinline__ void wino_note_sat(bool clamp, unsigned* sat) {
    if (sat && __builtin_amdgcn_ballot_w64(clamp) != 0ull && (threadIdx.x & 63) == 0) atomicOr(sat, 1u);
}

// Lane pairs.  A thread owns 4 channels, an hl group is [8 hi | 8 lo] (32 bytes): lanes 2k / 2k + 1 own channels 0-3 / 4-7 of the SAME group
// (C % 8 == 0 puts them in one tile).  Instead of two 8-byte accesses per lane (hi[0:4] and lo[0:4], resp. hi[4:8] and lo[4:8]) the even lane
// moves the 16 bytes of the hi terms and the odd lane the 16 bytes of the lo terms, and the pair trades halves with one DPP quad-permute per
// dword: half the vector-memory instructions (measured: input transform 3.0 -> TB/s of FETCH + WRITE, profiles/r06/wino4_layers.txt).
typedef unsigned int wu4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned wino_swap(unsigned v) { return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true); }   // lanes 2k <-> 2k + 1
__device__ __forceinline__ float wino_h2f(unsigned w, int hi16) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(hi16 ? w >> 16 : w & 0xFFFFu)); }

// the lane's 4 channels out of the 16 bytes `q` it loaded from its group (even lane: hi[0:8], odd lane: lo[0:8]) and the partner's
// 16 bytes, as 4 x = hi + lo (exact)
__device__ __forceinline__ f4 wino_pair_unpack(wu4 q, int par) {
    const unsigned k0 = par ? q[2] : q[0], k1 = par ? q[3] : q[1];     // own half: hi[0:4] (even) / lo[4:8] (odd)
    const unsigned r0 = wino_swap(par ? q[0] : q[2]), r1 = wino_swap(par ? q[1] : q[3]);   // the partner's: lo[0:4] / hi[4:8]
    return f4{wino_h2f(k0, 0) + wino_h2f(r0, 0), wino_h2f(k0, 1) + wino_h2f(r0, 1), wino_h2f(k1, 0) + wino_h2f(r1, 0), wino_h2f(k1, 1) + wino_h2f(r1, 1)};
}

// the 16 bytes a lane stores for its pair's group: the even lane hi[0:8], the odd lane lo[0:8]
__device__ __forceinline__ wu4 wino_pair_pack(int par, hf4 hi, hf4 lo) {
    const unsigned H0 = __builtin_bit_cast(unsigned, __builtin_shufflevector(hi, hi, 0, 1)), H1 = __builtin_bit_cast(unsigned, __builtin_shufflevector(hi, hi, 2, 3));
    const unsigned L0 = __builtin_bit_cast(unsigned, __builtin_shufflevector(lo, lo, 0, 1)), L1 = __builtin_bit_cast(unsigned, __builtin_shufflevector(lo, lo, 2, 3));
    const unsigned r0 = wino_swap(par ? H0 : L0), r1 = wino_swap(par ? H1 : L1);    // even receives the partner's hi[4:8], odd the partner's lo[0:4]
    return wu4{par ? r0 : H0, par ? r1 : H1, par ? L0 : r0, par ? L1 : r1};
}

// descriptor of frequency block xi of a (36, rows, ..) buffer: built from scalars next to its use (no 64-bit per-lane addresses: 36
// of them cost 72 registers and hipcc spilled)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wino_block_rsrc(const void* base, int xi, unsigned long long block_bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)base + (unsigned long long)xi * block_bytes), 0, (int)(unsigned)block_bytes, 0x00020000);
}

// store the lane's 4 channels (hi / lo terms) into the group at `grp`: the even lane writes hi[0:8], the odd lane lo[0:8]
__device__ __forceinline__ void wino_pair_store(_Float16* grp, int par, hf4 hi, hf4 lo) {
    const unsigned H0 = __builtin_bit_cast(unsigned, __builtin_shufflevector(hi, hi, 0, 1)), H1 = __builtin_bit_cast(unsigned, __builtin_shufflevector(hi, hi, 2, 3));
    const unsigned L0 = __builtin_bit_cast(unsigned, __builtin_shufflevector(lo, lo, 0, 1)), L1 = __builtin_bit_cast(unsigned, __builtin_shufflevector(lo, lo, 2, 3));
    const unsigned r0 = wino_swap(par ? H0 : L0), r1 = wino_swap(par ? H1 : L1);    // even receives the partner's hi[4:8], odd the partner's lo[0:4]
    const wu4 w = {par ? r0 : H0, par ? r1 : H1, par ? L0 : r0, par ? L1 : r1};
    *(wu4*)(grp + par * 8) = w;
}

// one thread = one 6x6 tile x 4 consecutive channels; tiles row-major over (B, H/4, W/4).  The 36 loads of a tile are bounded buffer
// loads issued back to back — a pixel outside the image is an out-of-range offset (zeros, no traffic, no branch): with a branch per
// padded pixel hipcc waited for every load before the next one (39 s_waitcnt, 80 branches) and the kernel ran at 3.0 TB/s.
template <bool RELU>
__global__ __launch_bounds__(256, 2) void wino4_input_kernel(const _Float16* __restrict__ x, unsigned x_bytes, int ld_x, long long bstride, int B, int H,
                                                             int W, int C, _Float16* __restrict__ U, long long Pp, unsigned* sat) {
    const int c4n = C >> 2;
    const int tw = W >> 2, th = H >> 2;
    const long long P = (long long)B * th * tw, total = P * c4n;
    const long long i = wino_logical_block() * 256 + threadIdx.x;
    if (i >= total) return;
    const long long p = i / c4n;
    const int c = (int)(i - p * c4n) * 4;
    const int b = (int)(p / ((long long)th * tw)), r = (int)(p - (long long)b * th * tw), ty = r / tw, tx = r - ty * tw;
    const int y0 = 4 * ty - 1, x0 = 4 * tx - 1;
    const int gcol = (c >> 3) << 4;                 // half index of the channel group (c & ~7, hi) inside an hl row
    const int par = (c >> 2) & 1;                   // = lane parity (c4n is even): which half of the group's channels this lane owns
    // address = base' + voffset (per lane: the tile's pixel (-1, -1), or out of range) + soffset (wave-uniform: (dy W + dx) pixels).  The
    // range check reads voffset alone, so the base is moved back by one row + one pixel and every in-image voffset is >= 0.
    const unsigned pix = (unsigned)ld_x * 4u;                                      // bytes per pixel row of the operand
    const unsigned back = (unsigned)(W + 1) * pix;
    const __amdgpu_buffer_rsrc_t R = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)x - back), 0, (int)(x_bytes + back), 0x00020000);
    const unsigned v00 = (unsigned)((b * bstride * 2 + gcol + par * 8) * 2) + (unsigned)((y0 + 1) * W + x0 + 1) * pix;
    bool rok[6], cok[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        rok[k] = (unsigned)(y0 + k) < (unsigned)H;
        cok[k] = (unsigned)(x0 + k) < (unsigned)W;
    }
    wu4 q[6][6];
#pragma unroll
    for (int dx = 0; dx < 6; ++dx)
#pragma unroll
        for (int dy = 0; dy < 6; ++dy)
            q[dy][dx] = __builtin_amdgcn_raw_buffer_load_b128(R, (rok[dy] && cok[dx]) ? v00 : 0xFFFFFFFFu, (dy * W + dx) * (int)pix, 0);
    f4 t[6][6];   // t = B^T d, built column by column
#pragma unroll
    for (int dx = 0; dx < 6; ++dx) {
        f4 d[6];
#pragma unroll
        for (int dy = 0; dy < 6; ++dy) {
            f4 v = wino_pair_unpack(q[dy][dx], par);       // = 4 x, exactly
            if (RELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            d[dy] = v;
        }
        PP_W4_BT(d[0], d[1], d[2], d[3], d[4], d[5], t[0][dx], t[1][dx], t[2][dx], t[3][dx], t[4][dx], t[5][dx])
    }
    float top = 0.f;
    const unsigned long long blk = (unsigned long long)Pp * C * 4;     // bytes of one frequency block (Pp = P rounded up to the engine's row tile)
    const unsigned uoff = (unsigned)((p * 2 * C + gcol + par * 8) * 2);
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        f4 u[6];
        PP_W4_BT(t[a][0], t[a][1], t[a][2], t[a][3], t[a][4], t[a][5], u[0], u[1], u[2], u[3], u[4], u[5])
#pragma unroll
        for (int bb = 0; bb < 6; ++bb) {
            hf4 hi, lo;
            wino_split4(u[bb] * (1.f / 64.f), hi, lo, top);       // (the 4 x of the source operand) / 64 = (B^T d B) / 16
            __builtin_amdgcn_raw_buffer_store_b128(wino_pair_pack(par, hi, lo), wino_block_rsrc(U, 6 * a + bb, blk), uoff, 0, PP_WINO_ST_AUX);
        }
    }
    wino_note_sat(!(top < 65504.f), sat);
}

// V_xi[co][ci] = (G g G^T)[a][b];  G = [[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]]
__global__ __launch_bounds__(256) void wino4_weight_kernel(const float* __restrict__ w, int Cout, int Cin, int ldw, float* __restrict__ V) {
    const long long total = (long long)Cout * Cin;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int co = (int)(i / Cin), ci = (int)(i - (long long)co * Cin);
    float g[3][3], t[6][3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) g[ky][kx] = w[(long long)co * ldw + (ky * 3 + kx) * Cin + ci];
    auto g6 = [](float g0, float g1, float g2, float (&o)[6]) {
        o[0] = 0.25f * g0;
        o[1] = (-1.f / 6.f) * ((g0 + g2) + g1);
        o[2] = (-1.f / 6.f) * ((g0 + g2) - g1);
        o[3] = (1.f / 24.f) * g0 + (1.f / 12.f) * g1 + (1.f / 6.f) * g2;
        o[4] = (1.f / 24.f) * g0 - (1.f / 12.f) * g1 + (1.f / 6.f) * g2;
        o[5] = g2;
    };
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        float o[6];
        g6(g[0][kx], g[1][kx], g[2][kx], o);
#pragma unroll
        for (int a = 0; a < 6; ++a) t[a][kx] = o[a];
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        float o[6];
        g6(t[a][0], t[a][1], t[a][2], o);
#pragma unroll
        for (int bb = 0; bb < 6; ++bb) V[((long long)(6 * a + bb) * Cout + co) * Cin + ci] = o[bb];
    }
}

// one 1-D output transform A^T (6 -> 4)
#define PP_W4_AT(y0, y1, y2, y3, y4, y5, o0, o1, o2, o3)                   \
    {                                                                      \
        const f4 s12 = y1 + y2, d12 = y1 - y2, s34 = y3 + y4, d34 = y3 - y4; \
        o0 = (y0 + s12) + s34;                                             \
        o1 = d12 + 2.f * d34;                                              \
        o2 = s12 + 4.f * s34;                                              \
        o3 = (d12 + 8.f * d34) + y5;                                       \
    }

// out(4x4) = A^T Y A, + bias, activation, (+ residuals for the fp32 map); fp32 map `out` and / or hl operand `out_hl` (of max(., 0) with
// c_relu); one thread = one tile x 4 channels
__global__ __launch_bounds__(256) void wino4_output_kernel(const float* __restrict__ Y, int B, int H, int W, int Cout, const float* __restrict__ bias,
                                                           int act, const float* __restrict__ residual, const float* __restrict__ residual2,
                                                           float* __restrict__ out, int ldc, _Float16* __restrict__ out_hl, int ld_h, int c_relu,
                                                           long long Pp, int ldy, unsigned* sat) {
    const int cn = Cout >> 2;
    const int tw = W >> 2, th = H >> 2;
    const long long P = (long long)B * th * tw, total = P * cn;
    const long long i = wino_logical_block() * 256 + threadIdx.x;
    if (i >= total) return;
    const long long p = i / cn;
    const int c = (int)(i - p * cn) * 4;
    const int b = (int)(p / ((long long)th * tw)), r = (int)(p - (long long)b * th * tw), ty = r / tw, tx = r - ty * tw;
    const unsigned long long blk = (unsigned long long)Pp * ldy * 4;     // (ldy >= Cout: Y may be a column slice of a wider product — two layers fused along N)
    const unsigned yoff = (unsigned)((p * ldy + c) * 4);
    f4 z[4][6];   // z = A^T Y, column by column
#pragma unroll
    for (int bb = 0; bb < 6; ++bb) {
        f4 y[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) y[a] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(wino_block_rsrc(Y, 6 * a + bb, blk), yoff, 0, PP_WINO_LD_AUX));
        PP_W4_AT(y[0], y[1], y[2], y[3], y[4], y[5], z[0][bb], z[1][bb], z[2][bb], z[3][bb])
    }
    const f4 bv = bias ? *(const f4*)(bias + c) : f4{0.f, 0.f, 0.f, 0.f};
    const float slope = act == PP_ACT_RELU ? 0.f : (act == PP_ACT_LEAKY01 ? 0.1f : 1.f);
    const float hfloor = c_relu ? 0.f : -INFINITY;
    const int gcol = (c >> 3) << 4, par = (c >> 2) & 1;     // (operand output: Cout % 8 == 0, lanes 2k / 2k + 1 share a group)
    float top = 0.f;
#pragma unroll
    for (int oy = 0; oy < 4; ++oy) {
        f4 o[4];
        PP_W4_AT(z[oy][0], z[oy][1], z[oy][2], z[oy][3], z[oy][4], z[oy][5], o[0], o[1], o[2], o[3])
#pragma unroll
        for (int ox = 0; ox < 4; ++ox) {
            const long long row = ((long long)b * H + 4 * ty + oy) * W + 4 * tx + ox;
            f4 v = o[ox] + bv;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], v[e] * slope);
            if (out) {
                f4 w = v;
                if (residual) w += *(const f4*)(residual + row * ldc + c);
                if (residual2) w += *(const f4*)(residual2 + row * ldc + c);
                *(f4*)(out + row * ldc + c) = w;
                v = w;
            }
            if (out_hl) {
                hf4 hi, lo;
                f4 x4;
#pragma unroll
                for (int e = 0; e < 4; ++e) x4[e] = fmaxf(v[e] * PP_A_SCALE, hfloor);
                wino_split4(x4, hi, lo, top);
                wino_pair_store(out_hl + row * 2 * ld_h + gcol, par, hi, lo);
            }
        }
    }
    wino_note_sat(!(top < 65504.f), sat);
}

// ---- output transform of layer k CHAINED into the input transform of layer k + 1 (round 6) ------------------------------------------------
// The decoder's heads are conv 3x3 -> ReLU -> conv 3x3 (raft_decoder.py:251-289), both by F(4x4, 3x3): the hidden map h = relu(A^T Y0 A + b)
// would be written as an operand (4 bytes per element) by the output transform and read back by the next input transform.  Here it never
// reaches HBM: a workgroup owns (image, 32-channel slice) and walks the image's tile rows; half of its threads transform tile row r of Y0
// into four pixel rows of h — rounded to the operand's 22 bits exactly as the operand store would, so the result is BIT-IDENTICAL to the two
// separate kernels — in an LDS ring of four tile rows, the other half takes tile row r - 2 (whose one-pixel halo above and below is in the
// ring by then) through B^T d B and writes U1.  One barrier per tile row.  Bytes per layer pair: Y0 once + U1 once instead of + 2 x h.
constexpr int CHN_PX = 36;     // floats per staged pixel: 32 channels + 4 (the 16-byte accesses of a tile's pixels fall on different banks)
// a frequency index the optimiser cannot see through: inside the tile-row loop the 72 buffer descriptors (36 of Y, 36 of U) are loop
// invariants, and hipcc hoisted all of them — 288 scalar registers, 460 spilled into vector lanes
__device__ __forceinline__ int wino_opaque(int xi) {
    int k;
    asm volatile("s_mov_b32 %0, %1" : "=s"(k) : "s"(xi));
    return k;
}

template <int W, int NB>      // NB: thread groups that share the input-transform side (1, or 2: three frequencies rows each)
__global__ __launch_bounds__((1 + NB) * ((W / 4) * 8 < 64 ? 64 : (W / 4) * 8)) void wino4_chain_kernel(const float* __restrict__ Y, int H, int C, const float* __restrict__ bias,
                                                                       int act, int c_relu, _Float16* __restrict__ U, long long Pp, int ldy, unsigned* sat) {
    constexpr int TW = W / 4, NI = TW * 8;                  // tiles per tile row; work items (tile, channel quad) per tile row
    constexpr int GS = NI < 64 ? 64 : NI;                   // threads per group: whole waves (a wave's group index is uniform)
    constexpr int SLOT = 4 * W * CHN_PX;                    // floats per ring slot (one tile row = 4 pixel rows)
    extern __shared__ __attribute__((aligned(16))) float chs[];
    const int th = H >> 2, nsl = C >> 5;
    const int b = blockIdx.x / nsl, cs = blockIdx.x - b * nsl;
    // (NB == 2: the group index picks the frequency rows and must be wave-uniform for the scalar descriptors; NB == 1: left per-lane —
    // made uniform there, hipcc laid both halves' live ranges over each other: 256 registers + spills instead of 176)
    const int half = NB == 2 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x / GS) : (int)threadIdx.x / GS;
    const int it = (int)threadIdx.x - half * GS, tx = it >> 3, q = it & 7;
    const bool live = NI >= GS || it < NI;                  // (W = 16: half of each 64-thread group idles)
    const int c = cs * 32 + 4 * q;                          // first of this thread's 4 channels
    const unsigned long long blk = (unsigned long long)Pp * C * 4, blky = (unsigned long long)Pp * ldy * 4;     // bytes of a frequency block of U / of Y
    const f4 bv = bias ? *(const f4*)(bias + c) : f4{0.f, 0.f, 0.f, 0.f};
    const float slope = act == PP_ACT_RELU ? 0.f : (act == PP_ACT_LEAKY01 ? 0.1f : 1.f);
    const int gcol = (c >> 3) << 4, par = q & 1;
    const float hfloor = c_relu ? 0.f : -INFINITY;          // the consumer's input ReLU folded into the operand, as pp_winograd4_output does
    float top = 0.f;
    for (int r = 0; r < th + 2; ++r) {
        if (half == 0 && live && r < th) {
            // ---- tile (b, r, tx): h(4x4) = relu'(A^T Y A + bias), as the operand's value 4 h (hi + lo), into ring slot r % 4
            const long long p = ((long long)b * th + r) * TW + tx;
            const unsigned yoff = (unsigned)((p * ldy + c) * 4);
            f4 z[4][6];
#pragma unroll
            for (int bb = 0; bb < 6; ++bb) {
                f4 y[6];
#pragma unroll
                for (int a = 0; a < 6; ++a) y[a] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(wino_block_rsrc(Y, wino_opaque(6 * a + bb), blky), yoff, 0, PP_WINO_LD_AUX));
                PP_W4_AT(y[0], y[1], y[2], y[3], y[4], y[5], z[0][bb], z[1][bb], z[2][bb], z[3][bb])
            }
            float* slot = chs + (r & 3) * SLOT;
#pragma unroll
            for (int oy = 0; oy < 4; ++oy) {
                f4 o[4];
                PP_W4_AT(z[oy][0], z[oy][1], z[oy][2], z[oy][3], z[oy][4], z[oy][5], o[0], o[1], o[2], o[3])
#pragma unroll
                for (int ox = 0; ox < 4; ++ox) {
                    f4 v = o[ox] + bv, x4, f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = fmaxf(v[e], v[e] * slope);
                        x4[e] = fmaxf(v[e] * PP_A_SCALE, hfloor);
                    }
                    hf4 hi, lo;
                    wino_split4(x4, hi, lo, top);
#pragma unroll
                    for (int e = 0; e < 4; ++e) f[e] = (float)hi[e] + (float)lo[e];     // what the operand would hold: exact
                    *(f4*)(slot + ((oy * W) + 4 * tx + ox) * CHN_PX + 4 * q) = f;
                }
            }
        }
        if (half >= 1 && live && r >= 2) {
            // ---- tile (b, R = r - 2, tx): U = B^T d B / 64 of the 6x6 pixels around it (rows 4 R - 1 .. 4 R + 4 of h: ring slots R - 1, R, R + 1).
            // TWO thread groups share a tile: both make the column transform t = B^T d (LDS reads, 84 packed operations), group 1 finishes
            // frequencies a = 0 .. 2, group 2 a = 3 .. 5 — the row transforms, splits and stores are the bulk of the work, and the output-
            // transform group is bound by its loads' latency: halving the other side's arithmetic shortens the tile row (NB = 2: W = 64 only)
            constexpr int RPG = 6 / NB;
            const int R = r - 2, a0 = NB == 2 ? RPG * (half - 1) : 0;
            f4 t[6][6];
#pragma unroll
            for (int dx = 0; dx < 6; ++dx) {
                f4 d[6];
                const int px = 4 * tx - 1 + dx;
#pragma unroll
                for (int dy = 0; dy < 6; ++dy) {
                    const int g = 4 * R - 1 + dy;                      // pixel row of the image
                    // (zero padding: the read runs on a clamped pixel and is masked afterwards — no branch per pixel)
                    const int gc = min(max(g, 0), H - 1), pc = min(max(px, 0), W - 1);
                    f4 v = *(const f4*)(chs + ((gc >> 2) & 3) * SLOT + ((gc & 3) * W + pc) * CHN_PX + 4 * q);
                    if (!(g >= 0 && g < H && px >= 0 && px < W)) v = f4{0.f, 0.f, 0.f, 0.f};
                    d[dy] = v;
                }
                PP_W4_BT(d[0], d[1], d[2], d[3], d[4], d[5], t[0][dx], t[1][dx], t[2][dx], t[3][dx], t[4][dx], t[5][dx])
            }
            const long long p = ((long long)b * th + R) * TW + tx;
            const unsigned uoff = (unsigned)((p * 2 * C + gcol + par * 8) * 2);
#pragma unroll
            for (int ai = 0; ai < RPG; ++ai) {
                f4 u[6];
                // (the group's rows: selected from the six by the wave-uniform a0 — both selections are compile-time register names)
                if constexpr (NB == 2) {
                    f4 t0 = t[ai][0], t1 = t[ai][1], t2 = t[ai][2], t3 = t[ai][3], t4 = t[ai][4], t5 = t[ai][5];
                    if (a0) {
                        t0 = t[3 + ai][0], t1 = t[3 + ai][1], t2 = t[3 + ai][2];
                        t3 = t[3 + ai][3], t4 = t[3 + ai][4], t5 = t[3 + ai][5];
                    }
                    PP_W4_BT(t0, t1, t2, t3, t4, t5, u[0], u[1], u[2], u[3], u[4], u[5])
                } else {
                    PP_W4_BT(t[ai][0], t[ai][1], t[ai][2], t[ai][3], t[ai][4], t[ai][5], u[0], u[1], u[2], u[3], u[4], u[5])
                }
#pragma unroll
                for (int bb = 0; bb < 6; ++bb) {
                    hf4 hi, lo;
                    wino_split4(u[bb] * (1.f / 64.f), hi, lo, top);
                    __builtin_amdgcn_raw_buffer_store_b128(wino_pair_pack(par, hi, lo), wino_block_rsrc(U, wino_opaque(6 * (a0 + ai) + bb), blk), uoff, 0, PP_WINO_ST_AUX);
                }
            }
        }
        __syncthreads();
    }
    wino_note_sat(!(top < 65504.f), sat);
}

// The same chain for the strict-fp32 mode's F(2x2, 3x3) (round 6): Y (16, P, C) fp32 of layer k -> U (16, P, C) fp32 of layer
// k + 1, h = act(A^T Y A + bias) (then the consumer's input ReLU) in an LDS ring of four tile rows of two pixel rows each (72 KB at W = 64:
// two workgroups per CU).  Every sum is the fp32 sum the two separate kernels make, in the same order: bit-identical.
template <int W>
__global__ __launch_bounds__(2 * (W / 2) * 8, 4) void wino2_chain_kernel(const float* __restrict__ Y, int H, int C, long long P, const float* __restrict__ bias,
                                                                       int act, int relu_next, float* __restrict__ U) {
    constexpr int TW = W / 2, NI = TW * 8;
    constexpr int SLOT = 2 * W * CHN_PX;
    extern __shared__ __attribute__((aligned(16))) float chs[];
    const int th = H >> 1, nsl = C >> 5;
    const int b = blockIdx.x / nsl, cs = blockIdx.x - b * nsl;
    const int half = threadIdx.x / NI, it = threadIdx.x - half * NI, tx = it >> 3, q = it & 7;
    const int c = cs * 32 + 4 * q;
    const f4 bv = bias ? *(const f4*)(bias + c) : f4{0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < th + 2; ++r) {
        if (half == 0 && r < th) {
            const long long p = ((long long)b * th + r) * TW + tx;
            f4 y[4][4];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) y[a][bb] = *(const f4*)(Y + ((long long)(4 * a + bb) * P + p) * C + c);
            float* slot = chs + (r & 3) * SLOT;
#pragma unroll
            for (int oy = 0; oy < 2; ++oy)
#pragma unroll
                for (int ox = 0; ox < 2; ++ox) {
                    f4 col[4], o;
#pragma unroll
                    for (int bb = 0; bb < 4; ++bb) col[bb] = oy == 0 ? (y[0][bb] + y[1][bb]) + y[2][bb] : (y[1][bb] - y[2][bb]) - y[3][bb];
                    o = ox == 0 ? (col[0] + col[1]) + col[2] : (col[1] - col[2]) - col[3];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = wino_act(o[e] + bv[e], act);
                        o[e] = relu_next ? (v > 0.f ? v : 0.f) : v;        // (wino_input_kernel's optional ReLU on its input)
                    }
                    *(f4*)(slot + (oy * W + 2 * tx + ox) * CHN_PX + 4 * q) = o;
                }
        }
        if (half == 1 && r >= 2) {
            const int R = r - 2;
            // t = B^T d column by column, then u = t B row by row, each row stored at once (the sums of wino_bt_d_b in its order; the tile's
            // 16 + 16 vectors are never all live: 128 registers keep two 512-thread workgroups on a CU)
            f4 t[4][4];
#pragma unroll
            for (int dx = 0; dx < 4; ++dx) {
                f4 d[4];
#pragma unroll
                for (int dy = 0; dy < 4; ++dy) {
                    const int g = 2 * R - 1 + dy, px = 2 * tx - 1 + dx;
                    const int gc = min(max(g, 0), H - 1), pc = min(max(px, 0), W - 1);
                    f4 v = *(const f4*)(chs + ((gc >> 1) & 3) * SLOT + ((gc & 1) * W + pc) * CHN_PX + 4 * q);
                    if (!(g >= 0 && g < H && px >= 0 && px < W)) v = f4{0.f, 0.f, 0.f, 0.f};
                    d[dy] = v;
                }
                t[0][dx] = d[0] - d[2];
                t[1][dx] = d[1] + d[2];
                t[2][dx] = d[2] - d[1];
                t[3][dx] = d[1] - d[3];
            }
            const long long p = ((long long)b * th + R) * TW + tx;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const f4 u0 = t[a][0] - t[a][2], u1 = t[a][1] + t[a][2], u2 = t[a][2] - t[a][1], u3 = t[a][1] - t[a][3];
                *(f4*)(U + ((long long)(4 * a + 0) * P + p) * C + c) = u0;
                *(f4*)(U + ((long long)(4 * a + 1) * P + p) * C + c) = u1;
                *(f4*)(U + ((long long)(4 * a + 2) * P + p) * C + c) = u2;
                *(f4*)(U + ((long long)(4 * a + 3) * P + p) * C + c) = u3;
            }
        }
        __syncthreads();
    }
}

template <int W>
static int wino2_chain_launch(const float* Y, int B, int H, int C, const float* bias, int act, int relu_next, float* U, hipStream_t st) {
    constexpr int threads = 2 * (W / 2) * 8;
    const size_t lds = (size_t)4 * 2 * W * CHN_PX * sizeof(float);
    static signed char attr[PP_MAX_DEVICES];
    signed char& ok = attr[pp_cur_device()];
    if (ok == 0)
        ok = hipFuncSetAttribute((const void*)wino2_chain_kernel<W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess ? 1 : -1;
    if (ok < 0) return PP_ELAUNCH;
    const long long P = (long long)B * (H / 2) * (W / 2);
    hipLaunchKernelGGL(wino2_chain_kernel<W>, dim3((unsigned)(B * (C / 32))), dim3(threads), lds, st, Y, H, C, P, bias, act, relu_next, U);
    return pp_last_launch();
}

template <int W>
static int wino4_chain_launch(const float* Y, int B, int H, int C, const float* bias, int act, int c_relu, void* U, long long Pp, int ldy, hipStream_t st) {
    // two input-transform groups at W = 64 (1.36 -> 1.26 ms per head: the ring leaves one workgroup per CU, more waves hide more); one at
    // W <= 32, where two workgroups share a CU and the third group's registers cost more than its arithmetic saves (0.35 vs 0.41 ms)
    constexpr int NB = W == 64 ? 2 : 1;
    constexpr int threads = (1 + NB) * ((W / 4) * 8 < 64 ? 64 : (W / 4) * 8);
    const size_t lds = (size_t)4 * 4 * W * CHN_PX * sizeof(float);
    static signed char attr[PP_MAX_DEVICES];
    signed char& ok = attr[pp_cur_device()];
    if (ok == 0)
        ok = hipFuncSetAttribute((const void*)wino4_chain_kernel<W, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess ? 1 : -1;
    if (ok < 0) return PP_ELAUNCH;
    hipLaunchKernelGGL((wino4_chain_kernel<W, NB>), dim3((unsigned)(B * (C / 32))), dim3(threads), lds, st, Y, H, C, bias, act, c_relu, (_Float16*)U, Pp,
                       ldy, pp_saturation_word());
    return pp_last_launch();
}

static inline int grid8_of(long long n) {   // blocks of 256 threads, a multiple of 8 (wino_logical_block)
    const long long g = ((n + 255) / 256 + 7) / 8 * 8;
    return (int)g;
}

static inline int grid_of(long long n) {
    const long long g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 65535 * 8 ? 65535 * 8 : g));
}

}  // namespace

extern "C" {

int pp_winograd_input_f32(const float* x, int ld_x, long long batch_stride, int B, int H, int W, int C, int relu, float* U, void* stream) {
    if (!x || !U || B <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1) || C <= 0 || (C & 3) || ld_x < C || (ld_x & 3) || (batch_stride & 3)) return PP_EINVAL;
    if (((uintptr_t)x & 15) || ((uintptr_t)U & 15)) return PP_EINVAL;
    const long long total = (long long)B * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(wino_input_kernel, dim3(grid_of(total)), dim3(256), 0, (hipStream_t)stream, x, ld_x, batch_stride, B, H, W, C, relu, U);
    return pp_last_launch();
}

int pp_winograd_weight_f32(const float* w, int Cout, int Cin, int ldw, float* V, void* stream) {
    if (!w || !V || Cout <= 0 || Cin <= 0 || ldw < 9 * Cin) return PP_EINVAL;
    hipLaunchKernelGGL(wino_weight_kernel, dim3(grid_of((long long)Cout * Cin)), dim3(256), 0, (hipStream_t)stream, w, Cout, Cin, ldw, V);
    return pp_last_launch();
}

int pp_winograd_output_f32(const float* Y, int B, int H, int W, int Cout, const float* bias, int act, const float* residual,
                           const float* residual2, float* out, int ldc, void* stream) {
    if (!Y || !out || B <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1) || Cout <= 0 || ldc < Cout) return PP_EINVAL;
    if (act != PP_ACT_NONE && act != PP_ACT_RELU && act != PP_ACT_LEAKY01) return PP_EINVAL;
    const bool vec = (Cout & 3) == 0 && (ldc & 3) == 0 && (((uintptr_t)Y | (uintptr_t)out | (uintptr_t)residual | (uintptr_t)residual2 | (uintptr_t)bias) & 15) == 0;
    const long long total = (long long)B * (H / 2) * (W / 2) * (vec ? Cout / 4 : Cout);
    if (vec)
        hipLaunchKernelGGL(wino_output_kernel<true>, dim3(grid_of(total)), dim3(256), 0, (hipStream_t)stream, Y, B, H, W, Cout, bias, act, residual,
                           residual2, out, ldc);
    else
        hipLaunchKernelGGL(wino_output_kernel<false>, dim3(grid_of(total)), dim3(256), 0, (hipStream_t)stream, Y, B, H, W, Cout, bias, act, residual,
                           residual2, out, ldc);
    return pp_last_launch();
}


/* ---- F(4x4, 3x3) on the f16x3 engine ---- */
int pp_winograd4_input_hl(const void* x_hl, int ld_x, long long batch_stride, int B, int H, int W, int C, int relu, void* U_hl, long long P_pad,
                          void* stream) {
    if (!x_hl || !U_hl || B <= 0 || H < 4 || W < 4 || (H & 3) || (W & 3) || C <= 0 || (C & 7) || ld_x < C || (ld_x & 7) || (batch_stride & 7)) return PP_EINVAL;
    if (((uintptr_t)x_hl & 15) || ((uintptr_t)U_hl & 15)) return PP_EINVAL;
    const long long total = (long long)B * (H / 4) * (W / 4) * (C / 4);
    if ((total + 255) / 256 + 8 >= (1LL << 31) || P_pad < (long long)B * (H / 4) * (W / 4)) return PP_EINVAL;
    // extent of the operand from x_hl on (32-bit byte offsets of the bounded loads; 0xFFFFFFFF is the "reads zero" offset)
    const long long x_bytes = ((long long)(B - 1) * batch_stride + (long long)(H * W - 1) * ld_x + C) * 4;
    if (x_bytes + (long long)(W + 1) * ld_x * 4 >= 0xFFFFFF00LL || P_pad * C * 4 >= 0xFFFFFF00LL) return PP_EINVAL;
    if (relu)
        hipLaunchKernelGGL(wino4_input_kernel<true>, dim3(grid8_of(total)), dim3(256), 0, (hipStream_t)stream, (const _Float16*)x_hl, (unsigned)x_bytes,
                           ld_x, batch_stride, B, H, W, C, (_Float16*)U_hl, P_pad, pp_saturation_word());
    else
        hipLaunchKernelGGL(wino4_input_kernel<false>, dim3(grid8_of(total)), dim3(256), 0, (hipStream_t)stream, (const _Float16*)x_hl, (unsigned)x_bytes,
                           ld_x, batch_stride, B, H, W, C, (_Float16*)U_hl, P_pad, pp_saturation_word());
    return pp_last_launch();
}

int pp_winograd4_weight_f32(const float* w, int Cout, int Cin, int ldw, float* V, void* stream) {
    if (!w || !V || Cout <= 0 || Cin <= 0 || ldw < 9 * Cin) return PP_EINVAL;
    const long long total = (long long)Cout * Cin;
    hipLaunchKernelGGL(wino4_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, Cout, Cin, ldw, V);
    return pp_last_launch();
}

int pp_winograd4_output(const float* Y, int ld_y, int B, int H, int W, int Cout, const float* bias, int act, const float* residual,
                        const float* residual2, float* out, int ldc, void* out_hl, int ld_h, int c_relu, long long P_pad, void* stream) {
    if (P_pad < (long long)B * (H / 4) * (W / 4) || ld_y < Cout || (ld_y & 3) || P_pad * ld_y * 4 >= 0xFFFFFF00LL) return PP_EINVAL;
    if (!Y || (!out && !out_hl) || B <= 0 || H < 4 || W < 4 || (H & 3) || (W & 3) || Cout <= 0 || (Cout & 3)) return PP_EINVAL;
    if (act != PP_ACT_NONE && act != PP_ACT_RELU && act != PP_ACT_LEAKY01) return PP_EINVAL;
    if (out && (ldc < Cout || (ldc & 3))) return PP_EINVAL;
    if (!out && (residual || residual2)) return PP_EINVAL;
    if (out_hl && ((Cout & 7) || ld_h < Cout || (ld_h & 7))) return PP_EINVAL;
    if (((uintptr_t)Y | (uintptr_t)out | (uintptr_t)residual | (uintptr_t)residual2 | (uintptr_t)bias | (uintptr_t)out_hl) & 15) return PP_EINVAL;
    const long long total = (long long)B * (H / 4) * (W / 4) * (Cout / 4);
    if ((total + 255) / 256 + 8 >= (1LL << 31)) return PP_EINVAL;
    hipLaunchKernelGGL(wino4_output_kernel, dim3(grid8_of(total)), dim3(256), 0, (hipStream_t)stream, Y, B, H, W, Cout, bias, act, residual, residual2,
                       out, ldc, (_Float16*)out_hl, ld_h, c_relu, P_pad, ld_y, pp_saturation_word());
    return pp_last_launch();
}


int pp_winograd4_chain(const float* Y, int ld_y, int B, int H, int W, int C, const float* bias, int act, int c_relu, void* U_hl, long long P_pad,
                       void* stream) {
    if (!Y || !U_hl || B <= 0 || H < 4 || (H & 3) || (W != 16 && W != 32 && W != 64) || C <= 0 || (C & 31) || ld_y < C || (ld_y & 3)) return PP_EINVAL;
    if (P_pad * ld_y * 4 >= 0xFFFFFF00LL) return PP_EINVAL;
    if (act != PP_ACT_NONE && act != PP_ACT_RELU && act != PP_ACT_LEAKY01) return PP_EINVAL;
    if (P_pad < (long long)B * (H / 4) * (W / 4) || P_pad * C * 4 >= 0xFFFFFF00LL || (long long)B * (C / 32) >= (1LL << 31)) return PP_EINVAL;
    if (((uintptr_t)Y | (uintptr_t)U_hl | (uintptr_t)bias) & 15) return PP_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (W == 64) return wino4_chain_launch<64>(Y, B, H, C, bias, act, c_relu, U_hl, P_pad, ld_y, st);
    if (W == 32) return wino4_chain_launch<32>(Y, B, H, C, bias, act, c_relu, U_hl, P_pad, ld_y, st);
    return wino4_chain_launch<16>(Y, B, H, C, bias, act, c_relu, U_hl, P_pad, ld_y, st);
}


int pp_winograd_chain_f32(const float* Y, int B, int H, int W, int C, const float* bias, int act, int relu_next, float* U, void* stream) {
    if (!Y || !U || B <= 0 || H < 2 || (H & 1) || (W != 16 && W != 32 && W != 64) || C <= 0 || (C & 31)) return PP_EINVAL;
    if (act != PP_ACT_NONE && act != PP_ACT_RELU && act != PP_ACT_LEAKY01) return PP_EINVAL;
    if ((long long)B * (C / 32) >= (1LL << 31) || (((uintptr_t)Y | (uintptr_t)U | (uintptr_t)bias) & 15)) return PP_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (W == 64) return wino2_chain_launch<64>(Y, B, H, C, bias, act, relu_next, U, st);
    if (W == 32) return wino2_chain_launch<32>(Y, B, H, C, bias, act, relu_next, U, st);
    return wino2_chain_launch<16>(Y, B, H, C, bias, act, relu_next, U, st);
}

}  // extern "C"
