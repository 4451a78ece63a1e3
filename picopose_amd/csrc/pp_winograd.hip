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

}  // extern "C"
