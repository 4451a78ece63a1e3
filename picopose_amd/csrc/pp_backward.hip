// First backward slice of the training path (utils/lite.py:33-49 calls loss.backward() on the sum of the
// `loss*` entries of model/picopose.py:114-137): the row-wise / element-wise pieces whose matrix products run on the GEMM engine
// (picopose_amd/autograd.py: dgrad = dz W, wgrad = dz^T x as pp_gemm launches).  Everything IN THIS FILE is deterministic: reductions
// run in a fixed order (no atomics), so a gradient does not depend on the launch configuration.  (The two scatter adjoints of
// pp_backward3.hip — feature warp, correlation lookup — use fp32 atomics by default and 64-bit fixed-point integer atomics with
// autograd.DETERMINISTIC = True; only then is the whole backward bit-reproducible.)
//   scope: InfoNCE (utils/loss_utils.py:144-175) -> the last ViT block (layers/block.py:82-107, attention.py:49-62, mlp.py:35-41,
//   layer_scale.py:27-28, nn.LayerNorm) and the stage-2 losses (:177-186) -> AffineRegressor (model/stage2/affine_regressor.py).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "../../include/picopose_hip.h"
#include "pp_common.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// out[c] = sum_r x[r][c] (x rows of ld floats), rows cut into gridDim.y slabs summed in slab order: part[slab][c]
__global__ __launch_bounds__(256) void colsum_part_kernel(const float* __restrict__ x, long long rows, int cols, int ld, float* __restrict__ part) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    const long long per = (rows + gridDim.y - 1) / gridDim.y, r0 = per * blockIdx.y, r1 = r0 + per < rows ? r0 + per : rows;
    float s = 0.f;
    for (long long r = r0; r < r1; ++r) s += x[r * ld + c];
    part[(size_t)blockIdx.y * cols + c] = s;
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, int slabs, int cols, float* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    float s = 0.f;
    for (int k = 0; k < slabs; ++k) s += part[(size_t)k * cols + c];
    out[c] = s;
}

// the same with 16-byte loads: a thread owns 4 consecutive columns and one of 4 row lanes of its slab (rows rl, rl + 4, ... with
// four rows in flight), the row lanes are added in lane order through LDS.  cols % 4 == 0, ld % 4 == 0, 16-byte aligned x.
__global__ __launch_bounds__(256) void colsum_part4_kernel(const float* __restrict__ x, long long rows, int cols, int ld, int per, float* __restrict__ part) {
    __shared__ f4 red[4][64];
    const int q = threadIdx.x & 63, rl = threadIdx.x >> 6, c = (blockIdx.x * 64 + q) * 4;
    const long long r0 = (long long)per * blockIdx.y, r1 = r0 + per < rows ? r0 + per : rows;
    f4 s = {0.f, 0.f, 0.f, 0.f};
    if (c < cols) {
        long long r = r0 + rl;
        for (; r + 12 < r1; r += 16) {
            const f4 a = *(const f4*)(x + r * ld + c), b = *(const f4*)(x + (r + 4) * ld + c);
            const f4 e = *(const f4*)(x + (r + 8) * ld + c), g = *(const f4*)(x + (r + 12) * ld + c);
            s += a;
            s += b;
            s += e;
            s += g;
        }
        for (; r < r1; r += 4) s += *(const f4*)(x + r * ld + c);
    }
    red[rl][q] = s;
    __syncthreads();
    if (rl == 0 && c < cols) {
        const f4 t = (red[0][q] + red[1][q]) + (red[2][q] + red[3][q]);
        *(f4*)(part + (size_t)blockIdx.y * cols + c) = t;
    }
}

// fold of the slabs with SIXTEEN slab lanes per column (slabs sl, sl + 16, ... each with eight loads in flight, then the lanes in
// order): a workgroup = 16 columns.  (One thread per column walks up to 512 dependent-latency loads; four lanes still took 17 us.)
__global__ __launch_bounds__(256) void colsum_final4_kernel(const float* __restrict__ part, int slabs, int cols, float* __restrict__ out) {
    __shared__ float red[16][17];
    const int q = threadIdx.x & 15, sl = threadIdx.x >> 4, c = blockIdx.x * 16 + q;
    float s = 0.f;
    if (c < cols) {
        int k = sl;
        for (; k + 112 < slabs; k += 128) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = part[(size_t)(k + 16 * j) * cols + c];
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[j];
        }
        for (; k < slabs; k += 16) s += part[(size_t)k * cols + c];
    }
    red[sl][q] = s;
    __syncthreads();
    if (sl == 0 && c < cols) {
        float t = red[0][q];
#pragma unroll
        for (int j = 1; j < 16; ++j) t += red[j][q];
        out[c] = t;
    }
}

__device__ __forceinline__ float act_fwd(float z, int act) {
    switch (act) {
        case PP_ACT_RELU: return z > 0.f ? z : 0.f;
        case PP_ACT_GELU: return 0.5f * z * (1.0f + erff(z * 0.70710678118654752440f));
        case PP_ACT_LEAKY01: return z > 0.f ? z : 0.1f * z;
        case PP_ACT_TANH: return tanhf(z);
        default: return z;
    }
}
__device__ __forceinline__ float act_grad(float z, int act) {
    switch (act) {
        case PP_ACT_RELU: return z > 0.f ? 1.f : 0.f;
        case PP_ACT_GELU: return 0.5f * (1.0f + erff(z * 0.70710678118654752440f)) + z * 0.39894228040143267794f * expf(-0.5f * z * z);
        case PP_ACT_LEAKY01: return z > 0.f ? 1.f : 0.1f;
        case PP_ACT_TANH: {
            const float t = tanhf(z);
            return 1.f - t * t;
        }
        default: return 1.f;
    }
}
__global__ __launch_bounds__(256) void act_forward_kernel(const float* __restrict__ z, long long n, int act, float* __restrict__ y) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) y[i] = act_fwd(z[i], act);
}
__global__ __launch_bounds__(256) void act_backward_kernel(const float* __restrict__ z, const float* __restrict__ dy, long long n, int act,
                                                           float* __restrict__ dz) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) dz[i] = dy[i] * act_grad(z[i], act);
}

// element-wise helpers of the slice: op 0: out = a * b; 1: out = a * v[col]; 2: out = a + b
__global__ __launch_bounds__(256) void ew_kernel(int op, const float* __restrict__ a, const float* __restrict__ b, long long n, int cols,
                                                 float* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float x = a[i];
        out[i] = op == 0 ? x * b[i] : (op == 1 ? x * b[i % cols] : x + b[i]);
    }
}

// nn.LayerNorm backward, one wave per row: xhat = (x - mean) rstd; dxhat = dy gamma;
// dx = rstd (dxhat - mean(dxhat) - xhat mean(dxhat xhat)); gx = dy xhat (for dgamma = column sums of gx; dbeta = column sums of dy)
__global__ __launch_bounds__(256) void layernorm_backward_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                                 const float* __restrict__ dy, int rows, int C, float eps,
                                                                 float* __restrict__ dx, float* __restrict__ gx) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (size_t)row * C;
    const float* dr = dy + (size_t)row * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += xr[c];
    const float mean = wave_sum(s) / (float)C;
    float v = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float d = xr[c] - mean;
        v = fmaf(d, d, v);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(v) / (float)C + eps);
    float a = 0.f, b = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float xh = (xr[c] - mean) * rstd, dh = dr[c] * g[c];
        a += dh;
        b = fmaf(dh, xh, b);
    }
    a = wave_sum(a) / (float)C;
    b = wave_sum(b) / (float)C;
    for (int c = lane; c < C; c += 64) {
        const float xh = (xr[c] - mean) * rstd, dh = dr[c] * g[c];
        dx[(size_t)row * C + c] = rstd * (dh - a - xh * b);
        gx[(size_t)row * C + c] = dr[c] * xh;
    }
}

// nn.GroupNorm(G, C) (+ReLU) backward on NHWC, one workgroup per (image, group); y = relu?(xhat gamma + beta).
// dy is masked by the ReLU first (y_pre > 0), then the LayerNorm formulas over the group's HW x C/G elements.
__global__ __launch_bounds__(256) void groupnorm_backward_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                                 const float* __restrict__ bta, const float* __restrict__ dy, int HW, int C,
                                                                 int G, float eps, int relu, float* __restrict__ dx, float* __restrict__ gx,
                                                                 float* __restrict__ gy) {
    __shared__ float red[16];
    const int img = blockIdx.x / G, grp = blockIdx.x % G, cg = C / G, tid = threadIdx.x;
    const size_t base = (size_t)img * HW * C + grp * cg;
    const int n = HW * cg;
    auto block_sum2 = [&](float u, float w, float& su, float& sw) {
        u = wave_sum(u);
        w = wave_sum(w);
        __syncthreads();
        if ((tid & 63) == 0) {
            red[tid >> 6] = u;
            red[4 + (tid >> 6)] = w;
        }
        __syncthreads();
        su = red[0] + red[1] + red[2] + red[3];
        sw = red[4] + red[5] + red[6] + red[7];
    };
    float s = 0.f, dummy;
    for (int i = tid; i < n; i += 256) s += x[base + (size_t)(i / cg) * C + (i % cg)];
    float mean;
    block_sum2(s, 0.f, mean, dummy);
    mean /= (float)n;
    float v = 0.f;
    for (int i = tid; i < n; i += 256) {
        const float d = x[base + (size_t)(i / cg) * C + (i % cg)] - mean;
        v = fmaf(d, d, v);
    }
    float var;
    block_sum2(v, 0.f, var, dummy);
    const float rstd = 1.0f / sqrtf(var / (float)n + eps);
    float a = 0.f, b = 0.f;
    for (int i = tid; i < n; i += 256) {
        const int c = grp * cg + i % cg;
        const size_t off = base + (size_t)(i / cg) * C + (i % cg);
        const float xh = (x[off] - mean) * rstd;
        float d = dy[off];
        if (relu && !(xh * g[c] + bta[c] > 0.f)) d = 0.f;
        const float dh = d * g[c];
        a += dh;
        b = fmaf(dh, xh, b);
    }
    float sa, sb;
    block_sum2(a, b, sa, sb);
    sa /= (float)n;
    sb /= (float)n;
    for (int i = tid; i < n; i += 256) {
        const int c = grp * cg + i % cg;
        const size_t off = base + (size_t)(i / cg) * C + (i % cg);
        const float xh = (x[off] - mean) * rstd;
        float d = dy[off];
        if (relu && !(xh * g[c] + bta[c] > 0.f)) d = 0.f;
        dx[off] = rstd * (d * g[c] - sa - xh * sb);
        gx[off] = d * xh;
        gy[off] = d;
    }
}

// softmax backward per row: ds = p (dp - sum_j dp_j p_j)
__global__ __launch_bounds__(256) void softmax_backward_kernel(const float* __restrict__ p, const float* __restrict__ dp, long long rows, int n,
                                                               float* __restrict__ ds) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* pr = p + row * n;
    const float* dr = dp + row * n;
    float s = 0.f;
    for (int c = lane; c < n; c += 64) s = fmaf(pr[c], dr[c], s);
    s = wave_sum(s);
    for (int c = lane; c < n; c += 64) ds[row * n + c] = pr[c] * (dr[c] - s);
}

// d/dlogits of mean_i [logsumexp_j(scale L_ij) - scale L_ii]: (softmax_j(scale L_i.) - delta_ij) scale / n, times the upstream scalar
__global__ __launch_bounds__(256) void xent_diag_backward_kernel(const float* __restrict__ logits, int n, int ld, float scale, const float* __restrict__ up,
                                                                 float* __restrict__ dl) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n) return;
    const float* lr = logits + (size_t)row * ld;
    float mx = -INFINITY;
    for (int c = lane; c < n; c += 64) mx = fmaxf(mx, lr[c] * scale);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float s = 0.f;
    for (int c = lane; c < n; c += 64) s += expf(lr[c] * scale - mx);
    s = wave_sum(s);
    const float k = up[0] * scale / (float)n;
    for (int c = lane; c < n; c += 64) dl[(size_t)row * n + c] = (expf(lr[c] * scale - mx) / s - (c == row ? 1.f : 0.f)) * k;
}

// F.normalize backward per row: q = x / max(|x|, eps); dx = (dq - q (q . dq)) / max(|x|, eps)
__global__ __launch_bounds__(256) void normalize_backward_kernel(const float* __restrict__ x, long long row_stride, const long long* __restrict__ index,
                                                                 const float* __restrict__ dq, int rows, int C, float eps, float* __restrict__ dx) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (index ? index[row] : (long long)row) * row_stride;
    const float* dr = dq + (size_t)row * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s = fmaf(xr[c], xr[c], s);
    const float nrm = fmaxf(sqrtf(wave_sum(s)), eps);
    float d = 0.f;
    for (int c = lane; c < C; c += 64) d = fmaf(xr[c] / nrm, dr[c], d);
    d = wave_sum(d);
    for (int c = lane; c < C; c += 64) dx[(size_t)row * C + c] = (dr[c] - (xr[c] / nrm) * d) / nrm;
}

// dst[index[i]] += src[i] for i = 0 .. n-1, duplicates in `index` allowed, WITHOUT atomics: the wave of source row i checks whether an
// earlier source row has the same destination (then that row's wave does the work) and otherwise adds up, in ascending i, every
// source row of this destination — torch.gather's backward (utils/torch_utils.py:257-283 under autograd) with a fixed summation
// order.  O(n^2 / 64) index compares per wave: n is a few thousand key-points.
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const float* __restrict__ src, const long long* __restrict__ index, int n, int C,
                                                               float* __restrict__ dst) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n) return;
    const long long row = index[i];
    bool earlier = false;
    for (int j = lane; j < i; j += 64) earlier |= index[j] == row;
    if (__ballot(earlier)) return;                                   // not the first source row of this destination (wave-uniform)
    float* d = dst + row * (long long)C;
    for (int c = lane; c < C; c += 64) d[c] += src[(size_t)i * C + c];
    for (int j0 = i + 1; j0 < n; j0 += 64) {
        const int j = j0 + lane;
        unsigned long long hit = __ballot(j < n && index[j] == row);
        while (hit) {                                                 // ascending j
            const int k = __ffsll((long long)hit) - 1;
            hit &= hit - 1;
            for (int c = lane; c < C; c += 64) d[c] += src[(size_t)(j0 + k) * C + c];
        }
    }
}

// im2col of an NHWC image for a k x k / stride s / pad p convolution: col[(b, oy, ox)][(ky, kx, ci)]
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ x, int H, int W, int C, int k, int s, int p, int Ho, int Wo,
                                                     long long total, float* __restrict__ col) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ci = (int)(i % C);
        long long t = i / C;
        const int kx = (int)(t % k);
        t /= k;
        const int ky = (int)(t % k);
        t /= k;
        const int ox = (int)(t % Wo);
        t /= Wo;
        const int oy = (int)(t % Ho);
        const int b = (int)(t / Ho);
        const int iy = oy * s - p + ky, ix = ox * s - p + kx;
        col[i] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? x[(((size_t)b * H + iy) * W + ix) * C + ci] : 0.f;
    }
}
// ... and its adjoint in gather form (an input pixel sums the windows it belongs to, in a fixed order)
__global__ __launch_bounds__(256) void col2im_kernel(const float* __restrict__ col, int H, int W, int C, int k, int s, int p, int Ho, int Wo,
                                                     long long total, float* __restrict__ dx) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ci = (int)(i % C);
        long long t = i / C;
        const int ix = (int)(t % W);
        t /= W;
        const int iy = (int)(t % H);
        const int b = (int)(t / H);
        float acc = 0.f;
        for (int ky = 0; ky < k; ++ky) {
            const int oyn = iy + p - ky;
            if (oyn < 0 || oyn % s != 0 || oyn / s >= Ho) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int oxn = ix + p - kx;
                if (oxn < 0 || oxn % s != 0 || oxn / s >= Wo) continue;
                acc += col[((((size_t)b * Ho + oyn / s) * Wo + oxn / s) * k * k + ky * k + kx) * C + ci];
            }
        }
        dx[i] = acc;
    }
}

inline int grid_for(long long n) { return (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192); }

}  // namespace

// adjoint of the tail of matching_features_similarity (utils/matching.py:21-25): out[b][s][h][w] = max(S[b][t][s] m[b][s], 0) with
// t = w 16 + h and m the template mask sampled at the patch (F.interpolate nearest, :16)  ->  dS[b][t][s] = dout m [out > 0]
__global__ void simvol_backward_kernel(const float* __restrict__ out, const float* __restrict__ dout, const float* __restrict__ mask, int mh,
                                       int mw, long long total, float* __restrict__ dS) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // over dS: ((b 256 + t) 256 + s)
    if (i >= total) return;
    const int s = (int)(i & 255), t = (int)((i >> 8) & 255);
    const long long b = i >> 16;
    const int sy = s >> 4, sx = s & 15;
    int my = (int)floorf((float)sy * ((float)mh / 16.0f)), mx = (int)floorf((float)sx * ((float)mw / 16.0f));
    my = my < mh - 1 ? my : mh - 1;
    mx = mx < mw - 1 ? mx : mw - 1;
    const float m = mask[(b * mh + my) * mw + mx];
    const long long o = ((b * 256 + s) * 16 + (t & 15)) * 16 + (t >> 4);
    dS[i] = out[o] > 0.f ? dout[o] * m : 0.f;
}

// im2col written TRANSPOSED: colT[(ky, kx, c)][row], row = (b, oy, ox) — the K-major operand of the weight-gradient product
// dW = dz^T col (its K axis is the rows of the batch).  A 32 x 32 (pixels x channels) tile per tap goes through LDS so that both the
// reads (channels contiguous) and the writes (rows contiguous) are coalesced.
__global__ __launch_bounds__(256) void im2col_t_kernel(const float* __restrict__ x, int H, int W, int C, int k, int stride, int pad, int Ho, int Wo,
                                                       int rows, float* __restrict__ colT) {
    // a block = 32 pixels x 32 channels for ALL taps (the pixel decomposition is done once, in 32-bit arithmetic)
    __shared__ float tile[32][33];
    const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    int iy0[4], ix0[4];
    long long xb[4];
    bool ok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned row = (unsigned)(r0 + ty + 8 * j);
        ok[j] = row < (unsigned)rows && c0 + tx < C;
        const unsigned per = (unsigned)(Ho * Wo), b = row / per, rem = row - b * per, oy = rem / (unsigned)Wo, ox = rem - oy * (unsigned)Wo;
        iy0[j] = (int)oy * stride - pad;
        ix0[j] = (int)ox * stride - pad;
        xb[j] = (long long)b * H * W * C + c0 + tx;
    }
    int ky = 0, kx = 0;
    for (int tap = 0; tap < k * k; ++tap) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int iy = iy0[j] + ky, ix = ix0[j] + kx;
            tile[ty + 8 * j][tx] = (ok[j] && iy >= 0 && iy < H && ix >= 0 && ix < W) ? x[xb[j] + ((long long)iy * W + ix) * C] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c0 + ty + 8 * j;
            if (r0 + tx < rows && c < C) colT[((long long)tap * C + c) * rows + r0 + tx] = tile[tx][ty + 8 * j];
        }
        __syncthreads();
        if (++kx == k) {
            kx = 0;
            ++ky;
        }
    }
}

// x (rows, cols) fp32, row stride ld -> the engine operand of s x^T: `cols` operand rows with K = rows (hl: [cols][rows / 8][2][8]
// halfs; h: [cols][rows]), s = scale[0] (a device scalar, null = 1) x the operand scale 4.  One pass instead of three (scale copy,
// transposed copy, split) for the K-major operands of the backward products (dW = dz^T x: both dz and x are read "down the rows").
// A 64 (rows) x 32 (cols) tile goes through LDS; a thread then owns 8 consecutive rows of one column = one 32-byte operand group.
template <int TERMS>
__global__ __launch_bounds__(256) void split_transpose_kernel(const float* __restrict__ x, long long rows, int cols, int ld, const float* __restrict__ scale,
                                                              _Float16* __restrict__ out, long long ld_out) {
    __shared__ float tile[64][33];
    const long long r0 = (long long)blockIdx.x * 64;
    const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;    // 32 x 8
    const float s = (scale ? scale[0] : 1.f) * PP_A_SCALE;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const long long r = r0 + ty + 8 * j;
        tile[ty + 8 * j][tx] = (r < rows && c0 + tx < cols) ? x[r * ld + c0 + tx] * s : 0.f;
    }
    __syncthreads();
    const int c = threadIdx.x >> 3, g = threadIdx.x & 7;       // column of the tile, group of 8 rows
    if (c0 + c < cols && r0 + 8 * g < ld_out) {                // (rows % 8 == 0: a group is in or out as a whole; [rows, ld_out) is zeros)
        h8 hh, ll;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float v = tile[8 * g + k][c];
            const _Float16 h = (_Float16)fminf(fmaxf(v, -65504.f), 65504.f);
            hh[k] = h;
            ll[k] = (_Float16)fminf(fmaxf(v - (float)h, -65504.f), 65504.f);
        }
        _Float16* o = out + ((long long)(c0 + c) * ld_out + r0 + 8 * g) * TERMS;
        *(h8*)o = hh;
        if (TERMS == 2) *(h8*)(o + 8) = ll;
    }
}

// im2col_t_kernel writing the ENGINE OPERAND of colT directly (operand row = (tap, channel), K = the pixels of the batch, groups of 8
// pixels): the fp32 K-major matrix (k k C x rows x 4 bytes: 15 GB for the 640-channel decoder input at 64 x 64 x 160) is never
// stored or re-read by a split pass.  A 64 (pixels) x 32 (channels) tile per tap goes through LDS.
template <int TERMS>
__global__ __launch_bounds__(256) void im2col_t_operand_kernel(const float* __restrict__ x, int H, int W, int C, int k, int stride, int pad, int Ho, int Wo,
                                                               int rows, _Float16* __restrict__ out) {
    // a block = 64 pixels x 32 channels for ALL taps: the pixel decomposition (three divisions per pixel) is done once, the taps re-read
    // the same input rows out of L1 / L2; float4 loads along the channels (C % 4 == 0)
    __shared__ float tile[64][33];
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 32;
    const int lp = threadIdx.x >> 3, c4 = (threadIdx.x & 7) * 4;
    const bool cv = c0 + c4 < C;
    int iy0[2], ix0[2];
    long long xb[2];
    bool ok[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const unsigned row = (unsigned)(r0 + lp + 32 * j);
        ok[j] = cv && row < (unsigned)rows;
        const unsigned per = (unsigned)(Ho * Wo), b = row / per, rem = row - b * per, oy = rem / (unsigned)Wo, ox = rem - oy * (unsigned)Wo;
        iy0[j] = (int)oy * stride - pad;
        ix0[j] = (int)ox * stride - pad;
        xb[j] = (long long)b * H * W * C + c0 + c4;
    }
    const int wc = threadIdx.x >> 3, g = threadIdx.x & 7;
    const bool wv = c0 + wc < C && r0 + 8 * g < rows;
    int ky = 0, kx = 0;
    for (int tap = 0; tap < k * k; ++tap) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int iy = iy0[j] + ky, ix = ix0[j] + kx;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok[j] && iy >= 0 && iy < H && ix >= 0 && ix < W) v = *(const float4*)(x + xb[j] + ((long long)iy * W + ix) * C);
            float* t = &tile[lp + 32 * j][c4];
            t[0] = v.x * PP_A_SCALE;
            t[1] = v.y * PP_A_SCALE;
            t[2] = v.z * PP_A_SCALE;
            t[3] = v.w * PP_A_SCALE;
        }
        __syncthreads();
        if (wv) {
            h8 hh, ll;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float v = tile[8 * g + q][wc];
                const _Float16 h = (_Float16)fminf(fmaxf(v, -65504.f), 65504.f);
                hh[q] = h;
                ll[q] = (_Float16)fminf(fmaxf(v - (float)h, -65504.f), 65504.f);
            }
            _Float16* o = out + (((long long)tap * C + c0 + wc) * rows + r0 + 8 * g) * TERMS;
            *(h8*)o = hh;
            if (TERMS == 2) *(h8*)(o + 8) = ll;
        }
        __syncthreads();
        if (++kx == k) {
            kx = 0;
            ++ky;
        }
    }
}

extern "C" {

int pp_im2col_t_operand(const float* x, int B, int H, int W, int C, int ksize, int stride, int pad, void* out, int terms, void* stream) {
    if (!x || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4 != 0 || ksize <= 0 || stride <= 0 || pad < 0 || (terms != 1 && terms != 2) ||
        ((uintptr_t)out & 15) != 0 || ((uintptr_t)x & 15) != 0)
        return PP_EINVAL;
    const int Ho = (H + 2 * pad - ksize) / stride + 1, Wo = (W + 2 * pad - ksize) / stride + 1;
    const long long rows = (long long)B * Ho * Wo;
    if (rows % 8 != 0 || rows >= (1LL << 31) - 64 || (C + 31) / 32 > 65535) return PP_EINVAL;
    const dim3 grid((unsigned)((rows + 63) / 64), (unsigned)((C + 31) / 32));
    if (terms == 2)
        hipLaunchKernelGGL(im2col_t_operand_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, x, H, W, C, ksize, stride, pad, Ho, Wo, (int)rows, (_Float16*)out);
    else
        hipLaunchKernelGGL(im2col_t_operand_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, x, H, W, C, ksize, stride, pad, Ho, Wo, (int)rows, (_Float16*)out);
    return pp_last_launch();
}

int pp_split_transpose_ld(const float* x, long long rows, int cols, int ld, const float* scale, void* out, long long ld_out, int terms,
                          void* stream) {
    if (!x || !out || rows <= 0 || cols <= 0 || ld < cols || rows % 8 != 0 || ld_out < rows || ld_out % 8 != 0 || (terms != 1 && terms != 2) ||
        ((uintptr_t)out & 15) != 0)
        return PP_EINVAL;
    const dim3 grid((unsigned)((ld_out + 63) / 64), (unsigned)((cols + 31) / 32));
    if (grid.y > 65535) return PP_EINVAL;
    if (terms == 2) hipLaunchKernelGGL(split_transpose_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, x, rows, cols, ld, scale, (_Float16*)out, ld_out);
    else hipLaunchKernelGGL(split_transpose_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, x, rows, cols, ld, scale, (_Float16*)out, ld_out);
    return pp_last_launch();
}

int pp_split_transpose_t(const float* x, long long rows, int cols, int ld, const float* scale, void* out, int terms, void* stream) {
    return pp_split_transpose_ld(x, rows, cols, ld, scale, out, rows, terms, stream);
}

// slabs of at least 64 rows, enough of them to fill the chip whatever the width (a 256-column matrix is ONE column block: the
// slab count is all its parallelism), at most 2048; a function of the shape only, so the summation order is fixed
static int colsum_slabs(long long rows, int cols) {
    if (rows < 64) return 1;
    const long long want = 2048 / ((cols + 255) / 256) + 1, cap = rows / 64;
    long long s = want < cap ? want : cap;
    return (int)(s < 1 ? 1 : (s > 512 ? 512 : s));
}

size_t pp_colsum_workspace_bytes(long long rows, int cols) { return (size_t)colsum_slabs(rows, cols) * cols * sizeof(float); }

int pp_colsum(const float* x, long long rows, int cols, int ld, float* out, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !out || !workspace || rows <= 0 || cols <= 0 || ld < cols) return PP_EINVAL;
    const int slabs = colsum_slabs(rows, cols);
    if (workspace_bytes < (size_t)slabs * cols * sizeof(float)) return PP_EWORKSPACE;
    if (cols % 4 == 0 && ld % 4 == 0 && ((uintptr_t)x & 15) == 0) {
        const int per = (int)((rows + slabs - 1) / slabs);
        hipLaunchKernelGGL(colsum_part4_kernel, dim3((cols + 255) / 256, slabs), dim3(256), 0, (hipStream_t)stream, x, rows, cols, ld, per, (float*)workspace);
        hipLaunchKernelGGL(colsum_final4_kernel, dim3((cols + 15) / 16), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, slabs, cols, out);
        return pp_last_launch();
    }
    hipLaunchKernelGGL(colsum_part_kernel, dim3((cols + 255) / 256, slabs), dim3(256), 0, (hipStream_t)stream, x, rows, cols, ld, (float*)workspace);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((cols + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, slabs, cols, out);
    return pp_last_launch();
}

int pp_act_forward(const float* z, long long n, int act, float* y, void* stream) {
    if (!z || !y || n <= 0 || act < 0 || act > PP_ACT_TANH) return PP_EINVAL;
    hipLaunchKernelGGL(act_forward_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, z, n, act, y);
    return pp_last_launch();
}

int pp_act_backward(const float* z, const float* dy, long long n, int act, float* dz, void* stream) {
    if (!z || !dy || !dz || n <= 0 || act < 0 || act > PP_ACT_TANH) return PP_EINVAL;
    hipLaunchKernelGGL(act_backward_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, z, dy, n, act, dz);
    return pp_last_launch();
}

int pp_elementwise(int op, const float* a, const float* b, long long n, int cols, float* out, void* stream) {
    if (!a || !b || !out || n <= 0 || op < 0 || op > 2 || (op == 1 && cols <= 0)) return PP_EINVAL;
    hipLaunchKernelGGL(ew_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, op, a, b, n, cols > 0 ? cols : 1, out);
    return pp_last_launch();
}

int pp_layernorm_backward(const float* x, const float* gamma, const float* dy, int rows, int C, float eps, float* dx, float* gx, void* stream) {
    if (!x || !gamma || !dy || !dx || !gx || rows <= 0 || C <= 0) return PP_EINVAL;
    hipLaunchKernelGGL(layernorm_backward_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, gamma, dy, rows, C, eps, dx, gx);
    return pp_last_launch();
}

int pp_groupnorm_backward_nhwc(const float* x, const float* gamma, const float* beta, const float* dy, int B, int HW, int C, int groups,
                               float eps, int relu, float* dx, float* gx, float* gy, void* stream) {
    if (!x || !gamma || !beta || !dy || !dx || !gx || !gy || B <= 0 || HW <= 0 || C <= 0 || groups <= 0 || C % groups != 0) return PP_EINVAL;
    hipLaunchKernelGGL(groupnorm_backward_kernel, dim3(B * groups), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, dy, HW, C, groups, eps,
                       relu, dx, gx, gy);
    return pp_last_launch();
}

int pp_softmax_backward_rows(const float* p, const float* dp, long long rows, int n, float* ds, void* stream) {
    if (!p || !dp || !ds || rows <= 0 || n <= 0) return PP_EINVAL;
    hipLaunchKernelGGL(softmax_backward_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p, dp, rows, n, ds);
    return pp_last_launch();
}

int pp_xent_diag_backward(const float* logits, int n, int ld, float scale, const float* upstream, float* dlogits, void* stream) {
    if (!logits || !upstream || !dlogits || n <= 0 || ld < n) return PP_EINVAL;
    hipLaunchKernelGGL(xent_diag_backward_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, n, ld, scale, upstream, dlogits);
    return pp_last_launch();
}

int pp_normalize_rows_backward(const float* x, long long row_stride, const int64_t* index, const float* dq, int rows, int C, float eps, float* dx,
                               void* stream) {
    if (!x || !dq || !dx || rows <= 0 || C <= 0) return PP_EINVAL;
    hipLaunchKernelGGL(normalize_backward_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, row_stride, (const long long*)index, dq,
                       rows, C, eps, dx);
    return pp_last_launch();
}

int pp_scatter_add_rows(const float* src, const int64_t* index, int n, int C, float* dst, void* stream) {
    if (!src || !index || !dst || n <= 0 || C <= 0) return PP_EINVAL;
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, src, (const long long*)index, n, C, dst);
    return pp_last_launch();
}

int pp_im2col_nhwc(const float* x, int B, int H, int W, int C, int ksize, int stride, int pad, float* col, void* stream) {
    if (!x || !col || B <= 0 || H <= 0 || W <= 0 || C <= 0 || ksize <= 0 || stride <= 0 || pad < 0) return PP_EINVAL;
    const int Ho = (H + 2 * pad - ksize) / stride + 1, Wo = (W + 2 * pad - ksize) / stride + 1;
    const long long total = (long long)B * Ho * Wo * ksize * ksize * C;
    hipLaunchKernelGGL(im2col_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, H, W, C, ksize, stride, pad, Ho, Wo, total, col);
    return pp_last_launch();
}

int pp_col2im_nhwc(const float* col, int B, int H, int W, int C, int ksize, int stride, int pad, float* dx, void* stream) {
    if (!col || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0 || ksize <= 0 || stride <= 0 || pad < 0) return PP_EINVAL;
    const int Ho = (H + 2 * pad - ksize) / stride + 1, Wo = (W + 2 * pad - ksize) / stride + 1;
    const long long total = (long long)B * H * W * C;
    hipLaunchKernelGGL(col2im_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, col, H, W, C, ksize, stride, pad, Ho, Wo, total, dx);
    return pp_last_launch();
}

int pp_im2col_t_nhwc(const float* x, int B, int H, int W, int C, int ksize, int stride, int pad, float* colT, void* stream) {
    if (!x || !colT || B <= 0 || H <= 0 || W <= 0 || C <= 0 || ksize <= 0 || stride <= 0 || pad < 0) return PP_EINVAL;
    const int Ho = (H + 2 * pad - ksize) / stride + 1, Wo = (W + 2 * pad - ksize) / stride + 1;
    const long long rows = (long long)B * Ho * Wo;
    if (rows <= 0 || rows >= (1LL << 31) - 32) return PP_EINVAL;
    hipLaunchKernelGGL(im2col_t_kernel, dim3((unsigned)((rows + 31) / 32), (C + 31) / 32), dim3(256), 0, (hipStream_t)stream, x, H, W, C, ksize, stride,
                       pad, Ho, Wo, (int)rows, colT);
    return pp_last_launch();
}

int pp_simvol_backward(const float* out, const float* dout, const float* src_mask, int mask_h, int mask_w, int B, float* dS, void* stream) {
    if (!out || !dout || !src_mask || !dS || B <= 0 || mask_h <= 0 || mask_w <= 0) return PP_EINVAL;
    const long long total = (long long)B * 256 * 256;
    hipLaunchKernelGGL(simvol_backward_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, out, dout, src_mask, mask_h, mask_w, total, dS);
    return pp_last_launch();
}

}  // extern "C"
