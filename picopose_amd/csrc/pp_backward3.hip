// Adjoint kernels of stage 3's training path (picopose_amd/autograd.py: offset_regressor_forward): BatchNorm in training mode,
// bilinear resize (align_corners=True), the feature warp, the fused correlation pyramid + lookup, 2x2 average pooling and the
// flow / certainty losses.  Convolutions go through pp_im2col_nhwc / pp_col2im_nhwc + pp_gemm (pp_backward.hip).  Every kernel
// restates the FORWARD kernel's own coordinate arithmetic (pp_sample.hip, pp_train.hip) so that value and gradient describe the
// same function; reductions over rows run in double in a fixed order, scatters (warp / lookup into the sampled maps) are fp32
// atomic adds.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/picopose_hip.h"
#include "pp_common.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

inline int grid_for(long long n) { return (int)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384); }

// ---------------------------------------------------------------------------------------------------------- BatchNorm (training)
// y = relu?(gamma xhat + beta), xhat = (x - mean) rstd with the batch statistics (biased variance): per channel
//   g = dy [y > 0],  dgamma = sum g xhat,  dbeta = sum g,  dx = gamma rstd (g - mean(g) - xhat mean(g xhat))
constexpr int BNB_ROWS = 256;

// (both partial kernels: a workgroup = BNB_ROWS rows; a thread owns 4 consecutive channels (16-byte loads) and one of 4 row lanes —
// rows rl, rl + 4, ... —, the row lanes are added in lane order through LDS: the sums keep a fixed order.  C % 4 == 0.)
__global__ __launch_bounds__(256) void bnb_stats_kernel(const float* __restrict__ x, int rows, int C, double* __restrict__ part) {
    __shared__ double red[4][64][8];
    const int r0 = blockIdx.x * BNB_ROWS, r1 = min(rows, r0 + BNB_ROWS);
    const int q = threadIdx.x & 63, rl = threadIdx.x >> 6;
    for (int cb = 0; cb < C; cb += 256) {
        const int c = cb + 4 * q;
        double s[4] = {0.0, 0.0, 0.0, 0.0}, qq[4] = {0.0, 0.0, 0.0, 0.0};
        if (c < C)
            for (int r = r0 + rl; r < r1; r += 4) {
                const f4 v = *(const f4*)(x + (size_t)r * C + c);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const double d = (double)v[i];
                    s[i] += d;
                    qq[i] = fma(d, d, qq[i]);
                }
            }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            red[rl][q][2 * i] = s[i];
            red[rl][q][2 * i + 1] = qq[i];
        }
        __syncthreads();
        if (rl == 0 && c < C)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                part[((size_t)blockIdx.x * C + c + (i >> 1)) * 2 + (i & 1)] = ((red[0][q][i] + red[1][q][i]) + red[2][q][i]) + red[3][q][i];
        __syncthreads();
    }
}

// fold of the per-workgroup partial sums (s, q) of one channel with 16 slab lanes (slabs sl, sl + 16, ..., eight loads in flight, then the
// lanes in order — a fixed order): a workgroup = 16 channels.  (One thread per channel walked up to 512 dependent loads: 27 us per launch.)
__device__ __forceinline__ bool bn_fold16(const double* __restrict__ part, int nblk, int C, int& c, double& s, double& q) {
    __shared__ double red[16][16][2];
    const int ql = threadIdx.x & 15, sl = threadIdx.x >> 4;
    c = blockIdx.x * 16 + ql;
    double ss = 0.0, qq = 0.0;
    if (c < C) {
        int k = sl;
        for (; k + 112 < nblk; k += 128) {
            double a[8], b[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                a[j] = part[((size_t)(k + 16 * j) * C + c) * 2];
                b[j] = part[((size_t)(k + 16 * j) * C + c) * 2 + 1];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                ss += a[j];
                qq += b[j];
            }
        }
        for (; k < nblk; k += 16) {
            ss += part[((size_t)k * C + c) * 2];
            qq += part[((size_t)k * C + c) * 2 + 1];
        }
    }
    red[sl][ql][0] = ss;
    red[sl][ql][1] = qq;
    __syncthreads();
    if (sl != 0 || c >= C) return false;
    s = red[0][ql][0];
    q = red[0][ql][1];
#pragma unroll
    for (int j = 1; j < 16; ++j) {
        s += red[j][ql][0];
        q += red[j][ql][1];
    }
    return true;
}

__global__ __launch_bounds__(256) void bnb_stats_finish_kernel(const double* __restrict__ part, int nblk, int rows, int C, float eps,
                                                               float* __restrict__ mean, float* __restrict__ rstd) {
    int c;
    double s, q;
    if (!bn_fold16(part, nblk, C, c, s, q)) return;
    const double m = s / rows;
    double var = q / rows - m * m;
    var = var > 0.0 ? var : 0.0;
    mean[c] = (float)m;
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
}

__global__ __launch_bounds__(256) void bnb_sums_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, int rows, int C, int relu, double* __restrict__ part) {
    __shared__ double red[4][64][8];
    const int r0 = blockIdx.x * BNB_ROWS, r1 = min(rows, r0 + BNB_ROWS);
    const int q = threadIdx.x & 63, rl = threadIdx.x >> 6;
    for (int cb = 0; cb < C; cb += 256) {
        const int c = cb + 4 * q;
        double s[4] = {0.0, 0.0, 0.0, 0.0}, qq[4] = {0.0, 0.0, 0.0, 0.0};
        if (c < C) {
            const f4 m = *(const f4*)(mean + c), rs = *(const f4*)(rstd + c), ga = *(const f4*)(gamma + c), be = *(const f4*)(beta + c);
            for (int r = r0 + rl; r < r1; r += 4) {
                const f4 xv = *(const f4*)(x + (size_t)r * C + c), gv = *(const f4*)(dy + (size_t)r * C + c);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float xh = (xv[i] - m[i]) * rs[i];
                    float g = gv[i];
                    if (relu && !(fmaf(xh, ga[i], be[i]) > 0.f)) g = 0.f;
                    s[i] += (double)g;
                    qq[i] = fma((double)g, (double)xh, qq[i]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            red[rl][q][2 * i] = s[i];
            red[rl][q][2 * i + 1] = qq[i];
        }
        __syncthreads();
        if (rl == 0 && c < C)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                part[((size_t)blockIdx.x * C + c + (i >> 1)) * 2 + (i & 1)] = ((red[0][q][i] + red[1][q][i]) + red[2][q][i]) + red[3][q][i];
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void bnb_sums_finish_kernel(const double* __restrict__ part, int nblk, int rows, int C,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              double* __restrict__ mg, double* __restrict__ mgx) {
    int c;
    double s, q;
    if (!bn_fold16(part, nblk, C, c, s, q)) return;
    dbeta[c] = (float)s;
    dgamma[c] = (float)q;
    mg[c] = s / rows;       // (kept in double: see bnb_dx_kernel)
    mgx[c] = q / rows;
}

__global__ __launch_bounds__(256) void bnb_dx_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const double* __restrict__ mg,
                                                     const double* __restrict__ mgx, long long n, int C, int relu, float* __restrict__ dx) {
    // dx = gamma rstd (g - mean(g) - xhat mean(g xhat)) evaluated in DOUBLE from the double channel means, rounded once — as the
    // reference's CPU kernel does (torch's accumulate type of float on the CPU is double).  With the two means rounded to fp32 first,
    // every element of a channel carries the SAME offset (2^-24 of mean(g)): it is invisible per element, but dx sums to zero over a
    // channel analytically, and everything upstream that adds dx over the pixels (bias / BatchNorm-shift gradients, the weight
    // gradients of a convolution in front of a BatchNorm) is that sum — the offset x 8 192 pixels was 2e-3 .. 5e-3 of those
    // tensors' maxima against a float64 evaluation of the reference (tests/golden/train_grads_f64.npz; round 4 had restated their bar).
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const float xh = (x[i] - mean[c]) * rstd[c];
        float g = dy[i];
        if (relu && !(fmaf(xh, gamma[c], beta[c]) > 0.f)) g = 0.f;
        const double xhd = ((double)x[i] - (double)mean[c]) * (double)rstd[c];
        dx[i] = (float)((double)gamma[c] * (double)rstd[c] * ((double)g - mg[c] - xhd * mgx[c]));
    }
}

// ---------------------------------------------------------------------------------------------------------- bilinear resize
// adjoint of resize_kernel (pp_sample.hip: src = dst (in - 1) / (out - 1), taps i0 = (int)src and i1 = min(i0 + 1, in - 1)) in
// GATHER form: an input pixel collects from the output pixels whose two taps include it — a fixed order, no atomics.
__device__ __forceinline__ void resize_taps(int o, float s, int in, int& i0, int& i1, float& l) {
    const float f = s * (float)o;
    i0 = (int)f;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l = f - (float)i0;
}

__global__ __launch_bounds__(256) void resize_backward_kernel(const float* __restrict__ dy, int H, int W, int C, int Ho, int Wo, float mul,
                                                              long long total, float* __restrict__ dx) {
    const float sy = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
    const float sx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        long long q = i / C;
        const int ix = (int)(q % W);
        q /= W;
        const int iy = (int)(q % H);
        const long long b = q / H;
        // candidate outputs: src in (i - 1, i + 1)
        int oy0 = 0, oy1 = Ho - 1, ox0 = 0, ox1 = Wo - 1;
        if (sy > 0.f) {
            oy0 = max(0, (int)floorf(((float)iy - 1.f) / sy) - 1);
            oy1 = min(Ho - 1, (int)ceilf(((float)iy + 1.f) / sy) + 1);
        }
        if (sx > 0.f) {
            ox0 = max(0, (int)floorf(((float)ix - 1.f) / sx) - 1);
            ox1 = min(Wo - 1, (int)ceilf(((float)ix + 1.f) / sx) + 1);
        }
        float acc = 0.f;
        for (int oy = oy0; oy <= oy1; ++oy) {
            int y0, y1;
            float ly;
            resize_taps(oy, sy, H, y0, y1, ly);
            const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
            if (wy == 0.f) continue;
            for (int ox = ox0; ox <= ox1; ++ox) {
                int x0, x1;
                float lx;
                resize_taps(ox, sx, W, x0, x1, lx);
                const float wx = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
                if (wx != 0.f) acc = fmaf(wy * wx, dy[((b * Ho + oy) * Wo + ox) * C + c], acc);
            }
        }
        dx[i] = acc * mul;
    }
}

// ---------------------------------------------------------------------------------------------------------- 2x2 average pooling
__global__ __launch_bounds__(256) void avgpool2_backward_kernel(const float* __restrict__ dy, int H, int W, int C, long long total,
                                                                int accumulate, float* __restrict__ dx) {
    const int Ho = H / 2, Wo = W / 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        long long q = i / C;
        const int x = (int)(q % W);
        q /= W;
        const int y = (int)(q % H);
        const long long b = q / H;
        float v = 0.f;
        if ((y >> 1) < Ho && (x >> 1) < Wo) v = 0.25f * dy[((b * Ho + (y >> 1)) * Wo + (x >> 1)) * C + c];
        dx[i] = accumulate ? dx[i] + v : v;
    }
}

// coordinate round trip of the forward kernels (pp_sample.hip: roundtrip)
__device__ __forceinline__ float roundtrip(float x, int size) {
    const float n = x * 2.f / (float)(size - 1 > 1 ? size - 1 : 1) - 1.f;
    return ((n + 1.f) / 2.f) * (float)(size - 1);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Scatter accumulation of the two sampling adjoints.  FIXED = false: fp32 atomicAdd — fast, but the sum depends on the order the
// workgroups arrive in, so two runs differ in the last bits.  FIXED = true (the deterministic option, autograd.DETERMINISTIC): the
// contribution is rounded to a 64-bit fixed-point number (2^-40 units: 9e-13 absolute resolution, |sum| < 8e6) and added with an
// INTEGER atomic — integer addition is associative and commutative exactly, so the result is the same bits in any order; a second
// pass turns the accumulators into floats (pp_fixed_to_float).
constexpr float FX_SCALE = 1099511627776.f;          // 2^40
template <bool FIXED>
__device__ __forceinline__ void scatter_add(void* base, size_t idx, float v) {
    if (FIXED) atomicAdd((unsigned long long*)base + idx, (unsigned long long)__float2ll_rn(v * FX_SCALE));
    else atomicAdd((float*)base + idx, v);
}

__global__ __launch_bounds__(256) void fixed_to_float_kernel(const long long* __restrict__ acc, long long n, float* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        out[i] = (float)((double)acc[i] * (1.0 / 1099511627776.0));
}

// ---------------------------------------------------------------------------------------------------------- feature warp
// adjoint of warp_kernel: out[p] = bilinear(feat, p + flow[p]), zeros padding.  One wave per pixel, lanes over channels.
template <bool FIXED>
__global__ __launch_bounds__(256) void warp_backward_kernel(const float* __restrict__ feat, const float* __restrict__ flow,
                                                            const float* __restrict__ dy, int H, int W, int C, int ld_flow,
                                                            void* __restrict__ dfeat, float* __restrict__ dflow) {
    const int b = blockIdx.y, p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (p >= H * W) return;
    const int y = p / W, x = p - y * W;
    const float* fl = flow + ((size_t)b * H * W + p) * ld_flow;
    const float rx = roundtrip((float)x + fl[0], W), ry = roundtrip((float)y + fl[1], H);
    const float ix = fminf(fmaxf(rx, -2.f), (float)W + 1.f), iy = fminf(fmaxf(ry, -2.f), (float)H + 1.f);
    const float x0f = floorf(ix), y0f = floorf(iy);
    const int x0 = (int)x0f, y0 = (int)y0f;
    const float wx1 = ix - x0f, wx0 = (x0f + 1.f) - ix, wy1 = iy - y0f, wy0 = (y0f + 1.f) - iy;
    const bool vx0 = x0 >= 0 && x0 < W, vx1 = x0 + 1 >= 0 && x0 + 1 < W, vy0 = y0 >= 0 && y0 < H, vy1 = y0 + 1 >= 0 && y0 + 1 < H;
    const int xa = min(max(x0, 0), W - 1), xb = min(max(x0 + 1, 0), W - 1), ya = min(max(y0, 0), H - 1), yb = min(max(y0 + 1, 0), H - 1);
    const size_t base = (size_t)b * H * W * C;
    const float* g = dy + ((size_t)b * H * W + p) * C;
    float gx = 0.f, gy = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float d = g[c];
        const float t00 = vy0 && vx0 ? feat[base + ((size_t)ya * W + xa) * C + c] : 0.f;
        const float t01 = vy0 && vx1 ? feat[base + ((size_t)ya * W + xb) * C + c] : 0.f;
        const float t10 = vy1 && vx0 ? feat[base + ((size_t)yb * W + xa) * C + c] : 0.f;
        const float t11 = vy1 && vx1 ? feat[base + ((size_t)yb * W + xb) * C + c] : 0.f;
        gx = fmaf(d, (t01 - t00) * wy0 + (t11 - t10) * wy1, gx);
        gy = fmaf(d, (t10 - t00) * wx0 + (t11 - t01) * wx1, gy);
        if (vy0 && vx0) scatter_add<FIXED>(dfeat, base + ((size_t)ya * W + xa) * C + c, d * (wx0 * wy0));
        if (vy0 && vx1) scatter_add<FIXED>(dfeat, base + ((size_t)ya * W + xb) * C + c, d * (wx1 * wy0));
        if (vy1 && vx0) scatter_add<FIXED>(dfeat, base + ((size_t)yb * W + xa) * C + c, d * (wx0 * wy1));
        if (vy1 && vx1) scatter_add<FIXED>(dfeat, base + ((size_t)yb * W + xb) * C + c, d * (wx1 * wy1));
    }
    gx = wave_sum(gx);
    gy = wave_sum(gy);
    if (lane == 0) {   // (a clamped coordinate is outside every tap's range: all taps invalid, the sums above are zero)
        dflow[((size_t)b * H * W + p) * 2] = rx == ix ? gx : 0.f;
        dflow[((size_t)b * H * W + p) * 2 + 1] = ry == iy ? gy : 0.f;
    }
}

// ---------------------------------------------------------------------------------------------------------- correlation lookup
// adjoint of corr_lookup_kernel (pp_sample.hip): corr_l[p, q] = <f1[p], pool_l(f2)[q]> / sqrt(C), output channel
// l win^2 + a win + b = bilinear sample of corr_l[p, .] at ((p + flow[p]) / 2^l) + (a - r, b - r), zeros padding.
// One wave per pixel: the (2r+2)^2 table positions per level are visited one after the other, lanes over channels.
constexpr int CLB_MAXL = 3, CLB_TW = 8;   // levels, table width for r <= 3

template <bool FIXED, bool SCATTER = true>
__global__ __launch_bounds__(256) void corr_lookup_backward_kernel(const float* __restrict__ f1, const float* __restrict__ f2l0,
                                                                   const float* __restrict__ f2l1, const float* __restrict__ f2l2,
                                                                   const float* __restrict__ flow, const float* __restrict__ dout, int H,
                                                                   int W, int C, int L, int r, int ld_flow, int ld_dout, float inv_sqrt_c,
                                                                   float* __restrict__ df1, void* __restrict__ df2l0,
                                                                   void* __restrict__ df2l1, void* __restrict__ df2l2,
                                                                   float* __restrict__ dflow) {
    __shared__ float dtab_s[4][CLB_MAXL * CLB_TW * CLB_TW];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.y, p = blockIdx.x * 4 + wv;
    if (p >= H * W) return;
    float* dtab = dtab_s[wv];
    const int y = p / W, x = p - y * W;
    const int tw = 2 * r + 2, win = 2 * r + 1;
    const float* fl = flow + ((size_t)b * H * W + p) * ld_flow;
    const float gx0 = (float)x + fl[0], gy0 = (float)y + fl[1];
    const float* a = f1 + ((size_t)b * H * W + p) * C;
    const float* g = dout + ((size_t)b * H * W + p) * ld_dout;
    float dfx = 0.f, dfy = 0.f;           // gradient of the flow (lane-partial)
    // accumulators of df1[p]: up to 4 channels per lane (C <= 256)
    float acc1[4] = {0.f, 0.f, 0.f, 0.f};
    for (int l = 0; l < L; ++l) {
        const int Hl = H >> l, Wl = W >> l;
        const float sc = (float)(1 << l);
        const float rx = roundtrip(gx0 / sc, Wl), ry = roundtrip(gy0 / sc, Hl);
        const float cx = fminf(fmaxf(rx, -(float)(r + 2)), (float)(Wl + r + 1)), cy = fminf(fmaxf(ry, -(float)(r + 2)), (float)(Hl + r + 1));
        const int bx = (int)floorf(cx) - r, by = (int)floorf(cy) - r;
        const float wx1 = cx - floorf(cx), wy1 = cy - floorf(cy), wx0 = 1.f - wx1, wy0 = 1.f - wy1;
        const float* f2 = l == 0 ? f2l0 : (l == 1 ? f2l1 : f2l2);
        void* df2 = l == 0 ? df2l0 : (l == 1 ? df2l1 : df2l2);
        const size_t lb = (size_t)b * Hl * Wl * C;
        // dtab[dy][dx] = sum over the outputs (a, b) whose four corners include (dy, dx)
        for (int i = lane; i < tw * tw; i += 64) {
            const int dy_ = i / tw, dx_ = i - dy_ * tw;
            float s = 0.f;
#pragma unroll
            for (int cy_ = 0; cy_ < 2; ++cy_)
#pragma unroll
                for (int cx_ = 0; cx_ < 2; ++cx_) {
                    const int bi = dy_ - cy_, ai = dx_ - cx_;      // output (ai, bi) has this position as corner (cy_, cx_)
                    if (ai >= 0 && ai < win && bi >= 0 && bi < win)
                        s = fmaf(g[l * win * win + ai * win + bi], (cx_ ? wx1 : wx0) * (cy_ ? wy1 : wy0), s);
                }
            dtab[l * CLB_TW * CLB_TW + i] = s;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        // the table VALUES are needed for the coordinate gradient: value(dy, dx) = <f1[p], f2_l[q]> / sqrt(C) or 0 outside
        float gcx = 0.f, gcy = 0.f;
        for (int i = 0; i < tw * tw; ++i) {
            const int dy_ = i / tw, dx_ = i - dy_ * tw;
            const int qx = bx + dx_, qy = by + dy_;
            if (!(qx >= 0 && qx < Wl && qy >= 0 && qy < Hl)) continue;   // (wave-uniform)
            const float* qv = f2 + lb + ((size_t)qy * Wl + qx) * C;
            const size_t dq = lb + ((size_t)qy * Wl + qx) * C;
            const float dt = dtab[l * CLB_TW * CLB_TW + i] * inv_sqrt_c;
            float dot = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = lane + 64 * k;
                if (c < C) {
                    const float u = a[c], v = qv[c];
                    dot = fmaf(u, v, dot);
                    acc1[k] = fmaf(dt, v, acc1[k]);
                    if (SCATTER && dt != 0.f) scatter_add<FIXED>(df2, dq + c, dt * u);
                }
            }
            const float val = wave_sum(dot) * inv_sqrt_c;
            // d out(ai, bi) / d wx1 for the outputs that use this position: corner (cy_, cx_) carries sign (cx_ ? +1 : -1) wy
            float sx = 0.f, sy = 0.f;
#pragma unroll
            for (int cy_ = 0; cy_ < 2; ++cy_)
#pragma unroll
                for (int cx_ = 0; cx_ < 2; ++cx_) {
                    const int bi = dy_ - cy_, ai = dx_ - cx_;
                    if (ai >= 0 && ai < win && bi >= 0 && bi < win) {
                        const float go = g[l * win * win + ai * win + bi];
                        sx = fmaf(go, (cx_ ? 1.f : -1.f) * (cy_ ? wy1 : wy0), sx);
                        sy = fmaf(go, (cy_ ? 1.f : -1.f) * (cx_ ? wx1 : wx0), sy);
                    }
                }
            gcx = fmaf(val, sx, gcx);
            gcy = fmaf(val, sy, gcy);
        }
        if (rx == cx) dfx += gcx / sc;   // (every lane holds the same sums)
        if (ry == cy) dfy += gcy / sc;
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = lane + 64 * k;
        if (c < C) df1[((size_t)b * H * W + p) * C + c] = acc1[k];
    }
    if (lane == 0) {
        dflow[((size_t)b * H * W + p) * 2] = dfx;
        dflow[((size_t)b * H * W + p) * 2 + 1] = dfy;
    }
}

// The df2 half of the adjoint above with 6 x fewer global atomics (the per-pixel kernel issues one per (pixel, table position, channel):
// 1.2 G per level at 64 x 64 x 32 images x 256 channels — the L2 takes about one atomic per channel and clock: 2.5 ms per level whatever
// the flow).  A workgroup (16 waves) owns a 4 x 4 pixel patch and 128 channels.  The 16 windows of a level overlap (smooth flow: a
// (tw + 3)^2 union against 16 tw^2 visits): each pixel's wave writes its table dt[p, .] into column p of a small LDS matrix D (cells of
// a (tw + 4)^2 frame anchored at the patch's second pixel x 16 pixels; plain stores — a pixel's positions are distinct cells), then
// T = D F1 (cells x 128 channels, 16 terms each) is evaluated in registers — thread = (channel, every 8th cell) — and every non-zero
// entry leaves as ONE global atomic.  A position outside the frame (a flow that tears the patch apart) falls back to the direct
// atomics.  (LDS float atomics into a cells x channels table were tried first: slower than the kernel they replace.)  fp32,
// order-dependent in the last bits like that kernel; FIXED: T's entries are evaluated in a fixed order and added as 64-bit fixed point
// with integer atomics — the same bits on every run (the deterministic option).
constexpr int CS_CH = 128, CS_CELLS = (CLB_TW + 4) * (CLB_TW + 4);

template <bool FIXED>
__global__ __launch_bounds__(1024) void corr_lookup_scatter_kernel(const float* __restrict__ f1, const float* __restrict__ flow,
                                                                   const float* __restrict__ dout, int H, int W, int C, int L, int r, int ld_flow,
                                                                   int ld_dout, float inv_sqrt_c, void* __restrict__ df2l0,
                                                                   void* __restrict__ df2l1, void* __restrict__ df2l2) {
    __shared__ __attribute__((aligned(16))) float D[CS_CELLS][16];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.y, ch0 = blockIdx.z * CS_CH;
    const int pw = W >> 2, px0 = (blockIdx.x % pw) * 4, py0 = (blockIdx.x / pw) * 4;
    const int tw = 2 * r + 2, win = 2 * r + 1, twd = tw + 4, ncell = twd * twd;
    const int mc = threadIdx.x & (CS_CH - 1), mg = threadIdx.x >> 7;     // matrix phase: channel, cell group (8)
    float fk[16];                                                        // f1 of the patch's 16 pixels at this thread's channel
#pragma unroll
    for (int k = 0; k < 16; ++k) fk[k] = f1[((size_t)b * H * W + (size_t)(py0 + (k >> 2)) * W + px0 + (k & 3)) * C + ch0 + mc];
    const int x = px0 + (wv & 3), y = py0 + (wv >> 2), p = y * W + x;    // pixel phase: this wave's pixel
    const float* fl = flow + ((size_t)b * H * W + p) * ld_flow;
    const float* g = dout + ((size_t)b * H * W + p) * ld_dout;
    const float gx0 = (float)x + fl[0], gy0 = (float)y + fl[1];
    const float* fa = flow + ((size_t)b * H * W + (size_t)(py0 + 1) * W + px0 + 1) * ld_flow;
    const float agx = (float)(px0 + 1) + fa[0], agy = (float)(py0 + 1) + fa[1];
    for (int l = 0; l < L; ++l) {
        const int Hl = H >> l, Wl = W >> l;
        const float sc = (float)(1 << l);
        void* df2 = l == 0 ? df2l0 : (l == 1 ? df2l1 : df2l2);
        const size_t lb = (size_t)b * Hl * Wl * C;
        for (int i = threadIdx.x; i < ncell * 16; i += 1024) (&D[0][0])[i] = 0.f;
        int ax, ay;   // anchor: the window base of pixel (px0 + 1, py0 + 1), two positions up and left
        {
            const float rx = roundtrip(agx / sc, Wl), ry = roundtrip(agy / sc, Hl);
            const float cx = fminf(fmaxf(rx, -(float)(r + 2)), (float)(Wl + r + 1)), cy = fminf(fmaxf(ry, -(float)(r + 2)), (float)(Hl + r + 1));
            ax = (int)floorf(cx) - r - 2;
            ay = (int)floorf(cy) - r - 2;
        }
        __syncthreads();
        {
            const float rx = roundtrip(gx0 / sc, Wl), ry = roundtrip(gy0 / sc, Hl);
            const float cx = fminf(fmaxf(rx, -(float)(r + 2)), (float)(Wl + r + 1)), cy = fminf(fmaxf(ry, -(float)(r + 2)), (float)(Hl + r + 1));
            const int bx = (int)floorf(cx) - r, by = (int)floorf(cy) - r;
            const float wx1 = cx - floorf(cx), wy1 = cy - floorf(cy), wx0 = 1.f - wx1, wy0 = 1.f - wy1;
            float dt = 0.f;
            int qx = 0, qy = 0;
            bool far = false;
            if (lane < tw * tw) {      // lane = table position; dt as in corr_lookup_backward_kernel
                const int dy_ = lane / tw, dx_ = lane - dy_ * tw;
                float s = 0.f;
#pragma unroll
                for (int cy_ = 0; cy_ < 2; ++cy_)
#pragma unroll
                    for (int cx_ = 0; cx_ < 2; ++cx_) {
                        const int bi = dy_ - cy_, ai = dx_ - cx_;
                        if (ai >= 0 && ai < win && bi >= 0 && bi < win)
                            s = fmaf(g[l * win * win + ai * win + bi], (cx_ ? wx1 : wx0) * (cy_ ? wy1 : wy0), s);
                    }
                qx = bx + dx_;
                qy = by + dy_;
                dt = (qx >= 0 && qx < Wl && qy >= 0 && qy < Hl) ? s * inv_sqrt_c : 0.f;
                const int lx = qx - ax, ly = qy - ay;
                if (dt != 0.f) {
                    if (lx >= 0 && lx < twd && ly >= 0 && ly < twd) D[ly * twd + lx][wv] = dt;
                    else far = true;
                }
            }
            unsigned long long m = __ballot(far);
            while (m) {                // positions outside the frame: direct atomics, all lanes over the channels
                const int src = __ffsll((long long)m) - 1;
                m &= m - 1;
                const float dts = __shfl(dt, src);
                const int sx = __shfl(qx, src), sy = __shfl(qy, src);
                const float* a = f1 + ((size_t)b * H * W + p) * C + ch0;
                const size_t t = lb + ((size_t)sy * Wl + sx) * C + ch0;
                scatter_add<FIXED>(df2, t + lane, dts * a[lane]);
                scatter_add<FIXED>(df2, t + lane + 64, dts * a[lane + 64]);
            }
        }
        __syncthreads();
        for (int cell = mg; cell < ncell; cell += 8) {
            const f4 d0 = *(const f4*)&D[cell][0], d1 = *(const f4*)&D[cell][4], d2 = *(const f4*)&D[cell][8], d3 = *(const f4*)&D[cell][12];
            float acc = d0.x * fk[0];
            acc = fmaf(d0.y, fk[1], acc);
            acc = fmaf(d0.z, fk[2], acc);
            acc = fmaf(d0.w, fk[3], acc);
            acc = fmaf(d1.x, fk[4], acc);
            acc = fmaf(d1.y, fk[5], acc);
            acc = fmaf(d1.z, fk[6], acc);
            acc = fmaf(d1.w, fk[7], acc);
            acc = fmaf(d2.x, fk[8], acc);
            acc = fmaf(d2.y, fk[9], acc);
            acc = fmaf(d2.z, fk[10], acc);
            acc = fmaf(d2.w, fk[11], acc);
            acc = fmaf(d3.x, fk[12], acc);
            acc = fmaf(d3.y, fk[13], acc);
            acc = fmaf(d3.z, fk[14], acc);
            acc = fmaf(d3.w, fk[15], acc);
            if (acc != 0.f) {          // (with finite data a non-zero cell is inside the map: only such positions are stored)
                const int qx = ax + cell % twd, qy = ay + cell / twd;
                // ... but 0 * inf = NaN != 0 for a cell that was never stored (a diverged run): the range is checked, not inferred
                if (qx >= 0 && qx < Wl && qy >= 0 && qy < Hl) scatter_add<FIXED>(df2, lb + ((size_t)qy * Wl + qx) * C + ch0 + mc, acc);
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------- flow / certainty losses
// adjoint of flow_loss_kernel (pp_train.hip; utils/loss_utils.py:119-125, RAFTLoss :24-39): gf = upstream * weight / (count + eps),
// gc = upstream * weight / (B H W) are formed by the caller.
constexpr int KP_GRID = 64, KP_N = KP_GRID * KP_GRID;

__global__ __launch_bounds__(256) void flow_loss_backward_kernel(const float* __restrict__ flow, const float* __restrict__ cert,
                                                                 const float* __restrict__ tar_pts, int B, int H, int W, float max_flow,
                                                                 const float* __restrict__ gf, const float* __restrict__ gc,
                                                                 float* __restrict__ dflow, float* __restrict__ dcert) {
    const long long total = (long long)B * H * W;
    const float kf = gf[0], kc = gc[0];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % W), y = (int)((i / W) % H), b = (int)(i / ((long long)W * H));
        const int iy = min((int)floorf((float)y * ((float)KP_GRID / H)), KP_GRID - 1);
        const int ix = min((int)floorf((float)x * ((float)KP_GRID / W)), KP_GRID - 1);
        const float* p = tar_pts + ((size_t)b * KP_N + (size_t)ix * KP_GRID + iy) * 2;
        const bool valid = p[0] != -1.f && p[1] != -1.f;
        const float k = (float)H / KP_GRID;
        const float g0 = (valid ? k * p[0] : 0.f) - (float)x, g1 = (valid ? k * p[1] : 0.f) - (float)y;
        const float z = cert[i], t = valid ? 1.f : 0.f;
        dcert[i] = kc * (1.f / (1.f + expf(-z)) - t);
        float d0 = 0.f, d1 = 0.f;
        if (valid && sqrtf(g0 * g0 + g1 * g1) < max_flow) {
            const float e0 = flow[2 * i] - g0, e1 = flow[2 * i + 1] - g1;
            d0 = e0 > 0.f ? kf : (e0 < 0.f ? -kf : 0.f);
            d1 = e1 > 0.f ? kf : (e1 < 0.f ? -kf : 0.f);
        }
        dflow[2 * i] = d0;
        dflow[2 * i + 1] = d1;
    }
}

}  // namespace

extern "C" {

size_t pp_batchnorm_train_backward_workspace_bytes(long long rows, int C) {
    const long long nblk = (rows + BNB_ROWS - 1) / BNB_ROWS;
    return (size_t)nblk * C * 2 * sizeof(double) + (size_t)2 * C * sizeof(double) + (size_t)2 * C * sizeof(float);
}

int pp_batchnorm_train_backward(const float* x, const float* gamma, const float* beta, const float* dy, long long rows, int C, float eps,
                                int relu, float* dx, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !gamma || !beta || !dy || !dx || !dgamma || !dbeta || !workspace || rows <= 0 || C <= 0 || C % 4 != 0) return PP_EINVAL;
    if ((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)gamma | (uintptr_t)beta) & 15) != 0) return PP_EINVAL;
    if (workspace_bytes < pp_batchnorm_train_backward_workspace_bytes(rows, C)) return PP_EWORKSPACE;
    const int nblk = (int)((rows + BNB_ROWS - 1) / BNB_ROWS);
    double* part = (double*)workspace;
    double *mg = part + (size_t)nblk * C * 2, *mgx = mg + C;
    float* mean = (float*)(mgx + C);
    float* rstd = mean + C;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bnb_stats_kernel, dim3(nblk), dim3(256), 0, st, x, (int)rows, C, part);
    hipLaunchKernelGGL(bnb_stats_finish_kernel, dim3((C + 15) / 16), dim3(256), 0, st, (const double*)part, nblk, (int)rows, C, eps, mean, rstd);
    hipLaunchKernelGGL(bnb_sums_kernel, dim3(nblk), dim3(256), 0, st, x, dy, gamma, beta, (const float*)mean, (const float*)rstd, (int)rows, C, relu, part);
    hipLaunchKernelGGL(bnb_sums_finish_kernel, dim3((C + 15) / 16), dim3(256), 0, st, (const double*)part, nblk, (int)rows, C, dgamma, dbeta, mg, mgx);
    const long long n = rows * C;
    hipLaunchKernelGGL(bnb_dx_kernel, dim3(grid_for(n)), dim3(256), 0, st, x, dy, gamma, beta, (const float*)mean, (const float*)rstd,
                       (const double*)mg, (const double*)mgx, n, C, relu, dx);
    return pp_last_launch();
}

int pp_resize_bilinear_backward_nhwc(const float* dy, int B, int H, int W, int C, int Ho, int Wo, float mul, float* dx, void* stream) {
    if (!dy || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0 || Ho <= 0 || Wo <= 0) return PP_EINVAL;
    const long long total = (long long)B * H * W * C;
    hipLaunchKernelGGL(resize_backward_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dy, H, W, C, Ho, Wo, mul, total, dx);
    return pp_last_launch();
}

int pp_avgpool2_backward_nhwc(const float* dy, int B, int H, int W, int C, int accumulate, float* dx, void* stream) {
    if (!dy || !dx || B <= 0 || H < 2 || W < 2 || C <= 0) return PP_EINVAL;
    const long long total = (long long)B * H * W * C;
    hipLaunchKernelGGL(avgpool2_backward_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dy, H, W, C, total, accumulate, dx);
    return pp_last_launch();
}

int pp_fixed_to_float(const long long* acc, long long n, float* out, void* stream) {
    if (!acc || !out || n <= 0) return PP_EINVAL;
    hipLaunchKernelGGL(fixed_to_float_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, acc, n, out);
    return pp_last_launch();
}

static int warp_backward(const float* feat, const float* flow, const float* dy, int B, int H, int W, int C, int ld_flow, void* dfeat,
                         float* dflow, bool fixed, void* stream) {
    if (!feat || !flow || !dy || !dfeat || !dflow || B <= 0 || H <= 0 || W <= 0 || C <= 0 || ld_flow < 2) return PP_EINVAL;
    if (fixed)
        hipLaunchKernelGGL(warp_backward_kernel<true>, dim3((H * W + 3) / 4, B), dim3(256), 0, (hipStream_t)stream, feat, flow, dy, H, W, C, ld_flow,
                           dfeat, dflow);
    else
        hipLaunchKernelGGL(warp_backward_kernel<false>, dim3((H * W + 3) / 4, B), dim3(256), 0, (hipStream_t)stream, feat, flow, dy, H, W, C, ld_flow,
                           dfeat, dflow);
    return pp_last_launch();
}

int pp_warp_backward_nhwc(const float* feat, const float* flow, const float* dy, int B, int H, int W, int C, int ld_flow, float* dfeat,
                          float* dflow, void* stream) {
    return warp_backward(feat, flow, dy, B, H, W, C, ld_flow, dfeat, dflow, false, stream);
}

int pp_warp_backward_nhwc_fixed(const float* feat, const float* flow, const float* dy, int B, int H, int W, int C, int ld_flow,
                                long long* dfeat_acc, float* dflow, void* stream) {
    return warp_backward(feat, flow, dy, B, H, W, C, ld_flow, dfeat_acc, dflow, true, stream);
}

static int corr_lookup_backward(const float* f1, const float* const* f2_levels, const float* flow, const float* dout, int B, int H, int W,
                                int C, int levels, int radius, int ld_flow, int ld_dout, float* df1, void* const* df2_levels, float* dflow,
                                bool fixed, void* stream) {
    if (!f1 || !f2_levels || !flow || !dout || !df1 || !df2_levels || !dflow || B <= 0 || H <= 0 || W <= 0) return PP_EINVAL;
    if (levels < 1 || levels > CLB_MAXL || radius < 0 || 2 * radius + 2 > CLB_TW || C <= 0 || C > 256 || ld_flow < 2) return PP_EINVAL;
    if (ld_dout < levels * (2 * radius + 1) * (2 * radius + 1)) return PP_EINVAL;
    for (int l = 0; l < levels; ++l)
        if (!f2_levels[l] || !df2_levels[l]) return PP_EINVAL;
    const dim3 pgrid((H * W + 3) / 4, B);
    const float isc = 1.0f / sqrtf((float)C);
    const float *f20 = f2_levels[0], *f21 = levels > 1 ? f2_levels[1] : nullptr, *f22 = levels > 2 ? f2_levels[2] : nullptr;
    void *d0 = df2_levels[0], *d1 = levels > 1 ? df2_levels[1] : nullptr, *d2 = levels > 2 ? df2_levels[2] : nullptr;
    hipStream_t st = (hipStream_t)stream;
    if (H % 4 == 0 && W % 4 == 0 && C % CS_CH == 0 && !getenv("PP_CORR_SCATTER_PER_PIXEL")) {
        // values / df1 / dflow per pixel without the scatter, df2 by patches (corr_lookup_scatter_kernel)
        const dim3 sgrid((H / 4) * (W / 4), B, C / CS_CH);
        if (fixed) {
            hipLaunchKernelGGL((corr_lookup_backward_kernel<true, false>), pgrid, dim3(256), 0, st, f1, f20, f21, f22, flow, dout, H, W, C, levels, radius,
                               ld_flow, ld_dout, isc, df1, d0, d1, d2, dflow);
            hipLaunchKernelGGL(corr_lookup_scatter_kernel<true>, sgrid, dim3(1024), 0, st, f1, flow, dout, H, W, C, levels, radius, ld_flow, ld_dout, isc, d0,
                               d1, d2);
        } else {
            hipLaunchKernelGGL((corr_lookup_backward_kernel<false, false>), pgrid, dim3(256), 0, st, f1, f20, f21, f22, flow, dout, H, W, C, levels, radius,
                               ld_flow, ld_dout, isc, df1, d0, d1, d2, dflow);
            hipLaunchKernelGGL(corr_lookup_scatter_kernel<false>, sgrid, dim3(1024), 0, st, f1, flow, dout, H, W, C, levels, radius, ld_flow, ld_dout, isc, d0,
                               d1, d2);
        }
    } else if (fixed) {
        hipLaunchKernelGGL(corr_lookup_backward_kernel<true>, pgrid, dim3(256), 0, st, f1, f20, f21, f22, flow, dout, H, W, C, levels, radius, ld_flow, ld_dout,
                           isc, df1, d0, d1, d2, dflow);
    } else {
        hipLaunchKernelGGL(corr_lookup_backward_kernel<false>, pgrid, dim3(256), 0, st, f1, f20, f21, f22, flow, dout, H, W, C, levels, radius, ld_flow, ld_dout,
                           isc, df1, d0, d1, d2, dflow);
    }
    return pp_last_launch();
}

int pp_corr_lookup_backward_nhwc(const float* f1, const float* const* f2_levels, const float* flow, const float* dout, int B, int H, int W,
                                 int C, int levels, int radius, int ld_flow, int ld_dout, float* df1, float* const* df2_levels, float* dflow,
                                 void* stream) {
    return corr_lookup_backward(f1, f2_levels, flow, dout, B, H, W, C, levels, radius, ld_flow, ld_dout, df1, (void* const*)df2_levels, dflow, false,
                                stream);
}

int pp_corr_lookup_backward_nhwc_fixed(const float* f1, const float* const* f2_levels, const float* flow, const float* dout, int B, int H, int W,
                                       int C, int levels, int radius, int ld_flow, int ld_dout, float* df1, long long* const* df2_acc_levels,
                                       float* dflow, void* stream) {
    return corr_lookup_backward(f1, f2_levels, flow, dout, B, H, W, C, levels, radius, ld_flow, ld_dout, df1, (void* const*)df2_acc_levels, dflow, true,
                                stream);
}

int pp_flow_loss_backward(const float* flow, const float* certainty, const float* tar_pts, int B, int H, int W, float max_flow,
                          const float* g_flow, const float* g_cert, float* dflow, float* dcertainty, void* stream) {
    if (!flow || !certainty || !tar_pts || !g_flow || !g_cert || !dflow || !dcertainty || B <= 0 || H <= 0 || W <= 0) return PP_EINVAL;
    const long long total = (long long)B * H * W;
    hipLaunchKernelGGL(flow_loss_backward_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, flow, certainty, tar_pts, B, H, W,
                       max_flow, g_flow, g_cert, dflow, dcertainty);
    return pp_last_launch();
}

}  // extern "C"
