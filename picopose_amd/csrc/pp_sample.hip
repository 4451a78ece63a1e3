// Sampling kernels of stage 3 (NHWC): bilinear resize (align_corners=True), the feature warp
// (grid_sample, zeros padding) and the on-demand local correlation lookup that replaces the
// reference's materialised correlation pyramid.

#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>
#include "../../include/picopose_hip.h"
#include "pp_common.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

// F.interpolate(mode="bilinear", align_corners=True) — dpt.py:150-152, flow_decoder.py:88-92.
// ATen: src = dst * (in-1)/(out-1); i0 = floor, i1 = min(i0+1, in-1), lambda = src - i0.
__global__ __launch_bounds__(256) void resize_kernel(const float* __restrict__ in, int H, int W, int C,
                                                     int Ho, int Wo, float mul, float* __restrict__ out,
                                                     _Float16* __restrict__ out_hl, int terms) {
    const int b = blockIdx.z, oy = blockIdx.y, tid = threadIdx.x;
    const float sy = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
    const float sx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    const float fy = sy * (float)oy;
    const int y0 = (int)fy, y1 = y0 + (y0 < H - 1 ? 1 : 0);
    const float ly = fy - (float)y0, hy = 1.f - ly;
    const float* ib = in + (size_t)b * H * W * C;
    if (out_hl) {  // the result as the f16x3 operand of the following 1x1 convolution (and, `out` given, as the fp32 map too): 8 channels per thread
        _Float16* oh = out_hl + ((size_t)b * Ho + oy) * Wo * terms * C;   // terms = 2: hl format, 1: h format
        float* of = out ? out + ((size_t)b * Ho + oy) * Wo * C : nullptr;
        const int C8 = C >> 3;
        for (int i = blockIdx.x * 256 + tid; i < Wo * C8; i += gridDim.x * 256) {
            const int ox = i / C8, c = (i - ox * C8) * 8;
            const float fx = sx * (float)ox;
            const int x0 = (int)fx, x1 = x0 + (x0 < W - 1 ? 1 : 0);
            const float lx = fx - (float)x0, hx = 1.f - lx;
            typedef _Float16 h8 __attribute__((ext_vector_type(8)));
            h8 hh, ll;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int cc = c + 4 * q;
                const f4 a = *(const f4*)(ib + ((size_t)y0 * W + x0) * C + cc), bq = *(const f4*)(ib + ((size_t)y0 * W + x1) * C + cc);
                const f4 cq = *(const f4*)(ib + ((size_t)y1 * W + x0) * C + cc), dq = *(const f4*)(ib + ((size_t)y1 * W + x1) * C + cc);
                f4 vq;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float v = (hy * (hx * a[k] + lx * bq[k]) + ly * (hx * cq[k] + lx * dq[k])) * mul;
                    _Float16 h, l;
                    pp_split_f16_chk(v, h, l);
                    hh[4 * q + k] = h;
                    ll[4 * q + k] = l;
                    vq[k] = v;
                }
                if (of) *(f4*)(of + (size_t)ox * C + cc) = vq;
            }
            *(h8*)(oh + ((size_t)ox * C + c) * terms) = hh;
            if (terms == 2) *(h8*)(oh + (size_t)ox * 2 * C + 2 * c + 8) = ll;
        }
        return;
    }
    float* ob = out + ((size_t)b * Ho + oy) * Wo * C;
    if ((C & 3) == 0 && (((uintptr_t)in | (uintptr_t)out) & 15) == 0) {  // 4 channels per thread: same arithmetic per element
        const int C4 = C >> 2;
        for (int i = blockIdx.x * 256 + tid; i < Wo * C4; i += gridDim.x * 256) {
            const int ox = i / C4, c = (i - ox * C4) * 4;
            const float fx = sx * (float)ox;
            const int x0 = (int)fx, x1 = x0 + (x0 < W - 1 ? 1 : 0);
            const float lx = fx - (float)x0, hx = 1.f - lx;
            const f4 a = *(const f4*)(ib + ((size_t)y0 * W + x0) * C + c), bq = *(const f4*)(ib + ((size_t)y0 * W + x1) * C + c);
            const f4 cq = *(const f4*)(ib + ((size_t)y1 * W + x0) * C + c), dq = *(const f4*)(ib + ((size_t)y1 * W + x1) * C + c);
            f4 v;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = (hy * (hx * a[k] + lx * bq[k]) + ly * (hx * cq[k] + lx * dq[k])) * mul;
            *(f4*)(ob + (size_t)ox * C + c) = v;
        }
        return;
    }
    for (int i = blockIdx.x * 256 + tid; i < Wo * C; i += gridDim.x * 256) {
        const int ox = i / C, c = i - ox * C;
        const float fx = sx * (float)ox;
        const int x0 = (int)fx, x1 = x0 + (x0 < W - 1 ? 1 : 0);
        const float lx = fx - (float)x0, hx = 1.f - lx;
        const float v = hy * (hx * ib[((size_t)y0 * W + x0) * C + c] + lx * ib[((size_t)y0 * W + x1) * C + c]) +
                        ly * (hx * ib[((size_t)y1 * W + x0) * C + c] + lx * ib[((size_t)y1 * W + x1) * C + c]);
        ob[i] = v * mul;
    }
}

// coordinate round trip of the reference: bilinear_sample scales pixel coords to [-1,1]
// (corr_lookup.py:61-63) and grid_sample(align_corners=True) maps them back
__device__ __forceinline__ float roundtrip(float x, int size) {
    const float n = x * 2.f / (float)(size - 1 > 1 ? size - 1 : 1) - 1.f;
    return ((n + 1.f) / 2.f) * (float)(size - 1);
}

// FlowDecoder.feature_sample — flow_decoder.py:49-56: out[p] = bilinear(feat, p + flow[p]), zeros padding.
// One wave per pixel, lanes over channels (float4).
// HL: the result leaves only as columns of an hl operand (rows of ld_out ELEMENTS = 2 ld_out halfs): a lane's 4 channels
// are half of a group of 8 — 4 hi terms and 4 lo terms, 8 bytes each.
template <bool HL>
__global__ __launch_bounds__(256) void warp_kernel(const float* __restrict__ feat, int feat_batch,
                                                   const float* __restrict__ flow, int H, int W, int C, int ld_flow,
                                                   float* __restrict__ out, int ld_out, int terms) {
    const int b = blockIdx.y, p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (p >= H * W) return;
    const int y = p / W, x = p - y * W;
    const float* fl = flow + ((size_t)b * H * W + p) * ld_flow;
    // (clamped as floats BEFORE the integer conversion: a huge or NaN flow lands two pixels outside the image — every tap
    // invalid, like any other out-of-image sample — instead of in an undefined float -> int conversion; fmaxf drops a NaN)
    const float ix = fminf(fmaxf(roundtrip((float)x + fl[0], W), -2.f), (float)W + 1.f);
    const float iy = fminf(fmaxf(roundtrip((float)y + fl[1], H), -2.f), (float)H + 1.f);
    const float x0f = floorf(ix), y0f = floorf(iy);
    const int x0 = (int)x0f, y0 = (int)y0f;
    const float wx1 = ix - x0f, wx0 = (x0f + 1.f) - ix, wy1 = iy - y0f, wy0 = (y0f + 1.f) - iy;
    const float* fb = feat + (size_t)(b % feat_batch) * H * W * C;  // feat given once for several hypotheses
    float* o = HL ? nullptr : out + ((size_t)b * H * W + p) * ld_out;
    _Float16* oh = HL ? (_Float16*)out + ((size_t)b * H * W + p) * terms * ld_out : nullptr;   // terms = 2: hl, 1: h format
    const bool vx0 = x0 >= 0 && x0 < W, vx1 = x0 + 1 >= 0 && x0 + 1 < W;
    const bool vy0 = y0 >= 0 && y0 < H, vy1 = y0 + 1 >= 0 && y0 + 1 < H;
    // Taps outside the image read a clamped pixel whose VALUE is then masked to zero (bitwise: all-ones / all-zeros mask) —
    // zeros padding exactly like grid_sample, also when the clamped pixel holds Inf / NaN (a zero WEIGHT would turn those
    // into NaN) — instead of being skipped by a lane-masked branch: no per-tap lane masks held in SGPR pairs across the
    // loads.  With two busy PROCESSES on one card bits 48..63 of such masks were seen corrupted (lanes 48-63 of single
    // waves lost taps: tests/stress_pc.py); the branch-free form ran 12 000 iterations under that load clean.
    typedef unsigned u4w __attribute__((ext_vector_type(4)));
    const unsigned m00 = 0u - (unsigned)(vy0 & vx0), m01 = 0u - (unsigned)(vy0 & vx1), m10 = 0u - (unsigned)(vy1 & vx0),
                   m11 = 0u - (unsigned)(vy1 & vx1);
    const int xa = min(max(x0, 0), W - 1), xb = min(max(x0 + 1, 0), W - 1), ya = min(max(y0, 0), H - 1), yb = min(max(y0 + 1, 0), H - 1);
    auto tap = [&](int yy, int xx, int c, unsigned m) __attribute__((always_inline)) -> f4 {
        const u4w raw = *(const u4w*)(fb + ((size_t)yy * W + xx) * C + c) & m;
        return __builtin_bit_cast(f4, raw);
    };
    for (int c = lane * 4; c < C; c += 256) {
        f4 acc = {0.f, 0.f, 0.f, 0.f};
        acc += tap(ya, xa, c, m00) * (wx0 * wy0);
        acc += tap(ya, xb, c, m01) * (wx1 * wy0);
        acc += tap(yb, xa, c, m10) * (wx0 * wy1);
        acc += tap(yb, xb, c, m11) * (wx1 * wy1);
        if (HL) {
            typedef _Float16 h4w __attribute__((ext_vector_type(4)));
            _Float16 h0, h1, h2, h3, l0, l1, l2, l3;
            pp_split_f16_chk(acc.x, h0, l0);
            pp_split_f16_chk(acc.y, h1, l1);
            pp_split_f16_chk(acc.z, h2, l2);
            pp_split_f16_chk(acc.w, h3, l3);
            if (terms == 2) {
                _Float16* q = oh + pp_hl_col(c, 0);
                *(h4w*)q = h4w{h0, h1, h2, h3};
                *(h4w*)(q + 8) = h4w{l0, l1, l2, l3};
            } else {
                *(h4w*)(oh + c) = h4w{h0, h1, h2, h3};
            }
        } else {
            *(f4*)(o + c) = acc;
        }
    }
}

// nn.AvgPool2d(2, 2) on NHWC (the pyramid levels of raft_decoder.py:49-51, applied to the
// feature map instead of the correlation volume: the mean commutes with the dot product)
__global__ __launch_bounds__(256) void avgpool2_kernel(const float* __restrict__ in, int H, int W, int C,
                                                       float* __restrict__ out) {
    const int Ho = H / 2, Wo = W / 2, b = blockIdx.y;
    const size_t n = (size_t)Ho * Wo * C;
    const float* ib = in + (size_t)b * H * W * C;
    float* ob = out + (size_t)b * n;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const size_t q = i / C;
        const int ox = (int)(q % Wo), oy = (int)(q / Wo);
        const float* p = ib + ((size_t)(2 * oy) * W + 2 * ox) * C + c;
        ob[i] = ((p[0] + p[C]) + (p[(size_t)W * C] + p[(size_t)W * C + C])) * 0.25f;
    }
}

// CorrelationPyramid + CorrLookup fused — raft_decoder.py:30-53, corr_lookup.py:100-134.
// corr_l[p, q] = <f1[p], pool_l(f2)[q]> / sqrt(C); for every pixel p and level l the (2r+1)^2
// window around (p + flow[p]) / 2^l is bilinearly sampled (zeros padding).  Output channel
// l*(2r+1)^2 + a*(2r+1) + b samples at x offset a-r, y offset b-r (the reference's transposed
// window order).  One wave per pixel: the (2r+2)^2 integer neighbours per level are dotted by
// one lane each (C sequential fmas, f1[p] broadcast from LDS), then lanes blend the 4 corners.
constexpr int MAXR = 2, TW = 2 * MAXR + 2, MAXL = 3;
__global__ __launch_bounds__(256) void corr_lookup_kernel(const float* __restrict__ f1, int ld_f1,
                                                          int f2_batch, const float* __restrict__ f2l0,
                                                          const float* __restrict__ f2l1,
                                                          const float* __restrict__ f2l2,
                                                          const float* __restrict__ flow, int H, int W, int C,
                                                          int L, int r, int ld_flow, float inv_sqrt_c,
                                                          float* __restrict__ out, int ld_out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* a = sm + (size_t)wv * (C + MAXL * TW * TW);  // f1[p] then the corr tables
    float* tab = a + C;
    const int b = blockIdx.y, p = blockIdx.x * 4 + wv;
    const bool live = p < H * W;
    const int y = live ? p / W : 0, x = live ? p - y * W : 0;
    const int tw = 2 * r + 2, win = 2 * r + 1;
    if (live) {
        const float* src = f1 + ((size_t)b * H * W + p) * ld_f1;
        for (int c = lane * 4; c < C; c += 256) *(f4*)(a + c) = *(const f4*)(src + c);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    float cx[MAXL], cy[MAXL];
    int bx[MAXL], by[MAXL];
    if (live) {
        const float* fl = flow + ((size_t)b * H * W + p) * ld_flow;
        const float gx = (float)x + fl[0], gy = (float)y + fl[1];
        for (int l = 0; l < L; ++l) {
            const int Hl = H >> l, Wl = W >> l;
            const float sc = (float)(1 << l);
            // centre sample (offset 0): its integer corner anchors the table, offsets shift by integers
            // (clamped as floats before the integer conversion: a huge / NaN flow is an out-of-image sample, all zeros)
            cx[l] = fminf(fmaxf(roundtrip(gx / sc, Wl), -(float)(r + 2)), (float)(Wl + r + 1));
            cy[l] = fminf(fmaxf(roundtrip(gy / sc, Hl), -(float)(r + 2)), (float)(Hl + r + 1));
            bx[l] = (int)floorf(cx[l]) - r;
            by[l] = (int)floorf(cy[l]) - r;
        }
        const int npos = L * tw * tw;
        for (int i = lane; i < npos; i += 64) {
            const int l = i / (tw * tw), rem = i - l * tw * tw, dy = rem / tw, dx = rem - dy * tw;
            const int Hl = H >> l, Wl = W >> l;
            const int qx = bx[l] + dx, qy = by[l] + dy;
            float dot = 0.f;
            // positions outside the (pooled) map: the dot product runs on a clamped position and its VALUE is then replaced
            // by zero (one select, no lane-masked branch around the loads — DESIGN section 6)
            const bool inside = qx >= 0 && qx < Wl && qy >= 0 && qy < Hl;
            const int qxc = min(max(qx, 0), Wl - 1), qyc = min(max(qy, 0), Hl - 1);
            const float* f2 = l == 0 ? f2l0 : (l == 1 ? f2l1 : f2l2);
            const float* q = f2 + (((size_t)(b % f2_batch) * Hl + qyc) * Wl + qxc) * C;
            for (int c = 0; c < C; c += 4) {
                const f4 u = *(const f4*)(a + c), v = *(const f4*)(q + c);
                dot = fmaf(u.x, v.x, dot);
                dot = fmaf(u.y, v.y, dot);
                dot = fmaf(u.z, v.z, dot);
                dot = fmaf(u.w, v.w, dot);
            }
            tab[i] = inside ? dot * inv_sqrt_c : 0.f;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    if (live) {
        float* o = out + ((size_t)b * H * W + p) * ld_out;
        const int nout = L * win * win;
        for (int i = lane; i < nout; i += 64) {
            const int l = i / (win * win), rem = i - l * win * win, ai = rem / win, bi = rem - ai * win;
            const float wx1 = cx[l] - floorf(cx[l]), wy1 = cy[l] - floorf(cy[l]);
            const float wx0 = 1.f - wx1, wy0 = 1.f - wy1;
            const float* t = tab + l * tw * tw + bi * tw + ai;  // x offset a-r -> dx = ai, y offset b-r -> dy = bi
            o[i] = t[0] * (wx0 * wy0) + t[1] * (wx1 * wy0) + t[tw] * (wx0 * wy1) + t[tw + 1] * (wx1 * wy1);
        }
    }
}

// ---- the same lookup, tiled on the matrix cores --------------------------------------------------------------
// With one lane per neighbour position the kernel above streams 108 separate 1 KB rows of f2 per pixel through the
// texture path (64 distinct cache lines per load instruction): 9.4 ms per step once the flows are realistic.  Here a
// workgroup takes an 8 x 8 tile of source pixels.  Their targets lie close together (a flow field is smooth), so the
// 6 x 6 neighbourhoods of all 64 pixels fall into one 16 x 16 REGION of the (pooled) f2 map: the workgroup computes the
// local correlation S[64 pixels][256 region positions] = F1_tile . F2_region^T on the matrix cores — in the engine's
// f16x3 arithmetic: every fp32 value is split into two fp16 terms as it is staged into LDS (hi = f16(4x), lo = f16(4x -
// hi)) and a product is hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_f16 with fp32 accumulation (22 operand bits; 5x
// the rate of the fp32 MFMA, which made this kernel matrix-bound).  K = C in chunks of 32 channels, coalesced 128-byte
// row segments.  S stays in LDS, and every pixel then blends its 25 bilinear samples per level out of S.  Pixels whose
// neighbourhood does not fit the region are served by further passes with a new region (a pixel with the smallest
// origin is always covered, so the loop ends; smooth flows need one pass), neighbourhoods entirely outside the image
// are zeros without any work.  f2 rows outside the image enter as zero rows (grid_sample's zeros padding).
constexpr int CT = 8, CM = CT * CT;            // source-pixel tile
constexpr int CRW = 16, CN = CRW * CRW;        // region of f2 positions
constexpr int CK = 32, CKP = CK + 4;           // channels per chunk; LDS row = 32 hi + 32 lo halfs + pad = 36 dwords
                                               // (144 B: conflict-free b128 reads)
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
constexpr int CSP = CN + 4;                    // row pitch of S
typedef float f32x16_t __attribute__((ext_vector_type(16)));

// HLIN: f1 / f2 arrive as the engine's hl operand (fp16 [rows][2 ld]: per 8 channels 8 hi then 8 lo terms — what the
// producing convolution's epilogue writes): staging is 16-byte copies, no split arithmetic (it was as long as the MFMAs of a
// chunk).  Same bits either way: the split of a value does not depend on who performs it.
// EXACT (fp32 input only; ops.PRECISION = "f32" / bench.py --mode exact): the chunk is staged as fp32 (a 128-byte row is 32 floats
// instead of 32 hi + 32 lo halfs) and S is accumulated by v_mfma_f32_32x32x2_f32 — exact fp32 products, a k-ordered fma chain per
// entry like the lane-per-position kernel's, 5.3 x the matrix time of the f16x3 form but a third less than that kernel's 108 row
// streams per pixel (9.6 -> measured below, profiles/r05/exact).
// ONE (fp32 input only; ops.PRECISION = "f16" / bench.py --mode fp16): plain fp16 operands — a value is staged as its hi term alone and a
// product is ONE MFMA (the arithmetic of that mode's convolutions: 11 operand bits, fp32 accumulation); a third of the matrix work.
template <bool HLIN, bool EXACT = false, bool ONE = false>
__global__ __launch_bounds__(256, 2) void corr_lookup_mfma_kernel(const void* __restrict__ f1v, int ld_f1, int f2_batch,
                                                                  const void* __restrict__ f2l0, const void* __restrict__ f2l1,
                                                                  const void* __restrict__ f2l2, const float* __restrict__ flow,
                                                                  int B, int H, int W, int C, int L, int r, int ld_flow,
                                                                  float inv_sqrt_c, float* __restrict__ out, int ld_out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* As = sm;                        // [CM][CKP]
    float* Bs = sm + CM * CKP;             // [CN][CKP]
    float* S = sm;                         // [CM][CSP], aliases As/Bs once the K loop is over
    __shared__ float s_cx[CM], s_cy[CM];
    __shared__ int s_bx[CM], s_by[CM], s_state[CM], s_ox, s_oy, s_todo;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int tiles_x = W / CT, tiles = tiles_x * (H / CT), total = tiles * B;
    // workgroups are dealt round-robin to the 8 XCDs: make consecutive tiles (same image, overlapping regions) share an L2
    int g = blockIdx.x;
    if (total % 8 == 0) g = (g % 8) * (total / 8) + g / 8;
    const int b = g / tiles, tile = g - b * tiles;
    const int y0 = (tile / tiles_x) * CT, x0 = (tile % tiles_x) * CT;
    const int tw = 2 * r + 2, win = 2 * r + 1, fit = CRW - tw;
    const int my = tid >> 3, mx = tid & 7;                  // tid < 64: the pixel this thread describes
    float gx = 0.f, gy = 0.f;
    if (tid < CM) {
        const float* fl = flow + ((size_t)b * H * W + (size_t)(y0 + my) * W + x0 + mx) * ld_flow;
        gx = (float)(x0 + mx) + fl[0];
        gy = (float)(y0 + my) + fl[1];
    }
    for (int l = 0; l < L; ++l) {
        const int Hl = H >> l, Wl = W >> l;
        const void* f2v = l == 0 ? f2l0 : (l == 1 ? f2l1 : f2l2);
        const float* f2 = (const float*)f2v + (size_t)(b % f2_batch) * Hl * Wl * C;                 // fp32 input
        const _Float16* f2h = (const _Float16*)f2v + (size_t)(b % f2_batch) * Hl * Wl * C * 2;      // hl input
        const float* f1 = (const float*)f1v;
        const _Float16* f1h = (const _Float16*)f1v;
        if (tid == 0) s_todo = 0;
        __syncthreads();
        if (tid < CM) {
            const float sc = (float)(1 << l);
            // centre sample (offset 0): its integer corner anchors the neighbourhood (clamped far outside: no overflow)
            const float cx = fminf(fmaxf(roundtrip(gx / sc, Wl), -1.0e6f), 1.0e6f);
            const float cy = fminf(fmaxf(roundtrip(gy / sc, Hl), -1.0e6f), 1.0e6f);
            const int bx = (int)floorf(cx) - r, by = (int)floorf(cy) - r;
            s_cx[tid] = cx; s_cy[tid] = cy; s_bx[tid] = bx; s_by[tid] = by;
            const bool outside = bx >= Wl || by >= Hl || bx + tw <= 0 || by + tw <= 0;
            s_state[tid] = outside ? 2 : 0;                // 0 = to do, 1 = done, 2 = all zeros
            if (!outside) atomicAdd(&s_todo, 1);
        }
        __syncthreads();
        for (int idx = tid; idx < CM * win * win; idx += 256) {   // neighbourhoods outside the image: zeros
            const int m = idx / (win * win), o = idx - m * (win * win);
            if (s_state[m] == 2)
                out[((size_t)b * H * W + (size_t)(y0 + (m >> 3)) * W + x0 + (m & 7)) * ld_out + l * win * win + o] = 0.f;
        }
        while (s_todo > 0) {                                // (uniform: read after a barrier, written before the next)
            // ---- region of this pass: origin = smallest bx still to do, then the smallest by among its column band
            if (tid == 0) { s_ox = 0x7fffffff; s_oy = 0x7fffffff; }
            __syncthreads();
            if (tid < CM && s_state[tid] == 0) atomicMin(&s_ox, s_bx[tid]);
            __syncthreads();
            if (tid < CM && s_state[tid] == 0 && s_bx[tid] - s_ox <= fit) atomicMin(&s_oy, s_by[tid]);
            __syncthreads();
            const int ox = s_ox, oy = s_oy;
            // ---- S = F1_tile . F2_region^T
            f32x16_t acc[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
            // Two chunks of register prefetch: chunk kc + 2 is requested as soon as chunk kc has been stored to LDS, so a load has
            // two chunk periods (MFMAs, fragment reads, two barriers each) to arrive; with one chunk ahead the L2 latency showed.
            f4 ra0[2], rb0[8], ra1[2], rb1[8];   // (HLIN: the 16 bytes are 8 halfs — piece `part` of the row's 128-byte chunk)
            const int part = tid & 7, row0 = tid >> 3;
            auto load_chunk = [&](int kc, f4 (&ra)[2], f4 (&rb)[8]) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int m = row0 + 32 * i;
                    const size_t px = (size_t)b * H * W + (size_t)(y0 + (m >> 3)) * W + x0 + (m & 7);
                    ra[i] = HLIN ? *(const f4*)(f1h + (px * ld_f1 + kc * CK) * 2 + 8 * part) : *(const f4*)(f1 + px * ld_f1 + kc * CK + 4 * part);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int n = row0 + 32 * i, qy = oy + (n >> 4), qx = ox + (n & 15);
                    const bool in = qx >= 0 && qx < Wl && qy >= 0 && qy < Hl;
                    const size_t px = (size_t)qy * Wl + qx;
                    rb[i] = !in ? f4{0.f, 0.f, 0.f, 0.f}
                                : (HLIN ? *(const f4*)(f2h + (px * C + kc * CK) * 2 + 8 * part) : *(const f4*)(f2 + px * C + kc * CK + 4 * part));
                }
            };
            // LDS rows hold the chunk in hl order: per 8 channels 8 hi then 8 lo halfs (fp32 input: split here)
            auto put = [&](float* rowp, f4 v) __attribute__((always_inline)) {
                if (EXACT) {
                    *(f4*)(rowp + 4 * part) = v;
                    return;
                }
                if (HLIN) {
                    *(f4*)((_Float16*)rowp + 8 * part) = v;
                    return;
                }
                // (staging into LDS, not an operand buffer: the same maps' saturation is reported by the epilogue that wrote their operand form)
                _Float16* g8 = (_Float16*)rowp + 16 * (part >> 1) + 4 * (part & 1);   // channels 4 part .. + 3 of group part / 2
                if (ONE) {   // the hi plane alone (the lo plane of the row is never read)
                    const h4_t hi = {pp_to_f16(v.x), pp_to_f16(v.y), pp_to_f16(v.z), pp_to_f16(v.w)};
                    *(h4_t*)g8 = hi;
                    return;
                }
                _Float16 h0, h1, h2, h3, l0, l1, l2, l3;
                pp_split_f16(v.x, h0, l0);
                pp_split_f16(v.y, h1, l1);
                pp_split_f16(v.z, h2, l2);
                pp_split_f16(v.w, h3, l3);
                const h4_t hi = {h0, h1, h2, h3}, lo = {l0, l1, l2, l3};
                *(h4_t*)g8 = hi;
                *(h4_t*)(g8 + 8) = lo;
            };
#ifdef PP_STUDY_CORR_NK1   // (timing-only study build: one channel chunk instead of C / 32 — sizes the K loop's share)
            const int nk = 1;
#else
            const int nk = C / CK;
#endif
            auto chunk = [&](int kc, f4 (&ra)[2], f4 (&rb)[8]) __attribute__((always_inline)) {
                __syncthreads();                            // the previous chunk (or the previous pass's S) has been read
#pragma unroll
                for (int i = 0; i < 2; ++i) put(As + (row0 + 32 * i) * CKP, ra[i]);
#pragma unroll
                for (int i = 0; i < 8; ++i) put(Bs + (row0 + 32 * i) * CKP, rb[i]);
                __syncthreads();
                if (kc + 2 < nk) load_chunk(kc + 2, ra, rb);    // in flight under two chunks of MFMAs
                if constexpr (EXACT) {
                    // lane (l31, lh) feeds row / column l31 with channels 16 lh + 4 q .. + 3: MFMA e of quad q multiplies the pair
                    // (4 q + e, 16 + 4 q + e) — the order of the fp32 engine (csrc/pp_gemm_f.hip)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f4 af[2], bf[2];
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            af[i] = *(const f4*)(As + (i * 32 + l31) * CKP + 16 * lh + 4 * q);
                            bf[i] = *(const f4*)(Bs + (wv * 64 + i * 32 + l31) * CKP + 16 * lh + 4 * q);
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e)
#pragma unroll
                            for (int i = 0; i < 2; ++i)
#pragma unroll
                                for (int j = 0; j < 2; ++j)
                                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
                    }
                    return;
                }
                // lane (l31, lh) feeds row / column l31 with channels 16 q + 8 lh .. + 7 of the chunk in k-step q
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    h8_t ah[2], al[2], bh[2], bl[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const _Float16* ar = (const _Float16*)(As + (i * 32 + l31) * CKP) + 16 * (2 * q + lh);
                        const _Float16* br = (const _Float16*)(Bs + (wv * 64 + i * 32 + l31) * CKP) + 16 * (2 * q + lh);
                        ah[i] = *(const h8_t*)ar;
                        bh[i] = *(const h8_t*)br;
                        if (!ONE) {
                            al[i] = *(const h8_t*)(ar + 8);
                            bl[i] = *(const h8_t*)(br + 8);
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                            if (!ONE) {
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
                            }
                        }
                }
            };
            load_chunk(0, ra0, rb0);
            if (nk > 1) load_chunk(1, ra1, rb1);
            for (int kc = 0; kc < nk; kc += 2) {
                chunk(kc, ra0, rb0);
                if (kc + 1 < nk) chunk(kc + 1, ra1, rb1);
            }
            __syncthreads();                                // every wave is done reading As/Bs: S may overwrite them
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e)            // accumulator (e, lane): row (e/4)*8 + lh*4 + e%4, column l31
                        S[(i * 32 + (e >> 2) * 8 + lh * 4 + (e & 3)) * CSP + wv * 64 + j * 32 + l31] =
                            acc[i][j][e] * (EXACT ? inv_sqrt_c : inv_sqrt_c / (PP_A_SCALE * PP_A_SCALE));
            __syncthreads();
            // ---- blend the 25 samples of every pixel whose neighbourhood lies in the region
            for (int idx = tid; idx < CM * win * win; idx += 256) {
                const int m = idx / (win * win), o = idx - m * (win * win), ai = o / win, bi = o - ai * win;
                const int dx = s_bx[m] - ox, dy = s_by[m] - oy;
                if (s_state[m] == 0 && dx >= 0 && dx <= fit && dy >= 0 && dy <= fit) {
                    const float cx = s_cx[m], cy = s_cy[m];
                    const float wx1 = cx - floorf(cx), wy1 = cy - floorf(cy), wx0 = 1.f - wx1, wy0 = 1.f - wy1;
                    const float* t = S + m * CSP + (dy + bi) * CRW + dx + ai;   // x offset a-r -> +ai, y offset b-r -> +bi
                    out[((size_t)b * H * W + (size_t)(y0 + (m >> 3)) * W + x0 + (m & 7)) * ld_out + l * win * win + o] =
                        t[0] * (wx0 * wy0) + t[1] * (wx1 * wy0) + t[CRW] * (wx0 * wy1) + t[CRW + 1] * (wx1 * wy1);
                }
            }
            __syncthreads();
            if (tid < CM && s_state[tid] == 0) {
                const int dx = s_bx[tid] - ox, dy = s_by[tid] - oy;
                if (dx >= 0 && dx <= fit && dy >= 0 && dy <= fit) {
                    s_state[tid] = 1;
                    atomicSub(&s_todo, 1);
                }
            }
            __syncthreads();
        }
        __syncthreads();
    }
}

// Row gather: dst[i] = src[index[i]] for rows of `row` floats (row % 4 == 0, 16-byte aligned): the selection of the
// top-k templates' data (model/picopose.py:55-62 — torch.gather with an expanded index over (B,N,3,224,224) etc.)
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, const long long* __restrict__ index,
                                                          long long row, long long nsrc, float* __restrict__ dst) {
    const long long r = index[blockIdx.y];
    if (r < 0 || r >= nsrc) return;  // (validated on the host side of the mirrors; never expected)
    const f4* s4 = (const f4*)(src + r * row);
    f4* d4 = (f4*)(dst + (long long)blockIdx.y * row);
    const long long n4 = row >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) d4[i] = s4[i];
}

// Crop + resize + normalise of one detection (provider/bop_test_dataset.py:162-177 with utils/data_utils.py:231-250):
// out_rgb[c] = (resize_linear(image[y1:y2, x1:x2, 2-c] / 255 [* (mask > 0)]) - mean[c]) / std[c], out_mask =
// resize_nearest(mask[y1:y2, x1:x2]).  cv2.resize semantics on a float image (pixel centres, edge clamp), in double
// like OpenCV's CV_64F path and torchvision's Normalize on the float64 tensor.  One thread per output pixel.
__global__ __launch_bounds__(256) void crop_resize_kernel(const unsigned char* __restrict__ img, int W,
                                                          const unsigned char* __restrict__ mask, int y1, int y2, int x1,
                                                          int x2, int S, int mask_rgb, double m0, double m1, double m2,
                                                          double s0, double s1, double s2, float* __restrict__ out_rgb,
                                                          float* __restrict__ out_mask) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= S * S) return;
    const int oy = i / S, ox = i - oy * S, h = y2 - y1, w = x2 - x1;
    auto taps = [](int o, int n, int S_, int& i0, int& i1, double& fr) {
        const double f = ((double)o + 0.5) * ((double)n / (double)S_) - 0.5;
        i0 = (int)floor(f);
        fr = f - (double)i0;
        if (i0 < 0) {
            i0 = 0;
            fr = 0.0;
        }
        if (i0 >= n - 1) {
            i0 = n - 1;
            fr = 0.0;
        }
        i1 = i0 + 1 < n ? i0 + 1 : n - 1;
    };
    int ya, yb, xa, xb;
    double fy, fx;
    taps(oy, h, S, ya, yb, fy);
    taps(ox, w, S, xa, xb, fx);
    const double mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
    auto px = [&](int yy, int xx, int c) -> double {  // channel c of the flipped ([..., ::-1]) crop, / 255, optionally masked
        const size_t p = (size_t)(y1 + yy) * W + (x1 + xx);
        double v = (double)img[p * 3 + (2 - c)] / 255.0;
        if (mask_rgb && mask && mask[p] == 0) v = 0.0;
        return v;
    };
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double top = px(ya, xa, c) * (1.0 - fx) + px(ya, xb, c) * fx;
        const double bot = px(yb, xa, c) * (1.0 - fx) + px(yb, xb, c) * fx;
        out_rgb[(size_t)c * S * S + i] = (float)(((top * (1.0 - fy) + bot * fy) - mean[c]) / stdv[c]);
    }
    if (out_mask) {
        int yi = (int)floor((double)oy * ((double)h / (double)S)), xi = (int)floor((double)ox * ((double)w / (double)S));
        yi = yi < h - 1 ? yi : h - 1;
        xi = xi < w - 1 ? xi : w - 1;
        out_mask[i] = mask ? (float)mask[(size_t)(y1 + yi) * W + (x1 + xi)] : 1.f;
    }
}

// Template lookup points (provider/bop_test_dataset.py:233-235 with utils/data_utils.py:97-115): the back-projected depth
// crop, INTER_NEAREST-resized to P x P: out[i, j] = ((x - cx) z / fx, (y - cy) z / fy, z) at the sampled crop pixel.
__global__ __launch_bounds__(256) void depth_points_kernel(const float* __restrict__ depth, int W, int y1, int y2, int x1,
                                                           int x2, int P, float fx, float fy, float cx, float cy,
                                                           float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P * P) return;
    const int oy = i / P, ox = i - oy * P, h = y2 - y1, w = x2 - x1;
    int yi = (int)floor((double)oy * ((double)h / (double)P)), xi = (int)floor((double)ox * ((double)w / (double)P));
    yi = yi < h - 1 ? yi : h - 1;
    xi = xi < w - 1 ? xi : w - 1;
    const int y = y1 + yi, x = x1 + xi;
    const float z = depth[(size_t)y * W + x];
    out[3 * i + 0] = ((float)x - cx) * z / fx;
    out[3 * i + 1] = ((float)y - cy) * z / fy;
    out[3 * i + 2] = z;
}


// ---- convolutions with one or two output channels (the flow / certainty predict layers, raft_decoder.py:287-289) -----
// On the GEMM engine such a layer pads its 2 (or 1) filters to a 64-wide tile and re-reads every pixel per tap from L2
// (0.6 ms for the 3x3 at 64 x 64 x 160, 10 TFLOP/s).  Here a workgroup takes a band of 256 / W full image rows (one output
// pixel per thread), stages the band plus a one-pixel halo per 32-channel slice in LDS (the input is the hl operand its
// producer wrote: 128 bytes per pixel and slice; x = hi + lo, the exact fp32 value x 4, is formed once per staged element) and
// every thread walks the taps of its pixel: the weights are the layer's fp32 filters (scalar loads: the index is uniform), fp32 fma.
constexpr int NRW_PITCH = 144;   // bytes per staged pixel: 128 + 16 (conflict-free 16-byte reads along a row of pixels)

// F32IN (round 5, ops.PRECISION = "f32"): the input is the fp32 NHWC map itself (ldx floats per pixel; a 32-channel slice of a pixel is
// the same 128 bytes), every product and sum fp32 — the exact-mode form of the same layers (they ran as GEMMs padded from 2 columns to 64).
template <int KS, int NOUT, bool F32IN = false>
__global__ __launch_bounds__(256) void conv_narrow_kernel(const _Float16* __restrict__ x, int ldx, int H, int W, int C,
                                                          const float* __restrict__ w, const float* __restrict__ bias,
                                                          const float* __restrict__ residual, float* __restrict__ out) {
    constexpr int R = KS / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char nsm[];
    const int TH = 256 / W, bands = H / TH;
    const int b = blockIdx.x / bands, y0 = (blockIdx.x % bands) * TH;
    const int tid = threadIdx.x, ty = tid / W, tx = tid % W;
    const int PW = W + 2 * R, PH = TH + 2 * R;       // staged pixels per row / rows
    float acc[NOUT];
#pragma unroll
    for (int n = 0; n < NOUT; ++n) acc[n] = 0.f;
    const int K = KS * KS * C;
    // staged items of this thread, fetched one slice ahead.  fp32 input: 16-byte pieces (at most 13 for the 6 x 66 band of a 64-wide
    // image).  hl input: 32-byte items — the hi and the lo terms of 8 channels — at most 7; they are summed to the fp32 value ONCE as
    // they are staged (round 6: every tap of every neighbour used to redo the two conversions and the add — 3 of the 5 vector
    // instructions per product), so LDS holds 32 floats per pixel and slice either way and the tap loop is pure fma.
    constexpr int MAXP = F32IN ? 13 : 7, PPX = F32IN ? 8 : 4;      // items per thread; items per pixel
    const int npiece = PH * PW * PPX;
    float4 pre[F32IN ? MAXP : 2 * MAXP];
    auto fetch = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < MAXP; ++j) {
            // halo pixels outside the image and slots past the band: the load runs on a clamped pixel and its value is masked
            // to zero bit-wise (zero padding) — no per-lane guarded load, no lane masks held across it
            const int i = min(tid + 256 * j, npiece - 1);
            const int piece = i % PPX, px = i / PPX, sy = px / PW, sx = px - sy * PW;
            const int iy = y0 + sy - R, ix = sx - R;
            const unsigned m = 0u - (unsigned)((tid + 256 * j < npiece) & (iy >= 0) & (iy < H) & (ix >= 0) & (ix < W));
            const int iyc = min(max(iy, 0), H - 1), ixc = min(max(ix, 0), W - 1);
            typedef unsigned u4n __attribute__((ext_vector_type(4)));
            if constexpr (F32IN) {
                const u4n raw = *(const u4n*)((const float*)x + (((size_t)b * H + iyc) * W + ixc) * (size_t)ldx + c0 + 4 * piece) & m;
                pre[j] = __builtin_bit_cast(float4, raw);
            } else {
                const u4n* src = (const u4n*)(x + (((size_t)b * H + iyc) * W + ixc) * (size_t)(2 * ldx) + 2 * c0 + 16 * piece);
                pre[2 * j] = __builtin_bit_cast(float4, src[0] & m);       // 8 hi terms
                pre[2 * j + 1] = __builtin_bit_cast(float4, src[1] & m);   // 8 lo terms
            }
        }
    };
    fetch(0);
    for (int c0 = 0; c0 < C; c0 += 32) {
        __syncthreads();                             // the previous slice has been read
#pragma unroll
        for (int j = 0; j < MAXP; ++j) {
            const int i = tid + 256 * j;
            if (i >= npiece) continue;
            if constexpr (F32IN) {
                *(float4*)(nsm + (size_t)(i >> 3) * NRW_PITCH + 16 * (i & 7)) = pre[j];
            } else {
                typedef _Float16 h8v __attribute__((ext_vector_type(8)));
                const h8v hi = __builtin_bit_cast(h8v, pre[2 * j]), lo = __builtin_bit_cast(h8v, pre[2 * j + 1]);
                float4 a, c;
                a.x = (float)hi[0] + (float)lo[0]; a.y = (float)hi[1] + (float)lo[1]; a.z = (float)hi[2] + (float)lo[2]; a.w = (float)hi[3] + (float)lo[3];
                c.x = (float)hi[4] + (float)lo[4]; c.y = (float)hi[5] + (float)lo[5]; c.z = (float)hi[6] + (float)lo[6]; c.w = (float)hi[7] + (float)lo[7];
                unsigned char* d = nsm + (size_t)(i >> 2) * NRW_PITCH + 32 * (i & 3);
                *(float4*)d = a;
                *(float4*)(d + 16) = c;
            }
        }
        __syncthreads();
        if (c0 + 32 < C) fetch(c0 + 32);             // in flight under this slice's arithmetic
#pragma unroll
        for (int dy = 0; dy < KS; ++dy)
#pragma unroll
            for (int dx = 0; dx < KS; ++dx) {
                const unsigned char* p = nsm + (size_t)((ty + dy) * PW + tx + dx) * NRW_PITCH;
                const float* wt = w + (dy * KS + dx) * C + c0;
#pragma unroll
                for (int g = 0; g < 8; ++g) {        // (channels in ascending order, as before the staging change: same bits)
                    const f4 xv = *(const f4*)(p + 16 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int n = 0; n < NOUT; ++n) acc[n] = fmaf(xv[e], wt[n * K + 4 * g + e], acc[n]);
                }
            }
    }
    const size_t o = (((size_t)b * H + y0 + ty) * W + tx) * NOUT;
#pragma unroll
    for (int n = 0; n < NOUT; ++n) {
        float v = (F32IN ? acc[n] : acc[n] * (1.0f / PP_A_SCALE)) + (bias ? bias[n] : 0.f);
        if (residual) v += residual[o + n];
        out[o + n] = v;
    }
}

}  // namespace

PP_SAT_SETTER(pp_sat_set_sample)

extern "C" {

int pp_resize_bilinear_nhwc(const float* in, int B, int H, int W, int C, int Ho, int Wo, float mul, float* out,
                            void* stream) {
    if (!in || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || Ho <= 0 || Wo <= 0) return PP_EINVAL;
    const bool vec = (C & 3) == 0 && (((uintptr_t)in | (uintptr_t)out) & 15) == 0;  // (the kernel makes the same test)
    const int work = vec ? Wo * (C / 4) : Wo * C;
    const int gx = (work + 1023) / 1024;  // four items per thread: one workgroup per output row for 64 x 256 channels
    hipLaunchKernelGGL(resize_kernel, dim3(gx < 64 ? gx : 64, Ho, B), dim3(256), 0, (hipStream_t)stream, in, H, W,
                       C, Ho, Wo, mul, out, (_Float16*)nullptr, 2);
    return pp_last_launch();
}

int pp_resize_bilinear_nhwc_t(const float* in, int B, int H, int W, int C, int Ho, int Wo, float mul, void* out_hl,
                              int terms, void* stream) {
    if (!in || !out_hl || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 != 0 || Ho <= 0 || Wo <= 0 ||
        (((uintptr_t)in | (uintptr_t)out_hl) & 15) != 0 || (terms != 1 && terms != 2))
        return PP_EINVAL;
    const int gx = (Wo * (C / 8) + 1023) / 1024;
    hipLaunchKernelGGL(resize_kernel, dim3(gx < 64 ? gx : 64, Ho, B), dim3(256), 0, (hipStream_t)stream, in, H, W,
                       C, Ho, Wo, mul, (float*)nullptr, (_Float16*)out_hl, terms);
    return pp_last_launch();
}

int pp_resize_bilinear_nhwc_dual(const float* in, int B, int H, int W, int C, int Ho, int Wo, float mul, float* out, void* out_hl,
                                 int terms, void* stream) {
    if (!in || !out || !out_hl || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 != 0 || Ho <= 0 || Wo <= 0 ||
        (((uintptr_t)in | (uintptr_t)out | (uintptr_t)out_hl) & 15) != 0 || (terms != 1 && terms != 2))
        return PP_EINVAL;
    const int gx = (Wo * (C / 8) + 1023) / 1024;
    hipLaunchKernelGGL(resize_kernel, dim3(gx < 64 ? gx : 64, Ho, B), dim3(256), 0, (hipStream_t)stream, in, H, W,
                       C, Ho, Wo, mul, out, (_Float16*)out_hl, terms);
    return pp_last_launch();
}

int pp_resize_bilinear_nhwc_hl(const float* in, int B, int H, int W, int C, int Ho, int Wo, float mul, void* out_hl,
                               void* stream) {
    return pp_resize_bilinear_nhwc_t(in, B, H, W, C, Ho, Wo, mul, out_hl, 2, stream);
}

int pp_warp_nhwc(const float* feat, int feat_batch, const float* flow, int B, int H, int W, int C, int ld_flow,
                 float* out, int ld_out, void* stream) {
    if (!feat || !flow || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4 != 0 || ld_flow < 2 || feat_batch <= 0 ||
        ld_out < C || ld_out % 4 != 0 || ((uintptr_t)out % 16) != 0)
        return PP_EINVAL;
    hipLaunchKernelGGL(warp_kernel<false>, dim3((H * W + 3) / 4, B), dim3(256), 0, (hipStream_t)stream, feat, feat_batch, flow,
                       H, W, C, ld_flow, out, ld_out, 2);
    return pp_last_launch();
}

int pp_warp_nhwc_t(const float* feat, int feat_batch, const float* flow, int B, int H, int W, int C, int ld_flow,
                   void* out_hl, int ld_h, int terms, void* stream) {
    if (!feat || !flow || !out_hl || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 != 0 || ld_flow < 2 || feat_batch <= 0 ||
        ld_h < C || ld_h % 8 != 0 || ((uintptr_t)out_hl % 16) != 0 || (terms != 1 && terms != 2))
        return PP_EINVAL;
    hipLaunchKernelGGL(warp_kernel<true>, dim3((H * W + 3) / 4, B), dim3(256), 0, (hipStream_t)stream, feat, feat_batch, flow,
                       H, W, C, ld_flow, (float*)out_hl, ld_h, terms);
    return pp_last_launch();
}

int pp_warp_nhwc_hl(const float* feat, int feat_batch, const float* flow, int B, int H, int W, int C, int ld_flow,
                    void* out_hl, int ld_h, void* stream) {
    return pp_warp_nhwc_t(feat, feat_batch, flow, B, H, W, C, ld_flow, out_hl, ld_h, 2, stream);
}

int pp_avgpool2_nhwc(const float* in, int B, int H, int W, int C, float* out, void* stream) {
    if (!in || !out || B <= 0 || H < 2 || W < 2 || C <= 0 || H % 2 != 0 || W % 2 != 0) return PP_EINVAL;
    const size_t n = (size_t)(H / 2) * (W / 2) * C;
    const int gx = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(avgpool2_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, in, H, W, C, out);
    return pp_last_launch();
}

static int corr_tiled_launch(bool hl, const void* f1, int ld_f1, const void* f2_l0, const void* f2_l1, const void* f2_l2, int f2_batch,
                             const float* flow, int B, int H, int W, int C, int levels, int radius, int ld_flow, float* out, int ld_out,
                             void* stream, bool exact = false, bool one = false) {
    // matrix-core version: one workgroup per 8 x 8 pixel tile
    const size_t lds = (size_t)(CM * CSP > (CM + CN) * CKP ? CM * CSP : (CM + CN) * CKP) * sizeof(float);
    static signed char attr[PP_MAX_DEVICES];
    signed char& ok = attr[pp_cur_device()];
    if (ok == 0)
        ok = hipFuncSetAttribute((const void*)corr_lookup_mfma_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
                     hipFuncSetAttribute((const void*)corr_lookup_mfma_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
                     hipFuncSetAttribute((const void*)corr_lookup_mfma_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
                     hipFuncSetAttribute((const void*)corr_lookup_mfma_kernel<false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess
                 ? 1 : -1;
    if (ok < 0) return PP_ELAUNCH;
    const dim3 grid((unsigned)((H / CT) * (W / CT) * B));
    if (exact)
        hipLaunchKernelGGL((corr_lookup_mfma_kernel<false, true>), grid, dim3(256), lds, (hipStream_t)stream, f1, ld_f1, f2_batch, f2_l0, f2_l1, f2_l2,
                           flow, B, H, W, C, levels, radius, ld_flow, 1.0f / sqrtf((float)C), out, ld_out);
    else if (one)
        hipLaunchKernelGGL((corr_lookup_mfma_kernel<false, false, true>), grid, dim3(256), lds, (hipStream_t)stream, f1, ld_f1, f2_batch, f2_l0, f2_l1,
                           f2_l2, flow, B, H, W, C, levels, radius, ld_flow, 1.0f / sqrtf((float)C), out, ld_out);
    else if (hl)
        hipLaunchKernelGGL(corr_lookup_mfma_kernel<true>, grid, dim3(256), lds, (hipStream_t)stream, f1, ld_f1, f2_batch, f2_l0, f2_l1, f2_l2,
                           flow, B, H, W, C, levels, radius, ld_flow, 1.0f / sqrtf((float)C), out, ld_out);
    else
        hipLaunchKernelGGL(corr_lookup_mfma_kernel<false>, grid, dim3(256), lds, (hipStream_t)stream, f1, ld_f1, f2_batch, f2_l0, f2_l1, f2_l2,
                           flow, B, H, W, C, levels, radius, ld_flow, 1.0f / sqrtf((float)C), out, ld_out);
    return pp_last_launch();
}

int pp_conv_narrow_f32(const float* x, int ld_x, int B, int H, int W, int C, const float* weight, const float* bias, int ksize,
                       int n_out, const float* residual, float* out, void* stream) {
    if (!x || !weight || !out || B <= 0 || H <= 0 || C <= 0 || C % 32 != 0 || ld_x < C || ld_x % 4 != 0) return PP_EINVAL;
    if ((W != 16 && W != 32 && W != 64) || H % (256 / W) != 0 || (ksize != 1 && ksize != 3) || (n_out != 1 && n_out != 2)) return PP_EINVAL;
    if (((uintptr_t)x % 16) != 0) return PP_EINVAL;
    const int TH = 256 / W, R = ksize / 2;
    const size_t lds = (size_t)(TH + 2 * R) * (W + 2 * R) * NRW_PITCH;
    const dim3 grid((unsigned)(B * (H / TH)));
    hipStream_t st = (hipStream_t)stream;
#define NRW_LAUNCH(KS_, N_)                                                                                                                  \
    hipLaunchKernelGGL((conv_narrow_kernel<KS_, N_, true>), grid, dim3(256), lds, st, (const _Float16*)x, ld_x, H, W, C, weight, bias, residual, out)
    if (ksize == 3 && n_out == 2) NRW_LAUNCH(3, 2);
    else if (ksize == 3) NRW_LAUNCH(3, 1);
    else if (n_out == 2) NRW_LAUNCH(1, 2);
    else NRW_LAUNCH(1, 1);
#undef NRW_LAUNCH
    return pp_last_launch();
}

int pp_corr_lookup_nhwc_ex(const float* f1, int ld_f1, const float* f2_l0, const float* f2_l1, const float* f2_l2,
                           int f2_batch, const float* flow, int B, int H, int W, int C, int levels, int radius,
                           int ld_flow, int prec, float* out, int ld_out, void* stream) {
    if (prec != PP_PREC_F32 && prec != PP_PREC_F16X3 && prec != PP_PREC_F16) return PP_EINVAL;
    if (!f1 || !f2_l0 || !flow || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4 != 0) return PP_EINVAL;
    if (ld_f1 < C || ld_f1 % 4 != 0 || ((uintptr_t)f1 % 16) != 0 || f2_batch <= 0) return PP_EINVAL;
    if (levels < 1 || levels > MAXL || radius < 1 || radius > MAXR) return PP_EINVAL;
    if ((levels > 1 && !f2_l1) || (levels > 2 && !f2_l2)) return PP_EINVAL;
    if ((H >> (levels - 1)) < 1 || (W >> (levels - 1)) < 1 || ld_flow < 2) return PP_EINVAL;
    const int win = 2 * radius + 1;
    if (ld_out < levels * win * win) return PP_EINVAL;
    const char* te = getenv("PP_CORR_TILED");   // read per call: the tests run both kernels in one process
    const bool tiled = !(te && te[0] == '0');   // (PP_PREC_F32: the tiled kernel on exact fp32 MFMAs — every product and sum fp32)
    // every pyramid level the tiled kernel reads goes through 16-byte vector loads: a misaligned one falls back (as the hl entry
    // point below rejects it) instead of faulting
    const bool aligned = (((uintptr_t)f2_l0 | (levels > 1 ? (uintptr_t)f2_l1 : 0) | (levels > 2 ? (uintptr_t)f2_l2 : 0)) % 16) == 0;
    if (tiled && H % CT == 0 && W % CT == 0 && C % CK == 0 && aligned)   // (PP_CORR_TILED=0 keeps the lane-per-position kernel)
        return corr_tiled_launch(false, f1, ld_f1, f2_l0, f2_l1, f2_l2, f2_batch, flow, B, H, W, C, levels, radius, ld_flow, out, ld_out, stream,
                                 prec == PP_PREC_F32, prec == PP_PREC_F16);
    // (PP_PREC_F16 on the lane-per-position kernel below: fp32 products — finer than the mode asks for)
    const size_t smem = (size_t)4 * (C + MAXL * TW * TW) * sizeof(float);
    hipLaunchKernelGGL(corr_lookup_kernel, dim3((H * W + 3) / 4, B), dim3(256), smem, (hipStream_t)stream, f1, ld_f1,
                       f2_batch, f2_l0, f2_l1, f2_l2, flow, H, W, C, levels, radius, ld_flow, 1.0f / sqrtf((float)C), out,
                       ld_out);
    return pp_last_launch();
}

int pp_corr_lookup_nhwc_hl(const void* f1_hl, int ld_f1, const void* f2_hl_l0, const void* f2_hl_l1, const void* f2_hl_l2,
                           int f2_batch, const float* flow, int B, int H, int W, int C, int levels, int radius, int ld_flow,
                           float* out, int ld_out, void* stream) {
    if (!f1_hl || !f2_hl_l0 || !flow || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || f2_batch <= 0) return PP_EINVAL;
    if (H % CT != 0 || W % CT != 0 || C % CK != 0 || ld_f1 < C || ld_f1 % 8 != 0) return PP_EINVAL;   // tiled shapes only
    if (((uintptr_t)f1_hl | (uintptr_t)f2_hl_l0 | (uintptr_t)f2_hl_l1 | (uintptr_t)f2_hl_l2) % 16 != 0) return PP_EINVAL;
    if (levels < 1 || levels > MAXL || radius < 1 || radius > MAXR) return PP_EINVAL;
    if ((levels > 1 && !f2_hl_l1) || (levels > 2 && !f2_hl_l2)) return PP_EINVAL;
    if ((H >> (levels - 1)) < 1 || (W >> (levels - 1)) < 1 || ld_flow < 2) return PP_EINVAL;
    const int win = 2 * radius + 1;
    if (ld_out < levels * win * win) return PP_EINVAL;
    return corr_tiled_launch(true, f1_hl, ld_f1, f2_hl_l0, f2_hl_l1, f2_hl_l2, f2_batch, flow, B, H, W, C, levels, radius, ld_flow, out, ld_out,
                             stream);
}

int pp_corr_lookup_nhwc(const float* f1, int ld_f1, const float* f2_l0, const float* f2_l1, const float* f2_l2,
                        int f2_batch, const float* flow, int B, int H, int W, int C, int levels, int radius,
                        int ld_flow, float* out, int ld_out, void* stream) {
    return pp_corr_lookup_nhwc_ex(f1, ld_f1, f2_l0, f2_l1, f2_l2, f2_batch, flow, B, H, W, C, levels, radius, ld_flow,
                                  PP_PREC_F16X3, out, ld_out, stream);
}

int pp_gather_rows(const float* src, const long long* index, long long n_src_rows, long long row_floats, int n, float* dst,
                   void* stream) {
    if (!src || !index || !dst || n <= 0 || n_src_rows <= 0 || row_floats <= 0 || row_floats % 4 != 0 ||
        ((uintptr_t)src % 16) != 0 || ((uintptr_t)dst % 16) != 0)
        return PP_EINVAL;
    const long long n4 = row_floats / 4;
    const int gx = (int)((n4 + 255) / 256 < 64 ? (n4 + 255) / 256 : 64);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(gx, n), dim3(256), 0, (hipStream_t)stream, src, index, row_floats, n_src_rows,
                       dst);
    return pp_last_launch();
}

int pp_crop_resize_normalize(const unsigned char* image, int H, int W, const unsigned char* mask, int y1, int y2, int x1,
                             int x2, int S, int rgb_mask_flag, const double* mean3, const double* std3, float* out_rgb,
                             float* out_mask, void* stream) {
    if (!image || !out_rgb || !mean3 || !std3 || H <= 0 || W <= 0 || S <= 0 || y1 < 0 || x1 < 0 || y2 > H || x2 > W ||
        y2 <= y1 || x2 <= x1)
        return PP_EINVAL;
    hipLaunchKernelGGL(crop_resize_kernel, dim3((S * S + 255) / 256), dim3(256), 0, (hipStream_t)stream, image, W, mask, y1, y2,
                       x1, x2, S, rgb_mask_flag, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], out_rgb, out_mask);
    return pp_last_launch();
}

int pp_depth_points_nearest(const float* depth_m, int H, int W, int y1, int y2, int x1, int x2, int P, float fx, float fy,
                            float cx, float cy, float* out_pts, void* stream) {
    if (!depth_m || !out_pts || H <= 0 || W <= 0 || P <= 0 || y1 < 0 || x1 < 0 || y2 > H || x2 > W || y2 <= y1 || x2 <= x1 ||
        fx == 0.f || fy == 0.f)
        return PP_EINVAL;
    hipLaunchKernelGGL(depth_points_kernel, dim3((P * P + 255) / 256), dim3(256), 0, (hipStream_t)stream, depth_m, W, y1, y2, x1,
                       x2, P, fx, fy, cx, cy, out_pts);
    return pp_last_launch();
}

int pp_conv_narrow_hl(const void* x_hl, int ld_x, int B, int H, int W, int C, const float* weight, const float* bias, int ksize,
                      int n_out, const float* residual, float* out, void* stream) {
    if (!x_hl || !weight || !out || B <= 0 || H <= 0 || C <= 0 || C % 32 != 0 || ld_x < C || ld_x % 8 != 0) return PP_EINVAL;
    if ((W != 16 && W != 32 && W != 64) || H % (256 / W) != 0 || (ksize != 1 && ksize != 3) || (n_out != 1 && n_out != 2)) return PP_EINVAL;
    if (((uintptr_t)x_hl % 16) != 0) return PP_EINVAL;
    const int TH = 256 / W, R = ksize / 2;
    const size_t lds = (size_t)(TH + 2 * R) * (W + 2 * R) * NRW_PITCH;
    const dim3 grid((unsigned)(B * (H / TH)));
    hipStream_t st = (hipStream_t)stream;
#define NRW_LAUNCH(KS_, N_)                                                                                                        \
    hipLaunchKernelGGL((conv_narrow_kernel<KS_, N_>), grid, dim3(256), lds, st, (const _Float16*)x_hl, ld_x, H, W, C, weight, bias, \
                       residual, out)
    if (ksize == 3 && n_out == 2) NRW_LAUNCH(3, 2);
    else if (ksize == 3) NRW_LAUNCH(3, 1);
    else if (n_out == 2) NRW_LAUNCH(1, 2);
    else NRW_LAUNCH(1, 1);
#undef NRW_LAUNCH
    return pp_last_launch();
}

}  // extern "C"
