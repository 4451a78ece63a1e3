// Kernels of the TRAINING forward (model/picopose.py:114-137) that the inference path does not
// have: the key-point sampler, BatchNorm on batch statistics, the InfoNCE rows and the flow / certainty loss sums.
// The networks themselves run on the same GEMM engine as in inference (csrc/pp_gemm.hip); gradients are not computed.

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "../../include/picopose_hip.h"
#include "pp_common.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int KP_GRID = 64, KP_N = KP_GRID * KP_GRID;   // utils/keypoints.py:97-111: 224 / 3.5 points per side
constexpr float KP_CELL = 3.5f;

struct Mat3 {
    float m[9];
};
struct Mat4 {
    float m[16];
};

// y = T (3x3) [x, y, 1]: the k-ordered fma chain of a GEMM row (first product rounded alone)
__device__ __forceinline__ void mul3(const float* T, float x, float y, float z, float& o0, float& o1, float& o2) {
    o0 = fmaf(T[2], z, fmaf(T[1], y, T[0] * x));
    o1 = fmaf(T[5], z, fmaf(T[4], y, T[3] * x));
    o2 = fmaf(T[8], z, fmaf(T[7], y, T[6] * x));
}

// Keypoint.mask (keypoints.py:47-68): truncate to integer pixels; outside the image or mask < 0.5 -> (-1, -1)
__device__ __forceinline__ bool mask_point(float fx, float fy, const float* mask, int mh, int mw, int& px, int& py) {
    px = (int)fx;   // (.long(): truncation toward zero)
    py = (int)fy;
    bool out = px < 0 || py < 0 || px >= mw || py >= mh;
    if (!out) out = mask[(size_t)py * mw + px] < 0.5f;
    if (out) px = py = -1;
    return !out;
}

// Keypoint.apply_affine (keypoints.py:85-92) on a point that may be the (-1, -1) marker
__device__ __forceinline__ void affine_point(const float* T, float x, float y, float& ox, float& oy) {
    if (x == -1.f) {
        ox = oy = -1.f;
        return;
    }
    float a, b, c;
    mul3(T, x, y, 1.f, a, b, c);
    ox = a / c;
    oy = b / c;
}

// One direction of KeyPointSampler.sample_pts (keypoints.py:138-175) for one grid point: crop pixel (int) -> image pixel
// -> unproject with the depth image -> rigid motion -> project into the other camera -> its crop -> its mask.
//   img_xy: the image-pixel point after unproject_points' in-place clamp (torch_utils.py:143-145)
__device__ __forceinline__ void reproject(int cx, int cy, const float* Minv, const float* Kinv, const float* depth, int dh, int dw,
                                          const float* T, const float* K2, const float* M2, const float* mask2, int mh, int mw,
                                          float& img_x, float& img_y, int& rx, int& ry) {
    affine_point(Minv, (float)cx, (float)cy, img_x, img_y);
    img_y = fminf(fmaxf(img_y, 0.f), (float)(dh - 1));
    img_x = fminf(fmaxf(img_x, 0.f), (float)(dw - 1));
    const float z = depth[(size_t)(int)img_y * dw + (int)img_x];
    float r0, r1, r2;
    mul3(Kinv, img_x, img_y, 1.f, r0, r1, r2);
    r0 *= z;
    r1 *= z;
    r2 *= z;
    // apply_3D_transform: T (4x4) [p, 1], rows 0..2
    const float q0 = fmaf(T[3], 1.f, fmaf(T[2], r2, fmaf(T[1], r1, T[0] * r0)));
    const float q1 = fmaf(T[7], 1.f, fmaf(T[6], r2, fmaf(T[5], r1, T[4] * r0)));
    const float q2 = fmaf(T[11], 1.f, fmaf(T[10], r2, fmaf(T[9], r1, T[8] * r0)));
    float u, v, w;
    mul3(K2, q0, q1, q2, u, v, w);
    float ax, ay;
    affine_point(M2, u / w, v / w, ax, ay);
    mask_point(ax, ay, mask2, mh, mw, rx, ry);
}

// Per grid point: both directions.  Outputs: src_crop / re_src (int2), tar_img (float2, clamped), bad_src / bad_tar flags.
__global__ __launch_bounds__(256) void keypoint_points_kernel(
    const float* __restrict__ src_mask, const float* __restrict__ tar_mask, int mh, int mw, const float* __restrict__ src_depth,
    const float* __restrict__ tar_depth, int dh, int dw, const float* __restrict__ src_Minv, const float* __restrict__ tar_Minv,
    const float* __restrict__ src_M, const float* __restrict__ tar_M, const float* __restrict__ src_Kinv,
    const float* __restrict__ tar_Kinv, const float* __restrict__ src_K, const float* __restrict__ tar_K,
    const float* __restrict__ T_s2t, const float* __restrict__ T_t2s, int B, int2* __restrict__ src_crop, int2* __restrict__ re_src,
    float2* __restrict__ tar_img, unsigned char* __restrict__ bad_src, unsigned char* __restrict__ bad_tar) {
    const int b = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x;
    if (n >= KP_N) return;
    // grid point n = (g[n / 64], g[n % 64]) with column 0 USED as x (keypoints.py:100-111, :47-68)
    const float gx = (float)(n / KP_GRID) * KP_CELL + KP_CELL / 2, gy = (float)(n % KP_GRID) * KP_CELL + KP_CELL / 2;
    const float* sm = src_mask + (size_t)b * mh * mw;
    const float* tm = tar_mask + (size_t)b * mh * mw;
    int sx, sy, tx, ty;
    const bool s_ok = mask_point(gx, gy, sm, mh, mw, sx, sy);
    const bool t_ok = mask_point(gx, gy, tm, mh, mw, tx, ty);
    float six, siy, tix, tiy;
    int rsx, rsy, rtx, rty;
    reproject(sx, sy, src_Minv + b * 9, src_Kinv + b * 9, src_depth + (size_t)b * dh * dw, dh, dw, T_s2t + b * 16, tar_K + b * 9,
              tar_M + b * 9, tm, mh, mw, six, siy, rsx, rsy);
    reproject(tx, ty, tar_Minv + b * 9, tar_Kinv + b * 9, tar_depth + (size_t)b * dh * dw, dh, dw, T_t2s + b * 16, src_K + b * 9,
              src_M + b * 9, sm, mh, mw, tix, tiy, rtx, rty);
    const size_t o = (size_t)b * KP_N + n;
    src_crop[o] = make_int2(sx, sy);
    re_src[o] = make_int2(rsx, rsy);
    tar_img[o] = make_float2(tix, tiy);
    bad_src[o] = !s_ok || rsx == -1;
    bad_tar[o] = !t_ok || rtx == -1;
}

// keypoints.py:177-190: a template point stays if some valid real grid point (IMAGE pixels) lies within 1000 px of its
// re-projection (CROP pixels) — evaluated literally.  Then convert_to_patch_coordinates (:113-117).
__global__ __launch_bounds__(256) void keypoint_visibility_kernel(const int2* __restrict__ src_crop, const int2* __restrict__ re_src,
                                                                   const float2* __restrict__ tar_img,
                                                                   const unsigned char* __restrict__ bad_src,
                                                                   const unsigned char* __restrict__ bad_tar,
                                                                   float* __restrict__ src_pts, float* __restrict__ tar_pts) {
    __shared__ float2 tp[KP_N];
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    for (int j = threadIdx.x; j < KP_N; j += 256) {
        const size_t o = (size_t)b * KP_N + j;
        tp[j] = bad_tar[o] ? make_float2(INFINITY, INFINITY) : tar_img[o];
    }
    __syncthreads();
    const size_t o = (size_t)b * KP_N + i;
    int2 sc = src_crop[o], rs = re_src[o];
    bool keep = false;
    if (!bad_src[o]) {
        const float x = (float)rs.x, y = (float)rs.y;
        float best = INFINITY;
        for (int j = 0; j < KP_N; ++j) {
            const float dx = x - tp[j].x, dy = y - tp[j].y;
            best = fminf(best, fmaf(dx, dx, dy * dy));   // (inf for masked columns)
        }
        keep = sqrtf(best) < 1000.0f;
    }
    if (!keep) sc = rs = make_int2(-1, -1);
    src_pts[2 * o] = sc.x == -1 ? -1.f : (float)sc.x / KP_CELL;
    src_pts[2 * o + 1] = sc.x == -1 ? -1.f : (float)sc.y / KP_CELL;
    tar_pts[2 * o] = rs.x == -1 ? -1.f : (float)rs.x / KP_CELL;
    tar_pts[2 * o + 1] = rs.x == -1 ? -1.f : (float)rs.y / KP_CELL;
}

// ---------------------------------------------------------------------------
// nn.BatchNorm2d in training mode on an NHWC map viewed as (rows, C): statistics of the batch (biased variance for the
// normalisation, unbiased for the running buffer), momentum update of the running buffers, then y = x * scale + shift.
// Sums are carried in double: E[x^2] - E[x]^2 is then exact to fp32 for any mean / spread a layer produces.
// ---------------------------------------------------------------------------
constexpr int BN_ROWS = 256;   // rows per statistics workgroup

// a workgroup = BN_ROWS rows; a thread owns 4 consecutive channels (16-byte loads) and one of 4 row lanes (rows rl, rl + 4, ...), the
// row lanes are added in lane order through LDS (fixed order).  C % 4 == 0.
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ x, int rows, int C, double* __restrict__ part) {
    __shared__ double red[4][64][8];
    const int r0 = blockIdx.x * BN_ROWS, r1 = min(rows, r0 + BN_ROWS);
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

__global__ __launch_bounds__(256) void bn_finish_kernel(const double* __restrict__ part, int nblk, int rows, int C,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                        float momentum, float* __restrict__ running_mean,
                                                        float* __restrict__ running_var, float* __restrict__ scale,
                                                        float* __restrict__ shift) {
    int c;
    double s, q;
    if (!bn_fold16(part, nblk, C, c, s, q)) return;
    const double mean = s / rows;
    double var = q / rows - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const float sc = gamma[c] * (float)(1.0 / sqrt(var + (double)eps));
    scale[c] = sc;
    shift[c] = beta[c] - (float)mean * sc;
    if (running_mean) {
        const double unbiased = rows > 1 ? var * ((double)rows / (rows - 1)) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

// y = relu?(x * scale + shift) + r1 + r2   (rows x C, C % 4 == 0)
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const float* __restrict__ r1,
                                                       const float* __restrict__ r2, long long n4, int C, int relu,
                                                       float* __restrict__ y) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int c = (int)((i * 4) % C);
    f4 v = *(const f4*)(x + i * 4);
    const f4 sc = *(const f4*)(scale + c), sh = *(const f4*)(shift + c);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        v[k] = fmaf(v[k], sc[k], sh[k]);
        if (relu) v[k] = fmaxf(v[k], 0.f);
    }
    if (r1) v += *(const f4*)(r1 + i * 4);
    if (r2) v += *(const f4*)(r2 + i * 4);
    *(f4*)(y + i * 4) = v;
}

// ---------------------------------------------------------------------------
// InfoNCE pieces (utils/loss_utils.py:163-175): gather + F.normalize of feature rows; per-row cross entropy with the
// diagonal as the label.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_normalize_kernel(const float* __restrict__ src, long long row_stride,
                                                               const int64_t* __restrict__ index, int n, int C, float eps,
                                                               float* __restrict__ out) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n) return;
    const float* s = src + index[row] * row_stride;
    float q = 0.f;
    for (int c = lane; c < C; c += 64) q = fmaf(s[c], s[c], q);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float inv = 1.0f / fmaxf(sqrtf(q), eps);
    for (int c = lane; c < C; c += 64) out[(size_t)row * C + c] = s[c] * inv;
}

// loss[i] = logsumexp_j(scale * L[i, j]) - scale * L[i, i]: one wave per row
__global__ __launch_bounds__(256) void xent_diag_kernel(const float* __restrict__ L, int n, int ld, float scale,
                                                        float* __restrict__ loss) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n) return;
    const float* r = L + (size_t)row * ld;
    float mx = -INFINITY;
    for (int j = lane; j < n; j += 64) mx = fmaxf(mx, r[j] * scale);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float s = 0.f;
    for (int j = lane; j < n; j += 64) s += expf(r[j] * scale - mx);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) loss[row] = mx + logf(s) - r[row] * scale;
}

// ---------------------------------------------------------------------------
// compute_stage_three_loss for one level (utils/loss_utils.py:188-202, compute_flow_loss :119-125, RAFTLoss :24-39):
// per workgroup the three sums  [ BCE-with-logits terms, valid-weighted |flow - gt|, number of valid pixels ].
//   gt at map pixel (y, x) = key-point n = ix * 64 + iy, (iy, ix) = nearest source cell of (y, x)   ('b (h w) c -> b w h c')
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void flow_loss_kernel(const float* __restrict__ flow, const float* __restrict__ cert,
                                                        const float* __restrict__ tar_pts, int B, int H, int W, float max_flow,
                                                        double* __restrict__ part) {
    __shared__ double red[3][4];
    const long long total = (long long)B * H * W;
    double bce = 0.0, l1 = 0.0, cnt = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % W), y = (int)((i / W) % H), b = (int)(i / ((long long)W * H));
        const int iy = min((int)floorf((float)y * ((float)KP_GRID / H)), KP_GRID - 1);
        const int ix = min((int)floorf((float)x * ((float)KP_GRID / W)), KP_GRID - 1);
        const float* p = tar_pts + ((size_t)b * KP_N + (size_t)ix * KP_GRID + iy) * 2;
        const bool valid = p[0] != -1.f && p[1] != -1.f;
        const float k = (float)H / KP_GRID;
        const float g0 = (valid ? k * p[0] : 0.f) - (float)x, g1 = (valid ? k * p[1] : 0.f) - (float)y;
        const float z = cert[i], t = valid ? 1.f : 0.f;
        bce += (double)(fmaxf(z, 0.f) - z * t + log1pf(expf(-fabsf(z))));
        if (valid && sqrtf(g0 * g0 + g1 * g1) < max_flow) {
            l1 += (double)(fabsf(flow[2 * i] - g0) + fabsf(flow[2 * i + 1] - g1));
            cnt += 1.0;
        }
    }
    double v[3] = {bce, l1, cnt};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < 3) part[(size_t)blockIdx.x * 3 + threadIdx.x] = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}

}  // namespace

extern "C" {

size_t pp_train_keypoints_workspace_bytes(int B) {
    return B <= 0 ? 0 : (size_t)B * KP_N * (2 * sizeof(int2) + sizeof(float2) + 2) + 256;
}

int pp_train_keypoints(const float* src_mask, const float* tar_mask, int mask_h, int mask_w, const float* src_depth,
                       const float* tar_depth, int depth_h, int depth_w, const float* src_Minv, const float* tar_Minv,
                       const float* src_M, const float* tar_M, const float* src_Kinv, const float* tar_Kinv, const float* src_K,
                       const float* tar_K, const float* T_src2tar, const float* T_tar2src, int B, float* src_pts, float* tar_pts,
                       void* workspace, size_t workspace_bytes, void* stream) {
    if (!src_mask || !tar_mask || !src_depth || !tar_depth || !src_Minv || !tar_Minv || !src_M || !tar_M || !src_Kinv || !tar_Kinv ||
        !src_K || !tar_K || !T_src2tar || !T_tar2src || !src_pts || !tar_pts || !workspace)
        return PP_EINVAL;
    if (B <= 0 || B > 65535 || mask_h <= 0 || mask_w <= 0 || depth_h <= 0 || depth_w <= 0) return PP_EINVAL;
    if (workspace_bytes < pp_train_keypoints_workspace_bytes(B)) return PP_EINVAL;
    const size_t n = (size_t)B * KP_N;
    int2* src_crop = (int2*)workspace;
    int2* re_src = src_crop + n;
    float2* tar_img = (float2*)(re_src + n);
    unsigned char* bad_src = (unsigned char*)(tar_img + n);
    unsigned char* bad_tar = bad_src + n;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(keypoint_points_kernel, dim3(KP_N / 256, B), dim3(256), 0, st, src_mask, tar_mask, mask_h, mask_w, src_depth,
                       tar_depth, depth_h, depth_w, src_Minv, tar_Minv, src_M, tar_M, src_Kinv, tar_Kinv, src_K, tar_K, T_src2tar,
                       T_tar2src, B, src_crop, re_src, tar_img, bad_src, bad_tar);
    hipLaunchKernelGGL(keypoint_visibility_kernel, dim3(KP_N / 256, B), dim3(256), 0, st, src_crop, re_src, tar_img, bad_src, bad_tar,
                       src_pts, tar_pts);
    return pp_last_launch();
}

size_t pp_batchnorm_train_workspace_bytes(int rows, int C) {
    if (rows <= 0 || C <= 0) return 0;
    return (size_t)((rows + BN_ROWS - 1) / BN_ROWS) * C * 2 * sizeof(double) + (size_t)2 * C * sizeof(float) + 256;
}

int pp_batchnorm_train(const float* x, const float* gamma, const float* beta, int rows, int C, float eps, float momentum,
                       float* running_mean, float* running_var, int relu, const float* residual, const float* residual2, float* y,
                       void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !gamma || !beta || !y || !workspace || rows <= 0 || C <= 0 || C % 4 != 0) return PP_EINVAL;
    if ((running_mean == nullptr) != (running_var == nullptr)) return PP_EINVAL;
    if (workspace_bytes < pp_batchnorm_train_workspace_bytes(rows, C)) return PP_EINVAL;
    if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)residual | (uintptr_t)residual2) % 16 != 0) return PP_EINVAL;
    const int nblk = (rows + BN_ROWS - 1) / BN_ROWS;
    double* part = (double*)workspace;
    float* scale = (float*)(part + (size_t)nblk * C * 2);
    float* shift = scale + C;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_partial_kernel, dim3(nblk), dim3(256), 0, st, x, rows, C, part);
    hipLaunchKernelGGL(bn_finish_kernel, dim3((C + 15) / 16), dim3(256), 0, st, part, nblk, rows, C, gamma, beta, eps, momentum,
                       running_mean, running_var, scale, shift);
    const long long n4 = (long long)rows * C / 4;
    hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, x, scale, shift, residual, residual2, n4, C,
                       relu, y);
    return pp_last_launch();
}

int pp_gather_normalize_rows(const float* src, long long row_stride, const int64_t* index, int n, int C, float eps, float* out,
                             void* stream) {
    if (!src || !index || !out || n <= 0 || C <= 0 || row_stride < C) return PP_EINVAL;
    hipLaunchKernelGGL(gather_normalize_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, src, row_stride, index, n, C, eps,
                       out);
    return pp_last_launch();
}

int pp_xent_diag_rows(const float* logits, int n, int ld, float scale, float* row_loss, void* stream) {
    if (!logits || !row_loss || n <= 0 || ld < n) return PP_EINVAL;
    hipLaunchKernelGGL(xent_diag_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, n, ld, scale, row_loss);
    return pp_last_launch();
}

int pp_flow_loss_blocks(void) { return 256; }

int pp_flow_loss_sums(const float* flow, const float* certainty, const float* tar_pts, int B, int H, int W, float max_flow,
                      double* partial_sums, void* stream) {
    if (!flow || !certainty || !tar_pts || !partial_sums || B <= 0 || H <= 0 || W <= 0) return PP_EINVAL;
    hipLaunchKernelGGL(flow_loss_kernel, dim3(pp_flow_loss_blocks()), dim3(256), 0, (hipStream_t)stream, flow, certainty, tar_pts, B, H, W,
                       max_flow, partial_sums);
    return pp_last_launch();
}

}  // extern "C"
