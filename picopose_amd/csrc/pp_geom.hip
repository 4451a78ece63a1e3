// Small kernels of stages 2/3 around the networks: the stage-2 similarity volume, the affine /
// pose closed forms, the initial correspondences, keypoint selection and the PnP gather.
// Each entry point replaces one reference function (cited per kernel); all are one launch, no
// host synchronisation (the reference's `.all()` asserts and `nonzero` syncs are gone).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/picopose_hip.h"
#include "pp_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));

// nearest resample index of F.interpolate(size=16): src = min(floor(dst * in/16), in-1)
__device__ __forceinline__ int nearest16(int dst, int in) {
    int i = (int)floorf((float)dst * ((float)in / 16.0f));
    return i < in - 1 ? i : in - 1;
}

// ---------------------------------------------------------------------------
// matching_features_similarity — utils/matching.py:6-26.
//   out[b,s,h,w] = relu( <q_hat[:,t], x_hat[:,s]> * mask_s ),  t = w*16 + h
// One workgroup = (b, 128 query patches, 128 template patches); 4 waves (2x2), 64x64 fp32
// tile per wave on v_mfma_f32_32x32x2_f32 (exact fp32, c-ordered fma chain).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void simvol_kernel(const float* __restrict__ src,
                                                     const float* __restrict__ tar,
                                                     const float* __restrict__ src_mask, int mh,
                                                     int mw, int C, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float Qs[16 * 128];
    __shared__ __attribute__((aligned(16))) float Xs[16 * 128];
    __shared__ float rq[128], rx[128];
    const int b = blockIdx.x, t0 = blockIdx.y * 128, s0 = blockIdx.z * 128;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wr = w >> 1, wc = w & 1, l31 = lane & 31, lh = lane >> 5;
    const float* qg = tar + (size_t)b * C * 256 + t0;
    const float* xg = src + (size_t)b * C * 256 + s0;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float sa[2] = {0.f, 0.f}, sb[2] = {0.f, 0.f};
    for (int c0 = 0; c0 < C; c0 += 16) {
        // 16 rows x 128 floats per operand = 512 float4: two per thread
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int idx = tid + 256 * j, row = idx >> 5, col = (idx & 31) * 4;
            *(f4*)(Qs + row * 128 + col) = *(const f4*)(qg + (size_t)(c0 + row) * 256 + col);
            *(f4*)(Xs + row * 128 + col) = *(const f4*)(xg + (size_t)(c0 + row) * 256 + col);
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            float a[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = Qs[(2 * p + lh) * 128 + wr * 64 + i * 32 + l31];
                bv[i] = Xs[(2 * p + lh) * 128 + wc * 64 + i * 32 + l31];
                sa[i] = fmaf(a[i], a[i], sa[i]);
                sb[i] = fmaf(bv[i], bv[i], sb[i]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bv[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    // norms: lanes l and l+32 hold the even / odd channel halves
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        sa[i] += __shfl_xor(sa[i], 32);
        sb[i] += __shfl_xor(sb[i], 32);
    }
    if (lh == 0) {
        if (wc == 0) {
            rq[wr * 64 + l31] = 1.0f / fmaxf(sqrtf(sa[0]), 1e-12f);
            rq[wr * 64 + 32 + l31] = 1.0f / fmaxf(sqrtf(sa[1]), 1e-12f);
        }
        if (wr == 0) {
            // template patch mask: nearest 16x16 resample (matching.py:16-17), folded into 1/||x||
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int s = s0 + wc * 64 + i * 32 + l31;
                const float m = src_mask[((size_t)b * mh + nearest16(s >> 4, mh)) * mw + nearest16(s & 15, mw)];
                rx[wc * 64 + i * 32 + l31] = m / fmaxf(sqrtf(sb[i]), 1e-12f);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int sl = wc * 64 + j * 32 + l31;
        const float cs = rx[sl];
        float* o = out + ((size_t)b * 256 + s0 + sl) * 256;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int tl = wr * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                const int t = t0 + tl;
                float v = acc[i][j][e] * rq[tl] * cs;
                v = v < 0.f ? 0.f : v;
                o[(t & 15) * 16 + (t >> 4)] = v;  // [h][w], t = w*16 + h
            }
    }
}

// ---------------------------------------------------------------------------
// 3x3 helpers (row-major)
// ---------------------------------------------------------------------------
struct M3 {
    float m[9];
};
__device__ __forceinline__ M3 ld3(const float* p) {
    M3 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.m[i] = p[i];
    return r;
}
__device__ __forceinline__ M3 mul3(const M3& a, const M3& b) {
    M3 r;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            r.m[i * 3 + j] = a.m[i * 3] * b.m[j] + a.m[i * 3 + 1] * b.m[3 + j] + a.m[i * 3 + 2] * b.m[6 + j];
    return r;
}
__device__ __forceinline__ void mulv3(const M3& a, const float* v, float* o) {
#pragma unroll
    for (int i = 0; i < 3; ++i) o[i] = a.m[i * 3] * v[0] + a.m[i * 3 + 1] * v[1] + a.m[i * 3 + 2] * v[2];
}

// calc_pred_Ms — utils/torch_utils.py:39-51 (affine_torch :53-73, apply_affine :114-135)
__global__ void pred_ms_kernel(const float* __restrict__ scale, const float* __restrict__ inplane,
                               const float* __restrict__ trans, const float* __restrict__ tem_pose,
                               const float* __restrict__ tem_K, const float* __restrict__ tem_M, int B,
                               float trans_scale, float* __restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float* P = tem_pose + (size_t)b * 16;
    const float t[3] = {P[3], P[7], P[11]};
    float c[3], cm[3];
    mulv3(ld3(tem_K + (size_t)b * 9), t, c);
    c[0] /= c[2];
    c[1] /= c[2];
    c[2] /= c[2];
    mulv3(ld3(tem_M + (size_t)b * 9), c, cm);
    const float s = scale[b], co = inplane[2 * b], si = inplane[2 * b + 1];
    const float m00 = co * s, m01 = -si * s, m10 = si * s, m11 = co * s;
    // apply the translation-free affine to the centre (homogeneous w = 1)
    const float wv = 0.f * cm[0] + 0.f * cm[1] + 1.f;
    const float ax = (m00 * cm[0] + m01 * cm[1] + 0.f) / wv;
    const float ay = (m10 * cm[0] + m11 * cm[1] + 0.f) / wv;
    const float tx = cm[0] + trans[2 * b] * trans_scale;
    const float ty = cm[1] + trans[2 * b + 1] * trans_scale;
    float* o = out + (size_t)b * 9;
    o[0] = m00; o[1] = m01; o[2] = tx - ax;
    o[3] = m10; o[4] = m11; o[5] = ty - ay;
    o[6] = 0.f; o[7] = 0.f; o[8] = 1.f;
}

// pose_recovery_2d_prediction — utils/pose_recovery.py:9-65
// (normalize_affine_transform torch_utils.py:228-240, inverse_affine :93-111; the reference's
// host-synchronising asserts on query_M are the caller's contract here)
__global__ void pose2d_kernel(const float* __restrict__ query_M, const float* __restrict__ query_K,
                              const float* __restrict__ pred_Ms, const float* __restrict__ tem_K,
                              const float* __restrict__ tem_M, const float* __restrict__ tem_pose, int B,
                              float* __restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float* P = tem_pose + (size_t)b * 16;
    const M3 pm = ld3(pred_Ms + (size_t)b * 9), qM = ld3(query_M + (size_t)b * 9);
    const M3 qK = ld3(query_K + (size_t)b * 9), tK = ld3(tem_K + (size_t)b * 9), tM = ld3(tem_M + (size_t)b * 9);
    // Step 1: R = R_inplane(normalised 2x2 of pred_Ms, embedded) @ R_template
    const float sc = sqrtf(pm.m[0] * pm.m[0] + pm.m[3] * pm.m[3]);
    M3 rin = {{pm.m[0] / sc, pm.m[1] / sc, 0.f, pm.m[3] / sc, pm.m[4] / sc, 0.f, 0.f, 0.f, 1.f}};
    M3 rt = {{P[0], P[1], P[2], P[4], P[5], P[6], P[8], P[9], P[10]}};
    const M3 R = mul3(rin, rt);
    // Step 2: projected template centre, moved by the full template->query 2-D affine
    const float t[3] = {P[3], P[7], P[11]};
    float c[3];
    mulv3(tK, t, c);
    const float cz = c[2];
    c[0] /= cz; c[1] /= cz; c[2] /= cz;
    const float s = qM.m[0];
    M3 inv = {{1.f / s, 0.f, -qM.m[2] / s, 0.f, 1.f / s, -qM.m[5] / s, 0.f, 0.f, 1.f}};
    const M3 aff = mul3(mul3(inv, pm), tM);
    float qc[3];
    mulv3(aff, c, qc);
    // inverse of query_K (adjugate / determinant)
    const float* k = qK.m;
    const float det = k[0] * (k[4] * k[8] - k[5] * k[7]) - k[1] * (k[3] * k[8] - k[5] * k[6]) +
                      k[2] * (k[3] * k[7] - k[4] * k[6]);
    const float id = 1.0f / det;
    M3 ik = {{(k[4] * k[8] - k[5] * k[7]) * id, (k[2] * k[7] - k[1] * k[8]) * id, (k[1] * k[5] - k[2] * k[4]) * id,
              (k[5] * k[6] - k[3] * k[8]) * id, (k[0] * k[8] - k[2] * k[6]) * id, (k[2] * k[3] - k[0] * k[5]) * id,
              (k[3] * k[7] - k[4] * k[6]) * id, (k[1] * k[6] - k[0] * k[7]) * id, (k[0] * k[4] - k[1] * k[3]) * id}};
    const float scale2d = sqrtf(aff.m[0] * aff.m[0] + aff.m[3] * aff.m[3]);
    const float focal = qK.m[0] / tK.m[0];
    const float qz = (t[2] / scale2d) * focal;
    float qt[3];
    mulv3(ik, qc, qt);
    const float z = qt[2];
    float* o = out + (size_t)b * 16;
    o[0] = R.m[0]; o[1] = R.m[1]; o[2] = R.m[2]; o[3] = (qt[0] / z) * qz;
    o[4] = R.m[3]; o[5] = R.m[4]; o[6] = R.m[5]; o[7] = (qt[1] / z) * qz;
    o[8] = R.m[6]; o[9] = R.m[7]; o[10] = R.m[8]; o[11] = (qt[2] / z) * qz;
    o[12] = P[12]; o[13] = P[13]; o[14] = P[14]; o[15] = P[15];
}

// compute_init_correspondences — utils/correspondence.py:10-26 (16x16 grid of patch centres
// 7,21,..; init_points2d_torch torch_utils.py:297-305; "b (w h) c -> b c h w")
__global__ __launch_bounds__(256) void init_corr_kernel(const float* __restrict__ pred_Ms,
                                                        const float* __restrict__ tem_mask, int mh,
                                                        int mw, float* __restrict__ flow,
                                                        float* __restrict__ cert) {
    const int b = blockIdx.x, tid = threadIdx.x, h = tid >> 4, w = tid & 15;
    const float patch = (float)(mh / 16);
    const float m = tem_mask[((size_t)b * mh + nearest16(h, mh)) * mw + nearest16(w, mw)];
    const float* M = pred_Ms + (size_t)b * 9;
    // point k = w*16 + h is (c[w], c[h]) with c[i] = i*patch + patch/2
    const float px = (float)w * patch + patch * 0.5f, py = (float)h * patch + patch * 0.5f;
    const float x = M[0] * px + M[1] * py + M[2];
    const float y = M[3] * px + M[4] * py + M[5];
    const float ww = M[6] * px + M[7] * py + M[8];
    const float fx = (x / ww) / patch, fy = (y / ww) / patch;
    flow[((size_t)b * 2 + 0) * 256 + tid] = fx * m - (float)w;
    flow[((size_t)b * 2 + 1) * 256 + tid] = fy * m - (float)h;
    cert[(size_t)b * 256 + tid] = m;
}

// compute_stage3_correspondences — utils/correspondence.py:28-59.  Output entry k = w*H + h
// holds (x=w, y=h) / trunc(flow + grid) or (-1,-1); dense, so no compaction and no host sync.
__global__ void stage3_corr_kernel(const float* __restrict__ flow, const float* __restrict__ cert, int H,
                                   int W, float thr, int64_t* __restrict__ tar_pts,
                                   int64_t* __restrict__ src_pts) {
    const int b = blockIdx.y;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;  // k = w*H + h
    if (k >= H * W) return;
    const int w = k / H, h = k - w * H;
    const float tx = flow[((size_t)b * 2 + 0) * H * W + h * W + w] + (float)w;
    const float ty = flow[((size_t)b * 2 + 1) * H * W + h * W + w] + (float)h;
    const float c = cert[(size_t)b * H * W + h * W + w];
    const float sg = 1.0f / (1.0f + expf(-c));
    const bool keep = sg > thr && tx > 0.f && ty > 0.f && tx < (float)(H - 1) && ty < (float)(W - 1);
    int64_t* tp = tar_pts + ((size_t)b * H * W + k) * 2;
    int64_t* sp = src_pts + ((size_t)b * H * W + k) * 2;
    tp[0] = keep ? (int64_t)tx : -1;
    tp[1] = keep ? (int64_t)ty : -1;
    sp[0] = keep ? w : -1;
    sp[1] = keep ? h : -1;
}

// gather — utils/torch_utils.py:257-284 as used at utils/pose_recovery.py:76-77: rows
// feat[:, y*W + x] of the entries != -1, order preserved.  One workgroup per batch item,
// block-wide exclusive scan of the validity flags.
__global__ __launch_bounds__(1024) void gather_valid_kernel(const float* __restrict__ feat,
                                                            const int64_t* __restrict__ idx, int C, int H,
                                                            int W, int N, float* __restrict__ out,
                                                            int32_t* __restrict__ count) {
    __shared__ int wsum[16];
    __shared__ int base;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t* ib = idx + (size_t)b * N * 2;
    const float* fb = feat + (size_t)b * C * H * W;
    float* ob = out + (size_t)b * N * C;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int n0 = 0; n0 < N; n0 += 1024) {
        const int n = n0 + tid;
        int64_t x = -1, y = -1;
        if (n < N) {
            x = ib[2 * n];
            y = ib[2 * n + 1];
        }
        const bool v = x != -1 && y != -1;
        const unsigned long long bal = __ballot(v);
        const int pre = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wv] = __popcll(bal);
        __syncthreads();
        int off = base;
        for (int i = 0; i < wv; ++i) off += wsum[i];
        if (v) {
            const int r = off + pre;
            const int64_t p = y * W + x;
            for (int c = 0; c < C; ++c) ob[(size_t)r * C + c] = fb[(size_t)c * H * W + p];
        }
        __syncthreads();
        if (tid == 0) {
            int t = base;
            for (int i = 0; i < 16; ++i) t += wsum[i];
            base = t;
        }
        __syncthreads();
    }
    if (tid == 0) count[b] = base;
}

}  // namespace

extern "C" {

int pp_similarity_volume(const float* src_feat, const float* tar_feat, const float* src_mask, int mask_h,
                         int mask_w, int B, int C, float* out, void* stream) {
    if (!src_feat || !tar_feat || !src_mask || !out) return PP_EINVAL;
    if (B <= 0 || C <= 0 || C % 16 != 0 || mask_h <= 0 || mask_w <= 0) return PP_EINVAL;
    hipLaunchKernelGGL(simvol_kernel, dim3(B, 2, 2), dim3(256), 0, (hipStream_t)stream, src_feat, tar_feat,
                       src_mask, mask_h, mask_w, C, out);
    return pp_last_launch();
}

int pp_calc_pred_Ms(const float* pred_scale, const float* pred_inplane, const float* pred_translation,
                    const float* tem_pose, const float* tem_K, const float* tem_M, int B, float trans_scale,
                    float* pred_Ms, void* stream) {
    if (!pred_scale || !pred_inplane || !pred_translation || !tem_pose || !tem_K || !tem_M || !pred_Ms || B <= 0)
        return PP_EINVAL;
    hipLaunchKernelGGL(pred_ms_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, pred_scale,
                       pred_inplane, pred_translation, tem_pose, tem_K, tem_M, B, trans_scale, pred_Ms);
    return pp_last_launch();
}

int pp_pose_recovery_2d(const float* query_M, const float* query_K, const float* pred_Ms, const float* tem_K,
                        const float* tem_M, const float* tem_pose, int B, float* pred_pose, void* stream) {
    if (!query_M || !query_K || !pred_Ms || !tem_K || !tem_M || !tem_pose || !pred_pose || B <= 0)
        return PP_EINVAL;
    hipLaunchKernelGGL(pose2d_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, query_M, query_K,
                       pred_Ms, tem_K, tem_M, tem_pose, B, pred_pose);
    return pp_last_launch();
}

int pp_init_correspondences(const float* pred_Ms, const float* tem_mask, int mask_h, int mask_w, int B,
                            float* init_flow, float* init_certainty, void* stream) {
    if (!pred_Ms || !tem_mask || !init_flow || !init_certainty || B <= 0) return PP_EINVAL;
    if (mask_h != mask_w || mask_h < 16) return PP_EINVAL;
    hipLaunchKernelGGL(init_corr_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, pred_Ms, tem_mask,
                       mask_h, mask_w, init_flow, init_certainty);
    return pp_last_launch();
}

int pp_stage3_correspondences(const float* pred_flow, const float* pred_certainty, int B, int H, int W,
                              float threshold, int64_t* tar_pts, int64_t* src_pts, void* stream) {
    if (!pred_flow || !pred_certainty || !tar_pts || !src_pts || B <= 0 || H <= 0 || W <= 0) return PP_EINVAL;
    hipLaunchKernelGGL(stage3_corr_kernel, dim3((H * W + 255) / 256, B), dim3(256), 0, (hipStream_t)stream,
                       pred_flow, pred_certainty, H, W, threshold, tar_pts, src_pts);
    return pp_last_launch();
}

int pp_gather_valid(const float* features, const int64_t* index_patches, int B, int C, int H, int W, int N,
                    float* out, int32_t* count, void* stream) {
    if (!features || !index_patches || !out || !count || B <= 0 || C <= 0 || H <= 0 || W <= 0 || N <= 0)
        return PP_EINVAL;
    hipLaunchKernelGGL(gather_valid_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, features,
                       index_patches, C, H, W, N, out, count);
    return pp_last_launch();
}

}  // extern "C"
