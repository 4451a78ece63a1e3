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
#include "../../include/picopose_hip.h"
#include "pp_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int HD = 64, KC = 32, KLD = 68;  // head dim, keys per chunk, floats per LDS row of the K chunk

__global__ __launch_bounds__(256) void attn_kernel(const float* __restrict__ qkv, int T, int heads, float scale,
                                                   float* __restrict__ out, _Float16* __restrict__ out_hi,
                                                   _Float16* __restrict__ out_lo) {
    __shared__ __attribute__((aligned(16))) float Ks[KC * KLD];
    __shared__ __attribute__((aligned(16))) float Vs[KC * HD];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int b = blockIdx.y / heads, h = blockIdx.y % heads;
    const int C3 = 3 * heads * HD;
    const float* base = qkv + (size_t)b * T * C3 + h * HD;  // q at +0, k at +heads*HD, v at +2*heads*HD, token stride C3
    const int q = blockIdx.x * 128 + w * 32 + l31;           // this lane's query row
    const int qc = q < T ? q : T - 1;

    // Q as the B operand of S^T = K Q^T: lane (d part lh, query l31) holds Q[q][32 lh + p] * scale, p = 0..31
    float qr[32];
    {
        const float* qp = base + (size_t)qc * C3 + 32 * lh;
#pragma unroll
        for (int p = 0; p < 32; p += 4) {
            const f4 v = *(const f4*)(qp + p);
            qr[p] = v.x * scale; qr[p + 1] = v.y * scale; qr[p + 2] = v.z * scale; qr[p + 3] = v.w * scale;
        }
    }
    f32x16 o0, o1;  // O^T: rows d (0..31 / 32..63), column = query (lane)
#pragma unroll
    for (int e = 0; e < 16; ++e) o0[e] = o1[e] = 0.f;
    float mrun = -INFINITY, lrun = 0.f;

    const float* kp = base + heads * HD;
    const float* vp = base + 2 * heads * HD;
    for (int k0 = 0; k0 < T; k0 += KC) {
        __syncthreads();  // previous chunk fully consumed
        // stage K and V chunks: 32 keys x 64 floats each = 512 float4: two per thread per tensor
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int idx = tid + 256 * j, row = idx >> 4, c4 = (idx & 15) * 4;
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
        // register e of this lane = key k0 + (e&3) + 8(e>>2) + 4 lh, query l31; mask keys past T
        float mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int key = k0 + (e & 3) + 8 * (e >> 2) + 4 * lh;
            s[e] = key < T ? s[e] : -INFINITY;
            mx = fmaxf(mx, s[e]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float mnew = fmaxf(mrun, mx);
        const float alpha = expf(mrun - mnew);  // 0 for the first chunk (mrun = -inf)
        float ls = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            s[e] = expf(s[e] - mnew);
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
            if (out_hi) {  // f16x3 operand planes of the output projection
                _Float16 hh, ll;
                pp_split_f16(a0, hh, ll);
                out_hi[obase + d] = hh;
                out_lo[obase + d] = ll;
                pp_split_f16(a1, hh, ll);
                out_hi[obase + 32 + d] = hh;
                out_lo[obase + 32 + d] = ll;
            }
        }
    }
}

}  // namespace

extern "C" {

int pp_attention(const float* qkv, int B, int T, int heads, int head_dim, float scale, float* out, void* stream) {
    if (!qkv || !out || B <= 0 || T <= 0 || heads <= 0) return PP_EINVAL;
    if (head_dim != HD || ((uintptr_t)qkv % 16) != 0) return PP_EINVAL;
    hipLaunchKernelGGL(attn_kernel, dim3((T + 127) / 128, B * heads), dim3(256), 0, (hipStream_t)stream, qkv, T, heads,
                       scale, out, (_Float16*)nullptr, (_Float16*)nullptr);
    return pp_last_launch();
}

int pp_attention_split(const float* qkv, int B, int T, int heads, int head_dim, float scale, float* out, void* out_hi,
                       void* out_lo, void* stream) {
    if (!qkv || !out_hi || !out_lo || B <= 0 || T <= 0 || heads <= 0) return PP_EINVAL;
    if (head_dim != HD || ((uintptr_t)qkv % 16) != 0) return PP_EINVAL;
    hipLaunchKernelGGL(attn_kernel, dim3((T + 127) / 128, B * heads), dim3(256), 0, (hipStream_t)stream, qkv, T, heads,
                       scale, out, (_Float16*)out_hi, (_Float16*)out_lo);
    return pp_last_launch();
}

}  // extern "C"
