/*
 * picopose_hip.h — C ABI of libpicopose_hip.so (gfx950 / MI355X).
 *
 * This is the drop-in boundary of the PicoPose correspondence hot path.  The
 * reference (foollh/PicoPose) is pure Python and has no FFI of its own; every
 * entry point below replaces the stock-torch arithmetic of one reference
 * function (cited as file:line relative to the reference tree) and is bound
 * from Python with ctypes (picopose_amd/_lib.py).  INTEGRATION.md shows the
 * binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (tensor.data_ptr()), fp32 unless the
 *     name says otherwise, dense row-major in the layout given per argument;
 *   - the caller owns all buffers (inputs, outputs, workspace); inputs are
 *     never written;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*) and the
 *     call returns without synchronising;
 *   - return value: PP_OK (0) or a negative PP_E* code; nothing throws across
 *     the ABI.  pp_strerror() maps a code to text.
 */
#ifndef PICOPOSE_HIP_H
#define PICOPOSE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PP_OK 0
#define PP_EINVAL (-1)     /* unsupported shape / null pointer / bad mode      */
#define PP_EWORKSPACE (-2) /* workspace too small or misaligned (256 B)        */
#define PP_ELAUNCH (-3)    /* hipGetLastError() != hipSuccess after a launch   */

/* arithmetic used for the 256x256xC similarity contraction of stage 1 */
#define PP_MATCH_EXACT 0 /* v_mfma_f32_32x32x2_f32: bit-for-bit an fp32 fma chain */
#define PP_MATCH_FAST 1  /* v_mfma_f32_32x32x16_f16 + exact fp32 re-evaluation of  \
                            every row/column whose index-0 decision is within eps */

const char* pp_strerror(int code);
int pp_version(void);

/* Measurement hooks (bench.py): after pp_prof_enable(n) every call that launches a
 * roofline kernel (stage 1: the fused similarity kernel) brackets exactly that launch with
 * two hipEvents on the caller's stream, up to n records; pp_prof_collect waits for them and
 * returns the durations in ms.  pp_prof_enable(0) switches the hooks off. */
int pp_prof_enable(int max_records);
int pp_prof_collect(float* out_ms, int max_out, int* count);
/* The same for the contraction engine: after pp_prof_gemm_enable(n) every pp_gemm launch (up to n; the
 * autotuner's trial launches excluded) is bracketed by two hipEvents on its stream and its 2*M*N*K*batch flop
 * count is kept.  pp_prof_gemm_collect sums durations (ms), flops and launches into 2-element arrays:
 * [0] = launches of the pre-split f16x3 kernel (the dominant kernel of the full path), [1] = all other
 * GEMM kernels.  pp_prof_gemm_enable(0) switches the hooks off. */
int pp_prof_gemm_enable(int max_records);
int pp_prof_gemm_collect(double* ms, double* flops, int* launches);
/* Per-launch view of the same records (call BEFORE pp_prof_gemm_collect, which resets them): for record i < *count,
 * shape[6 i ..] = {M, N, K, conv kernel size (0: dense), tile configuration the launch used (PP_GEMM_FORCE_CFG numbering),
 * kind (0 = both operands pre-split, 1 = other)}, ms[i] its duration, flops[i] = 2 M N K batch. */
int pp_prof_gemm_records(int max_records, int* shape, float* ms, double* flops, int* count);
/* The same with 8 ints per record — shape[8 i ..] = {M, N, K, conv kernel size, cfg, kind, A-delivery mode of the pre-split kernel
 * (0 dense, 1 convolution in channel-slice-major K order, 2 natural order), 0} — plus bytes[i] = the launch's ALGORITHMIC bytes: every element of A (a convolution: of its input image, not of the im2col),
 * B, the output(s) and the residual(s) once, in the format the launch reads / writes them (4 B fp32; 4 B per element of a 2-term
 * operand, 2 B of a 1-term one) — what a per-kernel traffic ratio (PMC FETCH + WRITE over this) is taken against. */
int pp_prof_gemm_records2(int max_records, int* shape, float* ms, double* flops, double* bytes, int* count);
/* The contraction engine's autotuner (pp_gemm: per problem shape the fastest tile configuration, measured once per process) as a
 * table that can be written and read back, so that separate processes — the passes of one profiling set, a serving fleet — run the
 * SAME configuration per shape: pp_gemm_tune_save writes "key cfg" lines and returns the entry count, pp_gemm_tune_load merges a
 * file into the table (shapes it lacks are still tuned on first use) and returns the entries read (PP_EINVAL: unreadable);
 * the environment variable PP_GEMM_TUNE_FILE loads a file before the first pp_gemm call.  Results do not depend on the table
 * (every configuration accumulates in the same order), only speed does. */
int pp_gemm_tune_save(const char* path);
int pp_gemm_tune_load(const char* path);
int pp_gemm_tune_entries(void);

/* ------------------------------------------------------------------------- *
 * Stage 1: template matching — utils/matching.py:29-69 (matching_templates)
 * called from model/picopose.py:102-104.
 *
 *   bank   (B,N,C,16,16)  template patch features (un-normalised is fine)
 *   query  (B,C,16,16)    query patch features
 *   mask   (B,mh,mw)      query mask; sampled nearest to 16x16 exactly as
 *                         F.interpolate(mask, size=(16,16)) does
 *   sim_avg (B,N)         out: masked mean of the best-match scores
 *
 * pp_stage1_scores computes sim_avg (matching.py:38-66); pp_topk performs
 * torch.topk(sim_avg, k, dim=1) (matching.py:68) with ties broken toward the
 * lower template id; pp_stage1_match is scores followed by topk.
 * `eps` is the half-width of the fast mode's re-evaluation band (ignored in
 * exact mode; <=0 selects the default 2e-4).
 * stats (optional, may be NULL): device int32[4] = {rows re-evaluated on their
 * candidate columns, rows re-evaluated in full, columns re-evaluated, 0}.
 * ------------------------------------------------------------------------- */
int pp_stage1_workspace_bytes(int B, int N, int C, size_t* bytes);

int pp_stage1_scores(const float* bank, const float* query, const float* mask,
                     int mask_h, int mask_w, int B, int N, int C, int mode,
                     float eps, void* workspace, size_t workspace_bytes,
                     float* sim_avg, int32_t* stats, void* stream);

/* The same with the bank's storage type as an argument (BASELINE configs[4]: a template bank kept in half precision —
 * half the HBM bytes per template).  bank_dtype = PP_BANK_F32: bank is float (B,N,C,16,16), exactly pp_stage1_scores;
 * PP_BANK_F16: bank is IEEE half (B,N,C,16,16) and the result is what pp_stage1_scores returns on those values
 * widened to float (the rounding to half happened when the bank was stored, outside this library). */
#define PP_BANK_F32 0
#define PP_BANK_F16 1
int pp_stage1_scores_ex(const void* bank, int bank_dtype, const float* query, const float* mask,
                        int mask_h, int mask_w, int B, int N, int C, int mode,
                        float eps, void* workspace, size_t workspace_bytes,
                        float* sim_avg, int32_t* stats, void* stream);

int pp_topk(const float* scores, int B, int N, int k, float* out_score,
            int64_t* out_index, void* stream);

int pp_stage1_match(const float* bank, const float* query, const float* mask,
                    int mask_h, int mask_w, int B, int N, int C, int k, int mode,
                    float eps, void* workspace, size_t workspace_bytes,
                    float* sim_avg, float* out_score, int64_t* out_index,
                    int32_t* stats, void* stream);
/* pp_stage1_match with the bank's storage type as an argument: one ABI call for scores + top-k (the host mirror's
 * matching_templates).  With fewer than two (crop, template) items per CU the main kernel runs its 4-wave shape on template
 * halves (shorter tail round: BASELINE configs[1] 73.2 -> 68.7 us per call).  Three launch fusions were built, measured SLOWER at
 * configs[1] and are off by default (profiles/r04/stage1_small.txt): PP_S1_QPREP=1 (query pre-pack as one launch),
 * PP_S1_TOPK_SMALL=1 (one wave per crop), PP_S1_FUSE_TOPK=1 (the last resolve workgroup of a crop ranks its scores through an
 * agent-scope arrival counter).  Same results as pp_stage1_scores_ex + pp_topk, bit for bit, in every form. */
int pp_stage1_match_ex(const void* bank, int bank_dtype, const float* query, const float* mask,
                       int mask_h, int mask_w, int B, int N, int C, int k, int mode,
                       float eps, void* workspace, size_t workspace_bytes, float* sim_avg,
                       float* out_score, int64_t* out_index, int32_t* stats, void* stream);


/* ------------------------------------------------------------------------- *
 * Stage 2/3 glue around the networks (picopose_amd/csrc/pp_geom.hip).
 * ------------------------------------------------------------------------- */

/* utils/matching.py:6-26 matching_features_similarity (model/picopose.py:81).
 * src_feat/tar_feat (B,C,16,16) template/query features, src_mask (B,mh,mw) template mask;
 * out (B,256,16,16): out[b,s,h,w] = relu(cos(query patch t=w*16+h, template patch s) * mask_s). */
int pp_similarity_volume(const float* src_feat, const float* tar_feat, const float* src_mask,
                         int mask_h, int mask_w, int B, int C, float* out, void* stream);

/* utils/torch_utils.py:39-51 calc_pred_Ms (model/picopose.py:84): pred_scale (B), pred_inplane
 * (B,2) cos/sin, pred_translation (B,2), tem_pose (B,4,4), tem_K/tem_M (B,3,3) -> pred_Ms (B,3,3). */
int pp_calc_pred_Ms(const float* pred_scale, const float* pred_inplane, const float* pred_translation,
                    const float* tem_pose, const float* tem_K, const float* tem_M, int B,
                    float trans_scale, float* pred_Ms, void* stream);

/* utils/pose_recovery.py:9-65 pose_recovery_2d_prediction (model/picopose.py:86-89) -> (B,4,4).
 * Contract (asserted on the host by the reference, torch_utils.py:100-101): query_M is a crop
 * affine with M01 = M10 = 0 and M00 = M11. */
int pp_pose_recovery_2d(const float* query_M, const float* query_K, const float* pred_Ms,
                        const float* tem_K, const float* tem_M, const float* tem_pose, int B,
                        float* pred_pose, void* stream);

/* utils/correspondence.py:10-26 compute_init_correspondences (model/picopose.py:91):
 * pred_Ms (B,3,3), tem_mask (B,mh,mw) square -> init_flow (B,2,16,16) [ch0 = x], init_certainty
 * (B,1,16,16). */
int pp_init_correspondences(const float* pred_Ms, const float* tem_mask, int mask_h, int mask_w, int B,
                            float* init_flow, float* init_certainty, void* stream);

/* utils/correspondence.py:28-59 compute_stage3_correspondences (model/picopose.py:93):
 * pred_flow (B,2,H,W), pred_certainty (B,1,H,W) -> tar_pts, src_pts (B,H*W,2) int64, entry
 * k = w*H + h = (x,y) or (-1,-1). */
int pp_stage3_correspondences(const float* pred_flow, const float* pred_certainty, int B, int H, int W,
                              float threshold, int64_t* tar_pts, int64_t* src_pts, void* stream);

/* utils/torch_utils.py:257-284 gather, as used at utils/pose_recovery.py:76-77: features
 * (B,C,H,W), index_patches (B,N,2) int64 (x,y) with -1 padding -> out (B,N,C) whose first count[b]
 * rows are features[b,:,y,x] of the valid entries in order; count (B) int32. */
int pp_gather_valid(const float* features, const int64_t* index_patches, int B, int C, int H, int W,
                    int N, float* out, int32_t* count, void* stream);

/* ------------------------------------------------------------------------- *
 * Network engine (picopose_amd/csrc/pp_gemm.hip): fp32 MFMA GEMM / implicit-GEMM convolution
 * and the row-wise kernels around it.  Device tensors are token-major / NHWC: a row is a token
 * or a pixel, channels are contiguous.  These entries replace the stock torch ops (nn.Linear,
 * nn.Conv2d, nn.ConvTranspose2d, nn.LayerNorm, softmax, nn.GroupNorm) the reference's modules
 * call: model/stage1/layers/{attention.py:44-62,mlp.py:30-41,patch_embed.py:66-82,block.py:56-106},
 * model/stage2/affine_regressor.py:72-84, model/stage3/{dpt.py:252-272,flow_decoder.py:58-94,
 * raft_decoder.py:147-161,287-289}.
 * ------------------------------------------------------------------------- */
#define PP_ACT_NONE 0
#define PP_ACT_RELU 1
#define PP_ACT_GELU 2    /* exact erf GELU (nn.GELU default) */
#define PP_ACT_LEAKY01 3 /* LeakyReLU(0.1) */
#define PP_ACT_TANH 4

/* arithmetic of the GEMM engine */
#define PP_PREC_F32 0   /* v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate                       */
#define PP_PREC_F16X3 1 /* operands scaled and split into 2 fp16 terms, 3 x v_mfma_f32_16x16x32_f16, fp32 acc. */
#define PP_PREC_F16 2   /* plain fp16 operands f16(4 x) [rows][ld] ("h" format), ONE v_mfma_f32_16x16x32_f16 per product, fp32
                           accumulate — the arithmetic BASELINE configs[4] names; pre-split operands only (A_hl / B_hl / C_hl
                           then hold the h format) */

/* C[m,n] = residual[m,n] + gamma[n] * act(alpha * sum_k A(m,k) * B(n,k) + bias[n])
 * for every batch index z = z0*batch1 + z1 (operand offsets z0*bs0 + z1*bs1, in floats). */
typedef struct PpGemmDesc {
    const float* A;        /* dense: [M][lda]; conv: NHWC image (conv_b, conv_h, conv_w, lda>=conv_cin) */
    const float* B;        /* [N][ldb] (k contiguous) or, if b_kn, [K][ldb] (n contiguous)            */
    float* C;              /* [M][ldc]; with shuffle_r: NHWC image (b, h*r, w*r, ldc)                 */
    const float* bias;     /* [N] or NULL */
    const float* gamma;    /* [N] or NULL (LayerScale) */
    const float* residual; /* laid out like C, or NULL */
    const float* residual2; /* second addend laid out like C, or NULL (FeatureFusionBlock, dpt.py:139-141) */
    int M, N, K;
    int lda, ldb, ldc;
    int b_kn;
    int batch0, batch1;
    long long a_bs0, a_bs1, b_bs0, b_bs1, c_bs0, c_bs1;
    float alpha;
    int act;
    int relu_in;           /* apply ReLU to A while loading (ResidualConvUnit, dpt.py:82-86)           */
    /* implicit im2col (conv_kh == 0: dense A): k = (ky*conv_kw + kx)*conv_cin + ci                    */
    int conv_kh, conv_kw, conv_cin, conv_stride, conv_pad, conv_h, conv_w, conv_ho, conv_wo;
    long long conv_bstride; /* floats between consecutive images of A (0: conv_h*conv_w*lda)           */
    /* ConvTranspose2d(kernel = stride = shuffle_r): rows are the pixels of (b, shuffle_h, shuffle_w),
     * column n = (dy*r + dx)*Cout + co is stored at pixel (y*r+dy, x*r+dx), channel co               */
    int shuffle_r, shuffle_h, shuffle_w;
    int prec;              /* PP_PREC_*                                                               */
    /* Pre-split f16x3 operands ("hl" format): a matrix [rows][ld] of fp32 becomes fp16 [rows][ld/8][2][8] —
     * for every group of 8 consecutive k the 8 hi terms then the 8 lo terms (32 contiguous bytes), so that a K
     * tile of 32 is one 128-byte segment per row.  element (r, k), term p: r*2*ld + (k/8)*16 + p*8 + k%8.
     * ld % 8 == 0; the buffers are indexed exactly like the fp32 operand would be (dense / NHWC image).   */
    const void* B_hl;      /* optional pre-split weights for PP_PREC_F16X3 ([N][ldb], pp_split_f16x3)  */
    float b_scale;         /* power-of-two scale the pre-split weights were multiplied by             */
    const void* A_hl;      /* optional pre-split activation operand (pp_split_activation); A may be NULL */
    long long a_hl_bytes, b_hl_bytes; /* filled in by pp_gemm (extent of the buffers)                  */
    void* C_hl;            /* optional: ALSO (or, with C == NULL, only) write the output as an hl operand */
    int ldc_h;             /* [M][ldc_h] for the next GEMM (no pixel shuffle, no batch); ldc_h % 8 == 0  */
    int c_relu;            /* C_hl holds split(max(out, 0)): the consumer's input ReLU folded in        */
    const float* alpha_dev;  /* optional DEVICE scalars: alpha is multiplied by alpha_dev[0] (and alpha_dev2[0]) when the kernel   */
    const float* alpha_dev2; /* runs — the inverse range scales of backward operands, chosen on the device (pp_pow2_scale_ws)      */
    /* K slices (weight gradients: few output tiles over a very long K leave most CUs idle).  ksplit = S > 1, both operands pre-split,
     * dense, M % 256 == 0, K % (64 S) == 0: C holds S*M rows — rows s*M .. of it receive the product over k in [s K/S, (s+1) K/S)
     * (one launch of S * tiles work items); add the S slices in index order (pp_sum_slices).  No bias / activation / residual.  */
    int ksplit;
    int ks_rows;             /* filled in by pp_gemm (the rows of one slice)                                                       */
    /* filled in by pp_gemm — a batch of fp32 products whose A and C blocks lie one behind the other (a_bs0 = M lda, c_bs0 = M ldc,
     * batch1 = 1, M % 256 == 0, no residual: the sixteen products of a Winograd convolution) runs as ONE persistent launch of the fp32
     * engine over batch0 * M rows; row tile r reads the weights of group r BM / grp_rows, grp_b_bytes further on.                     */
    int grp_rows;
    long long grp_b_bytes;
} PpGemmDesc;

int pp_gemm(const PpGemmDesc* desc, void* stream);
/* Split n fp32 weights (rows of a multiple of 8 elements) once at load time into an hl buffer of 2n halfs:
 * scale[0] = 2^k with max|scale*w| in [512,1024) (device float), hi = f16(scale*w), lo = f16(scale*w - hi). */
int pp_split_f16x3(const float* w, long long n, void* hl, float* scale, void* stream);
/* The same split with the scale left ON THE DEVICE: scale2[0] = 2^k, scale2[1] = 2^-k (pass scale2 + 1 as PpGemmDesc.alpha_dev with
 * b_scale = 1: no host read-back — the training step re-splits every parameter after every optimizer step); partials: 1024 floats of
 * scratch; two launches. */
int pp_split_weights_ws(const float* w, long long n, int terms, void* out, float* scale2, float* partials, void* stream);
/* Split an activation tensor x (B, P, C) fp32 (batch / row strides in floats, channels contiguous, C % 8 == 0)
 * once into a contiguous hl buffer (B*P rows, ld = C) for PP_PREC_F16X3 (activation scale 4; optional ReLU first). */
int pp_split_activation(const float* x, long long batch_stride, int B, int P, int row_stride, int C, int relu,
                        void* hl, void* stream);
/* The same into C columns of a wider hl operand whose rows hold ld_h elements (hl = address of the first column's
 * group: a multiple of 8 columns into the row): the channel concatenation of operands without an fp32 concat buffer. */
int pp_split_activation_ld(const float* x, long long batch_stride, int B, int P, int row_stride, int C, int relu,
                           void* hl, int ld_h, void* stream);
/* Backward-product operands (picopose_amd/autograd.py): gradients span 1e-9 .. 1, so an operand is multiplied by a power of two chosen
 * on the DEVICE from its max before it is split, and the product is scaled back through PpGemmDesc.alpha_dev — no host sync, and
 * (round 4) no separate scaling / transposing passes:
 *   pp_pow2_scale_ws   scale2[0] = 2^k with max|2^k x| in [512, 1024), scale2[1] = 2^-k; partials: >= 1024 floats of scratch
 *                      (two launches: per-workgroup maxima, one fold — no atomics, no init launch);
 *   pp_split_scaled_t  x (rows, C) fp32 with row pitch ld -> contiguous operand (rows, C) of scale[0] * x;
 *   pp_split_transpose_t  x (rows, cols) fp32 with row pitch ld -> contiguous operand (cols, rows) of scale[0] * x^T (scale NULL = 1;
 *                      rows % 8 == 0): the K-major operands of dW = dz^T x in ONE pass instead of scale copy + transposed copy + split. */
int pp_pow2_scale_ws(const float* x, long long n, float* scale2, float* partials, void* stream);
int pp_split_scaled_t(const float* x, long long rows, int ld, int C, const float* scale, void* out, int terms, void* stream);
int pp_split_transpose_t(const float* x, long long rows, int cols, int ld, const float* scale, void* out, int terms, void* stream);
/* ... with an operand row pitch ld_out >= rows (% 8 == 0), the k in [rows, ld_out) zero: K padded to a multiple of the K slices */
int pp_split_transpose_ld(const float* x, long long rows, int cols, int ld, const float* scale, void* out, long long ld_out, int terms,
                          void* stream);
/* The K-major im2col of an NHWC image (pp_im2col_t_nhwc below) written straight as the engine operand: (ksize^2 C) operand rows with
 * K = B Ho Wo pixels (a multiple of 8) — the B operand of a convolution's weight gradient dW = dz^T colT without the fp32 matrix. */
int pp_im2col_t_operand(const float* x, int B, int H, int W, int C, int ksize, int stride, int pad, void* out, int terms, void* stream);
/* Second half of a split-K linear layer (a few rows against a long K — stage 2's fc1, affine_regressor.py:77: 160 x 16384 x
 * 1024 fills 8 workgroups as one GEMM): part (S, M, N) fp32 = the S K-slices' products from one batched pp_gemm;
 * out[m, n] = act(sum_s part[s, m, n] + bias[n]), slices added in index order. */
int pp_sum_slices(const float* part, int S, int M, int N, const float* bias, int act, float* out, void* stream);
/* Columns col0 .. col0 + c - 1 (any alignment, c <= 64) of every row of an existing hl operand with rows of ld_h channels
 * <- x (rows, c) fp32 with row pitch ld_x: the narrow member of a channel concatenation (raft_decoder.py:161, [out | flow]). */
int pp_hl_patch_columns(const float* x, int ld_x, int c, long long rows, void* hl, int ld_h, int col0, void* stream);

/* ---- operand-format ("terms") forms of the producers above.  terms = 2: the hl format of PP_PREC_F16X3 (fp16 [rows][ld/8][2][8]);
 * terms = 1: the h format of PP_PREC_F16 (plain fp16 f16(4 x) [rows][ld]) — BASELINE configs[4]'s "fp16 storage / MFMA with fp32
 * accumulate".  Same arguments as the entries they generalise, which are these with terms = 2. */
int pp_split_weights_t(const float* w, long long n, int terms, void* out, float* scale, void* stream);
int pp_split_activation_t(const float* x, long long batch_stride, int B, int P, int row_stride, int C, int relu, void* out,
                          int ld_h, int terms, void* stream);
int pp_hl_patch_columns_t(const float* x, int ld_x, int c, long long rows, void* out, int ld_h, int col0, int terms, void* stream);
int pp_layernorm_t(const float* x, const float* gamma, const float* beta, int rows, int C, float eps, float* y, void* out, int terms,
                   void* stream);
int pp_resize_bilinear_nhwc_t(const float* in, int B, int H, int W, int C, int Ho, int Wo, float mul, void* out, int terms,
                              void* stream);
/* ... the same resize with BOTH results: the fp32 map `out` (B, Ho, Wo, C) and its operand form `out_operand` (the DPT fusion block's
 * path map, dpt.py:150-155 with out_conv moved in front of the interpolation: an output and the flow decoder's projection input). */
int pp_resize_bilinear_nhwc_dual(const float* in, int B, int H, int W, int C, int Ho, int Wo, float mul, float* out, void* out_operand,
                                 int terms, void* stream);
int pp_warp_nhwc_t(const float* feat, int feat_batch, const float* flow, int B, int H, int W, int C, int ld_flow, void* out,
                   int ld_h, int terms, void* stream);
/* attention on the operand the qkv GEMM wrote, result as fp32 (out) and / or as an operand (out_operand), both optional */
int pp_attention_t(const void* qkv_operand, int terms, int B, int T, int heads, int head_dim, float scale, float* out,
                   void* out_operand, void* stream);

/* Fused multi-head self-attention (model/stage1/layers/attention.py:49-62): qkv (B,T,3,heads,64) as the qkv
 * linear produces it -> out (B,T,heads*64) = softmax((q*scale) k^T) v per head; exact fp32 MFMA, flash style. */
int pp_attention(const float* qkv, int B, int T, int heads, int head_dim, float scale, float* out, void* stream);
/* Same, writing the output (also) as an hl operand (B*T rows, ld = heads*head_dim) for the projection GEMM;
 * out may be NULL. */
int pp_attention_split(const float* qkv, int B, int T, int heads, int head_dim, float scale, float* out, void* out_hl,
                       void* stream);
/* General form: prec = PP_PREC_F32 (exact fp32 products, as the two entries above) or PP_PREC_F16X3 (q, k, v and
 * the probabilities split into 2 fp16 terms, 3 fp16 MFMAs per product, fp32 soft-max statistics and accumulation);
 * out and/or the hl operand out_hl. */
int pp_attention_ex(const float* qkv, int B, int T, int heads, int head_dim, float scale, int prec, float* out, void* out_hl,
                    void* stream);
/* PP_PREC_F16X3 attention whose input is already the hl operand (B*T rows, ld = 3*heads*head_dim) that the qkv GEMM
 * wrote (PpGemmDesc.C_hl): no split work inside the kernel; bit-identical to pp_attention_ex on the fp32 qkv. */
int pp_attention_hl(const void* qkv_hl, int B, int T, int heads, int head_dim, float scale, float* out, void* out_hl,
                    void* stream);
/* Training forward of the same attention (PP_PREC_F16X3): out as above plus lse2 (B*heads, T), the base-2 log-sum-exp of every
 * query's scaled scores — all pp_attention_backward needs; the T x T probabilities are never stored. */
int pp_attention_train(const float* qkv, int B, int T, int heads, int head_dim, float scale, float* out, float* lse2, void* stream);
/* Adjoint of attention.py:49-62 (what torch.autograd derives for the reference): dqkv (B*T, 3*heads*64) from qkv, the forward's out
 * and lse2, and dout (B*T, heads*64).  gpair = device (2^k, 2^-k) with max|dout| 2^k in [512, 1024) (pp_pow2_scale_ws); Dws = B*heads*T
 * floats of scratch.  Scores and probabilities are recomputed tile by tile; two kernels (dq | dk, dv), no atomics: bit-reproducible. */
int pp_attention_backward(const float* qkv, const float* out, const float* dout, const float* lse2, const float* gpair, int B, int T,
                          int heads, int head_dim, float scale, float* Dws, float* dqkv, void* stream);

/* nn.LayerNorm(C, eps) over rows of a [rows][C] matrix. */
int pp_layernorm(const float* x, const float* gamma, const float* beta, int rows, int C, float eps,
                 float* y, void* stream);
/* Same, writing the result (also) as an hl operand (rows, ld = C; C % 8 == 0); y may be NULL. */
int pp_layernorm_split(const float* x, const float* gamma, const float* beta, int rows, int C, float eps, float* y,
                       void* hl, void* stream);
/* softmax(dim=-1) in place over rows of a [rows][ld] matrix (n valid columns). */
int pp_softmax_rows(float* x, int rows, int n, int ld, void* stream);
/* nn.GroupNorm(groups, C, eps) (+ReLU when relu != 0) on an NHWC tensor (B, HW, C). */
int pp_groupnorm_nhwc(const float* x, const float* gamma, const float* beta, int B, int HW, int C,
                      int groups, float eps, int relu, float* y, void* stream);
/* out[b, c, col_off + r] = in[b, r, c]: (B,R,C) -> (B,C,ld_out); NCHW <-> NHWC at the API boundary.
 * Batch strides in floats (0: dense). */
int pp_transpose_batched(const float* in, long long in_batch_stride, int B, int R, int C, float* out,
                         long long out_batch_stride, int ld_out, int col_off, void* stream);
/* vision_transformer.py:209-216: tokens (B,T+1,C) = [cls; patches (B,T,C)] + pos (T+1,C). */
int pp_assemble_tokens(const float* patches, const float* cls_token, const float* pos, int B, int T, int C,
                       float* tokens, void* stream);
/* F.normalize(x, dim=1) for [rows][n<=64]. */
int pp_normalize_rows(const float* x, int rows, int n, float eps, float* y, void* stream);

/* ------------------------------------------------------------------------- *
 * Stage-3 sampling kernels (picopose_amd/csrc/pp_sample.hip), NHWC.
 * ------------------------------------------------------------------------- */
/* F.interpolate(mode="bilinear", align_corners=True) (dpt.py:150-152, flow_decoder.py:88-92); the
 * result is multiplied by `mul` (the 2x of the flow up-sampling). */
int pp_resize_bilinear_nhwc(const float* in, int B, int H, int W, int C, int Ho, int Wo, float mul,
                            float* out, void* stream);
/* Same, writing the result only as an hl operand (B*Ho*Wo rows, ld = C, C % 8 == 0) for the convolution that follows. */
int pp_resize_bilinear_nhwc_hl(const float* in, int B, int H, int W, int C, int Ho, int Wo, float mul, void* out_hl,
                               void* stream);
/* FlowDecoder.feature_sample (flow_decoder.py:49-56): out[b,p,:] = bilinear(feat[b % feat_batch], p + flow[b,p]),
 * zeros padding, align_corners=True.  flow rows have ld_flow floats (x, y first), out rows ld_out.  feat holds
 * feat_batch images (= B, or the query maps given once for the B / feat_batch hypotheses of a hypothesis-major batch). */
int pp_warp_nhwc(const float* feat, int feat_batch, const float* flow, int B, int H, int W, int C, int ld_flow,
                 float* out, int ld_out, void* stream);
/* Same, writing the result only as C columns of an hl operand with rows of ld_h elements (C % 8 == 0). */
int pp_warp_nhwc_hl(const float* feat, int feat_batch, const float* flow, int B, int H, int W, int C, int ld_flow,
                    void* out_hl, int ld_h, void* stream);
/* nn.AvgPool2d(2,2) on NHWC. */
int pp_avgpool2_nhwc(const float* in, int B, int H, int W, int C, float* out, void* stream);
/* dst[i] = src[index[i]] for i < n: rows of row_floats fp32 (a multiple of 4; both buffers 16-byte aligned), index int64
 * into the n_src_rows rows of src — the selection of the top-k templates' rgb / mask / pts3d / pose / K / M rows
 * (model/picopose.py:55-62). */
int pp_gather_rows(const float* src, const long long* index, long long n_src_rows, long long row_floats, int n, float* dst,
                   void* stream);

/* -------------------------------------------------------------------------
 * Crop preprocessing of one detection (provider/bop_test_dataset.py:162-177, utils/data_utils.py:231-250):
 * image (H, W, 3) uint8 as loaded, optional full-frame binary mask (H, W) uint8 (device pointers); crop rows [y1, y2),
 * columns [x1, x2); out_rgb (3, S, S) fp32 = (cv2-style INTER_LINEAR resize of the channel-flipped crop / 255
 * [* (mask > 0) if rgb_mask_flag] - mean) / std (mean3 / std3: host pointers to 3 doubles, in output-channel order);
 * out_mask (S, S) fp32 = INTER_NEAREST resize of the cropped mask (may be NULL).  cv2 is not available to pin the
 * interpolation bit-for-bit: it follows OpenCV's published definition (oracle/preprocess.py).
 * ------------------------------------------------------------------------- */
int pp_crop_resize_normalize(const unsigned char* image, int H, int W, const unsigned char* mask, int y1, int y2, int x1,
                             int x2, int S, int rgb_mask_flag, const double* mean3, const double* std3, float* out_rgb,
                             float* out_mask, void* stream);
/* Template lookup points (bop_test_dataset.py:233-235, data_utils.py:97-115): depth (H, W) fp32 metres -> out (P, P, 3):
 * the back-projection ((x-cx) z/fx, (y-cy) z/fy, z) of the crop rows [y1,y2) x columns [x1,x2), INTER_NEAREST-resized. */
int pp_depth_points_nearest(const float* depth_m, int H, int W, int y1, int y2, int x1, int x2, int P, float fx, float fy,
                            float cx, float cy, float* out_pts, void* stream);
/* CorrelationPyramid (raft_decoder.py:30-53) + CorrLookup (corr_lookup.py:100-134) without the
 * (B*HW, HW) volume: f1 (B,H,W,C) with rows of ld_f1 floats, f2_l{0,1,2} = f2 and its 2x2 average pools holding
 * f2_batch images (image b reads f2[b % f2_batch]), flow (B,H,W,ld_flow);
 * out (B,H,W,ld_out) with channel l*(2r+1)^2 + a*(2r+1) + b = corr_l sampled at x offset a-r,
 * y offset b-r around (p + flow)/2^l.
 * When H, W are multiples of 8 and C of 32 the local correlations of an 8x8 pixel tile against a 16x16 region of f2
 * are computed on the matrix cores in the engine's f16x3 arithmetic (22 operand bits, fp32 accumulation; values with
 * |x| >= 16376 saturate); otherwise (and with PP_CORR_TILED=0 in the environment) by exact fp32 fmas, one lane per
 * neighbour position. */
int pp_corr_lookup_nhwc(const float* f1, int ld_f1, const float* f2_l0, const float* f2_l1, const float* f2_l2,
                        int f2_batch, const float* flow, int B, int H, int W, int C, int levels, int radius,
                        int ld_flow, float* out, int ld_out, void* stream);
/* The same with the arithmetic as an argument: PP_PREC_F16X3 = the default above, PP_PREC_F32 = exact fp32 products and sums
 * whatever the shape — what `ops.PRECISION = "f32"` / `bench.py --mode exact` runs — PP_PREC_F16 = plain fp16 operands, one MFMA
 * per product, fp32 accumulation (`--mode fp16`: the arithmetic of that mode's convolutions). */
int pp_corr_lookup_nhwc_ex(const float* f1, int ld_f1, const float* f2_l0, const float* f2_l1, const float* f2_l2,
                           int f2_batch, const float* flow, int B, int H, int W, int C, int levels, int radius,
                           int ld_flow, int prec, float* out, int ld_out, void* stream);

/* Convolution with ONE or TWO output channels on an hl operand (the flow / certainty predict layers of the decoder heads,
 * raft_decoder.py:287-289): x_hl (B,H,W) pixels with rows of ld_x channels (C of them read, C % 32 == 0), stride 1,
 * padding ksize / 2, ksize 1 or 3, W in {16, 32, 64} and H a multiple of 256 / W; weight (n_out, ksize*ksize*C) fp32 in the
 * engine's k order (tap-major, then channel), bias (n_out) or NULL, residual (B,H,W,n_out) or NULL added to the result;
 * out (B,H,W,n_out) fp32.  Exact fp32 products of the operand's value (hi + lo) with the fp32 filters, fp32 accumulation —
 * at least the accuracy of the f16x3 engine, independent of the batch size. */
int pp_conv_narrow_hl(const void* x_hl, int ld_x, int B, int H, int W, int C, const float* weight, const float* bias, int ksize,
                      int n_out, const float* residual, float* out, void* stream);
/* The same layers on the fp32 NHWC map itself (x: (B,H,W) pixels with rows of ld_x floats, ld_x % 4 == 0, 16-byte aligned; images
 * contiguous) — the strict-fp32 mode's form (PP_PREC_F32 networks: raft_decoder.py:287-289 in the reference's own arithmetic):
 * every product and sum fp32, taps walked in the engine's k order. */
int pp_conv_narrow_f32(const float* x, int ld_x, int B, int H, int W, int C, const float* weight, const float* bias, int ksize,
                       int n_out, const float* residual, float* out, void* stream);

/* Winograd F(2x2, 3x3) for the 3x3 / stride 1 / padding 1 convolutions of the strict-fp32 mode (raft_decoder.py:251-289, dpt.py:72-95 in
 * the reference's own arithmetic; csrc/pp_winograd.hip): 16 dense products Y_xi (P, Cout) = U_xi (P, Cin) V_xi (Cout, Cin)^T over the
 * P = B H W / 4 output tiles run through pp_gemm (PP_PREC_F32, dense); these are the three fp32 transforms around them.
 *   pp_winograd_input_f32   x: NHWC image (B, H, W) with rows of ld_x floats (C of them read; H, W even, C % 4 == 0, 16-byte aligned,
 *                           images batch_stride floats apart), relu != 0: max(x, 0) first (ResidualConvUnit)  ->  U (16, P, C)
 *   pp_winograd_weight_f32  w: (Cout, 9 Cin) in the engine's k order (tap-major, then channel), rows of ldw floats  ->  V (16, Cout, Cin)
 *   pp_winograd_output_f32  Y (16, P, Cout)  ->  out (B, H, W) with rows of ldc floats: A^T Y A + bias, act (none / ReLU / LeakyReLU),
 *                           + residual + residual2 (laid out like out) */
int pp_winograd_input_f32(const float* x, int ld_x, long long batch_stride, int B, int H, int W, int C, int relu, float* U, void* stream);
int pp_winograd_weight_f32(const float* w, int Cout, int Cin, int ldw, float* V, void* stream);
int pp_winograd_output_f32(const float* Y, int B, int H, int W, int Cout, const float* bias, int act, const float* residual,
                           const float* residual2, float* out, int ldc, void* stream);

/* Winograd F(4x4, 3x3) on the f16x3 engine (PP_PREC_F16X3; round 6) for the large 3x3 / stride 1 / padding 1 convolutions of the flow
 * decoder's heads (raft_decoder.py:251-289): 36 dense products Y_xi (P, Cout) = U_xi (P, Cin) V_xi (Cout, Cin)^T over the P = B H W / 16
 * tiles — four times fewer multiplications than the direct convolution — run by pp_gemm as a batch of pre-split products (A_hl, B_hl,
 * batch0 = the frequencies of the launch, a_bs0 = P lda, b_bs0 = Cout Cin, c_bs0 = P ldc, alpha = 64: ONE persistent launch).
 *   pp_winograd4_input_hl    x_hl: hl operand image (B, H, W), rows of ld_x channels (C of them read from the pointer's column on; H, W
 *                            % 4 == 0, C % 8 == 0); batch_stride in ELEMENTS  ->  U_hl (36, P, C) hl operand of (B^T d B) / 16
 *   pp_winograd4_weight_f32  w (Cout, 9 Cin) in the engine's k order, rows of ldw floats  ->  V (36, Cout, Cin) fp32 (split it with
 *                            pp_split_weights_t as ONE matrix of 36 Cout rows)
 *   pp_winograd4_output      Y (36, P, Cout) fp32  ->  A^T Y A + bias, act (none / ReLU / LeakyReLU): as fp32 NHWC map `out` (rows of ldc
 *                            floats; + residual + residual2 laid out like out) and / or as hl operand `out_hl` (rows of ld_h channels,
 *                            the pointer at the first column's group; of max(., 0) with c_relu).  Cout % 4 == 0 (% 8 for out_hl).
 * ld_y >= Cout: the row pitch of Y in floats (Y may be a column slice of a wider product: two layers that read the same U, fused along N).
 * P_pad >= P: the rows of one frequency block of U and Y (P rounded up to a multiple of 256, the engine's row tile, so that a row tile
 * lies inside one frequency; the pad rows are never read by the output transform). */
int pp_winograd4_input_hl(const void* x_hl, int ld_x, long long batch_stride, int B, int H, int W, int C, int relu, void* U_hl, long long P_pad,
                          void* stream);
int pp_winograd4_weight_f32(const float* w, int Cout, int Cin, int ldw, float* V, void* stream);
int pp_winograd4_output(const float* Y, int ld_y, int B, int H, int W, int Cout, const float* bias, int act, const float* residual,
                        const float* residual2, float* out, int ldc, void* out_hl, int ld_h, int c_relu, long long P_pad, void* stream);

/* Output transform of a Winograd F(4x4, 3x3) layer CHAINED into the input transform of the next one (conv 3x3 -> ReLU -> conv 3x3,
 * raft_decoder.py:251-289): Y (36, P_pad, C) fp32 of the first layer -> U_hl (36, P_pad, C) hl operand of B^T relu'(A^T Y A + bias) B / 16 of the
 * second, the hidden map never stored; bit-identical to pp_winograd4_output (operand output, c_relu) followed by pp_winograd4_input_hl.
 * W in {16, 32, 64}, H % 4 == 0, C % 32 == 0; act none / ReLU / LeakyReLU, then (c_relu) the consumer's input ReLU. */
int pp_winograd4_chain(const float* Y, int ld_y, int B, int H, int W, int C, const float* bias, int act, int c_relu, void* U_hl, long long P_pad,
                       void* stream);

/* The same chain for the strict-fp32 mode's F(2x2, 3x3): Y (16, P, C) fp32 of layer k -> U (16, P, C) fp32 of layer k + 1 (P = B H W / 4),
 * h = act(A^T Y A + bias), then max(h, 0) with relu_next (the next layer's input ReLU: dpt.py:82-86), never stored; bit-identical to
 * pp_winograd_output_f32 followed by pp_winograd_input_f32.  W in {16, 32, 64}, H even, C % 32 == 0. */
int pp_winograd_chain_f32(const float* Y, int B, int H, int W, int C, const float* bias, int act, int relu_next, float* U, void* stream);

/* Sticky operand-saturation word.  The f16x3 / f16 operand formats clamp at the fp16 range (|4 x| >= 65504): a clamped term is finite but
 * WRONG.  With a device word registered here, every kernel that writes operand terms ORs bit 0 into it when a term hit the clamp (one
 * atomic per wave that saw one; nothing when none did) — the host reads the word together with its results (no extra synchronisation)
 * and raises.  word = NULL switches the reporting off.  Per device (the current one).  block.py:104-106 / layer_scale.py:27-28: trained
 * DINOv2 residual streams hold a few very large channels. */
int pp_set_saturation_word(unsigned int* word);


/* The tiled lookup on operands the producers already hold in the engine's hl format (fp16 [pixels][2 ld]: per 8 channels
 * the 8 hi then the 8 lo terms; include "hl" above): f1_hl with rows of ld_f1 channels (a column block of a wider operand
 * is fine), f2_hl_l{0,1,2} contiguous (f2_batch, H >> l, W >> l, C).  H, W multiples of 8 and C of 32 (else PP_EINVAL: use
 * pp_corr_lookup_nhwc).  Same values as pp_corr_lookup_nhwc on the fp32 maps, without the split work per staged chunk. */
int pp_corr_lookup_nhwc_hl(const void* f1_hl, int ld_f1, const void* f2_hl_l0, const void* f2_hl_l1, const void* f2_hl_l2,
                           int f2_batch, const float* flow, int B, int H, int W, int C, int levels, int radius, int ld_flow,
                           float* out, int ld_out, void* stream);

/* ------------------------------------------------------------------------- *
 * utils/pose_recovery.py:68-105 pose_recovery_ransac_pnp, batched over P = instances x hypotheses
 * (run_test.py:168-184 calls it once per pair): gather of the valid 2D/3D correspondences, object-frame
 * transform, RANSAC (5-point samples, `iterations`, squared reprojection error <= threshold^2) with EPnP
 * as the model solver and an EPnP refit on the inliers (picopose_amd/csrc/pp_pnp.hip).  One 512-thread workgroup per
 * problem, fp64; hypothesis h of problem p draws its sample from a counter-based hash of (p, h), so a problem's result
 * depends on its index in the batch (as OpenCV's depends on its RNG state), not on the launch configuration.
 *   tar_pts_2d (P,2,H,W), src_pts_3d (P,3,H,W), K (P,3,3), tem_pose (P,4,4) fp32;
 *   tar_pts, src_pts (P,N,2) int64 (x,y) with -1 padding, N <= 4096;
 *   out: rot (P,3,3) f64, tvec (P,3) f64, inlier_ratio (P) f64, success (P) int32 (0: the reference's
 *   failure outputs I, [0,0,1], 0.0), num_points (P) int32.
 * ------------------------------------------------------------------------- */
int pp_pnp_ransac(const float* tar_pts_2d, const float* src_pts_3d, const float* K, const float* tem_pose,
                  const int64_t* tar_pts, const int64_t* src_pts, int P, int H, int W, int N, int iterations,
                  float reproj_threshold, double* rot, double* tvec, double* inlier_ratio, int32_t* success,
                  int32_t* num_points, void* stream);

/* pp_pnp_ransac with one more output for parity work: refit_branches (P, 40) f64 = for the final EPnP refit on the consensus
 * set, the candidate pose of each of EPnP's three beta initialisations (N = 1 / 2 / 3 null-space vectors; OpenCV's epnp.cpp
 * find_betas_approx_1 / _2 / _3 behind cv2.solvePnPRansac, utils/pose_recovery.py:93-95) as [R (9, row-major), t (3), mean
 * reprojection error in px (1e300: the branch gave no pose)] x 3, then the index of the branch the refit kept (-1: none, the
 * RANSAC winner's pose is returned).  Two correct EPnP implementations may keep different branches when their errors are
 * within rounding of each other; compared branch by branch they must agree to solver accuracy (tests/test_e2e.py). */
int pp_pnp_ransac_debug(const float* tar_pts_2d, const float* src_pts_3d, const float* K, const float* tem_pose,
                        const int64_t* tar_pts, const int64_t* src_pts, int P, int H, int W, int N, int iterations,
                        float reproj_threshold, double* rot, double* tvec, double* inlier_ratio, int32_t* success,
                        int32_t* num_points, double* refit_branches, void* stream);

/* ------------------------------------------------------------------------- *
 * Training forward (model/picopose.py:114-137 — losses only, no gradients; csrc/pp_train.hip)
 * ------------------------------------------------------------------------- */
/* KeyPointSampler.sample_pts (utils/keypoints.py:120-205, Keypoint :47-92, torch_utils.py unproject_points :138-151,
 * project_points :154-161) for B (template = "src", real = "tar") pairs: crop masks (B, mask_h, mask_w), full depth images
 * (B, depth_h, depth_w), crop affines M and their inverses (torch_utils.inverse_affine :93-111), intrinsics K and their
 * inverses, the rigid motions between the two cameras (B,4,4), all fp32 row-major.  The small matrix inverses are the
 * caller's (the reference's torch.inverse); every per-point step is evaluated here as the reference evaluates it,
 * including the comparison of re-projections in CROP pixels with grid points in IMAGE pixels against 1000 px.
 * out: src_pts, tar_pts (B, 4096, 2) fp32 patch coordinates (integer pixel / 3.5), -1 where invalid. */
size_t pp_train_keypoints_workspace_bytes(int B);
int pp_train_keypoints(const float* src_mask, const float* tar_mask, int mask_h, int mask_w, const float* src_depth,
                       const float* tar_depth, int depth_h, int depth_w, const float* src_Minv, const float* tar_Minv,
                       const float* src_M, const float* tar_M, const float* src_Kinv, const float* tar_Kinv, const float* src_K,
                       const float* tar_K, const float* T_src2tar, const float* T_tar2src, int B, float* src_pts, float* tar_pts,
                       void* workspace, size_t workspace_bytes, void* stream);
/* nn.BatchNorm2d in TRAINING mode (model/stage3/dpt.py:64-66,84-90, flow_decoder.py:22) on an NHWC map viewed as
 * (rows, C), C % 4 == 0: y = relu?((x - mean_batch) / sqrt(var_batch + eps) * gamma + beta) + residual + residual2
 * (residuals may be NULL); running_mean / running_var (may both be NULL) get the momentum update with the unbiased
 * variance, as the module does.  Statistics are summed in fp64. */
size_t pp_batchnorm_train_workspace_bytes(int rows, int C);
int pp_batchnorm_train(const float* x, const float* gamma, const float* beta, int rows, int C, float eps, float momentum,
                       float* running_mean, float* running_var, int relu, const float* residual, const float* residual2, float* y,
                       void* workspace, size_t workspace_bytes, void* stream);
/* compute_stage_one_loss (utils/loss_utils.py:144-175): out[i] = F.normalize(src[index[i] * row_stride ...][:C]) — the
 * gather of torch_utils.gather (:257-284) on a token-major feature map and the normalisation of :169-170 in one pass */
int pp_gather_normalize_rows(const float* src, long long row_stride, const int64_t* index, int n, int C, float eps, float* out,
                             void* stream);
/* ... and F.cross_entropy(scale * logits, arange(n)) per row (:171-174): row_loss[i] = logsumexp_j - the diagonal entry */
int pp_xent_diag_rows(const float* logits, int n, int ld, float scale, float* row_loss, void* stream);
/* compute_stage_three_loss for one level (utils/loss_utils.py:188-202 with compute_flow_loss :119-125, RAFTLoss :24-39):
 * flow (B,H,W,2), certainty logits (B,H,W) NHWC, tar_pts (B,4096,2) from pp_train_keypoints.  Writes
 * pp_flow_loss_blocks() x 3 doubles: per workgroup [sum of BCE-with-logits terms, sum over valid pixels with
 * |gt flow| < max_flow of |flow - gt|_1, number of those pixels]; the caller adds the rows and forms the two means. */
int pp_flow_loss_blocks(void);
int pp_flow_loss_sums(const float* flow, const float* certainty, const float* tar_pts, int B, int H, int W, float max_flow,
                      double* partial_sums, void* stream);

/* ------------------------------------------------------------------------- *
 * First backward slice of the training path (utils/lite.py:33-49 -> loss.backward()): the row-wise /
 * element-wise adjoints whose matrix products run on pp_gemm (picopose_amd/autograd.py: dgrad = dz W, wgrad = dz^T x).  Scope:
 * InfoNCE (utils/loss_utils.py:144-175) -> the last ViT block (layers/block.py:82-107); the stage-2 losses (:177-186) -> the
 * AffineRegressor (model/stage2/affine_regressor.py:72-84).  Deterministic (fixed-order reductions, no atomics); csrc/pp_backward.hip.
 * ------------------------------------------------------------------------- */
/* out[c] = sum_r x[r][c] over `rows` rows of ld floats (bias / LayerScale / norm-parameter gradients) */
size_t pp_colsum_workspace_bytes(long long rows, int cols);
int pp_colsum(const float* x, long long rows, int cols, int ld, float* out, void* workspace, size_t workspace_bytes, void* stream);
/* y = act(z) and dz = dy * act'(z) for the PP_ACT_* activations (exact erf GELU, as nn.GELU) */
int pp_act_forward(const float* z, long long n, int act, float* y, void* stream);
int pp_act_backward(const float* z, const float* dy, long long n, int act, float* dz, void* stream);
/* op 0: out = a * b; 1: out = a * b[col] (b a vector of `cols`); 2: out = a + b — n elements, rows of `cols` */
int pp_elementwise(int op, const float* a, const float* b, long long n, int cols, float* out, void* stream);
/* nn.LayerNorm backward: dx (rows, C) and gx = dy * xhat (dgamma = column sums of gx, dbeta = column sums of dy) */
int pp_layernorm_backward(const float* x, const float* gamma, const float* dy, int rows, int C, float eps, float* dx, float* gx, void* stream);
/* nn.GroupNorm(groups, C) (+ReLU when relu != 0) backward on NHWC (B, HW, C): dx, gx = d * xhat and gy = d with d = dy masked by the
 * ReLU (dgamma / dbeta = their column sums) */
int pp_groupnorm_backward_nhwc(const float* x, const float* gamma, const float* beta, const float* dy, int B, int HW, int C, int groups,
                               float eps, int relu, float* dx, float* gx, float* gy, void* stream);
/* softmax backward per row: ds = p * (dp - sum_j dp_j p_j) */
int pp_softmax_backward_rows(const float* p, const float* dp, long long rows, int n, float* ds, void* stream);
/* gradient of pp_xent_diag_rows's mean: dlogits[i][j] = upstream[0] * scale / n * (softmax_j(scale * logits[i]) - [i == j]) */
int pp_xent_diag_backward(const float* logits, int n, int ld, float scale, const float* upstream, float* dlogits, void* stream);
/* dst[index[i]][0..C) += src[i][0..C) for i = 0 .. n-1 — the backward of a row gather (torch.gather under autograd,
 * utils/torch_utils.py:257-283 as used by InfoNCE, utils/loss_utils.py:163-175).  `index` may repeat (several key-points in one
 * feature-grid cell): their rows are added in ascending i, no atomics, so the result does not depend on the launch.  dst (rows, C)
 * contiguous, zeroed or holding a sum to extend; every index must be a valid row of dst. */
int pp_scatter_add_rows(const float* src, const int64_t* index, int n, int C, float* dst, void* stream);
/* F.normalize backward for the rows x[index[i] * row_stride ...] (index NULL: row i): dx (rows, C) contiguous */
int pp_normalize_rows_backward(const float* x, long long row_stride, const int64_t* index, const float* dq, int rows, int C, float eps,
                               float* dx, void* stream);
/* im2col of an NHWC image for a ksize x ksize / stride / pad convolution (k order (ky, kx, ci), as pack_conv_weight) and its adjoint */
int pp_im2col_nhwc(const float* x, int B, int H, int W, int C, int ksize, int stride, int pad, float* col, void* stream);
int pp_col2im_nhwc(const float* col, int B, int H, int W, int C, int ksize, int stride, int pad, float* dx, void* stream);
/* pp_split_weights_t with a GIVEN scale (device scalar; a power of two): the operand of a weight whose scale the host already knows
 * (the training graph re-splits every weight each step with the scale read back one step earlier: no host wait) */
int pp_split_with_scale_t(const float* w, long long n, int terms, const float* scale, void* out, void* stream);
/* scale2[0] = the power of two s with max|x| s in [512, 1024) (1 for an all-zero x), scale2[1] = 1 / s — the range normalisation of a
 * gradient operand before a backward product (picopose_amd/autograd.py: _ranged); device scalars, no host sync */
int pp_pow2_scale(const float* x, long long n, float* scale2, void* stream);
/* im2col written transposed: colT (ksize^2 C, rows), rows = B Ho Wo — the K-major operand of the weight-gradient product */
int pp_im2col_t_nhwc(const float* x, int B, int H, int W, int C, int ksize, int stride, int pad, float* colT, void* stream);
/* adjoint of pp_similarity_volume's tail (mask, clamp at 0, the [s][h][w] layout with t = w 16 + h; utils/matching.py:21-25):
 * out / dout (B,256,16,16) -> dS (B, t = 256, s = 256), the gradient of the cosine matrix <tar_hat[t], src_hat[s]> */
int pp_simvol_backward(const float* out, const float* dout, const float* src_mask, int mask_h, int mask_w, int B, float* dS, void* stream);

/* ---- adjoints of stage 3's training path (csrc/pp_backward3.hip; each restates its forward kernel's coordinate arithmetic) ---- */
/* nn.BatchNorm2d in training mode (pp_batchnorm_train: batch statistics, biased variance) on (rows, C), y = relu?(bn(x)):
 * dx, dgamma, dbeta from dy; the ReLU mask is recomputed from x (beta is needed for it) */
size_t pp_batchnorm_train_backward_workspace_bytes(long long rows, int C);
int pp_batchnorm_train_backward(const float* x, const float* gamma, const float* beta, const float* dy, long long rows, int C, float eps,
                                int relu, float* dx, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, void* stream);
/* adjoint of pp_resize_bilinear_nhwc (align_corners = True, result scaled by mul): dy (B,Ho,Wo,C) -> dx (B,H,W,C), gather form */
int pp_resize_bilinear_backward_nhwc(const float* dy, int B, int H, int W, int C, int Ho, int Wo, float mul, float* dx, void* stream);
/* adjoint of pp_avgpool2_nhwc: dy (B,H/2,W/2,C) -> dx (B,H,W,C) (accumulate != 0: added to dx) */
int pp_avgpool2_backward_nhwc(const float* dy, int B, int H, int W, int C, int accumulate, float* dx, void* stream);
/* adjoint of pp_warp_nhwc (FlowDecoder.feature_sample): dfeat (B,H,W,C) must be ZERO on entry (atomic scatter), dflow (B,H,W,2) */
int pp_warp_backward_nhwc(const float* feat, const float* flow, const float* dy, int B, int H, int W, int C, int ld_flow, float* dfeat,
                          float* dflow, void* stream);
/* adjoint of pp_corr_lookup_nhwc (correlation pyramid + lookup): f2_levels[l] = the query map pooled l times (B, H>>l, W>>l, C);
 * df1 (B,H,W,C) is written, df2_levels[l] must be ZERO on entry (atomic scatter), dflow (B,H,W,2).  C <= 256, levels <= 3 */
int pp_corr_lookup_backward_nhwc(const float* f1, const float* const* f2_levels, const float* flow, const float* dout, int B, int H, int W,
                                 int C, int levels, int radius, int ld_flow, int ld_dout, float* df1, float* const* df2_levels, float* dflow,
                                 void* stream);
/* The deterministic forms of the two scatter adjoints: the scattered sums (dfeat; df2 per level) accumulate in 64-bit FIXED POINT
 * (2^-40 units, integer atomics: the same bits whatever order the workgroups arrive in) into caller-zeroed long long buffers of the
 * same element counts; pp_fixed_to_float turns them into the fp32 gradients.  df1 / dflow are written directly as before. */
int pp_warp_backward_nhwc_fixed(const float* feat, const float* flow, const float* dy, int B, int H, int W, int C, int ld_flow,
                                long long* dfeat_acc, float* dflow, void* stream);
int pp_corr_lookup_backward_nhwc_fixed(const float* f1, const float* const* f2_levels, const float* flow, const float* dout, int B, int H,
                                       int W, int C, int levels, int radius, int ld_flow, int ld_dout, float* df1,
                                       long long* const* df2_acc_levels, float* dflow, void* stream);
int pp_fixed_to_float(const long long* acc, long long n, float* out, void* stream);
/* adjoint of pp_flow_loss_sums for one level: g_flow[0] = upstream * flow_weight / (count + eps), g_cert[0] = upstream * mask_weight /
 * (B H W) (device scalars) -> dflow (B,H,W,2), dcertainty (B,H,W) */
int pp_flow_loss_backward(const float* flow, const float* certainty, const float* tar_pts, int B, int H, int W, float max_flow,
                          const float* g_flow, const float* g_cert, float* dflow, float* dcertainty, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PICOPOSE_HIP_H */
