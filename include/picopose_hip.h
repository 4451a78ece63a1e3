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

int pp_topk(const float* scores, int B, int N, int k, float* out_score,
            int64_t* out_index, void* stream);

int pp_stage1_match(const float* bank, const float* query, const float* mask,
                    int mask_h, int mask_w, int B, int N, int C, int k, int mode,
                    float eps, void* workspace, size_t workspace_bytes,
                    float* sim_avg, float* out_score, int64_t* out_index,
                    int32_t* stats, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PICOPOSE_HIP_H */
