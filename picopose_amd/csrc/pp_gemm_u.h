// Host-side interface of the unified pre-split contraction kernel (pp_gemm_u.hip), used by the dispatch in pp_gemm.hip.
#ifndef PP_GEMM_U_H
#define PP_GEMM_U_H
#include <hip/hip_runtime.h>
#include "../../include/picopose_hip.h"

// block tiles the kernel is instantiated for
enum { PP_U_128x64 = 0, PP_U_128x128 = 1, PP_U_256x128 = 2, PP_U_256x256 = 3, PP_F_256x192 = 4 /* fp32 engine only */ };

// A-delivery mode the kernel will use for this problem: 0 dense, 1 convolution in channel-slice-major K order, 2 natural order
int pp_gemm_u_mode(const PpGemmDesc& d, int terms);
// block tile shape and resident workgroups per CU of a tile id
void pp_gemm_u_tile_shape(int tile, int& bm, int& bn, int& per_cu);
// Launch on `st`: d.A_hl / d.B_hl (+ a_hl_bytes / b_hl_bytes) in the operand format of `terms` (2: hl, 1: h), persistent over
// min(tiles, slots) workgroups.  Returns PP_OK / PP_E*.
int pp_gemm_u_launch(const PpGemmDesc& d, int tile, int terms, int cus, hipStream_t st);
// 3x3 / stride 1 / pad 1 convolutions on the 256x256 tile with row-shared A delivery (see pp_gemm_u.hip); shape test + launch
bool pp_gemm_uh_shape_ok(const PpGemmDesc& d, int terms);
bool pp_gemm_u_vec_ok(const PpGemmDesc& d);
int pp_gemm_uh_launch(const PpGemmDesc& d, int terms, int cus, hipStream_t st);
// The fp32-operand engine (pp_gemm_f.hip: LDS-DMA ring + v_mfma_f32_32x32x2_f32, the same tile ids): eligibility test (fills the
// operand extents a_hl_bytes / b_hl_bytes), A-delivery mode (0 dense, 1 / 2 convolution) and launch
bool pp_gemm_f_ok(PpGemmDesc& d);
int pp_gemm_f_mode(const PpGemmDesc& d);
int pp_gemm_f_launch(const PpGemmDesc& d, int tile, int cus, hipStream_t st);
#endif
