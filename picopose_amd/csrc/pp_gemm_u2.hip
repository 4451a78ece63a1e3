// pp_gemm_u_kernel instantiated for the hl operand format (PP_PREC_F16X3: two fp16 terms, three MFMAs per product)
#include "pp_gemm_u_kernel.h"
int pp_gemm_u_launch_t2(const PpGemmDesc& d, int tile, int mode, bool vec, int cus, hipStream_t st) {
    return pp_u_launch_terms<2>(d, tile, mode, vec, cus, st);
}
PP_SAT_SETTER(pp_sat_set_gemm_u2)
