// pp_gemm_u_kernel instantiated for the h operand format (PP_PREC_F16: plain fp16 operands, one MFMA per product)
#include "pp_gemm_u_kernel.h"
int pp_gemm_u_launch_t1(const PpGemmDesc& d, int tile, int mode, bool vec, int cus, hipStream_t st) {
    return pp_u_launch_terms<1>(d, tile, mode, vec, cus, st);
}
PP_SAT_SETTER(pp_sat_set_gemm_u1)
