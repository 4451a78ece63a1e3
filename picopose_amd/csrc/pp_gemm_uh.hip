// Row-shared 3x3 convolution kernel (both operand formats) and the host-side entry points of the unified pre-split kernels.
#include "pp_gemm_u_kernel.h"

int pp_gemm_u_launch_t1(const PpGemmDesc& d, int tile, int mode, bool vec, int cus, hipStream_t st);
int pp_gemm_u_launch_t2(const PpGemmDesc& d, int tile, int mode, bool vec, int cus, hipStream_t st);

int pp_gemm_u_mode(const PpGemmDesc& d, int terms) {
    const int kt = 64 / terms;
    if (d.conv_kh == 0) return d.ks_rows > 0 ? 3 : 0;
    return (d.conv_cin % kt == 0 && d.conv_kh * d.conv_kw <= 32) ? 1 : 2;
}

void pp_gemm_u_tile_shape(int tile, int& bm, int& bn, int& per_cu) {
    switch (tile) {
        case PP_U_256x256: bm = 256, bn = 256, per_cu = 1; break;
        case PP_U_256x128: bm = 256, bn = 128, per_cu = 1; break;
        case PP_U_128x128: bm = 128, bn = 128, per_cu = 2; break;
        default: bm = 128, bn = 64, per_cu = 2; break;
    }
}


bool pp_gemm_uh_shape_ok(const PpGemmDesc& d, int terms) {
    const int kt = 64 / terms;
    return d.A_hl && d.conv_kh == 3 && d.conv_kw == 3 && d.conv_stride == 1 && d.conv_pad == 1 && d.conv_cin % kt == 0 &&
           d.conv_ho == d.conv_h && d.conv_wo == d.conv_w && d.conv_w >= 16 && d.conv_w <= H_BM && (d.conv_w & (d.conv_w - 1)) == 0 &&
           d.lda == d.conv_cin && d.K == 9 * d.conv_cin && d.conv_bstride == (long long)d.conv_h * d.conv_w * d.lda;
}

int pp_gemm_uh_launch(const PpGemmDesc& d, int terms, int cus, hipStream_t st) {
    static signed char attr_state[PP_MAX_DEVICES];
    signed char& ok = attr_state[pp_cur_device()];
    if (ok == 0)
        ok = (hipFuncSetAttribute((const void*)pp_gemm_uh_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, H_LDS_BYTES) == hipSuccess &&
              hipFuncSetAttribute((const void*)pp_gemm_uh_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, H_LDS_BYTES) == hipSuccess) ? 1 : -1;
    if (ok < 0) return PP_ELAUNCH;
    const int gx = (d.N + H_BN - 1) / H_BN, gy = (d.M + H_BM - 1) / H_BM;
    const int nt = gx * gy, g = nt < cus ? (nt + 7) / 8 * 8 : cus / 8 * 8;
    if (terms == 2) hipLaunchKernelGGL(pp_gemm_uh_kernel<2>, dim3(g), dim3(512), H_LDS_BYTES, st, d, gx, gy);
    else hipLaunchKernelGGL(pp_gemm_uh_kernel<1>, dim3(g), dim3(512), H_LDS_BYTES, st, d, gx, gy);
    return PP_OK;
}


// the vector epilogue's conditions (pp_gemm_dev.h epilogue_wave16), evaluated on the host
bool pp_gemm_u_vec_ok(const PpGemmDesc& d) {
    const int r2 = d.shuffle_r > 0 ? d.shuffle_r * d.shuffle_r : 1;
    const bool shuffle_vec = d.shuffle_r == 0 || ((d.N / r2) & 7) == 0;
    auto al = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    // the epilogue addresses its outputs with 32-bit byte offsets: the fp32 image (C and the residuals) and the operand image
    // (4 bytes per element in the hl format, 2 in the h format) must stay below 4 GB
    const long long out_rows = (long long)d.M * r2, out_cols = d.N / r2;
    const bool small = (out_rows - 1) * d.ldc * 4 + out_cols * 4 < 0xFFFFFF00LL && (!d.C_hl || (out_rows - 1) * d.ldc_h * 4 + out_cols * 4 < 0xFFFFFF00LL);
    return shuffle_vec && small && (d.N & 7) == 0 && (d.ldc & 3) == 0 && al(d.C) && al(d.residual) && al(d.residual2) && al(d.bias) && al(d.gamma) &&
           al(d.C_hl) && (!d.C_hl || (d.ldc_h & 7) == 0);
}

int pp_gemm_u_launch(const PpGemmDesc& d, int tile, int terms, int cus, hipStream_t st) {
    const int mode = pp_gemm_u_mode(d, terms);
    const bool vec = pp_gemm_u_vec_ok(d);
    if (!vec && tile != PP_U_128x64) tile = PP_U_128x128;   // the element-wise epilogue exists for the two small tiles
    return terms == 2 ? pp_gemm_u_launch_t2(d, tile, mode, vec, cus, st) : pp_gemm_u_launch_t1(d, tile, mode, vec, cus, st);
}
PP_SAT_SETTER(pp_sat_set_gemm_uh)
