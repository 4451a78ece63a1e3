"""ctypes binding of libpicopose_hip.so — the C ABI declared in include/picopose_hip.h.

There is deliberately no CPU or eager-torch fallback: if the HIP library is
missing the import of any compute entry point raises.
"""
import ctypes
import os
import re

from .build import LIB

_HEADER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include", "picopose_hip.h")

PP_OK = 0
PP_MATCH_EXACT = 0
PP_MATCH_FAST = 1
PP_BANK_F32 = 0
PP_BANK_F16 = 1

_lib = None


class PpGemmDesc(ctypes.Structure):
    _fields_ = [
        ("A", ctypes.c_void_p), ("B", ctypes.c_void_p), ("C", ctypes.c_void_p), ("bias", ctypes.c_void_p),
        ("gamma", ctypes.c_void_p), ("residual", ctypes.c_void_p), ("residual2", ctypes.c_void_p),
        ("M", ctypes.c_int), ("N", ctypes.c_int), ("K", ctypes.c_int),
        ("lda", ctypes.c_int), ("ldb", ctypes.c_int), ("ldc", ctypes.c_int), ("b_kn", ctypes.c_int),
        ("batch0", ctypes.c_int), ("batch1", ctypes.c_int),
        ("a_bs0", ctypes.c_longlong), ("a_bs1", ctypes.c_longlong), ("b_bs0", ctypes.c_longlong),
        ("b_bs1", ctypes.c_longlong), ("c_bs0", ctypes.c_longlong), ("c_bs1", ctypes.c_longlong),
        ("alpha", ctypes.c_float), ("act", ctypes.c_int), ("relu_in", ctypes.c_int),
        ("conv_kh", ctypes.c_int), ("conv_kw", ctypes.c_int), ("conv_cin", ctypes.c_int),
        ("conv_stride", ctypes.c_int), ("conv_pad", ctypes.c_int), ("conv_h", ctypes.c_int),
        ("conv_w", ctypes.c_int), ("conv_ho", ctypes.c_int), ("conv_wo", ctypes.c_int),
        ("conv_bstride", ctypes.c_longlong),
        ("shuffle_r", ctypes.c_int), ("shuffle_h", ctypes.c_int), ("shuffle_w", ctypes.c_int),
        ("prec", ctypes.c_int), ("B_hl", ctypes.c_void_p), ("b_scale", ctypes.c_float), ("A_hl", ctypes.c_void_p),
        ("a_hl_bytes", ctypes.c_longlong), ("b_hl_bytes", ctypes.c_longlong),
        ("C_hl", ctypes.c_void_p), ("ldc_h", ctypes.c_int), ("c_relu", ctypes.c_int),
        ("alpha_dev", ctypes.c_void_p), ("alpha_dev2", ctypes.c_void_p),
        ("ksplit", ctypes.c_int), ("ks_rows", ctypes.c_int), ("grp_rows", ctypes.c_int), ("grp_b_bytes", ctypes.c_longlong),
    ]


class PicoPoseHipError(RuntimeError):
    pass


def declared_symbols():
    """Every function name declared in include/picopose_hip.h."""
    text = open(_HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pp_[a-z0-9_]+)\s*\(", text)))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            raise PicoPoseHipError(
                f"{LIB} not found: build it with `python -m picopose_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback."
            )
        # torch first: PyTorch-ROCm bundles its own libamdhip64, and this library must bind to THAT runtime (it launches on
        # torch's streams and takes torch's device pointers).  Loaded before torch, it would pull in /opt/rocm's copy — two
        # HIP runtimes in one process, and every launch fails (seen on the GPU box with build() followed by smoke()).
        import torch  # noqa: F401

        L = ctypes.CDLL(LIB)
        c = ctypes
        vp, i32, f32, sz = c.c_void_p, c.c_int, c.c_float, c.c_size_t
        L.pp_strerror.restype = c.c_char_p
        L.pp_strerror.argtypes = [i32]
        L.pp_version.restype = i32
        L.pp_prof_enable.argtypes = [i32]
        L.pp_prof_collect.argtypes = [c.POINTER(f32), i32, c.POINTER(i32)]
        L.pp_prof_gemm_enable.argtypes = [i32]
        L.pp_prof_gemm_collect.argtypes = [c.POINTER(c.c_double), c.POINTER(c.c_double), c.POINTER(i32)]
        L.pp_prof_gemm_records.argtypes = [i32, c.POINTER(i32), c.POINTER(f32), c.POINTER(c.c_double), c.POINTER(i32)]
        L.pp_prof_gemm_records2.argtypes = [i32, c.POINTER(i32), c.POINTER(f32), c.POINTER(c.c_double), c.POINTER(c.c_double), c.POINTER(i32)]
        L.pp_gemm_tune_save.argtypes = [c.c_char_p]
        L.pp_gemm_tune_load.argtypes = [c.c_char_p]
        L.pp_gemm_tune_entries.argtypes = []
        L.pp_stage1_workspace_bytes.argtypes = [i32, i32, i32, c.POINTER(sz)]
        L.pp_stage1_scores.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, vp, sz, vp, vp, vp]
        L.pp_stage1_scores_ex.argtypes = [vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, f32, vp, sz, vp, vp, vp]
        L.pp_topk.argtypes = [vp, i32, i32, i32, vp, vp, vp]
        L.pp_stage1_match.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, vp, sz,
                                      vp, vp, vp, vp, vp]
        L.pp_stage1_match_ex.argtypes = [vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, vp, sz, vp, vp, vp, vp, vp]
        L.pp_similarity_volume.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp, vp]
        L.pp_calc_pred_Ms.argtypes = [vp, vp, vp, vp, vp, vp, i32, f32, vp, vp]
        L.pp_pose_recovery_2d.argtypes = [vp, vp, vp, vp, vp, vp, i32, vp, vp]
        L.pp_init_correspondences.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp]
        L.pp_stage3_correspondences.argtypes = [vp, vp, i32, i32, i32, f32, vp, vp, vp]
        L.pp_gather_valid.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp, vp]
        L.pp_gemm.argtypes = [c.POINTER(PpGemmDesc), vp]
        L.pp_split_f16x3.argtypes = [vp, c.c_longlong, vp, vp, vp]
        L.pp_split_activation.argtypes = [vp, c.c_longlong, i32, i32, i32, i32, i32, vp, vp]
        L.pp_split_activation_ld.argtypes = [vp, c.c_longlong, i32, i32, i32, i32, i32, vp, i32, vp]
        L.pp_conv_narrow_hl.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp, i32, i32, vp, vp, vp]
        L.pp_conv_narrow_f32.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp, i32, i32, vp, vp, vp]
        L.pp_winograd_input_f32.argtypes = [vp, i32, c.c_longlong, i32, i32, i32, i32, i32, vp, vp]
        L.pp_winograd_weight_f32.argtypes = [vp, i32, i32, i32, vp, vp]
        L.pp_winograd_output_f32.argtypes = [vp, i32, i32, i32, i32, vp, i32, vp, vp, vp, i32, vp]
        L.pp_winograd4_input_hl.argtypes = [vp, i32, c.c_longlong, i32, i32, i32, i32, i32, vp, c.c_longlong, vp]
        L.pp_winograd4_weight_f32.argtypes = [vp, i32, i32, i32, vp, vp]
        L.pp_winograd4_output.argtypes = [vp, i32, i32, i32, i32, i32, vp, i32, vp, vp, vp, i32, vp, i32, i32, c.c_longlong, vp]
        L.pp_winograd4_chain.argtypes = [vp, i32, i32, i32, i32, i32, vp, i32, i32, vp, c.c_longlong, vp]
        L.pp_winograd_chain_f32.argtypes = [vp, i32, i32, i32, i32, vp, i32, i32, vp, vp]
        L.pp_set_saturation_word.argtypes = [vp]
        L.pp_split_weights_t.argtypes = [vp, c.c_longlong, i32, vp, vp, vp]
        L.pp_split_weights_ws.argtypes = [vp, c.c_longlong, i32, vp, vp, vp, vp]
        L.pp_split_activation_t.argtypes = [vp, c.c_longlong, i32, i32, i32, i32, i32, vp, i32, i32, vp]
        L.pp_hl_patch_columns_t.argtypes = [vp, i32, i32, c.c_longlong, vp, i32, i32, i32, vp]
        L.pp_layernorm_t.argtypes = [vp, vp, vp, i32, i32, f32, vp, vp, i32, vp]
        L.pp_resize_bilinear_nhwc_t.argtypes = [vp, i32, i32, i32, i32, i32, i32, f32, vp, i32, vp]
        L.pp_resize_bilinear_nhwc_dual.argtypes = [vp, i32, i32, i32, i32, i32, i32, f32, vp, vp, i32, vp]
        L.pp_warp_nhwc_t.argtypes = [vp, i32, vp, i32, i32, i32, i32, i32, vp, i32, i32, vp]
        L.pp_attention_t.argtypes = [vp, i32, i32, i32, i32, i32, f32, vp, vp, vp]
        L.pp_sum_slices.argtypes = [vp, i32, i32, i32, vp, i32, vp, vp]
        L.pp_hl_patch_columns.argtypes = [vp, i32, i32, c.c_longlong, vp, i32, i32, vp]
        L.pp_warp_nhwc_hl.argtypes = [vp, i32, vp, i32, i32, i32, i32, i32, vp, i32, vp]
        L.pp_attention.argtypes = [vp, i32, i32, i32, i32, f32, vp, vp]
        L.pp_attention_split.argtypes = [vp, i32, i32, i32, i32, f32, vp, vp, vp]
        L.pp_attention_ex.argtypes = [vp, i32, i32, i32, i32, f32, i32, vp, vp, vp]
        L.pp_attention_hl.argtypes = [vp, i32, i32, i32, i32, f32, vp, vp, vp]
        L.pp_attention_train.argtypes = [vp, i32, i32, i32, i32, f32, vp, vp, vp]
        L.pp_attention_backward.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp, vp, vp]
        L.pp_layernorm.argtypes = [vp, vp, vp, i32, i32, f32, vp, vp]
        L.pp_layernorm_split.argtypes = [vp, vp, vp, i32, i32, f32, vp, vp, vp]
        L.pp_softmax_rows.argtypes = [vp, i32, i32, i32, vp]
        L.pp_groupnorm_nhwc.argtypes = [vp, vp, vp, i32, i32, i32, i32, f32, i32, vp, vp]
        L.pp_transpose_batched.argtypes = [vp, c.c_longlong, i32, i32, i32, vp, c.c_longlong, i32, i32, vp]
        L.pp_assemble_tokens.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp]
        L.pp_normalize_rows.argtypes = [vp, i32, i32, f32, vp, vp]
        L.pp_resize_bilinear_nhwc.argtypes = [vp, i32, i32, i32, i32, i32, i32, f32, vp, vp]
        L.pp_resize_bilinear_nhwc_hl.argtypes = [vp, i32, i32, i32, i32, i32, i32, f32, vp, vp]
        L.pp_warp_nhwc.argtypes = [vp, i32, vp, i32, i32, i32, i32, i32, vp, i32, vp]
        L.pp_avgpool2_nhwc.argtypes = [vp, i32, i32, i32, i32, vp, vp]
        L.pp_gather_rows.argtypes = [vp, vp, c.c_longlong, c.c_longlong, i32, vp, vp]
        L.pp_depth_points_nearest.argtypes = [vp, i32, i32, i32, i32, i32, i32, i32, f32, f32, f32, f32, vp, vp]
        L.pp_crop_resize_normalize.argtypes = [vp, i32, i32, vp, i32, i32, i32, i32, i32, i32, c.POINTER(c.c_double),
                                               c.POINTER(c.c_double), vp, vp, vp]
        L.pp_corr_lookup_nhwc.argtypes = [vp, i32, vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, vp, i32, vp]
        L.pp_corr_lookup_nhwc_ex.argtypes = [vp, i32, vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, i32, vp]
        L.pp_corr_lookup_nhwc_hl.argtypes = [vp, i32, vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, vp, i32, vp]
        L.pp_pnp_ransac.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, vp, vp, vp, vp, vp, vp]
        L.pp_pnp_ransac_debug.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, vp, vp, vp, vp, vp, vp, vp]
        L.pp_train_keypoints_workspace_bytes.restype = sz
        L.pp_train_keypoints_workspace_bytes.argtypes = [i32]
        L.pp_train_keypoints.argtypes = [vp, vp, i32, i32, vp, vp, i32, i32] + [vp] * 10 + [i32, vp, vp, vp, sz, vp]
        L.pp_batchnorm_train_workspace_bytes.restype = sz
        L.pp_batchnorm_train_workspace_bytes.argtypes = [i32, i32]
        L.pp_batchnorm_train.argtypes = [vp, vp, vp, i32, i32, f32, f32, vp, vp, i32, vp, vp, vp, vp, sz, vp]
        L.pp_gather_normalize_rows.argtypes = [vp, c.c_longlong, vp, i32, i32, f32, vp, vp]
        L.pp_xent_diag_rows.argtypes = [vp, i32, i32, f32, vp, vp]
        L.pp_flow_loss_blocks.argtypes = []
        L.pp_flow_loss_sums.argtypes = [vp, vp, vp, i32, i32, i32, f32, vp, vp]
        ll = c.c_longlong
        L.pp_colsum_workspace_bytes.restype = sz
        L.pp_colsum_workspace_bytes.argtypes = [ll, i32]
        L.pp_colsum.argtypes = [vp, ll, i32, i32, vp, vp, sz, vp]
        L.pp_act_forward.argtypes = [vp, ll, i32, vp, vp]
        L.pp_act_backward.argtypes = [vp, vp, ll, i32, vp, vp]
        L.pp_elementwise.argtypes = [i32, vp, vp, ll, i32, vp, vp]
        L.pp_layernorm_backward.argtypes = [vp, vp, vp, i32, i32, f32, vp, vp, vp]
        L.pp_groupnorm_backward_nhwc.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp, vp, vp, vp]
        L.pp_softmax_backward_rows.argtypes = [vp, vp, ll, i32, vp, vp]
        L.pp_xent_diag_backward.argtypes = [vp, i32, i32, f32, vp, vp, vp]
        L.pp_normalize_rows_backward.argtypes = [vp, ll, vp, vp, i32, i32, f32, vp, vp]
        L.pp_scatter_add_rows.argtypes = [vp, vp, i32, i32, vp, vp]
        L.pp_im2col_nhwc.argtypes = [vp, i32, i32, i32, i32, i32, i32, i32, vp, vp]
        L.pp_col2im_nhwc.argtypes = [vp, i32, i32, i32, i32, i32, i32, i32, vp, vp]
        L.pp_simvol_backward.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp]
        L.pp_im2col_t_nhwc.argtypes = [vp, i32, i32, i32, i32, i32, i32, i32, vp, vp]
        L.pp_pow2_scale.argtypes = [vp, ll, vp, vp]
        L.pp_pow2_scale_ws.argtypes = [vp, ll, vp, vp, vp]
        L.pp_split_scaled_t.argtypes = [vp, ll, i32, i32, vp, vp, i32, vp]
        L.pp_split_transpose_t.argtypes = [vp, ll, i32, i32, vp, vp, i32, vp]
        L.pp_split_transpose_ld.argtypes = [vp, ll, i32, i32, vp, vp, ll, i32, vp]
        L.pp_im2col_t_operand.argtypes = [vp, i32, i32, i32, i32, i32, i32, i32, vp, i32, vp]
        L.pp_split_with_scale_t.argtypes = [vp, ll, i32, vp, vp, vp]
        L.pp_batchnorm_train_backward_workspace_bytes.restype = sz
        L.pp_batchnorm_train_backward_workspace_bytes.argtypes = [ll, i32]
        L.pp_batchnorm_train_backward.argtypes = [vp, vp, vp, vp, ll, i32, f32, i32, vp, vp, vp, vp, sz, vp]
        L.pp_resize_bilinear_backward_nhwc.argtypes = [vp, i32, i32, i32, i32, i32, i32, f32, vp, vp]
        L.pp_avgpool2_backward_nhwc.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp]
        L.pp_warp_backward_nhwc.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp]
        L.pp_corr_lookup_backward_nhwc.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp]
        L.pp_warp_backward_nhwc_fixed.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp]
        L.pp_corr_lookup_backward_nhwc_fixed.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp]
        L.pp_fixed_to_float.argtypes = [vp, ll, vp, vp]
        L.pp_flow_loss_backward.argtypes = [vp, vp, vp, i32, i32, i32, f32, vp, vp, vp, vp, vp]
        _lib = L
    return _lib


def stream_ptr():
    import torch

    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def dev_f32(*tensors):
    """Contiguous fp32 device views of the inputs (the ABI takes raw device pointers)."""
    out = []
    for t in tensors:
        if not t.is_cuda:
            raise PicoPoseHipError("picopose_amd runs on the GPU only: inputs must be CUDA(HIP) tensors")
        out.append(t.contiguous().float())
    return out


def check(rc, what):
    if rc != PP_OK:
        raise PicoPoseHipError(f"{what} failed: {lib().pp_strerror(rc).decode()} (code {rc})")
