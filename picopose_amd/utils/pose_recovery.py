"""Host mirror of reference utils/pose_recovery.py (HIP through the C ABI)."""
import torch

from .. import _lib


def pose_recovery_2d_prediction(query_M, query_K, pred_Ms, template_K, template_Ms, template_poses):
    """Drop-in for reference utils/pose_recovery.py:9-65 -> pred_poses (B,4,4).

    The reference asserts (with a host sync) that query_M is a crop affine (M01 = M10 = 0,
    M00 = M11, torch_utils.py:100-101); here that is the caller's contract."""
    qM, qK, pM, tK, tM, tp = _lib.dev_f32(query_M, query_K, pred_Ms, template_K, template_Ms, template_poses)
    B = qM.shape[0]
    out = torch.empty(B, 4, 4, dtype=torch.float32, device=qM.device)
    rc = _lib.lib().pp_pose_recovery_2d(qM.data_ptr(), qK.data_ptr(), pM.data_ptr(), tK.data_ptr(), tM.data_ptr(),
                                        tp.data_ptr(), B, out.data_ptr(), _lib.stream_ptr())
    _lib.check(rc, "pp_pose_recovery_2d")
    return out
