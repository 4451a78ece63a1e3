"""Host mirror of reference utils/correspondence.py (HIP through the C ABI)."""
import torch

from .. import _lib


def compute_init_correspondences(pred_Ms, tem_mask, size=(16, 16)):
    """Drop-in for reference utils/correspondence.py:10-26 -> init_flow (B,2,16,16), init_certainty (B,1,16,16)."""
    B, H, W = tem_mask.shape
    assert H == W  # reference :12
    if tuple(size) != (16, 16):
        raise _lib.PicoPoseHipError("the HIP kernel is built for the reference's 16x16 grid")
    Ms, mask = _lib.dev_f32(pred_Ms, tem_mask)
    flow = torch.empty(B, 2, 16, 16, dtype=torch.float32, device=Ms.device)
    cert = torch.empty(B, 1, 16, 16, dtype=torch.float32, device=Ms.device)
    rc = _lib.lib().pp_init_correspondences(Ms.data_ptr(), mask.data_ptr(), H, W, B, flow.data_ptr(),
                                            cert.data_ptr(), _lib.stream_ptr())
    _lib.check(rc, "pp_init_correspondences")
    return flow, cert


def compute_stage3_correspondences(pred_flow, pred_certainty, threshold=0.5):
    """Drop-in for reference utils/correspondence.py:28-59 -> (tar_pts, src_pts), each (B,H*W,2) int64."""
    flow, cert = _lib.dev_f32(pred_flow, pred_certainty)
    B, _, H, W = flow.shape
    tar = torch.empty(B, H * W, 2, dtype=torch.int64, device=flow.device)
    src = torch.empty(B, H * W, 2, dtype=torch.int64, device=flow.device)
    rc = _lib.lib().pp_stage3_correspondences(flow.data_ptr(), cert.data_ptr(), B, H, W, float(threshold),
                                              tar.data_ptr(), src.data_ptr(), _lib.stream_ptr())
    _lib.check(rc, "pp_stage3_correspondences")
    return tar, src
