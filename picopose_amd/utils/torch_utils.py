"""Host mirror of the reference's utils/torch_utils.py entries on the hot path (HIP through the C ABI)."""
import torch

from .. import _lib


def calc_pred_Ms(pred_scale, pred_inplane, pred_translation, tem_pose, tem_K, tem_M, trans_scale=14):
    """Drop-in for reference utils/torch_utils.py:39-51 -> (B,3,3)."""
    s, ip, tr, pose, K, M = _lib.dev_f32(pred_scale, pred_inplane, pred_translation, tem_pose, tem_K, tem_M)
    B = s.shape[0]
    out = torch.empty(B, 3, 3, dtype=torch.float32, device=s.device)
    rc = _lib.lib().pp_calc_pred_Ms(s.data_ptr(), ip.data_ptr(), tr.data_ptr(), pose.data_ptr(), K.data_ptr(),
                                    M.data_ptr(), B, float(trans_scale), out.data_ptr(), _lib.stream_ptr())
    _lib.check(rc, "pp_calc_pred_Ms")
    return out


def gather(features, index_patches):
    """Drop-in for reference utils/torch_utils.py:257-284: (B,C,H,W), (B,N,2) -> (K,C) valid rows in order.

    The row count is data dependent, so (exactly like the reference's boolean-mask indexing) this
    synchronises once to read it."""
    (feat,) = _lib.dev_f32(features)
    idx = index_patches.contiguous().long()
    B, C, H, W = feat.shape
    N = idx.shape[1]
    out = torch.empty(B, N, C, dtype=torch.float32, device=feat.device)
    count = torch.empty(B, dtype=torch.int32, device=feat.device)
    rc = _lib.lib().pp_gather_valid(feat.data_ptr(), idx.data_ptr(), B, C, H, W, N, out.data_ptr(),
                                    count.data_ptr(), _lib.stream_ptr())
    _lib.check(rc, "pp_gather_valid")
    counts = count.tolist()
    return torch.cat([out[b, : counts[b]] for b in range(B)], dim=0)
