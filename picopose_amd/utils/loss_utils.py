"""Host mirror of the reference's utils/loss_utils.py (training forward): same function names and
return values; the per-pixel / per-row work runs in libpicopose_hip.so (csrc/pp_train.hip, the GEMM engine for the InfoNCE
logits), the B-sized scalar algebra in torch.  Called under `torch.no_grad()` these return forward values; in a live training step
`Net.forward_train` routes the same quantities through picopose_amd/autograd.py (infonce, flow_level_losses, ...), whose Functions
carry the graph — `Loss()(end_points)["loss"].backward()` is then the reference's training step."""
import torch
import torch.nn as nn

from .. import _lib, ops
from .augment import calc_gt_trans_scale_inplane


class Loss(nn.Module):
    """utils/loss_utils.py:10-21: `loss` = mean(clamp(sum of the `loss*` entries, max=100)); each entry reported as its mean."""

    def forward(self, end_points):
        out = {"loss": 0}
        for key in end_points.keys():
            if "loss" in key:
                out[key] = end_points[key].mean()
                out["loss"] = out["loss"] + end_points[key]
        out["loss"] = torch.clamp(out["loss"], max=100.0).mean()
        return out


def _feature_grid_index(pts, h):
    """loss_utils.py:148-161: (B,4096,2) patch coordinates -> (B,h*h) row index y*h + x of the nearest-sampled grid, -1 = invalid."""
    B = pts.shape[0]
    step = 64 // h
    p = pts.reshape(B, 64, 64, 2)[:, ::step, ::step]                  # F.interpolate(mode="nearest"): source cell = dst * step
    q = ((h / 64) * p).long()
    ok = (p[..., 0] != -1) & (q[..., 0] != -1) & (q[..., 1] != -1)
    return torch.where(ok, q[..., 1] * h + q[..., 0], torch.full_like(q[..., 0], -1)).reshape(B, h * h)


def infonce_index_rows(token_shape, src_pts, tar_pts):
    """Rows (into the (B*T, C) token matrix, cls row skipped) of the key-point pairs the InfoNCE loss uses, batch-major."""
    B, T, C = token_shape
    h = int(round((T - 1) ** 0.5))
    si, ti = _feature_grid_index(src_pts, h), _feature_grid_index(tar_pts, h)
    base = (torch.arange(B, device=si.device) * T + 1)[:, None]
    s_rows, t_rows = (si + base)[si >= 0], (ti + base)[ti >= 0]        # batch-major order of the valid entries (one sync)
    if s_rows.numel() != t_rows.numel():
        raise _lib.PicoPoseHipError("key-point lists disagree on which entries are valid")
    return s_rows.contiguous(), t_rows.contiguous()


def infonce_rows(tokens_src, tokens_tar, src_pts, tar_pts, tau=0.1):
    """InfoNCE on token-major features: tokens_* (B, 1 + h*h, C) (cls row first) -> scalar loss."""
    B, T, C = tokens_src.shape
    s_rows, t_rows = infonce_index_rows(tokens_src.shape, src_pts, tar_pts)
    n = s_rows.numel()
    if n == 0:
        return torch.full((), float("nan"), device=tokens_src.device)    # F.cross_entropy of an empty batch
    L = _lib.lib()
    (ts, tt) = _lib.dev_f32(tokens_src, tokens_tar)
    q = torch.empty(n, C, dtype=torch.float32, device=ts.device)
    r = torch.empty_like(q)
    _lib.check(L.pp_gather_normalize_rows(ts.data_ptr(), C, s_rows.contiguous().data_ptr(), n, C, 1e-12, q.data_ptr(), _lib.stream_ptr()),
               "pp_gather_normalize_rows")
    _lib.check(L.pp_gather_normalize_rows(tt.data_ptr(), C, t_rows.contiguous().data_ptr(), n, C, 1e-12, r.data_ptr(), _lib.stream_ptr()),
               "pp_gather_normalize_rows")
    logits = ops.bmm_nt(q[None, None], r[None, None])[0, 0]              # (n, n) = q r^T on the GEMM engine
    rows = torch.empty(n, dtype=torch.float32, device=ts.device)
    _lib.check(L.pp_xent_diag_rows(logits.data_ptr(), n, logits.stride(0), 1.0 / tau, rows.data_ptr(), _lib.stream_ptr()), "pp_xent_diag_rows")
    return rows.mean()


def compute_stage_one_loss(src_feat, tar_feat, src_pts, tar_pts, tau=0.1):
    """Drop-in for utils/loss_utils.py:144-175 (NCHW features)."""
    def tokens(f):
        B, C, h, w = f.shape
        t = torch.zeros(B, 1 + h * w, C, dtype=torch.float32, device=f.device)
        t[:, 1:] = ops.to_nhwc(f.float()).reshape(B, h * w, C)
        return t

    return infonce_rows(tokens(src_feat), tokens(tar_feat), src_pts, tar_pts, tau)


def geodesic(pred_cos_sin, gt_angle, eps=1e-6):
    c = pred_cos_sin[:, 0] * torch.cos(gt_angle) + pred_cos_sin[:, 1] * torch.sin(gt_angle)
    return torch.acos(torch.clamp(c, -1 + eps, 1 - eps)).mean()


def compute_stage_two_loss(end_points, pred_translation, pred_scale, pred_inplane, trans_scale=14):
    """Drop-in for utils/loss_utils.py:177-186 -> (l1 translation, log-l2 scale, geodesic in-plane) losses."""
    t, s, a = calc_gt_trans_scale_inplane(end_points)
    l_t = (pred_translation - t / trans_scale).abs().mean()
    l_s = ((torch.log(pred_scale.clamp(min=5e-3)) - torch.log(s)) ** 2).mean()
    assert not torch.isnan(l_t) and not torch.isnan(l_s)
    return l_t, l_s, geodesic(pred_inplane, a)


def flow_level_losses(flow_nhwc, cert_nhwc, tar_pts, mask_weight=1.0, flow_weight=0.1, max_flow=400.0, eps=1e-10):
    """One level of compute_stage_three_loss on NHWC maps: flow (B,H,W,2), certainty (B,H,W,1) -> (loss_flow, loss_certainty)."""
    (fl, ce, tp) = _lib.dev_f32(flow_nhwc, cert_nhwc, tar_pts)
    B, H, W, _ = fl.shape
    L = _lib.lib()
    part = torch.empty(L.pp_flow_loss_blocks(), 3, dtype=torch.float64, device=fl.device)
    _lib.check(L.pp_flow_loss_sums(fl.data_ptr(), ce.data_ptr(), tp.data_ptr(), B, H, W, float(max_flow), part.data_ptr(), _lib.stream_ptr()),
               "pp_flow_loss_sums")
    bce, l1, cnt = part.sum(0)
    return (flow_weight * l1 / (cnt + eps)).float(), (mask_weight * bce / (B * H * W)).float()


def compute_stage_three_loss(end_points, pred_flow, pred_certainty, tar_pts):
    """Drop-in for utils/loss_utils.py:188-202 (NCHW lists): adds loss_flow{l} / loss_certainty{l} to end_points."""
    for idx, (flow, cert) in enumerate(zip(pred_flow, pred_certainty)):
        end_points[f"loss_flow{idx}"], end_points[f"loss_certainty{idx}"] = flow_level_losses(ops.to_nhwc(flow), ops.to_nhwc(cert), tar_pts)
    return end_points
