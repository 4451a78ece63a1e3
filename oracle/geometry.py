"""Oracle for the affine / pose algebra and keypoint selection (TEST INFRASTRUCTURE only).

CPU restatement (plain torch, batched closed forms) of
  utils/torch_utils.py:39-51    calc_pred_Ms   (with affine_torch :53-73, apply_affine :114-135)
  utils/pose_recovery.py:9-65   pose_recovery_2d_prediction (normalize_affine_transform
                                torch_utils.py:228-240, inverse_affine :93-111)
  utils/correspondence.py:10-26 compute_init_correspondences (init_points2d_torch torch_utils.py:297-305)
  utils/correspondence.py:28-59 compute_stage3_correspondences
  utils/torch_utils.py:257-284  gather (as used at utils/pose_recovery.py:76-77)
Pinned by tests/golden/geometry.npz (reference outputs, oracle/gen_golden.py).
"""
import torch

from .matching import nearest_mask_16


def _project_center(K, pose, M):
    """K @ t, dehomogenise, then the crop affine M: template centre in crop pixels (B,3,1)."""
    t = pose[:, :3, 3:4]
    c = K @ t
    c = c / c[:, 2:3]
    return M @ c


def calc_pred_Ms(pred_scale, pred_inplane, pred_translation, tem_pose, tem_K, tem_M, trans_scale=14):
    B = pred_scale.shape[0]
    cos_t, sin_t = pred_inplane[:, 0], pred_inplane[:, 1]
    Ms = torch.zeros(B, 3, 3, dtype=pred_scale.dtype)
    Ms[:, 2, 2] = 1
    R = torch.stack([cos_t, -sin_t, sin_t, cos_t], dim=1).reshape(B, 2, 2)
    Ms[:, :2, :2] = R * pred_scale[:, None, None]
    center = _project_center(tem_K, tem_pose, tem_M)[:, :2, 0]          # (B,2)
    # apply_affine of the translation-free affine to the centre (divide by the homogeneous 1)
    h = torch.cat([center, torch.ones(B, 1)], dim=1)
    moved = torch.einsum("bhc,bc->bh", Ms, h)
    moved = moved[:, :2] / moved[:, 2:]
    target = center + pred_translation * trans_scale
    Ms[:, :2, 2] = target - moved
    return Ms


def pose_recovery_2d_prediction(query_M, query_K, pred_Ms, template_K, template_Ms, template_poses):
    B = query_M.shape[0]
    poses = template_poses.clone()
    scale = torch.norm(pred_Ms[:, :2, 0], dim=1)
    Rin = torch.zeros_like(pred_Ms)
    Rin[:, 2, 2] = 1
    Rin[:, :2, :2] = pred_Ms[:, :2, :2] / scale[:, None, None]
    poses[:, :3, :3] = Rin @ poses[:, :3, :3]
    z_tem = poses[:, 2, 3].clone()
    c = template_K @ poses[:, :3, 3:4]
    c = c / c[:, 2].unsqueeze(1)
    s = query_M[:, 0, 0]
    invM = torch.eye(3).repeat(B, 1, 1)
    invM[:, 0, 0] = 1 / s
    invM[:, 1, 1] = 1 / s
    invM[:, :2, 2] = -query_M[:, :2, 2] / s.unsqueeze(1)
    aff = (invM @ pred_Ms) @ template_Ms
    qc = aff @ c
    invK = torch.inverse(query_K)
    scale2d = torch.norm(aff[:, :2, 0], dim=1)
    focal = query_K[:, 0, 0] / template_K[:, 0, 0]
    qz = (z_tem / scale2d) * focal
    qt = (invK @ qc).squeeze(-1)
    qt = qt / qt[:, 2:3].clone()
    poses[:, :3, 3] = qt * qz.unsqueeze(-1)
    return poses


def compute_init_correspondences(pred_Ms, tem_mask, size=16):
    B, H, W = tem_mask.shape
    assert H == W
    patch = H // size
    m = nearest_mask_16(tem_mask.float(), size).reshape(B, 1, size, size)
    c = torch.arange(0, H, patch).float() + patch / 2
    # init_points2d_torch: meshgrid(y, x) 'ij', points = (yy, xx) flattened -> point k=(i*size+j) = (c[i], c[j])
    pts = torch.stack(torch.meshgrid(c, c, indexing="ij"), dim=-1).reshape(1, size * size, 2).repeat(B, 1, 1)
    h = torch.cat([pts, torch.ones(B, size * size, 1)], dim=2)
    moved = torch.einsum("bhc,bnc->bnh", pred_Ms, h)
    moved = moved[:, :, :2] / moved[:, :, 2:]
    moved = moved / patch
    # "b (w h) c -> b c h w": k = w*size + h
    moved = moved.reshape(B, size, size, 2).permute(0, 3, 2, 1)
    ys, xs = torch.meshgrid(torch.arange(size), torch.arange(size), indexing="ij")
    grid = torch.stack([xs, ys], dim=0).float()[None]
    flow = moved.float() * m - grid
    return flow, m


def compute_stage3_correspondences(pred_flow, pred_certainty, threshold=0.5):
    B, _, H, W = pred_flow.shape
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    grid = torch.stack([xs, ys], dim=-1).float()[None]                 # (1,H,W,2) = (x,y)
    tar = pred_flow.permute(0, 2, 3, 1) + grid
    inside = (tar[..., 0] > 0) & (tar[..., 1] > 0) & (tar[..., 0] < H - 1) & (tar[..., 1] < W - 1)
    keep = (pred_certainty.squeeze(1).sigmoid() > threshold) & inside   # (B,H,W)
    src = torch.stack([xs, ys], dim=-1)[None].repeat(B, 1, 1, 1).long()
    neg = torch.full((B, H, W, 2), -1, dtype=torch.long)
    src_pts = torch.where(keep[..., None], src, neg)
    tar_pts = torch.where(keep[..., None], tar.long(), neg)
    # "b h w c -> b (w h) c"
    return (tar_pts.permute(0, 2, 1, 3).reshape(B, H * W, 2), src_pts.permute(0, 2, 1, 3).reshape(B, H * W, 2))


def gather_valid(features, index_patches):
    """features (B,C,H,W), index_patches (B,N,2) (x,y) with -1 padding -> (K,C) rows, order kept."""
    B, C, H, W = features.shape
    f = features.reshape(B, C, H * W).permute(0, 2, 1)
    x, y = index_patches[..., 0], index_patches[..., 1]
    valid = (x != -1) & (y != -1)
    idx = torch.where(valid, y * W + x, torch.zeros_like(x))
    rows = torch.gather(f, 1, idx.unsqueeze(-1).expand(-1, -1, C)).reshape(-1, C)
    return rows[valid.reshape(-1)]
