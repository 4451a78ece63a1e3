"""Oracle for the training forward (TEST INFRASTRUCTURE only): plain-torch CPU restatement of

  model/picopose.py:29-50       Net.compute_keypoint_data
  utils/keypoints.py:120-205    KeyPointSampler.sample_pts (Keypoint.mask :47-68, .apply_affine :85-92,
                                .apply_3D_transform :79-83; torch_utils.py unproject_points :138-151, project_points :154-161)
  utils/loss_utils.py:144-175   compute_stage_one_loss (InfoNCE over gathered patch features, torch_utils.gather :257-284)
  utils/loss_utils.py:177-186   compute_stage_two_loss (TranslationLoss l1, ScaleLoss log-l2, InplaneLoss geodesic;
                                torch_utils.py calc_gt_trans_scale_inplane :17-37, get_relative_scale_inplane :168-183)
  utils/augment.py:6-55         aug_M_noise / aug_gtM_noise (torch_utils.get_relative_M :195-226)
  utils/loss_utils.py:188-202   compute_stage_three_loss (compute_flow_loss :119-125, RAFTLoss :24-39)
  utils/loss_utils.py:10-21     Loss.forward (sum of the `loss*` entries, clamp at 100)
  model/picopose.py:114-137     Net.forward_train on the oracle networks (oracle/nets.py) with BatchNorm in training mode

Pinned by tests/golden/train_forward.npz: outputs of the reference's own `Net.forward_train` / `Loss` (oracle/gen_golden.py,
gen_train_forward).  The reference evaluates several quirks literally and so does this file: grid "x" is the slowly varying
coordinate; key-points are truncated to integer pixels before they are mapped back to the image; the visibility test compares
re-projected points in CROP pixels with grid points in IMAGE pixels against a 1000-pixel threshold."""

import numpy as np
import torch
import torch.nn.functional as F

GRID = 64           # key-point grid per side (224 / 3.5)
CELL = 3.5          # keypoints.py:97 patch_size
FAR = 1e6           # keypoints.py:10 MAX_VALUES


# ------------------------------------------------------------------------------------------ key-point sampler
def grid_points():
    """keypoints.py:100-111: point n = (g[n // 64], g[n % 64]); column 0 is later USED as x."""
    g = torch.arange(0, 224, CELL).float() + CELL / 2
    return torch.stack([g.repeat_interleave(GRID), g.repeat(GRID)], dim=1)


def _mask_points(pts, mask):
    """Keypoint.mask: truncate to integer pixels, drop points outside the image or on mask < 0.5 -> int64 with -1."""
    p = pts.long()
    H, W = mask.shape[-2:]
    out = (p[..., 0] < 0) | (p[..., 1] < 0) | (p[..., 0] >= W) | (p[..., 1] >= H)
    q = p.clone()
    q[out] = 0
    b = torch.arange(p.shape[0])[:, None].expand(-1, p.shape[1])
    out = out | (mask[b, q[..., 1], q[..., 0]] < 0.5)
    p[out] = -1
    return p


def _affine_points(T, pts):
    """Keypoint.apply_affine: homogeneous 3x3 map of (B,N,2); entries whose x is exactly -1 stay -1."""
    gone = pts[..., 0] == -1
    h = torch.cat([pts.float(), torch.ones(*pts.shape[:2], 1)], dim=2)
    m = (T @ h.transpose(1, 2)).transpose(1, 2)
    out = m[..., :2] / m[..., 2:]
    out[gone] = -1
    return out


def _inverse_crop_affine(M):
    s = M[:, 0, 0]
    inv = torch.eye(3).repeat(M.shape[0], 1, 1)
    inv[:, 0, 0] = 1 / s
    inv[:, 1, 1] = 1 / s
    inv[:, :2, 2] = -M[:, :2, 2] / s[:, None]
    return inv


def _unproject(pts, K, depth):
    """unproject_points: clamps `pts` IN PLACE to the depth image, reads depth at the truncated pixel."""
    pts[..., 1] = pts[..., 1].clamp(0, depth.shape[1] - 1)
    pts[..., 0] = pts[..., 0].clamp(0, depth.shape[2] - 1)
    b = torch.arange(pts.shape[0])[:, None].expand(-1, pts.shape[1])
    z = depth[b, pts[..., 1].long(), pts[..., 0].long()]
    h = torch.cat([pts, torch.ones(*pts.shape[:2], 1)], dim=2).float()
    ray = (torch.inverse(K).float() @ h.transpose(1, 2)).transpose(1, 2)
    return ray * z[..., None]


def _rigid(T, p3):
    h = torch.cat([p3, torch.ones(*p3.shape[:2], 1)], dim=2)
    return (T @ h.transpose(1, 2)).transpose(1, 2)[..., :3]


def _project(p3, K):
    q = (K @ p3.transpose(1, 2)).transpose(1, 2)
    return q[..., :2] / q[..., 2:]


def keypoint_data(end_points):
    """-> {"src_pts", "tar_pts"}: (B,4096,2) float patch coordinates (pixels / 3.5) of the template key-points and of their
    re-projections into the real crop, -1 where invalid."""
    rel = end_points["tem_pose"] @ torch.inverse(end_points["real_pose"])        # real -> template
    T_src2tar, T_tar2src = torch.inverse(rel), rel
    sK, sM, smask, sdepth = (end_points[k] for k in ("tem_K", "tem_M", "tem_mask", "tem_full_depth"))
    tK, tM, tmask, tdepth = (end_points[k] for k in ("real_K", "real_M", "real_mask", "real_full_depth"))
    B = smask.shape[0]
    init = grid_points()[None].repeat(B, 1, 1)
    src_crop, tar_crop = _mask_points(init, smask), _mask_points(init, tmask)     # int64, crop pixels
    src_img = _affine_points(_inverse_crop_affine(sM), src_crop)                 # image pixels (float)
    tar_img = _affine_points(_inverse_crop_affine(tM), tar_crop)
    src3 = _rigid(T_src2tar, _unproject(src_img, sK, sdepth))                     # (src_img / tar_img are clamped in place)
    tar3 = _rigid(T_tar2src, _unproject(tar_img, tK, tdepth))
    re_src = _mask_points(_affine_points(tM, _project(src3, tK)), tmask)          # template points seen in the real crop
    re_tar = _mask_points(_affine_points(sM, _project(tar3, sK)), smask)
    bad_tar = (tar_crop[..., 0] == -1) | (re_tar[..., 0] == -1)
    bad_src = (src_crop[..., 0] == -1) | (re_src[..., 0] == -1)
    for b in range(B):
        d = torch.cdist(re_src[b].float(), tar_img[b].float())
        d[bad_src[b]] = FAR
        d[:, bad_tar[b]] = FAR
        keep = d.min(dim=1).values < 1000.0
        re_src[b, ~keep] = -1
        src_crop[b, ~keep] = -1

    def patches(p):
        out = p / CELL
        out[p[..., 0] == -1] = -1
        return out

    return {"src_pts": patches(src_crop), "tar_pts": patches(re_src)}


# ------------------------------------------------------------------------------------------ losses
def _gather_valid(feat, pts):
    """torch_utils.gather: rows feat[b, :, y, x] of the entries with x != -1 and y != -1, batch-major."""
    B, C, H, W = feat.shape
    x, y = pts[..., 0], pts[..., 1]
    ok = (x != -1) & (y != -1)
    idx = torch.where(ok, y * W + x, torch.zeros_like(x))
    rows = feat.reshape(B, C, H * W).transpose(1, 2)
    return torch.gather(rows, 1, idx[..., None].expand(-1, -1, C))[ok]


def _to_feature_grid(pts, h):
    """loss_utils.py:148-155: (B,4096,2) patch coordinates -> nearest-sampled (B,h*h,2) feature-map coordinates, -1 kept."""
    B = pts.shape[0]
    p = pts.reshape(B, GRID, GRID, 2)
    gone = F.interpolate((p[..., 0] == -1).float()[:, None], size=(h, h), mode="nearest")[:, 0].bool()
    q = (h / GRID) * F.interpolate(p.permute(0, 3, 1, 2), size=(h, h), mode="nearest").permute(0, 2, 3, 1)
    q[gone] = -1
    return q.reshape(B, -1, 2)


def stage_one_loss(src_feat, tar_feat, src_pts, tar_pts, tau=0.1):
    h = src_feat.shape[2]
    a = _gather_valid(src_feat, _to_feature_grid(src_pts, h).long())
    b = _gather_valid(tar_feat, _to_feature_grid(tar_pts, h).long())
    logits = F.normalize(a, dim=1) @ F.normalize(b, dim=1).t() / tau
    return F.cross_entropy(logits, torch.arange(a.shape[0]))


def _centre_in_crop(K, pose, M):
    c = K @ pose[:, :3, 3:4]
    return M @ (c / c[:, 2:3])


def relative_scale_inplane(end_points):
    """get_relative_scale_inplane with src = template, tar = real: scale and in-plane angle in [0, 2 pi)."""
    from scipy.spatial.transform import Rotation

    sp, tp = end_points["tem_pose"], end_points["real_pose"]
    crop = torch.norm(end_points["real_M"][:, :2, 0], dim=1) / torch.norm(end_points["tem_M"][:, :2, 0], dim=1)
    scale = (sp[:, 2, 3] / tp[:, 2, 3]) * crop / (end_points["tem_K"][:, 0, 0] / end_points["real_K"][:, 0, 0])
    relR = tp[:, :3, :3] @ sp[:, :3, :3].transpose(1, 2)
    ang = torch.from_numpy(Rotation.from_matrix(relR.numpy()).as_euler("zxy")[:, 0]).float()
    return scale, (ang + 2 * torch.pi) % (2 * torch.pi)


def stage_two_targets(end_points):
    """calc_gt_trans_scale_inplane: 2-D translation (crop pixels), relative scale, relative in-plane angle."""
    scale, ang = relative_scale_inplane(end_points)
    d = _centre_in_crop(end_points["real_K"], end_points["real_pose"], end_points["real_M"]) - \
        _centre_in_crop(end_points["tem_K"], end_points["tem_pose"], end_points["tem_M"])
    return d[:, :2, 0], scale, ang


def geodesic(pred_cos_sin, gt_angle, eps=1e-6):
    c = pred_cos_sin[:, 0] * torch.cos(gt_angle) + pred_cos_sin[:, 1] * torch.sin(gt_angle)
    return torch.acos(c.clamp(-1 + eps, 1 - eps)).mean()


def stage_two_loss(end_points, pred_translation, pred_scale, pred_inplane, trans_scale=14):
    t, s, a = stage_two_targets(end_points)
    l_t = (pred_translation - t / trans_scale).abs().mean()
    l_s = ((torch.log(pred_scale.clamp(min=5e-3)) - torch.log(s)) ** 2).mean()
    return l_t, l_s, geodesic(pred_inplane, a)


def relative_M(end_points):
    """get_relative_M: the ground-truth template-crop -> real-crop affine (scale, in-plane rotation, centre to centre)."""
    scale, ang = relative_scale_inplane(end_points)
    B = scale.shape[0]
    c, s = torch.cos(ang), torch.sin(ang)
    M = torch.eye(3).repeat(B, 1, 1)
    M[:, :2, :2] = torch.stack([c, -s, s, c], dim=1).reshape(B, 2, 2) * scale[:, None, None]
    src_c = _centre_in_crop(end_points["tem_K"], end_points["tem_pose"], end_points["tem_M"])[:, :2, 0]
    dst_c = _centre_in_crop(end_points["real_K"], end_points["real_pose"], end_points["real_M"])[:, :2, 0]
    moved = torch.einsum("bhc,bc->bh", M, torch.cat([src_c, torch.ones(B, 1)], dim=1))
    M[:, :2, 2] = dst_c - moved[:, :2] / moved[:, 2:]
    return M


def draw_noise(B):
    """The random draws of aug_M_noise in the reference's order (numpy choice, torch normal — CPU generators)."""
    std_scale = np.random.choice([0.01, 0.05, 0.1, 0.15, 0.2])
    k_scale = torch.normal(mean=torch.ones(B), std=torch.tensor(std_scale))
    std_rot = np.random.choice([1, 2, 5, 10, 15])
    k_rot = torch.normal(mean=0, std=std_rot, size=(B,))
    std_tran = np.random.choice([2, 5, 10, 15, 20])
    k_tran = torch.normal(mean=torch.zeros(B, 2), std=torch.tensor([std_tran, std_tran]).view(1, 2))
    return k_scale, k_rot, k_tran


def noisy_M(gt_M, k_scale, k_rot, k_tran, max_scales=1.5, min_scales=0.5, max_rot=45, max_trans=56):
    """aug_M_noise given its draws.  (The scale factor is clamped to [-0.5, 1.5] — `min=-min_scales` in the reference.)"""
    s0 = torch.norm(gt_M[:, 0, :2], dim=1)
    rot0 = torch.acos(gt_M[:, 0, 0] / s0)
    s = s0 * k_scale.clamp(min=-min_scales, max=max_scales)
    rot = rot0 + (k_rot.clamp(min=-max_rot, max=max_rot) / 180) * torch.pi
    t = gt_M[:, :2, 2] + k_tran.clamp(min=-max_trans, max=max_trans)
    a = (rot + 2 * torch.pi) % (2 * torch.pi)
    c, sn = torch.cos(a), torch.sin(a)
    M = torch.eye(3).repeat(gt_M.shape[0], 1, 1)
    M[:, :2, :2] = torch.stack([c, -sn, sn, c], dim=1).reshape(-1, 2, 2) * s[:, None, None]
    M[:, :2, 2] = t
    return M


def stage_three_loss(pred_flow, pred_certainty, tar_pts, max_flow=400, eps=1e-10):
    """-> [(loss_flow_l, loss_certainty_l)] per level.  tar_pts (B,4096,2) patch coordinates; the reference swaps the grid
    axes ('b (h w) c -> b w h c') because point n = (slow x, fast y)."""
    B = tar_pts.shape[0]
    pts = tar_pts.reshape(B, GRID, GRID, 2).transpose(1, 2)
    valid = ((pts[..., 0] != -1) & (pts[..., 1] != -1)).float()
    out = []
    for flow, cert in zip(pred_flow, pred_certainty):
        H, W = flow.shape[2:]
        ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
        grid = torch.stack([xs, ys], dim=-1).float()[None]
        gt_c = F.interpolate(valid[:, None], size=(H, W), mode="nearest")[:, 0].bool()
        gt_f = (H / GRID) * F.interpolate(pts.permute(0, 3, 1, 2), size=(H, W), mode="nearest").permute(0, 2, 3, 1)
        gt_f = gt_f * gt_c[..., None] - grid
        l_c = 1.0 * F.binary_cross_entropy_with_logits(cert[:, 0], gt_c.float())
        g = gt_f.permute(0, 3, 1, 2)
        w = (gt_c & (g.pow(2).sum(dim=1).sqrt() < max_flow)).float()
        l_f = 0.1 * (w[:, None] * (flow - g).abs()).sum() / (w.sum() + eps)
        out.append((l_f, l_c))
    return out


def total_loss(losses):
    """Loss.forward: `loss` = mean(clamp(sum of every entry, max=100)); every entry is also reported as its mean."""
    tot = sum(losses.values())
    return torch.clamp(tot, max=100.0).mean()


# ------------------------------------------------------------------------------------------ the training forward
def net_forward_train(sd, end_points, heads, blocks_to_take, pred_Ms, num_levels=3, radius=4):
    """model/picopose.py:114-137 on the oracle networks.  BatchNorm layers normalise with the statistics of their batch and
    update `sd`'s running buffers in place (two updates per step for the DPT head: template maps, then real maps).
    pred_Ms: the noisy ground-truth affines (noisy_M(relative_M(end_points), *draws)).  Returns (losses, aux)."""
    from . import geometry as og
    from . import matching as om
    from . import nets

    kp = keypoint_data(end_points)
    f_real = nets.vit_features(sd, end_points["real_rgb"], heads, blocks_to_take)
    f_tem = nets.vit_features(sd, end_points["tem_rgb"], heads, blocks_to_take)
    losses = {"loss_info": stage_one_loss(f_tem[-1], f_real[-1], kp["src_pts"], kp["tar_pts"])}
    sim = om.matching_features_similarity(f_tem[-1], f_real[-1], end_points["tem_mask"], None)
    t, s, ip = nets.affine_regressor(sd, sim)
    losses["loss_2d_trans"], losses["loss_scale"], losses["loss_inplane"] = stage_two_loss(end_points, t, s, ip)
    f0, c0 = og.compute_init_correspondences(pred_Ms, end_points["tem_mask"])
    fl, ce = nets.flow_decoder(sd, nets.dpt_head(sd, f_tem, train=True), nets.dpt_head(sd, f_real, train=True), f0, c0,
                               num_levels, radius, train=True)
    for i, (lf, lc) in enumerate(stage_three_loss(fl, ce, kp["tar_pts"])):
        losses[f"loss_flow{i}"], losses[f"loss_certainty{i}"] = lf, lc
    return losses, {"keypoints": kp, "flow": fl, "cert": ce, "stage2": (t, s, ip)}
