"""Host mirror of the reference's utils/augment.py (training forward): the noisy ground-truth affine that stands in for
the stage-2 prediction while stage 3 trains.  B-sized torch algebra on the inputs' device, random draws in the reference's
order (numpy choice, torch normal), scipy for the Euler angle exactly as the reference does."""
import numpy as np
import torch


def cosSin(angle):
    return torch.stack([torch.cos(angle), torch.sin(angle)], dim=1)


def _centre_in_crop(K, pose, M):
    c = K @ pose[:, :3, 3:4]
    return M @ (c / c[:, 2].unsqueeze(2))


def get_relative_scale_inplane(src_K, tar_K, src_pose, tar_pose, src_M, tar_M):
    """utils/torch_utils.py:168-183: scale(src -> tar) = (z_src / z_tar) (crop scale ratio) / (focal ratio); in-plane angle =
    first 'zxy' Euler angle of R_tar R_src^T, in [0, 2 pi)."""
    from scipy.spatial.transform import Rotation

    scale = (src_pose[:, 2, 3] / tar_pose[:, 2, 3]) * (torch.norm(tar_M[:, :2, 0], dim=1) / torch.norm(src_M[:, :2, 0], dim=1)) \
        / (src_K[:, 0, 0] / tar_K[:, 0, 0])
    relR = tar_pose[:, :3, :3] @ src_pose[:, :3, :3].transpose(1, 2)
    ang = torch.from_numpy(Rotation.from_matrix(relR.cpu().numpy()).as_euler("zxy")[:, 0]).float().to(relR.device)
    return scale, (ang + 2 * torch.pi) % (2 * torch.pi)


def calc_gt_trans_scale_inplane(end_points):
    """utils/torch_utils.py:17-37 -> (2-D translation in crop pixels (B,2), relative scale (B,), in-plane angle (B,))."""
    scale, ang = get_relative_scale_inplane(end_points["tem_K"], end_points["real_K"], end_points["tem_pose"], end_points["real_pose"],
                                            end_points["tem_M"], end_points["real_M"])
    d = _centre_in_crop(end_points["real_K"], end_points["real_pose"], end_points["real_M"]) - \
        _centre_in_crop(end_points["tem_K"], end_points["tem_pose"], end_points["tem_M"])
    return d[:, :2].squeeze(-1), scale, ang


def _similarity(cos_sin, scale, translation=None):
    c, s = cos_sin[:, 0], cos_sin[:, 1]
    M = torch.eye(3, device=scale.device, dtype=scale.dtype).repeat(scale.shape[0], 1, 1)
    M[:, :2, :2] = torch.stack([c, -s, s, c], dim=1).reshape(-1, 2, 2) * scale[:, None, None]
    if translation is not None:
        M[:, :2, 2] = translation
    return M


def get_relative_M(src_K, tar_K, src_pose, tar_pose, src_M, tar_M):
    """utils/torch_utils.py:195-226: the ground-truth src-crop -> tar-crop affine (object centre mapped onto object centre)."""
    scale, ang = get_relative_scale_inplane(src_K, tar_K, src_pose, tar_pose, src_M, tar_M)
    M = _similarity(cosSin(ang), scale)
    src_c = _centre_in_crop(src_K, src_pose, src_M)[:, :2, 0]
    dst_c = _centre_in_crop(tar_K, tar_pose, tar_M)[:, :2, 0]
    h = torch.cat([src_c, torch.ones_like(src_c[:, :1])], dim=1)
    moved = torch.einsum("bhc,bc->bh", M, h)
    M[:, :2, 2] = dst_c - moved[:, :2] / moved[:, 2:]
    return M


def aug_M_noise(gt_Ms, std_scales=(0.01, 0.05, 0.1, 0.15, 0.2), min_scales=0.5, max_scales=1.5, std_rots=(1, 2, 5, 10, 15), max_rot=45,
                std_trans=(2, 5, 10, 15, 20), max_trans=56):
    """utils/augment.py:6-44.  (The scale factor is clamped to [-min_scales, max_scales], as the reference writes it.)"""
    B, dev = gt_Ms.size(0), gt_Ms.device
    s0 = torch.norm(gt_Ms[:, 0, :2], dim=1)
    rot0 = torch.acos(gt_Ms[:, 0, 0] / s0)
    k = torch.normal(mean=torch.ones([B]).to(dev), std=torch.tensor(np.random.choice(list(std_scales)), device=dev))
    s = s0 * k.clamp(min=-min_scales, max=max_scales)
    r = torch.normal(mean=0, std=np.random.choice(list(std_rots)), size=(B,)).to(device=dev)
    rot = rot0 + (r.clamp(min=-max_rot, max=max_rot) / 180) * torch.pi
    st = np.random.choice(list(std_trans))
    t = torch.normal(mean=torch.zeros([B, 2]).to(dev), std=torch.tensor([st, st], device=dev).view(1, 2))
    t = gt_Ms[:, :2, 2] + torch.clamp(t, min=-max_trans, max=max_trans)
    return _similarity(cosSin((rot + 2 * torch.pi) % (2 * torch.pi)), s, t).detach()


def aug_gtM_noise(end_points):
    """utils/augment.py:46-55."""
    return aug_M_noise(get_relative_M(src_K=end_points["tem_K"], tar_K=end_points["real_K"], src_pose=end_points["tem_pose"],
                                      tar_pose=end_points["real_pose"], src_M=end_points["tem_M"], tar_M=end_points["real_M"]))
