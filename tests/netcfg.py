"""Shared test configuration of the network fixtures (ViT-S/14: the smallest architecture the reference's
FeatureExtractor can build, feature_extractor.py:12-18)."""
import types

ns = types.SimpleNamespace


def small_cfg():
    return ns(hypothesis=5,
              stage1=ns(vit_type="dinov2_vits14", pretrained=False, interaction_indexes=[[0, 2], [3, 5], [6, 8], [9, 11]]),
              stage2=ns(in_channel=256, hidden_dim=256),
              stage3=ns(nclass=1, in_channels=384, use_bn=True, out_channels=[256, 512, 1024, 1024], num_levels=3,
                        radius=4))


HEADS = 6
TAKE = [2, 5, 8, 11]


def dome_points(K, M, size=64, crop=224, z0=0.8, relief=0.05, radius=0.4):
    """Template-camera-frame 3-D points of a shallow dome seen through the crop affine M (crop px = M . image px) with
    intrinsics K: the 64x64 lookup map `tem_pts3d` of a geometrically consistent synthetic object.  A similarity
    between template crop and query crop is then (up to the dome's small parallax) a rigid motion, so the key-point
    lists of stage 3 give PnP/RANSAC a pose to find.  K, M: (3,3) tensors -> (size, size, 3)."""
    import torch

    c = torch.arange(size, dtype=torch.float32) * (crop / size) + crop / (2 * size)
    cy, cx = torch.meshgrid(c, c, indexing="ij")
    u, v = (cx - M[0, 2]) / M[0, 0], (cy - M[1, 2]) / M[1, 1]
    r2 = ((cx - (crop - 1) / 2) ** 2 + (cy - (crop - 1) / 2) ** 2) / (radius * crop) ** 2
    z = z0 - relief * (1 - r2).clamp_min(0)
    return torch.stack([(u - K[0, 2]) / K[0, 0] * z, (v - K[1, 2]) / K[1, 1] * z, z], dim=-1)


def make_end_points(B, N, seed, feature_fn=None, tem_pose=None, dome=False):
    """Synthetic eval `end_points`: N(0,1) crops and template renders, disk masks, BOP
    intrinsics, random template rotations at z = 0.8 m, crop affines that satisfy inverse_affine's asserts.
    `template_feature` = feature_fn(tem_rgb)[-1] per template (as run_test.py:120-134 precomputes it)."""
    import torch

    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(224.0), torch.arange(224.0), indexing="ij")
    disk = (((yy - 111.5) ** 2 + (xx - 111.5) ** 2) < (0.4 * 224) ** 2).float()
    K = torch.tensor([[572.4114, 0, 320], [0, 573.57043, 240], [0, 0, 1.0]])
    ep = {}
    ep["real_rgb"] = torch.randn(B, 3, 224, 224, generator=g)
    ep["real_mask"] = disk[None].repeat(B, 1, 1)
    ep["real_K"] = K[None].repeat(B, 1, 1)
    ep["real_M"] = torch.tensor([[2.0, 0, -100.0], [0, 2.0, -80.0], [0, 0, 1.0]])[None].repeat(B, 1, 1)
    ep["real_pose"] = torch.eye(4)[None].repeat(B, 1, 1)
    c = torch.arange(64).float() * 3.5 + 1.75                      # 64x64 lookup grid of the 224 crop
    gy, gx = torch.meshgrid(c, c, indexing="ij")
    pts = torch.stack([gx, gy], dim=-1)                             # (64,64,2) crop pixels (x,y)
    ep["real_pts2d"] = ((pts - torch.tensor([-100.0, -80.0])) / 2.0)[None].repeat(B, 1, 1, 1)  # inv(real_M) applied
    ep["tem_rgb"] = torch.randn(B, N, 3, 224, 224, generator=g)
    ep["tem_mask"] = disk[None, None].repeat(B, N, 1, 1)
    ep["tem_pts3d"] = (torch.rand(B, N, 64, 64, 3, generator=g) - 0.5) * 0.2
    if dome:   # (the random map above is still drawn, so both variants share every other tensor of a seed)
        tem_M = torch.tensor([[1.5, 0, -300.0], [0, 1.5, -200.0], [0, 0, 1.0]])
        ep["tem_pts3d"] = dome_points(K, tem_M)[None, None].repeat(B, N, 1, 1, 1)
    q, _ = torch.linalg.qr(torch.randn(B, N, 3, 3, generator=g))
    q = q * torch.sign(torch.det(q))[..., None, None]
    pose = torch.eye(4)[None, None].repeat(B, N, 1, 1)
    pose[..., :3, :3] = q
    pose[..., :3, 3] = torch.tensor([0.0, 0.0, 0.8])
    # LAPACK's QR is not bit-reproducible across hosts: fixtures carry the poses they were generated with
    ep["tem_pose"] = pose if tem_pose is None else tem_pose
    ep["tem_K"] = K[None, None].repeat(B, N, 1, 1)
    ep["tem_M"] = torch.tensor([[1.5, 0, -300.0], [0, 1.5, -200.0], [0, 0, 1.0]])[None, None].repeat(B, N, 1, 1)
    if feature_fn is not None:
        ep["template_feature"] = torch.stack([feature_fn(ep["tem_rgb"][b])[-1] for b in range(B)])
    return ep
