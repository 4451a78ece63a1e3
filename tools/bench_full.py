"""Full-path timing probe: Net.forward (eval) at a BASELINE config on one GPU, with per-stage event timing."""
import os, sys, time, types
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from picopose_amd.picopose import Net
ns = types.SimpleNamespace
B, N, hyp = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (32, 162, 5)
vit = sys.argv[4] if len(sys.argv) > 4 else "dinov2_vitb14"
C = {"dinov2_vits14": 384, "dinov2_vitb14": 768, "dinov2_vitl14": 1024}[vit]
idx = {"dinov2_vits14": [[0, 2], [3, 5], [6, 8], [9, 11]], "dinov2_vitb14": [[0, 2], [3, 5], [6, 8], [9, 11]],
       "dinov2_vitl14": [[0, 5], [6, 11], [12, 17], [18, 23]]}[vit]
cfg = ns(hypothesis=hyp, stage1=ns(vit_type=vit, pretrained=False, interaction_indexes=idx), stage2=ns(in_channel=256, hidden_dim=256),
         stage3=ns(nclass=1, in_channels=C, use_bn=True, out_channels=[256, 512, 1024, 1024], num_levels=3, radius=4))
torch.manual_seed(4)
net = Net(cfg)
with torch.no_grad():
    for n_, p in net.named_parameters():
        if p.dim() >= 2:
            fan = p[0].numel()
            p.copy_(torch.randn(p.shape) * (1.5 / fan) ** 0.5)
net = net.cuda().eval()
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
yy, xx = torch.meshgrid(torch.arange(224.0), torch.arange(224.0), indexing="ij")
disk = (((yy - 111.5) ** 2 + (xx - 111.5) ** 2) < (0.4 * 224) ** 2).float().to(dev)
K = torch.tensor([[572.4114, 0, 320], [0, 573.57043, 240], [0, 0, 1.0]], device=dev)
ep = {"real_rgb": torch.randn(B, 3, 224, 224, device=dev, generator=g), "real_mask": disk[None].repeat(B, 1, 1),
      "real_K": K[None].repeat(B, 1, 1), "real_M": torch.tensor([[2.0, 0, -100.0], [0, 2.0, -80.0], [0, 0, 1.0]], device=dev)[None].repeat(B, 1, 1),
      "real_pose": torch.eye(4, device=dev)[None].repeat(B, 1, 1), "real_pts2d": torch.rand(B, 64, 64, 2, device=dev, generator=g) * 100,
      "tem_rgb": torch.randn(B, N, 3, 224, 224, device=dev, generator=g), "tem_mask": disk[None, None].repeat(B, N, 1, 1),
      "tem_pts3d": (torch.rand(B, N, 64, 64, 3, device=dev, generator=g) - 0.5) * 0.2,
      "tem_K": K[None, None].repeat(B, N, 1, 1), "tem_M": torch.tensor([[1.5, 0, -300.0], [0, 1.5, -200.0], [0, 0, 1.0]], device=dev)[None, None].repeat(B, N, 1, 1)}
pose = torch.eye(4, device=dev)[None, None].repeat(B, N, 1, 1); pose[..., 2, 3] = 0.8
ep["tem_pose"] = pose
t0 = time.time()
feats = []
for b in range(B):  # bank precompute (outside the timed region, run_test.py:120-134)
    f = [net.feature_extractor(ep["tem_rgb"][b, s:s + 54])[-1] for s in range(0, N, 54)]
    feats.append(torch.cat(f))
ep["template_feature"] = torch.stack(feats)
torch.cuda.synchronize()
print(f"bank precompute: {time.time()-t0:.2f} s for {B*N} templates", flush=True)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    out = net(ep, hyp)
    torch.cuda.synchronize(); dt = time.time() - t0
    print(f"forward {it}: {dt*1e3:.1f} ms -> {B/dt:.1f} crops/s", flush=True)
