"""Time the TRAINING forward (Net.forward in train mode = model/picopose.py:114-137: key-point ground truth, both ViT passes,
losses, BatchNorm on batch statistics) on synthetic training batches.  usage: bench_train_forward.py [B=32] [vit=dinov2_vitb14]"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from netcfg import make_train_end_points  # noqa: E402

from picopose_amd.picopose import Net  # noqa: E402
from picopose_amd.utils.loss_utils import Loss  # noqa: E402
from picopose_amd.utils.seeding import calibrated_state_dict  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
vit = sys.argv[2] if len(sys.argv) > 2 else "dinov2_vitb14"
ns = types.SimpleNamespace
C, idx = {"dinov2_vits14": (384, [[0, 2], [3, 5], [6, 8], [9, 11]]), "dinov2_vitb14": (768, [[0, 2], [3, 5], [6, 8], [9, 11]]),
          "dinov2_vitl14": (1024, [[0, 5], [6, 11], [12, 17], [18, 23]])}[vit]
cfg = ns(hypothesis=5, stage1=ns(vit_type=vit, pretrained=False, interaction_indexes=idx), stage2=ns(in_channel=256, hidden_dim=256),
         stage3=ns(nclass=1, in_channels=C, use_bn=True, out_channels=[256, 512, 1024, 1024], num_levels=3, radius=4))
net = Net(cfg)
net.load_state_dict(calibrated_state_dict(net.state_dict(), 4, vit))
net = net.cuda().train()
ep = {k: v.cuda() for k, v in make_train_end_points(B, 11).items()}
np.random.seed(0)
torch.manual_seed(0)
net.train_backward = False   # this tool times the forward-only training step (fused kernels); bench_train_step.py the full step
for _ in range(3):
    out = net(dict(ep))
torch.cuda.synchronize()
e = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
e[0].record()
for i in range(10):
    out = net(dict(ep))
    e[i + 1].record()
torch.cuda.synchronize()
ms = sorted(e[i].elapsed_time(e[i + 1]) for i in range(10))
tot = Loss()(out)
print(f"training forward {vit} B={B}: median {ms[5]:.2f} ms/step ({B / ms[5] * 1e3:.0f} pairs/s), min {ms[0]:.2f}; "
      f"loss {float(tot['loss']):.4f}; valid key-points per pair "
      f"{float((net.compute_keypoint_data(ep)['src_pts'][..., 0] != -1).sum(1).float().mean()):.0f}")
