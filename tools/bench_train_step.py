"""Time the full TRAINING STEP (run_train.py:109-130: forward_train under autograd, Loss, backward, SGD step) on synthetic training
batches, per scope of Net.train_backward, with the peak device memory.  usage: bench_train_step.py [B=8] [vit=dinov2_vitb14]"""
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

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
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
opt = None
for scope in ("full", "vit+stage2", False):
    net.train_backward = scope
    torch.cuda.reset_peak_memory_stats()
    times = []
    for i in range(6):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = net(dict(ep))
        tot = Loss()(out)["loss"]
        if scope:
            tot.backward()
            if opt is None:
                opt = torch.optim.SGD([p for p in net.parameters() if p.grad is not None], lr=1e-6)
            opt.step()
            opt.zero_grad(set_to_none=True)
        e1.record()
        torch.cuda.synchronize()
        if i >= 2:
            times.append(e0.elapsed_time(e1))
    times.sort()
    print(f"training step {vit} B={B} scope={scope}: median {times[len(times) // 2]:.1f} ms ({B / times[len(times) // 2] * 1e3:.1f} pairs/s), "
          f"peak memory {torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GiB, loss {float(tot.detach()):.4f}", flush=True)
