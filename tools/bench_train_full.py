"""The full training step (scope "full": forward_train under autograd, Loss, backward, SGD) at B pairs, ViT-B: wall time per step, when
the host had finished launching it, peak memory.  usage: bench_train_full.py [B=32] [steps=6]   (PP_DETERMINISTIC=1: autograd.DETERMINISTIC; PP_TRAIN_MARK=1: a tiny marker kernel
— torch.zeros(1) — before every step, so a kernel trace can be cut into steps)"""
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from netcfg import make_train_end_points  # noqa: E402

from picopose_amd import autograd as _ag  # noqa: E402
from picopose_amd.picopose import Net  # noqa: E402
from picopose_amd.utils.loss_utils import Loss  # noqa: E402
from picopose_amd.utils.seeding import calibrated_state_dict  # noqa: E402

_ag.DETERMINISTIC = os.environ.get("PP_DETERMINISTIC", "0") == "1"     # the bit-reproducible scatter adjoints (autograd.DETERMINISTIC)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
vit = "dinov2_vitb14"
ns = types.SimpleNamespace
cfg = ns(hypothesis=5, stage1=ns(vit_type=vit, pretrained=False, interaction_indexes=[[0, 2], [3, 5], [6, 8], [9, 11]]), stage2=ns(in_channel=256, hidden_dim=256),
         stage3=ns(nclass=1, in_channels=768, use_bn=True, out_channels=[256, 512, 1024, 1024], num_levels=3, radius=4))
net = Net(cfg)
net.load_state_dict(calibrated_state_dict(net.state_dict(), 4, vit))
net = net.cuda().train()
ep = {k: v.cuda() for k, v in make_train_end_points(B, 11).items()}
np.random.seed(0)
torch.manual_seed(0)
opt = None
rows = []
mark = torch.zeros(7, device="cuda")
for i in range(steps):
    torch.cuda.synchronize()
    if os.environ.get("PP_TRAIN_MARK") == "1":
        mark.fill_(float(i))          # (a fill kernel on 7 floats: the step marker of tools/train_profile2.sh)
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = net(dict(ep))
    tot = Loss()(out)["loss"]
    t1 = time.perf_counter()
    tot.backward()
    t2 = time.perf_counter()
    if opt is None:
        opt = torch.optim.SGD([p for p in net.parameters() if p.grad is not None], lr=1e-6)
    opt.step()
    opt.zero_grad(set_to_none=True)
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    rows.append([1e3 * (t - t0) for t in (t1, t2, t3, t4)])
rows = rows[2:]
med = lambda k: sorted(r[k] for r in rows)[len(rows) // 2]  # noqa: E731
print(f"training step {vit} B={B} scope=full: {med(3):.1f} ms per step ({B / med(3) * 1e3:.1f} pairs/s); host: forward launched at {med(0):.1f} ms, "
      f"backward at {med(1):.1f}, optimizer at {med(2):.1f}; peak memory {torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GiB; loss {float(tot.detach()):.4f}", flush=True)
# the same steps back to back, ONE synchronisation at the end (a training loop that does not read the loss every step): the host runs ahead
# of the GPU, so the forward's launch time is hidden behind the previous step's backward
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    Loss()(net(dict(ep)))["loss"].backward()
    opt.step()
    opt.zero_grad(set_to_none=True)
torch.cuda.synchronize()
print(f"  {steps} steps back to back, one synchronisation: {1e3 * (time.perf_counter() - t0) / steps:.1f} ms per step", flush=True)
