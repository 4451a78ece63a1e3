"""STUDY: does any kernel of the training step read memory it (or a kernel before it) did not write?  Every torch.empty / empty_like /
new_empty made during the step is filled with a poison pattern (float NaN; bytes 0xFF = NaN as fp32 / fp16) before it is handed out, then the
ViT-S training step of tests/dist_worker_train_gpu.py runs (deterministic adjoints) and its loss and gradients are compared BITWISE with an
unpoisoned run of the same process.  usage: study_poison.py [train|infer]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from netcfg import make_train_end_points, small_cfg  # noqa: E402

from picopose_amd import autograd  # noqa: E402
from picopose_amd.picopose import Net  # noqa: E402
from picopose_amd.utils.loss_utils import Loss  # noqa: E402
from picopose_amd.utils.seeding import calibrated_state_dict  # noqa: E402

autograd.DETERMINISTIC = True
net = Net(small_cfg())
net.load_state_dict(calibrated_state_dict(net.state_dict(), 4, "dinov2_vits14"))
net = net.cuda().train()
ep = {k: v.cuda() for k, v in make_train_end_points(2, 100).items()}
sd0 = {k: v.clone() for k, v in net.state_dict().items()}


def step():
    net.load_state_dict(sd0)
    net.zero_grad(set_to_none=True)
    np.random.seed(700)
    torch.manual_seed(900)
    loss = Loss()(net(ep))["loss"]
    loss.backward()
    torch.cuda.synchronize()
    return {"loss": loss.detach().cpu(), **{n: p.grad.cpu().clone() for n, p in net.named_parameters() if p.grad is not None}}


step()
base = step()
_empty, _empty_like, _new_empty = torch.empty, torch.empty_like, torch.Tensor.new_empty
count = [0]


def poison(t):
    if t.is_cuda and t.numel():
        count[0] += 1
        if t.is_floating_point():
            t.fill_(float("nan"))
        elif t.dtype == torch.uint8:
            t.fill_(255)
        elif t.dtype in (torch.int32, torch.int64, torch.int16):
            t.fill_(-1)
    return t


torch.empty = lambda *a, **k: poison(_empty(*a, **k))
torch.empty_like = lambda *a, **k: poison(_empty_like(*a, **k))
torch.Tensor.new_empty = lambda self, *a, **k: poison(_new_empty(self, *a, **k))
try:
    got = step()
finally:
    torch.empty, torch.empty_like, torch.Tensor.new_empty = _empty, _empty_like, _new_empty
print(f"{count[0]} allocations poisoned; loss {float(base['loss']):.9f} -> {float(got['loss']):.9f}")
bad = [(k, bool(torch.isnan(got[k]).any()), float((got[k] - base[k]).abs().max()) / max(float(base[k].abs().max()), 1e-30)) for k in base if not torch.equal(got[k], base[k])]
print(len(bad), "of", len(base), "results differ from the unpoisoned step:", bad[:12])
