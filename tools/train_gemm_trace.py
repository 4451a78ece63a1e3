"""Which GEMM launches of ONE full training step take the time: per-launch records of the engine (pp_prof_gemm_records: shape,
tile configuration, kind 0 = pre-split engine / 1 = on-the-fly or fp32 kernels), grouped by (M, N, K, kind).
usage: train_gemm_trace.py [B=32] [vit=dinov2_vitb14]"""
import collections
import ctypes
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from netcfg import make_train_end_points  # noqa: E402

from picopose_amd import _lib  # noqa: E402
from picopose_amd.picopose import Net  # noqa: E402
from picopose_amd.utils.loss_utils import Loss  # noqa: E402
from picopose_amd.utils.seeding import calibrated_state_dict  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
vit = sys.argv[2] if len(sys.argv) > 2 else "dinov2_vitb14"
ns = types.SimpleNamespace
C, idx = {"dinov2_vits14": (384, [[0, 2], [3, 5], [6, 8], [9, 11]]), "dinov2_vitb14": (768, [[0, 2], [3, 5], [6, 8], [9, 11]])}[vit]
cfg = ns(hypothesis=5, stage1=ns(vit_type=vit, pretrained=False, interaction_indexes=idx), stage2=ns(in_channel=256, hidden_dim=256),
         stage3=ns(nclass=1, in_channels=C, use_bn=True, out_channels=[256, 512, 1024, 1024], num_levels=3, radius=4))
net = Net(cfg)
net.load_state_dict(calibrated_state_dict(net.state_dict(), 4, vit))
net = net.cuda().train()
ep = {k: v.cuda() for k, v in make_train_end_points(B, 11).items()}
np.random.seed(0)
torch.manual_seed(0)
L = _lib.lib()
for i in range(3):
    if i == 2:
        _lib.check(L.pp_prof_gemm_enable(16384), "pp_prof_gemm_enable")
    Loss()(net(dict(ep)))["loss"].backward()
    net.zero_grad(set_to_none=True)
torch.cuda.synchronize()
cap = 16384
shape, ms, fl, cnt = (ctypes.c_int * (6 * cap))(), (ctypes.c_float * cap)(), (ctypes.c_double * cap)(), ctypes.c_int()
_lib.check(L.pp_prof_gemm_records(cap, shape, ms, fl, ctypes.byref(cnt)), "pp_prof_gemm_records")
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for i in range(cnt.value):
    M, N, K, ck, cf, kind = (shape[6 * i + k] for k in range(6))
    a = agg[(M, N, K, ck, kind)]
    a[0] += 1
    a[1] += ms[i]
    a[2] += fl[i]
tot = sum(a[1] for a in agg.values())
print(f"{cnt.value} GEMM launches, {tot:.1f} ms in one full training step (B = {B}, {vit})")
for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get("PP_TRACE_TOP", "28"))]:
    M, N, K, ck, kind = key
    print(f"  {a[1]:7.2f} ms  x{a[0]:3d}  M={M:7d} N={N:5d} K={K:7d} {'conv' if ck else 'dense'} {'engine' if kind == 0 else 'fly/fp32'}  {a[2] / max(a[1], 1e-9) / 1e9:7.1f} useful TFLOP/s")
