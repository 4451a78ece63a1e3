"""STUDY TOOL (GPU box): the gradient of the training loss at interior nodes of the DPT head / flow decoder on the HIP backward, as
per-channel sums over batch and pixels, beside the reference's float64 / float32 values of the same nodes (oracle/grad_nodes.py ->
_dbg/sums_*.npz, maps_*.npz, fwd.npz: made here from its output, scratch, not committed).  Locates where the backward first departs from float64 (profiles/r05/grad_f64.txt).
usage: python tools/grad_nodes_hip.py [f16x3|f32]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_train_gpu as T  # noqa: E402
from netcfg import small_cfg  # noqa: E402

from picopose_amd import autograd as A  # noqa: E402
from picopose_amd import ops  # noqa: E402
from picopose_amd.picopose import Net  # noqa: E402
from picopose_amd.utils.loss_utils import Loss  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
ops.PRECISION = prec
z, ep, weights = T._load_grad_fixture(os.path.join(ROOT, "tests", "golden"))
net = Net(small_cfg())
net.load_state_dict(weights(net.state_dict()))
net = net.cuda().train()
names = {id(p): n for n, p in net.named_parameters()}
mods = {id(m): n for n, m in net.named_modules()}
got, calls, fwd = {}, {}, {}
WANT = set(np.load(os.path.join(ROOT, "_dbg", "maps_f64.npz")).files)


def short(n):
    n = n.replace("offset_regressor.dpt_head.scratch.", "").replace("resConfUnit", "rcu").replace(".weight", "")
    if n.startswith("offset_regressor.flow_decoder.proj."):
        l, k = n.split(".")[3:5]
        n = f"proj{l}." + ("conv" if k == "0" else "bn")
    return n


def keep(name):
    def hook(g):
        g = g.detach().double()
        got[f"sum/{name}"] = g.sum(dim=(0, 1, 2)).cpu().numpy()
        got[f"max/{name}"] = float(g.abs().max())
        if name in WANT:
            got[f"map/{name}"] = g.permute(0, 3, 1, 2).cpu().numpy()
    return hook


def tap(node, x, y):
    c = calls.get(node, 0)
    calls[node] = c + 1
    if y.requires_grad:
        y.register_hook(keep(f"{node}.out/{c}"))
    if x.requires_grad:
        x.register_hook(keep(f"{node}.in/{c}"))


conv0, bn0, rcu0 = A.conv2d, A.batchnorm_train, A._rcu


def conv2d(x, weight, bias, k, **kw):
    y = conv0(x, weight, bias, k, **kw)
    tap(short(names[id(weight)]), x, y)
    return y


def batchnorm_train(x, bn, relu=False):
    y = bn0(x, bn, relu=relu)
    node = short(mods[id(bn)])
    fwd[f"{node}/{calls.get(node, 0)}"] = y.detach().double().reshape(-1).cpu().numpy()
    tap(node, x, y)
    return y


def _rcu(u, x, extra=None):
    y = rcu0(u, x, extra=None)
    node = short(mods[id(u)])
    fwd[f"{node}/{calls.get(node, 0)}"] = x.detach().double().reshape(-1).cpu().numpy()
    tap(node, x, y)
    return y if extra is None else A.add(y, extra)


A.conv2d, A.batchnorm_train, A._rcu = conv2d, batchnorm_train, _rcu
res = net.forward_train(T._cuda(ep), pred_Ms=torch.from_numpy(z["pred_Ms"]).cuda())
tot = Loss()(res)["loss"]
tot.backward()
torch.cuda.synchronize()
r64, r32 = np.load(os.path.join(ROOT, "_dbg", "sums_f64.npz")), np.load(os.path.join(ROOT, "_dbg", "sums_f32.npz"))
print(f"precision {prec}; total loss {float(tot.detach()):.9f}")
print("%-34s %10s %10s %11s %11s %10s" % ("node", "max|g| 64", "max|S| 64", "|Ship-S64|", "|S32-S64|", "max|g| hip"))
for k in sorted(k[4:] for k in r64.files if k.startswith("sum/")):
    if "sum/" + k not in got:
        continue
    if k.split("/")[0].endswith("bn1.out"):
        continue                                   # (this build's bn1 output is behind its fused ReLU)
    s64, s32, sh = r64["sum/" + k], r32["sum/" + k], got["sum/" + k]
    print("%-34s %10.2e %10.2e %11.2e %11.2e %10.2e" % (k, float(r64["max/" + k]), np.abs(s64).max(), np.abs(sh - s64).max(), np.abs(s32 - s64).max(), got["max/" + k]))

m64, m32 = np.load(os.path.join(ROOT, "_dbg", "maps_f64.npz")), np.load(os.path.join(ROOT, "_dbg", "maps_f32.npz"))
print("per-element: node, max|g64|, max|hip - g64|, max|ref32 - g64|, rms(hip - g64), rms(ref32 - g64)")
for k in m64.files:
    g64, g32, gh = m64[k], m32[k].astype(np.float64), got["map/" + k]
    eh, e3 = gh - g64, g32 - g64
    print("%-30s %9.2e %9.2e %9.2e %9.2e %9.2e" % (k, np.abs(g64).max(), np.abs(eh).max(), np.abs(e3).max(), np.sqrt((eh ** 2).mean()), np.sqrt((e3 ** 2).mean())))
    c = np.abs(eh.sum(axis=(0, 2, 3))).argmax()
    e = eh[:, c]
    top = np.argsort(-np.abs(e).reshape(-1))[:4]
    print("    worst channel of the sum:", c, "sum err %.2e" % eh[:, c].sum(), "largest element errors", e.reshape(-1)[top], "of g", g64[:, c].reshape(-1)[top])

fz = np.load(os.path.join(ROOT, "_dbg", "fwd.npz"))
print("forward values: node, rms, max|hip - f64|, rms(hip - f64), max|ref32 - f64|, rms(ref32 - f64), sign flips hip / ref32 against f64")
for k in [k[8:] for k in fz.files if k.startswith("f64/fwd/")]:
    v64, v32, vh = fz["f64/fwd/" + k], fz["f32/fwd/" + k].astype(np.float64), fwd[k]
    if k.endswith("bn1/1"):
        v64, v32 = np.maximum(v64, 0), np.maximum(v32, 0)
    eh, e3 = vh - v64, v32 - v64
    print("%-26s %9.2e %9.2e %9.2e %9.2e %9.2e %d %d" % (k, np.sqrt((v64 ** 2).mean()), np.abs(eh).max(), np.sqrt((eh ** 2).mean()), np.abs(e3).max(), np.sqrt((e3 ** 2).mean()),
          int(((vh > 0) != (v64 > 0)).sum()), int(((v32 > 0) != (v64 > 0)).sum())))
