"""Locate run-to-run differences of the HIP forward under contention: every engine op's output is fingerprinted (exact integer
sum of its bit pattern) and compared with the first run; prints the first ops whose outputs differ."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from netcfg import make_end_points, small_cfg  # noqa: E402

from picopose_amd import ops  # noqa: E402
from picopose_amd.picopose import Net  # noqa: E402
from picopose_amd.utils.seeding import calibrated_state_dict  # noqa: E402

log = []


def fp(t):
    if isinstance(t, ops.Split):
        t = t.hl
    if not isinstance(t, torch.Tensor):
        return None
    t = t.contiguous()
    v = t.view(torch.int16) if t.dtype == torch.float16 else t.view(torch.int32)
    return int(v.to(torch.int64).sum())


def wrap(name):
    orig = getattr(ops, name)

    def f(*a, **k):
        out = orig(*a, **k)
        tgt = k.get("hl_into") or k.get("into")
        items = [out] + ([tgt[0]] if tgt else []) + ([getattr(out, "_hl", None), getattr(out, "_hl_relu", None)] if isinstance(out, torch.Tensor) else [])
        log.append((name, tuple(getattr(a[0], "shape", ())), tuple(fp(i) for i in items if i is not None)))
        return out

    setattr(ops, name, f)


_warp = ops.warp


def warp_checked(feat, flow, out=None, hl_into=None):
    r = _warp(feat, flow, out=out, hl_into=hl_into)
    if hl_into is not None:
        tgt, col0 = hl_into
        C = feat.shape[-1]
        a = tgt.hl[:, 2 * col0:2 * (col0 + C)].clone()
        _warp(feat, flow, hl_into=hl_into)
        b = tgt.hl[:, 2 * col0:2 * (col0 + C)].clone()
        log.append(("warp_inputs", tuple(feat.shape), (fp(feat), fp(flow))))
        log.append(("warp_cols", tuple(feat.shape), (fp(a),)))
        if not torch.equal(a, b):
            d = (a != b)
            rows = d.any(1).nonzero().flatten()
            cols = d.any(0).nonzero().flatten()
            _warp(feat, flow, hl_into=hl_into)
            c = tgt.hl[:, 2 * col0:2 * (col0 + C)].clone()
            H = feat.shape[1]
            print("WARP disagree:", int(d.sum()), "halfs; rows", rows[:6].tolist(), "..", int(rows[-1]), "n", rows.numel(), "(pixels per image", H * H, ") cols",
                  int(cols[0]), "..", int(cols[-1]), "n", cols.numel(), "| third launch equals first:", bool(torch.equal(c, a)), "second:", bool(torch.equal(c, b)), flush=True)
    return r


ops.warp = warp_checked

for n in ("linear", "conv2d", "conv_transpose2d", "attention", "layernorm", "corr_lookup", "warp", "resize_bilinear", "split_activation",
          "groupnorm", "bmm_nt", "linear_splitk", "split_image", "avgpool2", "hl_patch_columns", "to_nhwc", "to_nchw", "tokens_to_nchw"):
    if n == "warp":
        continue
    wrap(n)

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
net = Net(small_cfg())
net.load_state_dict(calibrated_state_dict(net.state_dict(), 4, "dinov2_vits14"))
net = net.cuda().eval()
ep = {k: v.cuda() for k, v in make_end_points(2, 7, 55, dome=True).items()}
with torch.no_grad():
    ep["template_feature"] = torch.stack([net.feature_extractor(ep["tem_rgb"][b])[-1] for b in range(2)])
net(ep, 3)
first = None
seen = {}
for r in range(reps):
    log.clear()
    net(ep, 3)
    cur = list(log)
    if first is None:
        first = cur
        continue
    for i, (a, b) in enumerate(zip(cur, first)):
        if a != b:
            key = (i, a[0], a[1])
            seen[key] = seen.get(key, 0) + 1
            break          # only the FIRST differing op of the run
print("first differing op per run (index, op, input shape): count")
for k, v in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(k, v)
print("runs:", reps - 1, "with a difference:", sum(seen.values()))
