import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from picopose_amd import autograd as ag
B, T, heads, hd = 4, 257, 12, 64
for gain in (1.0, 2.0, 4.0):
    g = torch.Generator().manual_seed(7)
    qkv = torch.randn(B * T, 3 * heads * hd, generator=g)
    qkv[:, :2 * heads * hd] *= gain
    x = qkv.double()
    q, k, v = x.view(B, T, 3, heads, hd).permute(2, 0, 3, 1, 4)
    ref = (torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5, dim=-1) @ v).permute(0, 2, 1, 3).reshape(B * T, heads * hd)
    q32, k32, v32 = (t.float() for t in (q, k, v))
    t32 = (torch.softmax(q32 @ k32.transpose(-1, -2) * hd ** -0.5, dim=-1) @ v32).permute(0, 2, 1, 3).reshape(B * T, heads * hd)
    out = {}
    for fused in (True, False):
        ag.FUSED_ATTENTION = fused
        with torch.no_grad():
            y = ag._Attention.apply(qkv.cuda(), B, T, heads, hd).cpu().double()
        out[fused] = y
    for name, y in (("fused", out[True]), ("unfused", out[False]), ("torch fp32 cpu", t32.double())):
        e = (y - ref).abs()
        print(f"gain {gain}: {name:15s} max {float(e.max() / ref.abs().max()):.2e} rms {float(e.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()):.2e}  last-row max {float(e.view(B, T, -1)[:, -1].max() / ref.abs().max()):.2e} row0 {float(e.view(B, T, -1)[:, 0].max() / ref.abs().max()):.2e}")
