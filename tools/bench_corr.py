"""Time the correlation lookup at the decoder's 64x64 level (160 images, 3 pyramid levels) on an affine flow field with
noise — the bench's regime — for the tiled kernel and (PP_CORR_TILED=0) the lane-per-position kernel."""
import math, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from picopose_amd import ops
B, H, C = 160, 64, 256
g = torch.Generator().manual_seed(0)
f1 = torch.randn(B, H, H, C, generator=g).cuda()
f2 = torch.randn(32, H, H, C, generator=g).cuda()
yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(H, dtype=torch.float32), indexing="ij")
for ang, sc, noise in ((0.0, 1.0, 0.0), (0.2, 1.05, 0.4), (0.6, 1.2, 0.4), (0.2, 1.05, 3.0)):
    c, s = sc * math.cos(ang), sc * math.sin(ang)
    tx, ty = c * (xx - 32) - s * (yy - 32) + 32, s * (xx - 32) + c * (yy - 32) + 32
    flow = (torch.stack([tx - xx, ty - yy], dim=-1)[None] + noise * torch.randn(B, H, H, 2, generator=g)).cuda().contiguous()
    line = f"rot {ang} scale {sc} noise {noise}:"
    for tiled in ("1", "0"):
        os.environ["PP_CORR_TILED"] = tiled
        for _ in range(2): ops.corr_lookup(f1, f2, flow, 3, 2, c_pad=80)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): ops.corr_lookup(f1, f2, flow, 3, 2, c_pad=80)
        e1.record(); torch.cuda.synchronize()
        line += f"  {'tiled' if tiled == '1' else 'lane-per-position'} {e0.elapsed_time(e1) / 5:.3f} ms"
    print(line, flush=True)
