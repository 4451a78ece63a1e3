"""Profiling target: a few fast-mode stage-1 calls at the bench shape (no timing, no oracle)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from picopose_amd.utils import matching as hm
B, N, C = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (32, 162, 768)
mode = sys.argv[4] if len(sys.argv) > 4 else "fast"
it = int(sys.argv[5]) if len(sys.argv) > 5 else 5
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
bank = torch.randn(B, N, C, 16, 16, device=dev, generator=g)
q = torch.randn(B, C, 16, 16, device=dev, generator=g)
yy, xx = torch.meshgrid(torch.arange(224.0), torch.arange(224.0), indexing="ij")
m = (((yy - 111.5) ** 2 + (xx - 111.5) ** 2) < (0.4 * 224) ** 2).float()[None].repeat(B, 1, 1).to(dev)
for _ in range(it):
    s, i = hm.matching_templates(bank, q, None, m, topk=5, mode=mode)
torch.cuda.synchronize()
print("done", s[0].tolist())
