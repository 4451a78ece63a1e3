"""One ViT linear at the batched-hypotheses size under a pinned configuration, a few launches (PMC passes: tools/pmc.sh)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from picopose_amd import ops
name = sys.argv[1] if len(sys.argv) > 1 else "qkv"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 41120
K, N, act = {"qkv": (768, 2304, None), "proj": (768, 768, None), "fc1": (768, 3072, "gelu"), "fc2": (3072, 768, None)}[name]
x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
xs = ops.Split(ops.split_activation(x, 1, M, K, 0, K))
for _ in range(6):
    y = ops.linear(xs, w, b, act=act, out_split=True)
torch.cuda.synchronize()
