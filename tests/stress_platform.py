"""Platform check: is a plain TORCH elementwise kernel deterministic when two processes share the GPU and each also runs LDS-heavy
GEMM launches (as load)?  y = x * 1.5 + 2 on (6*64*64, 256) fp32 into a fresh buffer, compared with a reference."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from picopose_amd import ops
g = torch.Generator().manual_seed(0)
x = torch.randn(6 * 64 * 64, 256, generator=g).cuda()
a = torch.randn(8192, 256, generator=g).cuda(); w = (torch.randn(256, 256, generator=g) / 16).cuda()
xs = ops.Split(ops.split_activation(a, 1, 8192, 256, 0, 256))
os.environ["PP_GEMM_FORCE_CFG"] = os.environ.get("CFG", "8")
load = os.environ.get("LOAD", "1") == "1"
ref = x * 1.5 + 2
torch.cuda.synchronize()
bad = 0
for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 1500):
    if load:
        ops.linear(xs, w, None)
    y = x * 1.5 + 2
    if not torch.equal(y, ref):
        d = y != ref
        bad += 1
        if bad <= 3:
            print("rep", r, int(d.sum()), "floats differ; rows", d.any(1).nonzero().flatten()[:4].tolist(), "cols", int(d.any(0).nonzero()[0]), "..", int(d.any(0).nonzero()[-1]), flush=True)
print("differing:", bad)
