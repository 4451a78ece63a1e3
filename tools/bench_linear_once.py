import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from picopose_amd import ops
d = "cuda"; M = 41120; K = int(os.environ.get("K", 768)); N = int(os.environ.get("N", 3072))
x = torch.randn(M, K, device=d); w = torch.randn(N, K, device=d) / K ** 0.5; b = torch.randn(N, device=d)
xs = ops.Split(ops.split_activation(x, 1, M, K, 0, K))
for _ in range(6): y = ops.linear(xs, w, b)
torch.cuda.synchronize(); print(float(y[0, 0]))
