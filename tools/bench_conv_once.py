import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from picopose_amd import ops
d = "cuda"; B = 32
x = torch.randn(B, 64, 64, 640, device=d); w = ops.pack_conv_weight(torch.randn(512, 640, 3, 3, device=d))
for _ in range(6): y = ops.conv2d(x, w, None, 3, 1, 1)
torch.cuda.synchronize(); print(float(y[0, 0, 0, 0]))
