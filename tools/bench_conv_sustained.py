"""Sustained vs short-burst rate of the big 3x3 convolution (is the step power/clock limited?)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from picopose_amd import ops
d = "cuda"; B = 32
x = torch.randn(B, 64, 64, 640, device=d); w = ops.pack_conv_weight(torch.randn(512, 640, 3, 3, device=d))
xs = ops.split_image(x)
for _ in range(3): y = ops.conv2d(xs, w, None, 3, 1, 1)
torch.cuda.synchronize()
for n in (5, 20, 100, 400):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): y = ops.conv2d(xs, w, None, 3, 1, 1)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"{n:4d} back-to-back launches: {ms:.3f} ms each, {2*B*4096*512*5760/ms/1e9:.0f} useful TFLOP/s", flush=True)
