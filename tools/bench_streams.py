"""Do two independent convolution chains (the decoder's flow / mask heads) gain from running on two HIP streams?  Serial vs
concurrent, at the 16x16 and 32x32 levels (mid-size launches with tails) and at 64x64 (launches that fill the chip)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from picopose_amd import ops
g = torch.Generator().manual_seed(0)
for hw in (16, 32, 64):
    B = 160
    x = torch.randn(B, hw, hw, 640, generator=g).cuda()
    xs = ops.split_image(x)
    w = [[ops.pack_conv_weight((torch.randn(co, ci, 3, 3, generator=g) / (ci * 9) ** 0.5).cuda()) for ci, co in ((640, 512), (512, 256))] for _ in range(2)]
    def chain(k):
        h = ops.conv2d(xs, w[k][0], None, 3, pad=1, act="relu", out_split=True)
        return ops.conv2d(h, w[k][1], None, 3, pad=1, act="relu", out_split=True)
    for _ in range(3): chain(0); chain(1)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    def serial():
        a = chain(0); b = chain(1); return a, b
    def concurrent():
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            b = chain(1)
        a = chain(0)
        main.wait_stream(side)
        return a, b
    for name, fn in (("serial", serial), ("two streams", concurrent), ("serial", serial), ("two streams", concurrent)):
        for _ in range(2): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        keep = [fn() for _ in range(6)]
        e1.record(); torch.cuda.synchronize()
        print(f"{hw}x{hw}: {name:12s} {e0.elapsed_time(e1) / 6:.3f} ms", flush=True)
