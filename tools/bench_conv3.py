"""Time the 3x3 convolutions of the flow decoder's heads at the headline batch under each big-kernel configuration."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from picopose_amd import ops
g = torch.Generator().manual_seed(0)
for B, cin, cout, hw in [(160, 640, 512, 64), (160, 512, 256, 64), (160, 256, 256, 64), (160, 640, 512, 32), (192, 256, 256, 32)]:
    x = torch.randn(B, hw, hw, cin, generator=g).cuda()
    w = ops.pack_conv_weight((torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).cuda())
    xs = ops.split_image(x)
    line = f"B={B} {cin}->{cout} @{hw}: "
    for cfg in ("4", "5", "6"):
        os.environ["PP_GEMM_FORCE_CFG"] = cfg
        for _ in range(2): ops.conv2d(xs, w, None, 3, pad=1, act="relu", out_split=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8): ops.conv2d(xs, w, None, 3, pad=1, act="relu", out_split=True)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 8
        line += f" cfg{cfg} {ms:.3f} ms ({2 * B * hw * hw * cout * cin * 9 / ms / 1e9:.0f} TF)"
    print(line, flush=True)
