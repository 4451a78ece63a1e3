"""The big 3x3 convolutions of the flow decoder at the batched-hypotheses size under each pinned configuration."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from picopose_amd import ops
d = "cuda"; B = int(sys.argv[1]) if len(sys.argv) > 1 else 160
for cin, cout, hw in ((640, 512, 64), (512, 256, 64), (256, 256, 64), (640, 512, 32)):
    x = torch.randn(B, hw, hw, cin, device=d); w = ops.pack_conv_weight(torch.randn(cout, cin, 3, 3, device=d) / 50)
    xs = ops.split_image(x)
    for cfg in os.environ.get("CFGS", "0,3,4").split(","):
        os.environ["PP_GEMM_FORCE_CFG"] = cfg
        for so in (False,):
            for _ in range(2): y = ops.conv2d(xs, w, None, 3, 1, 1, act="relu", out_split=so)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): y = ops.conv2d(xs, w, None, 3, 1, 1, act="relu", out_split=so)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            print(f"conv {cin}->{cout} {hw}x{hw} B={B} cfg={cfg} planes_out={int(so)}: {ms:.3f} ms {2*B*hw*hw*cout*cin*9/ms/1e9:.0f} TFLOP/s", flush=True)
    del x, xs
