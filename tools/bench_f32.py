"""The fp32 engine (csrc/pp_gemm_f.hip) on the headline step's heaviest shapes under pinned tile configurations (PP_GEMM_FORCE_CFG:
3 = 128x128, 4 = 256x128, 5 = 256x256, 6 = 128x64): ms and useful TFLOP/s against the 157.3 TFLOP/s fp32-MFMA peak.
usage: [PP_LIB_SUFFIX=_x] [CFGS=5,3] [REPS=8] python tools/bench_f32.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from picopose_amd import ops
ops.PRECISION = "f32"
g = torch.Generator().manual_seed(0)
cfgs = os.environ.get("CFGS", "5,3").split(",")
reps = int(os.environ.get("REPS", "8"))


def timed(fn):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for B, cin, cout, hw in [(40, 640, 512, 64), (40, 512, 256, 64), (48, 256, 256, 64)]:
    x = torch.randn(B, hw, hw, cin, generator=g).cuda()
    w = ops.pack_conv_weight((torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).cuda())
    b = torch.randn(cout, generator=g).cuda()
    line = f"conv3x3 B={B} {cin}->{cout} @{hw}: "
    for cfg in cfgs:
        os.environ["PP_GEMM_FORCE_CFG"] = cfg
        ms = timed(lambda: ops.conv2d(x, w, b, 3, pad=1, act="relu"))
        line += f" cfg{cfg} {ms:.3f} ms ({2 * B * hw * hw * cout * cin * 9 / ms / 1e9:.1f} TF)"
    print(line, flush=True)
M = 41120
x768, x3072 = torch.randn(M, 768, generator=g).cuda(), torch.randn(M, 3072, generator=g).cuda()
res = torch.randn(M, 768, generator=g).cuda()
for name, xin, N, K, kw in [("qkv", x768, 2304, 768, {}), ("proj", x768, 768, 768, {"residual": res, "gamma": torch.randn(768, generator=g).cuda()}),
                            ("fc1", x768, 3072, 768, {"act": "gelu"}), ("fc2", x3072, 768, 3072, {"residual": res, "gamma": torch.randn(768, generator=g).cuda()})]:
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    line = f"linear {name} M={M} N={N} K={K}: "
    for cfg in cfgs:
        os.environ["PP_GEMM_FORCE_CFG"] = cfg
        ms = timed(lambda: ops.linear(xin, w, b, **kw))
        line += f" cfg{cfg} {ms:.3f} ms ({2 * M * N * K / ms / 1e9:.1f} TF)"
    print(line, flush=True)
