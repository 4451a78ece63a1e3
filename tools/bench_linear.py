"""Time the ViT linears at the batched-hypotheses size under each pinned GEMM configuration."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from picopose_amd import ops
d = "cuda"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 41120
shapes = [("qkv", 768, 2304, None), ("proj", 768, 768, None), ("fc1", 768, 3072, "gelu"), ("fc1-noact", 768, 3072, None), ("fc2", 3072, 768, None)]
for name, K, N, act in shapes:
    x = torch.randn(M, K, device=d); w = torch.randn(N, K, device=d) / K ** 0.5; b = torch.randn(N, device=d)
    xs = ops.Split(ops.split_activation(x, 1, M, K, 0, K))
    for cfg in os.environ.get("CFGS", "0,2,3").split(","):
        os.environ["PP_GEMM_FORCE_CFG"] = cfg
        for split_out in ((False, True) if os.environ.get("PLANES") else (False,)):
            for _ in range(3): y = ops.linear(xs, w, b, act=act, out_split=split_out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): y = ops.linear(xs, w, b, act=act, out_split=split_out)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            print(f"{name:10s} M={M} K={K} N={N} cfg={cfg} planes_out={int(split_out)} {ms:.3f} ms {2*M*N*K/ms/1e9:.0f} TFLOP/s", flush=True)
