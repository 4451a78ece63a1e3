"""GEMM / conv engine throughput on the shapes of the path (fp32 MFMA peak 157 TFLOP/s)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from picopose_amd import ops
def timeit(fn, flops, name, it=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / it
    print(f"{name:46s} {ms:8.3f} ms  {flops/ms/1e9:7.1f} TFLOP/s", flush=True)
d = "cuda"
for (M, K, N, nm) in [(8224, 768, 2304, "qkv ViT-B B32"), (8224, 768, 3072, "fc1"), (8224, 3072, 768, "fc2"), (8224, 768, 768, "proj"),
                      (8192, 768, 1024, "dpt project 1x1")]:
    x, w = torch.randn(M, K, device=d), torch.randn(N, K, device=d)
    timeit(lambda: ops.linear(x, w), 2 * M * K * N, f"linear {nm} {M}x{K}x{N}")
B = 32
for (cin, cout, k, hw, nm) in [(640, 512, 3, 64, "xhead0 L2"), (512, 256, 3, 64, "xhead1 L2"), (256, 256, 3, 64, "rcu 64"),
                               (640, 512, 3, 32, "xhead0 L1"), (256, 256, 3, 16, "rcu 16"), (1024, 1024, 3, 16, "resize3 s2")]:
    x = torch.randn(B, hw, hw, cin, device=d); w = ops.pack_conv_weight(torch.randn(cout, cin, k, k, device=d))
    timeit(lambda: ops.conv2d(x, w, None, k, 1, k // 2), 2 * B * hw * hw * cin * k * k * cout, f"conv {nm} {cin}->{cout} k{k} {hw}x{hw} B{B}")
qkv = torch.randn(B, 257, 3, 12, 64, device=d)
q, kk, v = (qkv[:, :, j].permute(0, 2, 1, 3) for j in range(3))
timeit(lambda: ops.bmm_nt(q, kk, alpha=0.125), 2 * B * 12 * 257 * 257 * 64, "attention QK^T B32 h12")
s = torch.randn(B, 12, 257, 257, device=d); o = torch.empty(B, 257, 12, 64, device=d)
timeit(lambda: ops.bmm_nn(s, v, o.permute(0, 2, 1, 3)), 2 * B * 12 * 257 * 257 * 64, "attention PV")
