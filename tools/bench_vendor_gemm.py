"""Context for the engine's numbers: the vendor library's plain fp16 GEMM (torch.matmul on half tensors = hipBLASLt / rocBLAS) on
random data at the ViT-B shapes, (a) at the algorithmic size M x N x K and (b) at the size whose MFMA work equals the f16x3 engine's
(three fp16 MFMA products per fp32-grade product: K' = 3 K).  Not part of the product: measurement only."""
import sys
import torch

M = int(sys.argv[1]) if len(sys.argv) > 1 else 41120
for name, K, N in (("qkv", 768, 2304), ("proj", 768, 768), ("fc1", 768, 3072), ("fc2", 3072, 768)):
    for kk, tag in ((K, "algorithmic K"), (3 * K, "3K: the f16x3 engine's executed MFMA work")):
        a = torch.randn(M, kk, device="cuda").half()
        b = torch.randn(N, kk, device="cuda").half()
        for _ in range(5):
            c = a @ b.t()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            c = a @ b.t()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"{name:5s} M={M} N={N} K={kk:5d} ({tag}): {ms:.3f} ms, {2 * M * N * kk / ms / 1e9:.0f} TFLOP/s executed, fp16 in / fp16 out", flush=True)
