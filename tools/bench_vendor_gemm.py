"""Context for the engine's numbers: the vendor library's plain fp16 GEMM (torch.matmul on half tensors = hipBLASLt / rocBLAS) on
random data at the shapes the headline step runs — the ViT-B / ViT-L linears at this round's row count and the dense products of the
Winograd F(4x4, 3x3) convolutions — (a) at the algorithmic size M x N x K, which is also what `--mode fp16` executes, and (b) at the size
whose MFMA work equals the f16x3 engine's (three fp16 MFMA products per fp32-grade product: K' = 3 K); next to each, THIS engine on the
same shape (pre-split operand in, operand out — the form the ViT runs; fp32 out for the Winograd products).  Not part of the product:
measurement only (the library is not on the product path).

    python tools/bench_vendor_gemm.py [M]        (default M = 49 344: 160 template + 32 query crops x 257 tokens)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from picopose_amd import ops  # noqa: E402

M0 = int(sys.argv[1]) if len(sys.argv) > 1 else 49344
SHAPES = [("ViT-B qkv", M0, 768, 2304), ("ViT-B proj", M0, 768, 768), ("ViT-B fc1", M0, 768, 3072), ("ViT-B fc2", M0, 3072, 768),
          ("ViT-L qkv", M0, 1024, 3072), ("ViT-L proj", M0, 1024, 1024), ("ViT-L fc1", M0, 1024, 4096), ("ViT-L fc2", M0, 4096, 1024),
          # one frequency block of the flow decoder's heads at 64 x 64 x 160 (P = 40 960 tiles); the engine runs 36 of them as one launch
          ("wino 640->512 (1 of 36)", 40960, 640, 512), ("wino 512->256 (1 of 36)", 40960, 512, 256)]


def timed(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    print(f"{'shape':26s} {'M':>6s} {'N':>5s} {'K':>5s} | vendor fp16 @K: ms TF/s | vendor fp16 @3K: ms TF/s(executed) | engine f16x3: ms, useful TF/s, executed TF/s | "
          f"engine f16 (1 term): ms, TF/s")
    for name, M, K, N in SHAPES:
        row = f"{name:26s} {M:6d} {N:5d} {K:5d} |"
        for kk in (K, 3 * K):
            a = torch.randn(M, kk, device="cuda").half()
            b = torch.randn(N, kk, device="cuda").half()
            ms = timed(lambda: a @ b.t())
            row += f" {ms:7.3f} {2 * M * N * kk / ms / 1e9:6.0f} |"
            del a, b
        x = torch.randn(M, K, device="cuda")
        w = (torch.randn(N, K, device="cuda") / K ** 0.5).contiguous()
        for prec in ("f16x3", "f16"):
            ops.PRECISION = prec
            xs = ops.Split(ops.split_activation(x, 1, M, K, 0, K))
            fn = (lambda: ops.linear(xs, w, out_split=True)) if not name.startswith("wino") else (lambda: ops.linear(xs, w))
            ms = timed(fn)
            tf = 2 * M * N * K / ms / 1e9
            row += f" {ms:7.3f} {tf:6.0f}" + (f" {3 * tf:6.0f} |" if prec == "f16x3" else "")
        ops.PRECISION = "f16x3"
        print(row, flush=True)


if __name__ == "__main__":
    with torch.no_grad():
        main()
