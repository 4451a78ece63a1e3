"""STUDY: which kernels, if any, give different BITS when other processes keep the same GPU busy (time slicing / wave save-restore)?
`hammer` = an endless load (a GEMM + attention loop) to run in the background; `probe` runs each op N times on fixed inputs and counts the
iterations whose output differs from the first one.   usage: study_contention.py hammer | probe [iterations]"""
import os
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from picopose_amd import ops  # noqa: E402

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(3)
r = lambda *s: torch.randn(*s, device=dev, generator=g)  # noqa: E731

if sys.argv[1] == "hammer":
    x, w = r(16384, 768), r(3072, 768) / 28
    qkv = ops.linear(r(64 * 257, 768), r(2304, 768) / 28, out_split=True)
    t0 = time.time()
    while time.time() - t0 < float(sys.argv[2]) if len(sys.argv) > 2 else True:
        for _ in range(20):
            ops.linear(x, w, act="gelu")
            ops.attention(qkv, 64, 257, 12, 64, out_split=True)
        torch.cuda.synchronize()
    sys.exit(0)

N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
x768, w1, b1 = r(8224, 768), r(3072, 768) / 28, r(3072)
xs = ops.linear(r(8224, 768), r(768, 768) / 28, out_split=True)            # a Split operand (hl)
img = r(8, 32, 32, 256)
wc = ops.pack_conv_weight(r(256, 256, 3, 3) / 48)
qkv = ops.linear(r(16 * 257, 768), r(2304, 768) / 28, out_split=True)
lnw, lnb = r(768), r(768)


def f32(fn):
    def run():
        old = ops.PRECISION
        ops.PRECISION = "f32"
        try:
            return fn()
        finally:
            ops.PRECISION = old
    return run


OPS = {
    "linear f16x3, fp32 operand (split pass + dense engine)": lambda: ops.linear(x768, w1, b1, act="gelu"),
    "linear f16x3, operand input (dense engine, LDS-DMA)": lambda: ops.linear(xs, w1, b1, act="gelu"),
    "conv3x3 f16x3 (implicit GEMM, LDS-DMA)": lambda: ops.conv2d(img, wc, None, 3, pad=1, act="relu"),
    "attention f16x3 (LDS-DMA ring)": lambda: ops.attention(qkv, 16, 257, 12, 64, out_split=False),
    "layernorm": lambda: ops.layernorm(x768, lnw, lnb, 1e-6),
    "linear f32 engine (LDS-DMA)": f32(lambda: ops.linear(x768, w1, b1, act="gelu")),
    "conv3x3 f32 engine": f32(lambda: ops.conv2d(img, wc, None, 3, pad=1, act="relu")),
    "torch matmul (vendor library, for comparison)": lambda: x768 @ w1.t(),
}
for name, fn in OPS.items():
    first = fn()
    first = first.hl.clone() if isinstance(first, ops.Split) else first.clone()
    bad = 0
    for i in range(N):
        y = fn()
        y = y.hl if isinstance(y, ops.Split) else y
        if not torch.equal(y, first):
            bad += 1
    torch.cuda.synchronize()
    print(f"{name}: {bad} of {N} iterations with different bits", flush=True)

# ---- the training path's autograd functions: forward + backward on fixed inputs, every output compared with the first pass
from picopose_amd import autograd as A  # noqa: E402

A.DETERMINISTIC = True


def leaf(*s, scale=1.0):
    return (r(*s) * scale).requires_grad_(True)


bn = torch.nn.BatchNorm2d(256).to(dev).train()
x_img, w_c, b_c = leaf(4, 32, 32, 256), leaf(256, 256, 3, 3, scale=1 / 48), leaf(256)
x_tok, w_l, b_l = leaf(2056, 384), leaf(1536, 384, scale=1 / 20), leaf(1536)
qkv_t = leaf(8 * 257, 3 * 384)
f1, f2, flow = leaf(4, 32, 32, 256, scale=0.1), leaf(4, 32, 32, 256, scale=0.1), leaf(4, 32, 32, 2, scale=2.0)
lnw2, lnb2 = leaf(384), leaf(384)


def run_fb(fn, inputs):
    for t in inputs:
        t.grad = None
    y = fn()
    gy = torch.ones_like(y) * 0.01 + y.detach() * 1e-3
    y.backward(gy)
    return [y.detach().clone()] + [t.grad.clone() for t in inputs]


TRAIN = {
    "conv3x3 forward + backward": (lambda: A.conv2d(x_img, w_c, b_c, 3, pad=1, act="relu"), [x_img, w_c, b_c]),
    "batchnorm (batch statistics) + ReLU": (lambda: A.batchnorm_train(x_img, bn, relu=True), [x_img, bn.weight, bn.bias]),
    "linear + GELU": (lambda: A.linear(x_tok, w_l, b_l, "gelu"), [x_tok, w_l, b_l]),
    "layernorm": (lambda: A.layernorm(x_tok, lnw2, lnb2, 1e-6), [x_tok, lnw2, lnb2]),
    "attention (fused forward + backward)": (lambda: A._Attention.apply(qkv_t, 8, 257, 6, 64), [qkv_t]),
    "resize x2": (lambda: A.resize(x_img, 64, 64), [x_img]),
    "warp": (lambda: A._Warp.apply(f2, flow), [f2, flow]),
    "correlation lookup, 2 levels": (lambda: A._CorrLookup.apply(f1, f2, flow, 2, 2, 56), [f1, f2, flow]),
}
for name, (fn, inputs) in TRAIN.items():
    first = run_fb(fn, inputs)
    bad = 0
    for i in range(N // 4):
        got = run_fb(fn, inputs)
        if not all(torch.equal(a, b) for a, b in zip(first, got)):
            bad += 1
    torch.cuda.synchronize()
    print(f"train: {name}: {bad} of {N // 4} passes with different bits", flush=True)
