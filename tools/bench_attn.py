"""Time the fused attention of a ViT block, operand (hl / h) input.  usage: bench_attn.py [f16x3|f16] [B] [heads]
defaults: the headline batch of ViT-B (192 images x 257 tokens x 12 heads); `f16 384 16` = configs[4]'s ViT-L pass in the fp16 engine mode."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from picopose_amd import ops
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 192
heads = int(sys.argv[3]) if len(sys.argv) > 3 else 12
T, hd = 257, 64
ops.PRECISION = prec
C = heads * hd
x = torch.randn(B * T, C, device="cuda")
w = torch.randn(3 * C, C, device="cuda") / C ** 0.5
qkv = ops.linear(x, w, out_split=True)
for _ in range(3): ops.attention(qkv, B, T, heads, hd, out_split=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.attention(qkv, B, T, heads, hd, out_split=True)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
terms = 3 if prec == "f16x3" else 1
print(f"attention [{prec}, PP_ATTN_RING={os.environ.get('PP_ATTN_RING', 'default')}] B={B} T={T} heads={heads}: {ms * 1e3:.1f} us  ({terms * 4 * B * heads * T * T * hd / ms / 1e9:.0f} TFLOP/s executed)")
