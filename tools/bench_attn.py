"""Time the fused attention of a ViT-B block at the headline batch (192 images x 257 tokens x 12 heads), hl input."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from picopose_amd import ops
B, T, heads, hd = 192, 257, 12, 64
x = torch.randn(B * T, 768, device="cuda")
w = torch.randn(3 * 768, 768, device="cuda") / 28
qkv = ops.linear(x, w, out_split=True)
for _ in range(3): ops.attention(qkv, B, T, heads, hd, out_split=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.attention(qkv, B, T, heads, hd, out_split=True)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"attention B={B} T={T} heads={heads}: {ms * 1e3:.1f} us  ({3 * 4 * B * heads * T * T * hd / ms / 1e9:.0f} TFLOP/s executed)")
