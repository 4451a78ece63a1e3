import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from picopose_amd import ops
d = "cuda"; B = 32
x = torch.randn(B, 64, 64, 640, device=d); w = ops.pack_conv_weight(torch.randn(512, 640, 3, 3, device=d))
for _ in range(3): y = ops.conv2d(x, w, None, 3, 1, 1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): y = ops.conv2d(x, w, None, 3, 1, 1)
e1.record(); torch.cuda.synchronize()
print(f"dbg={os.environ.get('PP_GEMM_DBG','0')} conv 640->512 (incl. split pass): {e0.elapsed_time(e1)/10:.3f} ms")
