"""Time of pp_corr_lookup_backward_nhwc by number of pyramid levels (where do the 12 ms of the training step go?)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from picopose_amd import _lib, ops
B, H, W, C, r = 32, 64, 64, 256, 2
g = torch.Generator(device="cuda").manual_seed(0)
f1 = torch.randn(B, H, W, C, device="cuda", generator=g)
f2 = torch.randn(B, H, W, C, device="cuda", generator=g)
for fmag in (0.1, 1.0):
    flow = torch.randn(B, H, W, 2, device="cuda", generator=g) * fmag
    for levels in (1, 2, 3):
        win = (2 * r + 1) ** 2
        dout = torch.randn(B, H, W, levels * win, device="cuda", generator=g)
        pyr = [f2]
        for _ in range(levels - 1):
            pyr.append(ops.avgpool2(pyr[-1]))
        df1 = torch.empty_like(f1); dflow = torch.empty(B, H, W, 2, device="cuda")
        arr = ctypes.c_void_p * 3
        fl = arr(*[t.data_ptr() for t in pyr] + [None] * (3 - levels))
        dpyr = [torch.zeros_like(t) for t in pyr]
        dl = arr(*[t.data_ptr() for t in dpyr] + [None] * (3 - levels))
        def run():
            _lib.check(_lib.lib().pp_corr_lookup_backward_nhwc(f1.data_ptr(), fl, flow.data_ptr(), dout.data_ptr(), B, H, W, C, levels, r, 2, dout.shape[-1],
                                                               df1.data_ptr(), dl, dflow.data_ptr(), _lib.stream_ptr()), "x")
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): run()
        e1.record(); torch.cuda.synchronize()
        print(f"flow sigma {fmag}: levels {levels}: {e0.elapsed_time(e1) / 3:.2f} ms")
