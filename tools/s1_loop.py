"""Measurement aid: the stage-1 roofline kernel in a loop for a few seconds, kernel time per second of run."""
import ctypes, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from picopose_amd import _lib
from picopose_amd.utils import matching as hm
B, N, C = 32, 162, 768
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
if len(sys.argv) > 2 and sys.argv[2] != "default":  # side-by-side tuning build (PP_LIB_SUFFIX)
    _lib.LIB = _lib.LIB.replace(".so", "_" + sys.argv[2] + ".so")
g = torch.Generator(device="cuda").manual_seed(1)
bank = torch.randn(B, N, C, 16, 16, device="cuda", generator=g)
q = torch.randn(B, C, 16, 16, device="cuda", generator=g)
m = torch.ones(B, 224, 224, device="cuda")
L = _lib.lib()
t_end = time.perf_counter() + secs
while time.perf_counter() < t_end:
    L.pp_prof_enable(100)
    for _ in range(100):
        hm.matching_templates(bank, q, None, m, topk=5, mode="fast")
    torch.cuda.synchronize()
    buf = (ctypes.c_float * 100)(); cnt = ctypes.c_int()
    L.pp_prof_collect(buf, 100, ctypes.byref(cnt))
    v = sorted(buf[i] for i in range(cnt.value))
    last = f"s1_main median {v[len(v)//2]*1e3:.1f} us  min {v[0]*1e3:.1f} us"
print(sys.argv[2:] or "default", last, flush=True)
