"""Which GEMM / conv launches of ONE inference step (default bench workload) take the time: per-launch engine records grouped by
(M, N, K, conv, tile configuration).  usage: infer_gemm_trace.py [cached]    (cached: the extended template bank)"""
import collections
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from picopose_amd import _lib  # noqa: E402
from picopose_amd.picopose import Net  # noqa: E402

from picopose_amd import ops  # noqa: E402

if os.environ.get("PP_TRACE_MODE"):     # "f32" (exact) / "f16x3" (default) / "f16"
    ops.PRECISION = os.environ["PP_TRACE_MODE"]
dev = torch.device("cuda", 0)
vit, Bl, N = "dinov2_vitb14", 32, 162
net = Net(bench.make_cfg(vit))
bench.seeded_weights(net, 4, vit)
net = net.to(dev).eval()
ep = bench.make_end_points(Bl, N, dev, 100)
if "cached" in sys.argv[1:]:
    banks = [net.precompute_templates(ep["tem_rgb"][b]) for b in range(Bl)]
    ep["template_feature"] = torch.stack([bk["feature"] for bk in banks])
    ep["template_cache"] = {"obj_index": torch.arange(Bl, device=dev), "dpt": [torch.stack([bk["dpt"][k] for bk in banks]) for k in range(3)]}
    del banks
else:
  with torch.no_grad():
    ep["template_feature"] = torch.stack([torch.cat([net.feature_extractor(ep["tem_rgb"][b, s:min(s + 54, N)])[-1] for s in range(0, N, 54)]) for b in range(Bl)])
L = _lib.lib()
for i in range(3):
    if i == 2:
        _lib.check(L.pp_prof_gemm_enable(8192), "enable")
    net(ep, 5)
torch.cuda.synchronize()
cap = 8192
shape, ms, fl, cnt = (ctypes.c_int * (6 * cap))(), (ctypes.c_float * cap)(), (ctypes.c_double * cap)(), ctypes.c_int()
_lib.check(L.pp_prof_gemm_records(cap, shape, ms, fl, ctypes.byref(cnt)), "records")
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for i in range(cnt.value):
    M, Nn, K, ck, cf, kind = (shape[6 * i + k] for k in range(6))
    a = agg[(M, Nn, K, ck, cf, kind)]
    a[0] += 1; a[1] += ms[i]; a[2] += fl[i]
tot = sum(a[1] for a in agg.values())
print(f"{cnt.value} launches, {tot:.1f} ms")
for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    M, Nn, K, ck, cf, kind = key
    print(f"  {a[1]:7.3f} ms x{a[0]:3d}  M={M:7d} N={Nn:5d} K={K:6d} {'conv ' if ck else 'dense'} cfg={cf} {'engine' if kind == 0 else 'fly'}  {a[2] / max(a[1], 1e-9) / 1e9:6.1f} TF/s useful")
