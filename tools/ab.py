"""A/B harness: time the stage-1 roofline kernel of several library variants, interleaved in ONE
process on ONE device (guide rule 24).  usage: tools_ab.py B N C suffix1 suffix2 ...  ('' = default)"""
import ctypes, os, sys, statistics
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B, N, C = (int(x) for x in sys.argv[1:4])
variants = [v if v != "default" else "" for v in sys.argv[4:]] or [""]
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
bank = torch.randn(B, N, C, 16, 16, device=dev, generator=g)
if os.environ.get('AB_CONST') == '1':
    bank.fill_(0.1234)  # data-dependence check: constant bank
q = torch.randn(B, C, 16, 16, device=dev, generator=g)
yy, xx = torch.meshgrid(torch.arange(224.0), torch.arange(224.0), indexing="ij")
m = (((yy - 111.5) ** 2 + (xx - 111.5) ** 2) < (0.4 * 224) ** 2).float()[None].repeat(B, 1, 1).to(dev)
libs = {}
for v in variants:
    L = ctypes.CDLL(os.path.join(ROOT, "picopose_amd", "lib", f"libpicopose_hip{('_' + v) if v else ''}.so"))
    vp, i32, f32, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
    L.pp_stage1_scores.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, vp, sz, vp, vp, vp]
    L.pp_stage1_workspace_bytes.argtypes = [i32, i32, i32, ctypes.POINTER(sz)]
    L.pp_prof_collect.argtypes = [ctypes.POINTER(f32), i32, ctypes.POINTER(i32)]
    libs[v] = L
need = ctypes.c_size_t()
libs[variants[0]].pp_stage1_workspace_bytes(B, N, C, ctypes.byref(need))
ws = torch.empty(need.value + (64 << 20), dtype=torch.uint8, device=dev)
out = torch.empty(B, N, device=dev)
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(L, mode=1):
    rc = L.pp_stage1_scores(bank.data_ptr(), q.data_ptr(), m.data_ptr(), 224, 224, B, N, C, mode, 0.0,
                            ws.data_ptr(), ws.numel(), out.data_ptr(), None, stream)
    assert rc == 0, rc
res = {v: [] for v in variants}
tot = {v: [] for v in variants}
for v in variants:
    for _ in range(3): run(libs[v])
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.0:  # the clocks ramp over the first ~0.3 s of load
    for _ in range(50): run(libs[variants[0]])
    torch.cuda.synchronize()
ROUNDS, IT = 6, 40
for r in range(ROUNDS):
    for v in variants:
        L = libs[v]
        L.pp_prof_enable(IT)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(IT): run(L)
        e1.record(); torch.cuda.synchronize()
        buf = (ctypes.c_float * IT)(); cnt = ctypes.c_int()
        L.pp_prof_collect(buf, IT, ctypes.byref(cnt))
        L.pp_prof_enable(0)
        res[v] += [buf[i] for i in range(cnt.value)]
        tot[v].append(e0.elapsed_time(e1) / IT)
gb = B * N * C * 256 * 4 / 1e9
for v in variants:
    k = statistics.median(res[v]); kmin = min(res[v]); t = statistics.median(tot[v])
    print(f"{v or 'default':10s} kernel med {k*1e3:7.1f} us min {kmin*1e3:7.1f} us  ({gb/k:6.0f} GB/s)   whole call med {t*1e3:7.1f} us", flush=True)
