"""Time pp_pnp_ransac on a bench-like batch (160 problems x ~3000 correspondences, outliers + noise) and check a few
problems against the CPU oracle.  PP_LIB_SUFFIX selects a variant build of the library."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from pnp_problems import make_batch, pose_errors
from oracle import pnp as op
from picopose_amd.utils.pose_recovery import pnp_launch
P, n = 160, int(sys.argv[1]) if len(sys.argv) > 1 else 3000
rng = np.random.default_rng(5)
b = make_batch(rng, P, n, 0.4, 0.7)
d = {k: torch.from_numpy(b[k]).cuda() for k in ("tar2d", "src3d", "K", "pose", "tar_pts", "src_pts")}
args = (d["tar2d"], d["src3d"], d["K"], d["pose"], d["tar_pts"], d["src_pts"])
for _ in range(3): out = pnp_launch(*args)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): out = pnp_launch(*args)
e1.record(); torch.cuda.synchronize()
print(f"pnp kernel: {e0.elapsed_time(e1) / 10:.3f} ms for {P} problems x {n} pts (lib suffix '{os.environ.get('PP_LIB_SUFFIX', '')}')")
rot, tvec, ratio, ok, npts = [t.cpu().numpy() for t in out]
ang, dt = pose_errors(rot, tvec, b["R"], b["t"])
print("success", ok.mean(), "rot deg p50/p95", np.quantile(ang, 0.5), np.quantile(ang, 0.95), "ratio mean", ratio.mean())
for i in range(4):
    r, t, ra, su = op.pose_recovery_ransac_pnp(b["tar2d"][i], b["src3d"][i], b["K"][i], b["pose"][i], b["tar_pts"][i], b["src_pts"][i], prob=i)
    print(i, "oracle ok", su, "inliers", ra * n, "gpu", ratio[i] * n, "rot diff", np.abs(r - rot[i]).max(), "t diff", np.abs(t.ravel() - tvec[i]).max())
for it in (1, 32, 64, 128, 150, 256):
    for _ in range(2): pnp_launch(*args, iterations=it)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): pnp_launch(*args, iterations=it)
    e1.record(); torch.cuda.synchronize()
    print(f"iterations {it}: {e0.elapsed_time(e1) / 10:.3f} ms")
