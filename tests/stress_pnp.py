"""Run-to-run determinism of the PnP/RANSAC kernel (start two of these at once to share the GPU between processes)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pnp_problems import make_batch
from picopose_amd.utils.pose_recovery import pose_recovery_ransac_pnp_batched

rng = np.random.default_rng(3)
b = make_batch(rng, 64, 3000, 0.4, 0.3)
t = {k: torch.from_numpy(v).cuda() for k, v in b.items() if isinstance(v, np.ndarray) and k in ("tar2d", "src3d", "K", "pose", "tar_pts", "src_pts")}
def run():
    r = pose_recovery_ransac_pnp_batched(t["tar2d"], t["src3d"], t["K"], t["pose"], t["tar_pts"], t["src_pts"])
    return [np.array(x) for x in r]
ref = run()
bad = 0
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 200):
    cur = run()
    if not all(np.array_equal(a, c) for a, c in zip(cur, ref)):
        bad += 1
print("differing:", bad)
