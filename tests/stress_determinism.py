"""Run-to-run determinism of the HIP forward under contention: the same forward repeated, every output compared bit for bit
with the first run (start two of these at once to share the GPU between processes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from netcfg import make_end_points, small_cfg  # noqa: E402

from picopose_amd.picopose import Net  # noqa: E402
from picopose_amd.utils.seeding import calibrated_state_dict  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
net = Net(small_cfg())
net.load_state_dict(calibrated_state_dict(net.state_dict(), 4, "dinov2_vits14"))
net = net.cuda().eval()
ep = {k: v.cuda() for k, v in make_end_points(2, 7, 55, dome=True).items()}
with torch.no_grad():
    ep["template_feature"] = torch.stack([net.feature_extractor(ep["tem_rgb"][b])[-1] for b in range(2)])
net.keep_stage3 = True
first = None
bad = 0
for r in range(reps):
    outs = net(ep, 3)
    fl, ce = net.last_stage3
    cur = (torch.stack([o["pred_tar_pts"] for o in outs]), torch.stack([o["pred_poses"] for o in outs]), fl.clone(), ce.clone())
    if first is None:
        first = cur
        continue
    names = ("pred_tar_pts", "pred_poses", "flow", "certainty")
    for n, a, b in zip(names, cur, first):
        if not torch.equal(a, b):
            d = (a.double() - b.double()).abs()
            print(f"rep {r} {n}: {int((d > 0).sum())} entries differ, max {float(d.max()):.3e}", flush=True)
            bad += 1
print("runs with a difference:", bad, "of", reps - 1)
