"""STUDY: the whole inference step (Net.forward of the headline workload, the next batch's query crops riding along) captured as ONE HIP
graph and replayed, against the eager step.  usage: graph_full_try.py [steps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from picopose_amd.picopose import Net  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
vit, Bl, N = "dinov2_vitb14", 32, 162
net = Net(bench.make_cfg(vit))
bench.seeded_weights(net, 4, vit)
net = net.to(dev).eval()
ep = bench.make_end_points(Bl, N, dev, 100)
with torch.no_grad():
    ep["template_feature"] = torch.stack([torch.cat([net.feature_extractor(ep["tem_rgb"][b, s:min(s + 54, N)])[-1] for s in range(0, N, 54)]) for b in range(Bl)])


def fwd():
    with torch.no_grad():
        return net(ep, 5, next_real_rgb=ep["real_rgb"])


for _ in range(3):
    outs = fwd()
torch.cuda.synchronize()


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


eager = timed(fwd)
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    fwd()
torch.cuda.current_stream().wait_stream(side)
with torch.cuda.graph(g):
    gouts = fwd()
torch.cuda.synchronize()
g.replay()
torch.cuda.synchronize()
same = all(torch.equal(a[k], b[k]) for a, b in zip(outs, gouts) for k in ("pred_poses", "pred_tar_pts", "tem_pose"))
graph = timed(g.replay)
eager2 = timed(fwd)
print(f"forward only (no PnP), ms per step: eager {eager:.2f} / {eager2:.2f}   graph replay {graph:.2f}   outputs equal: {same}")
