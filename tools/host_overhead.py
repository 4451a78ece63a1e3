"""Is the full step host-bound?  Time to ENQUEUE one step (no sync) vs time to finish it."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from picopose_amd.picopose import Net
vit = "dinov2_vitb14"; B, N = 32, 162
net = Net(bench.make_cfg(vit)); bench.seeded_weights(net, 4, vit); net = net.cuda().eval()
ep = bench.make_end_points(B, N, torch.device("cuda"), 100)
ep["template_feature"] = torch.randn(B, N, 768, 16, 16, device="cuda")
for _ in range(2): net(ep, 5)
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); out = net(ep, 5); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"enqueue {1e3*(t1-t0):.1f} ms, finished {1e3*(t2-t0):.1f} ms", flush=True)
