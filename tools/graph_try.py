import os, sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
from picopose_amd.picopose import Net
from picopose_amd.pipeline import pnp_for_outputs
dev = torch.device("cuda", 0)
vit = "dinov2_vitb14"; Bl, N = 32, 162
net = Net(bench.make_cfg(vit)); bench.seeded_weights(net, 4, vit); net = net.to(dev).eval()
ep = bench.make_end_points(Bl, N, dev, 100)
fe = net.feature_extractor
with torch.no_grad():
    ep["template_feature"] = torch.stack([torch.cat([fe(ep["tem_rgb"][b, s:min(s + 54, N)])[-1] for s in range(0, N, 54)]) for b in range(Bl)])
def step_eager():
    outs = net(ep, 5)
    return outs, pnp_for_outputs(outs, ep["real_K"])
for _ in range(3): step_eager()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): step_eager()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print(f"eager: {dt*1e3:.2f} ms/step {Bl/dt:.1f} crops/s", flush=True)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): net(ep, 5)
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    outs = net(ep, 5)
def step_graph():
    g.replay()
    return outs, pnp_for_outputs(outs, ep["real_K"])
ref = step_eager()
got = step_graph()
torch.cuda.synchronize()
same = all(torch.equal(a[k], b[k]) for a, b in zip(ref[0], got[0]) for k in a)
print("graph outputs identical to eager:", same, flush=True)
t0 = time.perf_counter()
for _ in range(5): step_graph()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print(f"graph: {dt*1e3:.2f} ms/step {Bl/dt:.1f} crops/s", flush=True)
