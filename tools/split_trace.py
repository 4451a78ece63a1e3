import sys, os, collections, traceback, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from picopose_amd import ops
from picopose_amd.picopose import Net
from picopose_amd.pipeline import pnp_for_outputs
vit="dinov2_vitb14"
net=Net(bench.make_cfg(vit)); bench.seeded_weights(net, 4, vit); net=net.cuda().eval()
ep=bench.make_end_points(32,162,"cuda",100)
with torch.no_grad():
    fe=net.feature_extractor
    ep["template_feature"]=torch.stack([torch.cat([fe(ep["tem_rgb"][b,s:s+54])[-1] for s in range(0,162,54)]) for b in range(32)])
outs=net(ep,5); torch.cuda.synchronize()
cnt=collections.Counter()
orig=ops.split_activation
def traced(x,B,P,C,bs,rs,relu=False,into=None):
    st=traceback.extract_stack(limit=6)
    where=" <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in st[:-1][-4:])
    cnt[(B,P,C,relu,where)]+=1
    return orig(x,B,P,C,bs,rs,relu,into=into)
ops.split_activation=traced
outs=net(ep,5); torch.cuda.synchronize()
for k,v in sorted(cnt.items(), key=lambda kv:-kv[0][0]*kv[0][1]*kv[0][2]*kv[1]):
    print(v, k[:4], f"{k[0]*k[1]*k[2]*v*12/1e6:.0f} MB", k[4])
