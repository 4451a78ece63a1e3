"""Which call sites launch the NON-engine kernels of one headline forward (split passes, transposes, resizes, warps, gathers, pools):
name, integer arguments, call chain — the inventory behind DESIGN.md section 8 "non-engine tail" (round 6).  GPU box: python tools/trace_tail.py"""
import sys, os, collections, traceback, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from picopose_amd import ops, _lib
from picopose_amd.picopose import Net
vit="dinov2_vitb14"
net=Net(bench.make_cfg(vit)); bench.seeded_weights(net, 4, vit); net=net.cuda().eval()
ep=bench.make_end_points(32,162,"cuda",100)
with torch.no_grad():
    fe=net.feature_extractor
    ep["template_feature"]=torch.stack([torch.cat([fe(ep["tem_rgb"][b,s:s+54])[-1] for s in range(0,162,54)]) for b in range(32)])
outs=net(ep,5,next_real_rgb=ep["real_rgb"]); torch.cuda.synchronize()
cnt=collections.Counter()
L=_lib.lib()
names=["pp_split_activation_t","pp_transpose_batched","pp_resize_bilinear_nhwc_t","pp_resize_bilinear_nhwc","pp_warp_nhwc_t","pp_hl_patch_columns_t","pp_avgpool2_nhwc","pp_gather_rows"]
import ctypes
class Wrap:
    def __init__(self, f, name): self.f, self.name = f, name; self.argtypes=f.argtypes
    def __call__(self, *a):
        st=traceback.extract_stack(limit=8)
        where=" <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in st[:-1][-5:])
        ints=tuple(x for x in a if isinstance(x,int) and 0 < x < 10**7)[:7]
        cnt[(self.name, ints, where)]+=1
        return self.f(*a)
for n in names:
    setattr(L, n, Wrap(getattr(L,n), n))
# torch ops on the path
outs=net(ep,5,next_real_rgb=ep["real_rgb"]); torch.cuda.synchronize()
for k,v in sorted(cnt.items(), key=lambda kv:(kv[0][0], -kv[1])):
    print(v, k[0], k[1], k[2])
