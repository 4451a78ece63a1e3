"""ViT-L/14 (C 1024, 16 heads, 24 blocks) through the full training step at 4 pairs: finite loss and gradients, and the same gradients
(to rounding) with the round-4 training switches off (fused attention, K slices, single ViT pass, patch scatter)."""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from netcfg import make_train_end_points
from picopose_amd import autograd as ag, ops, picopose as pp
from picopose_amd.picopose import Net
from picopose_amd.utils.loss_utils import Loss
from picopose_amd.utils.seeding import calibrated_state_dict
ns = types.SimpleNamespace
vit = "dinov2_vitl14"
cfg = ns(hypothesis=5, stage1=ns(vit_type=vit, pretrained=False, interaction_indexes=[[0, 5], [6, 11], [12, 17], [18, 23]]), stage2=ns(in_channel=256, hidden_dim=256),
         stage3=ns(nclass=1, in_channels=1024, use_bn=True, out_channels=[256, 512, 1024, 1024], num_levels=3, radius=4))
net = Net(cfg)
net.load_state_dict(calibrated_state_dict(net.state_dict(), 4, vit))
net = net.cuda().train()
ep = {k: v.cuda() for k, v in make_train_end_points(4, 11).items()}
def run():
    np.random.seed(0); torch.manual_seed(0)
    net.zero_grad(set_to_none=True)
    loss = Loss()(net(dict(ep)))["loss"]
    loss.backward()
    return float(loss), {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
l1, g1 = run()
ag.FUSED_ATTENTION = False; ops.KSPLIT = False; pp.BATCH_VIT_TRAIN = False; os.environ["PP_CORR_SCATTER_PER_PIXEL"] = "1"
l0, g0 = run()
assert np.isfinite(l1) and all(torch.isfinite(g).all() for g in g1.values())
worst = max(((float((g1[n] - g0[n]).abs().max() / g0[n].abs().max().clamp_min(1e-30)), n) for n in g0 if float(g0[n].abs().max()) > 1e-8), key=lambda t: t[0])
nerr = max(abs(float(g1[n].double().norm()) - float(g0[n].double().norm())) / max(float(g0[n].double().norm()), 1e-30) for n in g0 if float(g0[n].abs().max()) > 1e-8)
print(f"ViT-L, 4 pairs: loss {l1:.6f} (switches off: {l0:.6f}); {len(g1)} gradient tensors finite; switches on vs off: worst max|diff| / max|grad| {worst[0]:.2e} ({worst[1]}), worst norm difference {nerr:.2e}")
