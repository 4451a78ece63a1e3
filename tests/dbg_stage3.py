"""Per-level comparison of the HIP stage 3 with the CPU oracle on the calibrated ViT-S fixture inputs (debug aid)."""
import os, sys
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from netcfg import make_end_points, small_cfg, HEADS, TAKE
from oracle import nets as on, geometry as og, matching as om
from oracle.weights import calibrated_state_dict
from picopose_amd import ops
from picopose_amd.picopose import Net
torch.set_num_threads(16)
mode = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
ops.PRECISION = mode
vit = "dinov2_vits14"
net = Net(small_cfg()); sd = calibrated_state_dict(net.state_dict(), 4, vit); ops.CHECK_SATURATION = True; net.load_state_dict(sd); net = net.cuda().eval()
B = 2
with torch.no_grad():
    ep = make_end_points(B, 4, 41, dome=True)
    fr = on.vit_features(sd, ep["real_rgb"], HEADS, TAKE); ft = on.vit_features(sd, ep["tem_rgb"][:, 0], HEADS, TAKE)
    sim = om.matching_features_similarity(ft[-1], fr[-1], ep["tem_mask"][:, 0], None)
    t, s, ip = on.affine_regressor(sd, sim)
    Ms = og.calc_pred_Ms(s, ip, t, ep["tem_pose"][:, 0], ep["tem_K"][:, 0], ep["tem_M"][:, 0])
    f0, c0 = og.compute_init_correspondences(Ms, ep["tem_mask"][:, 0])
    dt, dr = on.dpt_head(sd, ft), on.dpt_head(sd, fr)
    fl, ce = on.flow_decoder(sd, dt, dr, f0, c0)
    # HIP: same oracle inputs into each piece
    orr = net.offset_regressor
    gdt = orr.dpt_head([f.cuda() for f in ft]); gdr = orr.dpt_head([f.cuda() for f in fr])
    for l in range(3):
        print(f"dpt{l}: max|ref| {dt[l].abs().max():.1f} err tem {(gdt[l].cpu()-dt[l]).abs().max():.4f} real {(gdr[l].cpu()-dr[l]).abs().max():.4f}")
    # flow decoder fed with the ORACLE's dpt maps
    gfl, gce = orr.flow_decoder([d.cuda() for d in dt], [d.cuda() for d in dr], f0.cuda(), c0.cuda())
    for l in range(3):
        ef, ec = (gfl[l].cpu() - fl[l]).abs(), (gce[l].cpu() - ce[l]).abs()
        print(f"level {l}: flow max|ref| {fl[l].abs().max():.2f} err max {ef.max():.4f} mean {ef.mean():.5f} | cert max|ref| {ce[l].abs().max():.2f} err max {ec.max():.4f} mean {ec.mean():.5f}")
