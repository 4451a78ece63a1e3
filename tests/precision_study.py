"""Per-family precision study of the contraction engine's product forms.

The product build evaluates an fp32 product as hi.hi + hi.lo + lo.hi on fp16 MFMA (3 terms, 22 operand bits).  Two cheaper
forms are emulated bit for bit by variant builds of the library that zero the lo term of an operand class:
    PP_LIB_SUFFIX=_a1    -DPP_STUDY_ACT_LO_ZERO                      2 terms: weights split, activations plain fp16
    PP_LIB_SUFFIX=_a1w1  -DPP_STUDY_ACT_LO_ZERO -DPP_STUDY_W_LO_ZERO  1 term : plain fp16 x fp16
Run once per build (the library is chosen at import); every run prints one JSON object with, per layer family on the
reference-generated golden inputs (ViT-B/14 for the backbone): max |error| relative to max |reference tensor|, and for the
calibrated ViT-B end-to-end case the discrete outputs (template ids, key-point slots).  tools/precision_table.py merges
the runs into the table of DESIGN.md.
"""
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
G = os.path.join(ROOT, "tests", "golden")


def rel(a, ref):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else a
    return float(np.abs(a - ref).max() / max(1e-12, np.abs(ref).max()))


def main():
    from netcfg import small_cfg
    from oracle.weights import seeded_state_dict
    from picopose_amd import ops

    out = {"lib": os.environ.get("PP_LIB_SUFFIX", "") or "3-term (product)"}
    ns = types.SimpleNamespace
    # ---- ViT-B/14 and ViT-L/14 feature extractor (tests/golden/vit_wide.npz)
    from picopose_amd.model.stage1 import FeatureExtractor

    z = np.load(os.path.join(G, "vit_wide.npz"))
    for vit, idx in (("dinov2_vitb14", [[0, 2], [3, 5], [6, 8], [9, 11]]), ("dinov2_vitl14", [[0, 5], [6, 11], [12, 17], [18, 23]])):
        wseed, xseed = (int(v) for v in z[f"{vit}/seeds"])
        fe = FeatureExtractor(ns(vit_type=vit, pretrained=False, interaction_indexes=idx))
        fe.load_state_dict(seeded_state_dict(fe.state_dict(), wseed))
        fe = fe.cuda().eval()
        feats = fe(torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(xseed)).cuda())
        out[f"{vit}_levels"] = [float(np.abs(f[0, ::32].cpu().numpy() - z[f"{vit}/channel_probe"][l]).max() / z[f"{vit}/absmax"][l])
                                for l, f in enumerate(feats)]
    # ---- stage 2 and stage 3 modules (tests/golden/nets.npz, ViT-S width)
    from picopose_amd.model.stage2 import AffineRegressor
    from picopose_amd.model.stage3 import OffsetRegressor

    zn = np.load(os.path.join(G, "nets.npz"))
    t = {k: torch.from_numpy(zn[k]).cuda() for k in zn.files if zn[k].dtype == np.float32}
    cfg = small_cfg()
    ar = AffineRegressor(cfg.stage2)
    ar.load_state_dict(seeded_state_dict(ar.state_dict(), int(zn["aff/seed"])))
    ar = ar.cuda().eval()
    tr, sc, ip = ar(t["aff/sim"])
    out["affine_regressor"] = max(rel(tr, zn["aff/translation"]), rel(sc, zn["aff/scale"]), rel(ip, zn["aff/inplane"]))
    orr = OffsetRegressor(cfg.stage3)
    orr.load_state_dict(seeded_state_dict(orr.state_dict(), int(zn["s3/seed"])))
    orr = orr.cuda().eval()
    dt = orr.dpt_head([t[f"s3/ft{i}"] for i in range(4)])
    dr = orr.dpt_head([t[f"s3/fr{i}"] for i in range(4)])
    out["dpt_head"] = max(rel(dt[0], zn["s3/dpt_t_path4"]), rel(dt[1][0, :, ::8, ::8], zn["s3/dpt_t_path3_probe"]),
                          rel(dt[2][0, :, ::16, ::16], zn["s3/dpt_t_path2_probe"]))
    fl, ce = orr.flow_decoder(dt, dr, t["s3/init_flow"], t["s3/init_cert"])
    out["flow_decoder_flow"] = [rel(fl[i], zn[f"s3/flow{i}"]) for i in range(3)]
    out["flow_decoder_cert"] = [rel(ce[i], zn[f"s3/cert{i}"]) for i in range(3)]
    # ---- end to end, calibrated ViT-B (tests/golden/e2e_calibrated.npz): the discrete outputs
    import test_e2e as te

    zz, B, N, hyp, ref, ep, dev, outs, flow, cert = te._hip_calibrated_forward(G, "vitb_b2n6")
    same_t = all(np.array_equal(outs[h]["tem_pose"].cpu().numpy(), ref[h]["tem_pose"]) for h in range(hyp))
    slots = [float((outs[h]["pred_tar_pts"].cpu().numpy() == ref[h]["pred_tar_pts"]).all(-1).mean()) for h in range(hyp)]
    out["e2e_vitb"] = {"same_templates": bool(same_t), "pred_poses_max_abs": max(float(np.abs(outs[h]["pred_poses"].cpu().numpy() - ref[h]["pred_poses"]).max()) for h in range(hyp)),
                       "keypoint_slots_equal": slots, "keypoint_slots_differing": [int(round((1 - s) * 4096 * B)) for s in slots]}
    print("PRECISION_STUDY " + json.dumps(out))


if __name__ == "__main__":
    main()
