"""Generate tests/golden/*.npz by running the REFERENCE itself on CPU.

Run in the build container only (needs /root/reference); the fixtures it writes
are data (inputs + the reference's outputs), committed under tests/golden/.
    python oracle/gen_golden.py [--only stage1]
Nothing under tests/, bench.py or the package imports this module.
"""
import argparse
import os
import sys

import numpy as np
import torch

REF = os.environ.get("PICOPOSE_REFERENCE", "/root/reference")
# PICOPOSE_GOLDEN_OUT: write somewhere else (tests/test_fixture_provenance.py regenerates into a temp dir and compares)
OUT = os.environ.get("PICOPOSE_GOLDEN_OUT") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def _ref():
    if REF not in sys.path:
        sys.path.insert(0, REF)


def disk_mask(B, size=224, frac=0.4):
    yy, xx = torch.meshgrid(torch.arange(float(size)), torch.arange(float(size)), indexing="ij")
    c = (size - 1) / 2.0
    return (((yy - c) ** 2 + (xx - c) ** 2) < (frac * size) ** 2).float()[None].repeat(B, 1, 1)


def gen_stage1():
    _ref()
    from utils.matching import matching_features_similarity, matching_templates

    cases = {}

    def add(name, bank, query, mask, topk):
        score, idx = matching_templates(bank.clone(), query.clone(), None, mask.clone(), topk=topk)
        cases[name] = dict(bank=bank.numpy(), query=query.numpy(), mask=mask.numpy(),
                           topk=np.int64(topk), score=score.numpy(), index=idx.numpy())

    g = torch.Generator().manual_seed(1234)
    # random, Bernoulli mask (query patch 0 sometimes unmasked -> column decisions live)
    B, N, C = 2, 6, 64
    add("random_bernoulli", torch.randn(B, N, C, 16, 16, generator=g), torch.randn(B, C, 16, 16, generator=g),
        (torch.rand(B, 224, 224, generator=g) < 0.7).float(), 3)
    # disk mask (patch 0 is background: exercises the idx != 0 logic with sim[0,:] == 0)
    B, N, C = 2, 5, 128
    add("random_disk", torch.randn(B, N, C, 16, 16, generator=g), torch.randn(B, C, 16, 16, generator=g),
        disk_mask(B), 5)
    # fully masked query -> every sim_avg is 0, top-k order is torch's tie order
    B, N, C = 1, 4, 64
    add("all_masked", torch.randn(B, N, C, 16, 16, generator=g), torch.randn(B, C, 16, 16, generator=g),
        torch.zeros(B, 224, 224), 2)
    # patch-0 winners: template patch 0 is a copy of many query patches' direction, and query
    # patch 0 of many template patches' direction, so idx == 0 decisions occur often
    B, N, C = 2, 4, 64
    bank = torch.randn(B, N, C, 256, generator=g)
    query = torch.randn(B, C, 256, generator=g)
    for t in range(0, 256, 5):
        query[:, :, t] = bank[:, 1, :, 0] + 0.3 * torch.randn(B, C, generator=g)
    for s in range(0, 256, 7):
        bank[:, 2, :, s] = query[:, :, 0] + 0.3 * torch.randn(B, C, generator=g)
    add("patch0_winners", bank.reshape(B, N, C, 16, 16), query.reshape(B, C, 16, 16),
        torch.ones(B, 224, 224), 4)
    # all-negative similarities in some rows (negative scores enter sim_avg), non-square mask size
    B, N, C = 1, 3, 64
    query = torch.randn(B, C, 16, 16, generator=g)
    bank = -query[:, None].repeat(1, N, 1, 1, 1) + 0.5 * torch.randn(B, N, C, 16, 16, generator=g)
    add("negative_scores_mask100", bank, query, (torch.rand(B, 100, 100, generator=g) < 0.8).float(), 3)
    # un-normalised, badly scaled features (the reference normalises; so must we)
    B, N, C = 1, 4, 64
    add("scaled_features", 37.0 * torch.randn(B, N, C, 16, 16, generator=g),
        0.01 * torch.randn(B, C, 16, 16, generator=g), disk_mask(B), 4)

    flat = {}
    for name, d in cases.items():
        for k, v in d.items():
            flat[f"{name}/{k}"] = v
    np.savez_compressed(os.path.join(OUT, "stage1_matching_templates.npz"), **flat)

    # stage-2 similarity volume (matching.py:6-26)
    sims = {}
    B, C = 2, 64
    src, tar = torch.randn(B, C, 16, 16, generator=g), torch.randn(B, C, 16, 16, generator=g)
    sm = (torch.rand(B, 224, 224, generator=g) < 0.6).float()
    out = matching_features_similarity(src.clone(), tar.clone(), sm.clone(), None)
    sims.update({"random/src": src.numpy(), "random/tar": tar.numpy(), "random/src_mask": sm.numpy(),
                 "random/out": out.numpy()})
    np.savez_compressed(os.path.join(OUT, "stage2_similarity.npz"), **sims)
    print("stage1/stage2-similarity fixtures written")


def _rot(B, g):
    q, _ = torch.linalg.qr(torch.randn(B, 3, 3, generator=g))
    return q * torch.sign(torch.det(q))[:, None, None]


def gen_geometry():
    """calc_pred_Ms, pose_recovery_2d_prediction, compute_init_correspondences,
    compute_stage3_correspondences and gather — outputs of the reference functions."""
    _ref()
    import types

    sys.modules.setdefault("cv2", types.ModuleType("cv2"))  # utils/pose_recovery.py imports cv2 at module level
    from utils.correspondence import compute_init_correspondences, compute_stage3_correspondences
    from utils.pose_recovery import pose_recovery_2d_prediction
    from utils.torch_utils import calc_pred_Ms, gather

    g = torch.Generator().manual_seed(4321)
    B = 6
    scale = torch.rand(B, generator=g) + 0.5
    inplane = torch.nn.functional.normalize(torch.randn(B, 2, generator=g), dim=1)
    trans = torch.randn(B, 2, generator=g)
    pose = torch.eye(4).repeat(B, 1, 1)
    pose[:, :3, :3] = _rot(B, g)
    pose[:, :3, 3] = torch.tensor([0.01, -0.02, 0.8]) + 0.05 * torch.randn(B, 3, generator=g)
    K = torch.tensor([[572.4114, 0, 320], [0, 573.57043, 240], [0, 0, 1.0]]).repeat(B, 1, 1)
    K[:, 0, 0] += 10 * torch.randn(B, generator=g)
    tM = torch.eye(3).repeat(B, 1, 1)
    tM[:, 0, 0] = tM[:, 1, 1] = 1.5
    tM[:, :2, 2] = torch.tensor([-300.0, -200.0]) + 5 * torch.randn(B, 2, generator=g)
    qM = torch.eye(3).repeat(B, 1, 1)
    qM[:, 0, 0] = qM[:, 1, 1] = 2.0
    qM[:, :2, 2] = torch.tensor([-100.0, -80.0]) + 5 * torch.randn(B, 2, generator=g)
    Ms = calc_pred_Ms(scale, inplane, trans, pose, K, tM)
    poses = pose_recovery_2d_prediction(qM, K, Ms, K, tM, pose)
    mask = (torch.rand(B, 224, 224, generator=g) > 0.4).float()
    flow0, cert0 = compute_init_correspondences(Ms, mask)
    eye = torch.eye(3).repeat(2, 1, 1)
    flow_id, cert_id = compute_init_correspondences(eye, torch.ones(2, 224, 224))  # identity -> 0.5 everywhere
    # keypoint selection with edge cases: logits at +-eps, targets on the bounds, far outside
    flow = 3 * torch.randn(B, 2, 64, 64, generator=g)
    cert = torch.randn(B, 1, 64, 64, generator=g)
    cert[0, 0, 0, :8] = torch.tensor([0.0, 1e-9, -1e-9, 1e-6, -1e-6, 1e-3, -1e-3, 5e-8])
    flow[1, :, 10, 10] = torch.tensor([-10.0, -10.0])     # tar exactly (0,0): excluded (strict >)
    flow[1, :, 11, 11] = torch.tensor([52.0 - 11, 63.0 - 11])  # y == H-1: excluded (strict <)
    flow[1, :, 12, 12] = torch.tensor([0.25 - 12, 0.75 - 12])  # inside, truncates to (0,0)
    flow[1, :, 13, 13] = torch.tensor([62.999 - 13, 1.5 - 13])
    flow[2] = 200.0                                            # everything out of bounds
    cert[3] = -5.0                                             # nothing certain
    tar, src = compute_stage3_correspondences(flow, cert)
    feat2 = torch.randn(1, 2, 64, 64, generator=g)
    feat3 = torch.randn(1, 3, 64, 64, generator=g)
    g2 = gather(feat2, tar[:1])
    g3 = gather(feat3, src[:1])
    g_empty = gather(feat3, src[2:3])
    np.savez_compressed(
        os.path.join(OUT, "geometry.npz"),
        scale=scale.numpy(), inplane=inplane.numpy(), trans=trans.numpy(), tem_pose=pose.numpy(), K=K.numpy(),
        tem_M=tM.numpy(), query_M=qM.numpy(), pred_Ms=Ms.numpy(), pred_poses=poses.numpy(), mask=mask.numpy(),
        init_flow=flow0.numpy(), init_cert=cert0.numpy(), init_flow_identity=flow_id.numpy(),
        init_cert_identity=cert_id.numpy(), flow=flow.numpy(), cert=cert.numpy(), tar_pts=tar.numpy(),
        src_pts=src.numpy(), feat2=feat2.numpy(), feat3=feat3.numpy(), gather2=g2.numpy(), gather3=g3.numpy(),
        gather_empty=g_empty.numpy())
    print("geometry fixtures written")


def _small_cfg():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
    from netcfg import small_cfg

    return small_cfg()


def gen_nets():
    """Reference modules (FeatureExtractor, AffineRegressor, DPTHead, FlowDecoder, CorrLookup) with weights from
    oracle/weights.seeded_state_dict: only the seed, the small inputs and the outputs are stored."""
    _ref()
    sys.path.insert(0, os.path.join(REF, "model"))
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from oracle import ref_shims

    ref_shims.install()
    from model.stage1.feature_extractor import FeatureExtractor
    from model.stage2.affine_regressor import AffineRegressor
    from model.stage3.offset_regressor import OffsetRegressor
    from utils.corr_lookup import CorrLookup
    from model.stage3.raft_decoder import CorrelationPyramid

    from oracle.weights import seeded_state_dict

    cfg = _small_cfg()
    g = torch.Generator().manual_seed(99)
    out = {}
    with torch.no_grad():
        fe = FeatureExtractor(cfg.stage1).eval()
        fe.load_state_dict(seeded_state_dict(fe.state_dict(), 11))
        x = torch.randn(1, 3, 224, 224, generator=g)
        feats = fe(x)
        out["vit/seed"] = np.int64(11)
        out["vit/x"] = x.numpy()
        out["vit/feat_last"] = feats[-1].numpy()
        out["vit/feat_probe"] = torch.stack([f[0, :, 3, 5] for f in feats]).numpy()   # one pixel of every level

        ar = AffineRegressor(cfg.stage2).eval()
        ar.load_state_dict(seeded_state_dict(ar.state_dict(), 12))
        sim = torch.rand(3, 256, 16, 16, generator=g)
        t, s, ip = ar(sim)
        out.update({"aff/seed": np.int64(12), "aff/sim": sim.numpy(), "aff/translation": t.numpy(), "aff/scale": s.numpy(),
                    "aff/inplane": ip.numpy()})

        orr = OffsetRegressor(cfg.stage3).eval()
        orr.load_state_dict(seeded_state_dict(orr.state_dict(), 13))
        ft = [0.5 * torch.randn(1, 384, 16, 16, generator=g) for _ in range(4)]
        fr = [0.5 * torch.randn(1, 384, 16, 16, generator=g) for _ in range(4)]
        dt = orr.dpt_head(ft)
        dr = orr.dpt_head(fr)
        flow0 = 0.5 + torch.randn(1, 2, 16, 16, generator=g)
        cert0 = (torch.rand(1, 1, 16, 16, generator=g) > 0.3).float()
        fl, ce = orr.flow_decoder(dt, dr, flow0, cert0)
        out.update({"s3/seed": np.int64(13), "s3/init_flow": flow0.numpy(), "s3/init_cert": cert0.numpy()})
        for i in range(4):
            out[f"s3/ft{i}"] = ft[i].numpy()
            out[f"s3/fr{i}"] = fr[i].numpy()
        out["s3/dpt_t_path4"] = dt[0].numpy()
        out["s3/dpt_t_path3_probe"] = dt[1][0, :, ::8, ::8].numpy()
        out["s3/dpt_t_path2_probe"] = dt[2][0, :, ::16, ::16].numpy()
        for i in range(3):
            out[f"s3/flow{i}"] = fl[i].numpy()
            out[f"s3/cert{i}"] = ce[i].numpy()

        # correlation pyramid + lookup on its own (3 levels, radius 2, flows that leave the image)
        f1 = torch.randn(2, 64, 16, 16, generator=g)
        f2 = torch.randn(2, 64, 16, 16, generator=g)
        fw = 3 * torch.randn(2, 2, 16, 16, generator=g)
        look = CorrLookup(radius=2)(CorrelationPyramid(num_levels=3)(f1, f2), fw)
        out.update({"corr/f1": f1.numpy(), "corr/f2": f2.numpy(), "corr/flow": fw.numpy(), "corr/out": look.numpy()})
    np.savez_compressed(os.path.join(OUT, "nets.npz"), **out)
    print("nets fixtures written")


def gen_e2e():
    """Reference Net.forward (eval) end to end: BASELINE configs[0]-like (1 crop vs 4 templates, hyp 2, ViT-S).
    Inputs and weights are regenerated from seeds by tests/netcfg.make_end_points / oracle.weights."""
    _ref()
    sys.path.insert(0, os.path.join(REF, "model"))
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from oracle import ref_shims

    ref_shims.install()
    import picopose as ref_picopose

    from oracle.weights import seeded_state_dict

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
    from netcfg import make_end_points

    cfg = _small_cfg()
    out = {}
    with torch.no_grad():
        net = ref_picopose.Net(cfg).eval()
        net.load_state_dict(seeded_state_dict(net.state_dict(), 21))
        for tag, (B, N, hyp, seed) in {"b1n4": (1, 4, 2, 31), "b2n3": (2, 3, 3, 32)}.items():
            ep = make_end_points(B, N, seed, feature_fn=net.feature_extractor)
            res = net({k: v.clone() for k, v in ep.items()}, hyp)
            out[f"{tag}/meta"] = np.array([B, N, hyp, seed, 21], dtype=np.int64)
            out[f"{tag}/template_feature_probe"] = ep["template_feature"][:, :, ::16, 3, 5].numpy()
            out[f"{tag}/tem_pose_all"] = ep["tem_pose"].numpy()
            for k, o in enumerate(res):
                for key, val in o.items():
                    out[f"{tag}/h{k}/{key}"] = val.numpy()
    np.savez_compressed(os.path.join(OUT, "e2e.npz"), **out)
    print("e2e fixtures written")


def _cfg(vit):
    import types

    ns = types.SimpleNamespace
    C, idx = {"dinov2_vits14": (384, [[0, 2], [3, 5], [6, 8], [9, 11]]), "dinov2_vitb14": (768, [[0, 2], [3, 5], [6, 8], [9, 11]]),
              "dinov2_vitl14": (1024, [[0, 5], [6, 11], [12, 17], [18, 23]])}[vit]
    return ns(hypothesis=5, stage1=ns(vit_type=vit, pretrained=False, interaction_indexes=idx),
              stage2=ns(in_channel=256, hidden_dim=256),
              stage3=ns(nclass=1, in_channels=C, use_bn=True, out_channels=[256, 512, 1024, 1024], num_levels=3, radius=4))


def _cal_arrays(cal):
    """calibration dict -> arrays stored in a fixture (tests rebuild the weights from them, not from the table)."""
    return {"cal_flow": np.array(cal["flow"], np.float64), "cal_cert": np.array(cal["cert"], np.float64),
            "cal_proj_bn": np.float64(cal["proj_bn"]),
            **{f"cal_affine_{h}": np.array([g, *shift], np.float64) for h, (g, shift) in cal["affine"].items()}}


# (vit, B, N, hyp, seed).  vitb_b2n6: the width of BASELINE configs[2] with hyp = 5, so the top-5 ORDER of stage 1 is compared with
# the reference's at least once; vitl_b2n4: the backbone of configs[4] / config/base.yaml with a batch and two hypotheses.
E2E_CAL_CASES = {"vits_b2n4": ("dinov2_vits14", 2, 4, 3, 41), "vitb_b2n6": ("dinov2_vitb14", 2, 6, 5, 42),
                 "vitl_b2n4": ("dinov2_vitl14", 2, 4, 2, 43)}


def gen_e2e_calibrated(only=None):
    """Reference Net.forward (eval) with CALIBRATED heads and the dome geometry (tests/netcfg.dome_points): about half of
    the 4096 key-point slots are valid per hypothesis, so the network -> key-points -> PnP chain is compared on realistic
    occupancy.  ViT-S (B=2, N=4, hyp 3) and ViT-B (B=1, N=3, hyp 2: the 12-head / K=768/3072 shapes of
    BASELINE configs[2] inside a net-vs-reference comparison).  Weights = oracle.weights.seeded_state_dict(seed 4) with
    the stored calibration applied; mmcv.cnn.ConvModule is the stand-in of oracle/ref_shims.py (mmcv 2.0.0 arithmetic
    for norm_cfg=None: Conv2d + bias, ReLU)."""
    _ref()
    sys.path.insert(0, os.path.join(REF, "model"))
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from oracle import ref_shims

    ref_shims.install()
    import picopose as ref_picopose

    from oracle.weights import AFFINE_CALIBRATION, HEAD_CALIBRATION, PROJ_BN_GAIN, apply_head_calibration, seeded_state_dict

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
    from netcfg import make_end_points

    out = {}
    path = os.path.join(OUT, "e2e_calibrated.npz")
    if only is not None:   # add / refresh some cases, keep the stored arrays of the others
        old = np.load(path)
        out = {k: old[k] for k in old.files if k.split("/")[0] not in only}
    with torch.no_grad():
        for tag, (vit, B, N, hyp, seed) in E2E_CAL_CASES.items():
            if only is not None and tag not in only:
                continue
            net = ref_picopose.Net(_cfg(vit)).eval()
            cal = dict(HEAD_CALIBRATION[vit], affine=AFFINE_CALIBRATION, proj_bn=PROJ_BN_GAIN)
            net.load_state_dict(apply_head_calibration(seeded_state_dict(net.state_dict(), 4), cal))
            ep = make_end_points(B, N, seed, feature_fn=net.feature_extractor, dome=True)
            res = net({k: v.clone() for k, v in ep.items()}, hyp)
            out[f"{tag}/meta"] = np.array([B, N, hyp, seed, 4], dtype=np.int64)
            out[f"{tag}/vit"] = np.array(vit)
            for k, v in _cal_arrays(cal).items():
                out[f"{tag}/{k}"] = v
            out[f"{tag}/template_feature_probe"] = ep["template_feature"][:, :, ::16, 3, 5].numpy()
            out[f"{tag}/tem_pose_all"] = ep["tem_pose"].numpy()
            for k, o in enumerate(res):
                print(tag, "hyp", k, "valid key-points", (o["pred_tar_pts"][..., 0] >= 0).sum(1).tolist())
                for key, val in o.items():
                    if key in ("tar_pts_2d", "src_pts_3d"):   # pure functions of the inputs (a permute of real_pts2d / the
                        continue                              # selected template's dome): rebuilt by the test, not stored
                    out[f"{tag}/h{k}/{key}"] = val.numpy()
    np.savez_compressed(path, **out)
    print("calibrated e2e fixtures written")


def gen_e2e_calibrated_vitl():
    gen_e2e_calibrated(only=("vitl_b2n4",))


def gen_vit_wide():
    """FeatureExtractor of the reference at ViT-B/14 and ViT-L/14 (configs[2] / config/base.yaml widths): B=1, the input is
    regenerated from its seed, only probes of the 4 returned levels are stored."""
    _ref()
    sys.path.insert(0, os.path.join(REF, "model"))
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from model.stage1.feature_extractor import FeatureExtractor

    from oracle.weights import seeded_state_dict

    out = {}
    with torch.no_grad():
        for vit, wseed, xseed in (("dinov2_vitb14", 51, 52), ("dinov2_vitl14", 53, 54)):
            fe = FeatureExtractor(_cfg(vit).stage1).eval()
            fe.load_state_dict(seeded_state_dict(fe.state_dict(), wseed))
            x = torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(xseed))
            feats = fe(x)
            out[f"{vit}/seeds"] = np.array([wseed, xseed], dtype=np.int64)
            out[f"{vit}/pixel_probe"] = torch.stack([f[0, :, 3, 5] for f in feats]).numpy()          # (4, C)
            out[f"{vit}/channel_probe"] = torch.stack([f[0, ::32] for f in feats]).numpy()           # (4, C/32, 16, 16)
            out[f"{vit}/absmax"] = np.array([float(f.abs().max()) for f in feats])
    np.savez_compressed(os.path.join(OUT, "vit_wide.npz"), **out)
    print("ViT-B / ViT-L fixtures written")


def gen_state_dict():
    """Names, shapes and dtypes of the reference Net's state_dict for ViT-S/B/L (the authors' checkpoint must load)."""
    import json

    _ref()
    sys.path.insert(0, os.path.join(REF, "model"))
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from oracle import ref_shims

    ref_shims.install()
    import picopose as ref_picopose

    table = {}
    for vit in ("dinov2_vits14", "dinov2_vitb14", "dinov2_vitl14"):
        sd = ref_picopose.Net(_cfg(vit)).state_dict()
        table[vit] = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()]
        print(vit, len(sd), "tensors")
    with open(os.path.join(OUT, "state_dict_names.json"), "w") as f:
        json.dump(table, f, separators=(",", ":"))


def gen_preprocess():
    """get_bbox / get_square_bbox of the reference (utils/data_utils.py:131-196) on seeded masks and boxes, including the
    clamped-at-every-border cases.  data_utils imports cv2 / imageio at module level: empty in-memory modules stand in
    (neither function touches them)."""
    import types

    _ref()
    for name in ("cv2", "imageio"):
        sys.modules.setdefault(name, types.ModuleType(name))
    from utils.data_utils import get_bbox, get_square_bbox

    rng = np.random.RandomState(77)
    masks, boxes_in, sizes, ratios, out_mask, out_sq = [], [], [], [], [], []
    H, W = 48, 64
    for i in range(40):
        m = np.zeros((H, W), np.uint8)
        r0, c0 = rng.randint(0, H - 2), rng.randint(0, W - 2)
        r1, c1 = rng.randint(r0 + 1, H + 1), rng.randint(c0 + 1, W + 1)
        if i % 5 == 0:                       # thin / border-hugging / full-frame shapes
            r0, r1 = (0, H) if i % 10 == 0 else (H - 3, H)
        if i % 7 == 0:
            c0, c1 = 0, 2
        m[r0:r1, c0:c1] = rng.rand(r1 - r0, c1 - c0) < 0.8
        if not m.any():
            m[r0, c0] = 1
        ratio = [1.0, 1.2, 1.5][i % 3]
        masks.append(m)
        ratios.append(ratio)
        out_mask.append(get_bbox(m, ratio))
    for i in range(60):
        ih, iw = [(480, 640), (540, 720), (1080, 1920), (300, 200)][i % 4]
        r0, c0 = rng.randint(-20, ih), rng.randint(-20, iw)
        r1, c1 = r0 + rng.randint(1, ih), c0 + rng.randint(1, iw)
        ratio = [1.0, 1.2, 1.5][i % 3]
        boxes_in.append([r0, r1, c0, c1])
        sizes.append([ih, iw])
        out_sq.append(get_square_bbox([r0, r1, c0, c1], (ih, iw), ratio))
    np.savez_compressed(os.path.join(OUT, "preprocess_boxes.npz"), masks=np.stack(masks), mask_ratio=np.array(ratios),
                        mask_boxes=np.array(out_mask, np.int64), boxes=np.array(boxes_in, np.int64), sizes=np.array(sizes, np.int64),
                        box_ratio=np.array([[1.0, 1.2, 1.5][i % 3] for i in range(60)]), square_boxes=np.array(out_sq, np.int64))
    print("bbox fixtures written")


def synthetic_eval_case(seed=5, n_images=3, hyp=3):
    """Canned evaluator inputs shared by gen_run_test (which feeds them to the REFERENCE run_test.run_test) and by
    tests/test_pipeline_cpu.py (which feeds them to picopose_amd.pipeline): per image a few instances, per (instance,
    hypothesis) a canned PnP answer (R, t, inlier ratio, success) and a canned stage-2 pose.  The network outputs carry
    only an id that selects the canned answer — the loop semantics are what is under test, not the arithmetic."""
    rng = np.random.RandomState(seed)
    images = []
    uid = 0
    for i in range(n_images):
        n_inst = [3, 1, 5][i % 3]
        inst = []
        for k in range(n_inst):
            hyps = []
            for h in range(hyp):
                q, _ = np.linalg.qr(rng.randn(3, 3))
                ok = bool(rng.rand() > 0.35)
                hyps.append({"uid": uid, "R": q.tolist(), "t": (rng.randn(3) * 0.1 + [0, 0, 0.8]).tolist(),
                             "ratio": float(np.round(rng.rand(), 3)) if ok else 0.0, "ok": ok,
                             "stage2": np.vstack([np.hstack([np.linalg.qr(rng.randn(3, 3))[0], rng.randn(3, 1) * 0.1 + [[0], [0], [0.9]]]),
                                                  [[0, 0, 0, 1]]]).astype(np.float32).tolist()})
                uid += 1
            inst.append({"obj_id": int(rng.randint(1, 4)), "score": float(np.round(rng.rand(), 4)), "hyps": hyps})
        images.append({"scene_id": 10 + i, "img_id": 100 + 7 * i, "seg_time": 0.0, "instances": inst})
    return {"hyp": hyp, "bs": 2, "n_obj": 3, "images": images}


def gen_run_test():
    """The REFERENCE evaluator loop itself — run_test.run_test (run_test.py:100-221) — on the canned case above: instance
    mini-batches, PnP per (instance, hypothesis), stage-2 fallback, ranking by inlier ratio, the BOP csv rows.  It runs on
    the CPU with: a stub model and a stub dataset module (ours: they only carry ids), `pose_recovery_ransac_pnp` replaced
    by the canned table, `Tensor.cuda` / `torch.cuda.synchronize` neutralised, and empty in-memory modules for the
    third-party imports of run_test.py that this image lacks (omegaconf, pytorch_lightning, timm, cv2 — none is touched
    by the loop).  Stored: the case and the csv rows (without their last column, which is wall-clock time)."""
    import json
    import tempfile
    import types

    _ref()
    for name in ("omegaconf", "pytorch_lightning", "timm", "timm.scheduler", "cv2"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["omegaconf"].OmegaConf = object
    sys.modules["pytorch_lightning"].LightningModule = torch.nn.Module
    sys.modules["timm"].scheduler = sys.modules["timm.scheduler"]
    import run_test as rt

    case = synthetic_eval_case()
    hyp = case["hyp"]
    table = {h["uid"]: h for im in case["images"] for inst in im["instances"] for h in inst["hyps"]}

    def fake_pnp(tar_pts_2d, src_pts_3d, K, tem_pose, tar_pts, src_pts):
        h = table[int(tar_pts[0, 0])]
        return np.array(h["R"]), np.array(h["t"]).reshape(3, 1), h["ratio"], h["ok"]

    class Model(torch.nn.Module):
        def feature_extractor(self, x):
            return [torch.zeros(x.shape[0], 4, 16, 16)]

        def forward(self, inputs, hyp_):
            outs = []
            for tk in range(hyp_):
                uid = inputs["uid"][:, tk]
                outs.append({"tar_pts_2d": torch.zeros(len(uid), 2, 4, 4), "src_pts_3d": torch.zeros(len(uid), 3, 4, 4),
                             "tem_pose": torch.eye(4).repeat(len(uid), 1, 1), "pred_tar_pts": uid[:, None, None].repeat(1, 16, 2),
                             "pred_src_pts": uid[:, None, None].repeat(1, 16, 2),
                             "pred_poses": torch.tensor([table[int(u)]["stage2"] for u in uid], dtype=torch.float32)})
            return outs

    class BOPTestset(torch.utils.data.Dataset):
        def __init__(self, cfg, name, det):
            self.obj_idxs = {1: 0, 2: 1, 3: 2}
            self.n_template_view = 5

        def __len__(self):
            return len(case["images"])

        def __getitem__(self, i):
            im = case["images"][i]
            n = len(im["instances"])
            # tensor kinds and shapes of BOPTestset.__getitem__ (provider/bop_test_dataset.py:120-143)
            return {"scene_id": torch.IntTensor([im["scene_id"]]), "img_id": torch.IntTensor([im["img_id"]]),
                    "seg_time": torch.FloatTensor([im["seg_time"]]),
                    "score": torch.FloatTensor([[x["score"]] for x in im["instances"]]),
                    "obj_id": torch.IntTensor([[x["obj_id"]] for x in im["instances"]]),
                    "obj_idx": torch.IntTensor([[x["obj_id"] - 1] for x in im["instances"]]),
                    "real_K": torch.eye(3).repeat(n, 1, 1),
                    "uid": torch.tensor([[h["uid"] for h in x["hyps"]] for x in im["instances"]])}

        def get_templates(self, device):
            return {"tem_rgb": torch.zeros(case["n_obj"], self.n_template_view, 3, 14, 14)}

    mod = types.ModuleType("pp_fake_dataset")
    mod.BOPTestset = BOPTestset
    sys.modules["pp_fake_dataset"] = mod
    ns = types.SimpleNamespace
    cfg = ns(test_dataloader=ns(bs=case["bs"], num_workers=0, shuffle=False, drop_last=False, pin_memory=False),
             test_dataset=ns(name="pp_fake_dataset"), model=ns(hypothesis=hyp))
    rt.pose_recovery_ransac_pnp = fake_pnp
    saved = torch.Tensor.cuda, torch.cuda.synchronize
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.synchronize = lambda *a, **k: None
    try:
        with tempfile.TemporaryDirectory() as tmp:
            rt.run_test(Model(), cfg, tmp, "synthetic", None)
            (name,) = os.listdir(tmp)
            lines = open(os.path.join(tmp, name)).read().splitlines()
    finally:
        torch.Tensor.cuda, torch.cuda.synchronize = saved
    assert name == f"picopose-stage3-{hyp}hyp_synthetic-test.csv", name
    rows = [ln.rsplit(",", 1)[0] for ln in lines]            # drop the wall-clock column
    with open(os.path.join(OUT, "run_test_rows.json"), "w") as f:
        json.dump({"case": case, "csv_name": name, "rows": rows}, f, separators=(",", ":"))
    print(f"run_test fixture written: {len(rows)} rows")


def gen_train_forward(name="train_forward"):
    """The REFERENCE training forward itself: `Net.forward_train` (model/picopose.py:114-137) in train mode — key-point
    sampler, both ViT passes, InfoNCE / stage-2 / flow + certainty losses, BatchNorm on batch statistics with running-buffer
    updates — and `Loss.forward` (utils/loss_utils.py:10-21), on tests/netcfg.make_train_end_points (ViT-S, B=2, calibrated
    seeded weights).  The noisy affines `aug_gtM_noise` drew are recorded (product and oracle take them as an input) together
    with the numpy / torch seeds that reproduce them."""
    _ref()
    sys.path.insert(0, os.path.join(REF, "model"))
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from oracle import ref_shims

    ref_shims.install()
    import picopose as ref_picopose
    from utils.loss_utils import Loss

    from oracle.weights import AFFINE_CALIBRATION, HEAD_CALIBRATION, PROJ_BN_GAIN, apply_head_calibration, seeded_state_dict

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
    from netcfg import make_train_end_points, train_case

    B, seed, edit = train_case(name)
    vit, wseed = "dinov2_vits14", 4
    net = ref_picopose.Net(_cfg(vit)).train()
    cal = dict(HEAD_CALIBRATION[vit], affine=AFFINE_CALIBRATION, proj_bn=PROJ_BN_GAIN)
    net.load_state_dict(apply_head_calibration(seeded_state_dict(net.state_dict(), wseed), cal))
    ep = edit(make_train_end_points(B, seed))
    drawn = {}
    orig = ref_picopose.aug_gtM_noise

    def recording(end_points):
        drawn["pred_Ms"] = orig(end_points)
        return drawn["pred_Ms"]

    ref_picopose.aug_gtM_noise = recording
    np.random.seed(1000 + seed)
    torch.manual_seed(2000 + seed)
    with torch.no_grad():
        kp = net.compute_keypoint_data({k: v.clone() for k, v in ep.items()})
        res = net({k: v.clone() for k, v in ep.items()})
        tot = Loss()(res)
    ref_picopose.aug_gtM_noise = orig
    out = {"meta": np.array([B, seed, wseed, 1000 + seed, 2000 + seed], dtype=np.int64), "vit": np.array(vit)}
    for k, v in _cal_arrays(cal).items():
        out[k] = v
    out["real_pose"], out["tem_pose"] = ep["real_pose"].numpy(), ep["tem_pose"].numpy()
    out["pred_Ms"] = drawn["pred_Ms"].numpy()
    for k in ("src_pts", "tar_pts"):     # patch coordinates = integer pixels / 3.5 (or -1): stored as the integer pixels
        px = torch.where(kp[k] == -1, kp[k], (kp[k] * 3.5).round())
        assert px.abs().max() < 32000
        assert torch.equal(torch.where(px == -1, px, px / 3.5), kp[k])
        out[f"kp_{k}_px"] = px.numpy().astype(np.int16)
    for k, v in res.items():
        if "loss" in k:
            out[k] = v.numpy()
            print(k, float(v))
    out["total_loss"] = tot["loss"].numpy()
    print("total", float(tot["loss"]), "valid key-points", (kp["src_pts"][..., 0] != -1).sum(1).tolist())
    sd = net.state_dict()
    for layer in ("offset_regressor.dpt_head.scratch.refinenet4.resConfUnit2.bn1", "offset_regressor.dpt_head.scratch.refinenet2.resConfUnit1.bn2",
                  "offset_regressor.flow_decoder.proj.0.1", "offset_regressor.flow_decoder.proj.2.1"):
        for buf in ("running_mean", "running_var", "num_batches_tracked"):
            out[f"bn/{layer}.{buf}"] = sd[f"{layer}.{buf}"].numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("training-forward fixture written:", name)


def gen_train_forward_edge():
    gen_train_forward("train_forward_edge")


GRAD_SAMPLES = 65536   # entries kept of a gradient tensor larger than this (evenly strided over the flattened tensor)


def gen_train_grads(name="train_grads"):
    """The reference's OWN gradients for the first backward slice (picopose_amd/autograd.py): `Net.forward_train` in train mode
    under autograd on CPU (ViT-S, B = 2, the batch of train_forward.npz), then
      * d(loss_2d_trans + loss_scale + loss_inplane) / d(every parameter of affine_regressor)   (utils/loss_utils.py:177-186),
      * d(loss_info) / d(every parameter of the last ViT block)                                 (utils/loss_utils.py:144-175)
    by torch.autograd.grad.  Tensors above GRAD_SAMPLES entries are stored as an evenly strided sample of the flattened gradient
    plus its full L2 norm (affine_regressor.fc1.weight alone is 67 MB).
    name = "train_grads_dup": the batch of tests/netcfg.train_kwargs("train_grads_dup") (real crop = the smaller view: repeated rows
    in the InfoNCE gather of the real tokens) and ONE group, d(loss_info) / d(every parameter of dinov2) — keys gradi/..."""
    _ref()
    sys.path.insert(0, os.path.join(REF, "model"))
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from oracle import ref_shims

    ref_shims.install()
    import picopose as ref_picopose

    from oracle.weights import AFFINE_CALIBRATION, HEAD_CALIBRATION, PROJ_BN_GAIN, apply_head_calibration, seeded_state_dict

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
    from netcfg import make_train_end_points, train_case, train_kwargs

    dup = name == "train_grads_dup"
    B, seed, edit = train_case("train_grads_dup" if dup else "train_forward")
    vit, wseed = "dinov2_vits14", 4
    net = ref_picopose.Net(_cfg(vit)).train()
    cal = dict(HEAD_CALIBRATION[vit], affine=AFFINE_CALIBRATION, proj_bn=PROJ_BN_GAIN)
    net.load_state_dict(apply_head_calibration(seeded_state_dict(net.state_dict(), wseed), cal))
    ep = edit(make_train_end_points(B, seed, **train_kwargs(name)))
    drawn = {}
    orig = ref_picopose.aug_gtM_noise

    def recording(end_points):
        drawn["pred_Ms"] = orig(end_points)
        return drawn["pred_Ms"]

    ref_picopose.aug_gtM_noise = recording
    np.random.seed(1000 + seed)
    torch.manual_seed(2000 + seed)
    res = net({k: v.clone() for k, v in ep.items()})
    ref_picopose.aug_gtM_noise = orig
    out = {"meta": np.array([B, seed, wseed], dtype=np.int64), "vit": np.array(vit), "pred_Ms": drawn["pred_Ms"].detach().numpy()}
    for k, v in _cal_arrays(cal).items():
        out[k] = v
    out["real_pose"], out["tem_pose"] = ep["real_pose"].numpy(), ep["tem_pose"].numpy()
    from utils.loss_utils import Loss

    total = Loss()(res)["loss"]
    out["total_loss"] = total.detach().numpy()
    for k in [k for k in res if "loss" in k]:
        out[k] = res[k].detach().numpy()
        print(k, float(res[k]))
    groups = {"affine": ([(n, p) for n, p in net.named_parameters() if n.startswith("affine_regressor.")],
                         res["loss_2d_trans"] + res["loss_scale"] + res["loss_inplane"]),
              "vit_last": ([(n, p) for n, p in net.named_parameters() if n.startswith(f"feature_extractor.dinov2.blocks.{len(net.feature_extractor.dinov2.blocks) - 1}.")],
                           res["loss_info"])}
    # the wide slice ("vit+stage2"): what the stage-1 and stage-2 losses TOGETHER send into the whole ViT (InfoNCE directly, the
    # stage-2 losses through the similarity volume, utils/matching.py:6-26) — keys grad2/..., every parameter of dinov2
    groups["vit_all"] = ([(n, p) for n, p in net.named_parameters() if n.startswith("feature_extractor.dinov2.")],
                         res["loss_info"] + res["loss_2d_trans"] + res["loss_scale"] + res["loss_inplane"])
    # the full training step: d(Loss()(end_points)["loss"]) / d(EVERY parameter) — keys grad3/...
    groups["full"] = (list(net.named_parameters()), total)
    if dup:
        with torch.no_grad():   # how many real-token rows of the InfoNCE gather repeat (utils/loss_utils.py:150-160: cell = trunc(pt / 64 * 16))
            kp = net.compute_keypoint_data({k: v.clone() for k, v in ep.items()})
        # nearest 64 -> 16 sampling keeps every 4th key-point; its cell is trunc(pt * 16 / 64)  (pts are in 64-grid units)
        cells = []
        for b in range(B):
            t = kp["tar_pts"][b].reshape(64, 64, 2)[::4, ::4].reshape(-1, 2)
            cells.append((t[t[:, 0] != -1] * (16 / 64)).long())
        repeats = [int(len(c) - len(torch.unique(c[:, 1] * 16 + c[:, 0]))) for c in cells]
        print("InfoNCE rows per pair", [len(c) for c in cells], "of which repeats of an earlier cell", repeats)
        out["infonce_rows"], out["infonce_repeats"] = np.array([len(c) for c in cells]), np.array(repeats)
        assert min(repeats) > 10
        groups = {"vit_info": ([(n, p) for n, p in net.named_parameters() if n.startswith("feature_extractor.dinov2.")], res["loss_info"])}
    for gname, (params, loss) in groups.items():
        grads = torch.autograd.grad(loss, [p for _, p in params], retain_graph=True, allow_unused=True)
        prefix = {"vit_all": "grad2", "full": "grad3", "vit_info": "gradi"}.get(gname, "grad")
        for (n, p), g in zip(params, grads):
            out[f"{prefix}used/{n}"] = np.bool_(g is not None)
            g = torch.zeros_like(p) if g is None else g
            flat = g.detach().reshape(-1)
            stride = max(1, -(-flat.numel() // {"grad": GRAD_SAMPLES, "grad2": GRAD_SAMPLES // 16, "grad3": GRAD_SAMPLES // 32, "gradi": GRAD_SAMPLES // 16}[prefix]))   # (176 / 380 tensors)
            out[f"{prefix}/{n}"] = flat[::stride].numpy()
            out[f"{prefix}norm/{n}"] = np.float64(flat.double().norm())
            print(gname, n, tuple(p.shape), "stride", stride, "norm %.4g" % float(flat.double().norm()))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("training-gradient fixture written:", name)


def gen_train_grads_dup():
    gen_train_grads("train_grads_dup")


KINK_BAND = 5e-4


def gen_train_grads_f64():
    """The SAME training step as train_grads.npz (same weights, batch and augmentation draw: `pred_Ms` is taken from that fixture) with the
    reference evaluated in FLOAT64 — module in .double(), inputs in double, default dtype float64, and the reference's explicit `.float()`
    casts widened for the duration of this run — then d(Loss) / d(every parameter), stored as grad3f64/<name> at the strides of grad3/<name>.
    It is the arbiter for gradient tensors too ill-conditioned for an fp32-vs-fp32 bar (the DPT fusion blocks between
    batch-statistics BatchNorms); tests/test_train_gpu.py holds every tensor to a bar against these values, and — with the ReLU kinks
    recorded below pinned to float64's side — the fusion blocks to |HIP - f64| <= 2 |reference-fp32 - f64| + 1e-7 max|grad|."""
    _ref()
    sys.path.insert(0, os.path.join(REF, "model"))
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from oracle import ref_shims

    ref_shims.install()
    import picopose as ref_picopose

    from oracle.weights import AFFINE_CALIBRATION, HEAD_CALIBRATION, PROJ_BN_GAIN, apply_head_calibration, seeded_state_dict

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
    from netcfg import make_train_end_points, train_case, train_kwargs

    src = os.path.join(OUT, "train_grads.npz")     # (a regeneration into a scratch directory reads the committed fp32 fixture)
    if not os.path.exists(src):
        src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "train_grads.npz")
    f32 = np.load(src)
    B, seed, edit = train_case("train_forward")
    vit, wseed = "dinov2_vits14", 4
    net = ref_picopose.Net(_cfg(vit)).train()
    cal = dict(HEAD_CALIBRATION[vit], affine=AFFINE_CALIBRATION, proj_bn=PROJ_BN_GAIN)
    net.load_state_dict(apply_head_calibration(seeded_state_dict(net.state_dict(), wseed), cal))   # the fp32 weights, then widened
    net = net.double()
    ep = edit(make_train_end_points(B, seed, **train_kwargs("train_grads")))
    ep = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in ep.items()}
    orig_aug, orig_float, orig_dtype = ref_picopose.aug_gtM_noise, torch.Tensor.float, torch.get_default_dtype()
    ref_picopose.aug_gtM_noise = lambda end_points: torch.from_numpy(f32["pred_Ms"]).double()
    torch.Tensor.float = lambda self, *a, **k: self.double()
    torch.set_default_dtype(torch.float64)
    # The ReLU kinks of the DPT fusion blocks (dpt.py:82,87): every pre-activation the float64 forward holds within KINK_BAND of zero, as
    # (index into the NHWC-flattened map, float64 value) — keys kink_idx/<module>/<call>, kink_val/... with <module> a ResidualConvUnit
    # (its INPUT, the ReLU in front of conv1) or its bn1 (its OUTPUT, the ReLU in front of conv2) and <call> 0 = templates, 1 = real
    # crops.  An fp32 forward within rounding of float64 may hold such an element on the other side of zero; the gradient of that element
    # is then wholly different, and these blocks' parameter gradients are sums that cancel to 1e-7 of their terms
    # (profiles/r05/grad_f64.txt).  tests/test_train_gpu.py pins these elements to float64's values for the float64 comparison.
    kinks, calls = {}, {}

    def kink_site(name, take_output):
        def fwd(m, inp, o):
            c = calls.get(name, 0)
            calls[name] = c + 1
            v = (o if take_output else inp[0]).detach().permute(0, 2, 3, 1).reshape(-1)
            idx = torch.nonzero(v.abs() < KINK_BAND).reshape(-1)
            kinks[f"kink_idx/{name}/{c}"] = idx.numpy().astype(np.int64)
            kinks[f"kink_val/{name}/{c}"] = v[idx].numpy().astype(np.float64)
        return fwd

    for mname, m in net.named_modules():
        if ".scratch.refinenet" in mname and (mname.endswith(".resConfUnit1") or mname.endswith(".resConfUnit2")):
            m.register_forward_hook(kink_site(mname, False))
            m.bn1.register_forward_hook(kink_site(mname + ".bn1", True))
    try:
        np.random.seed(1000 + seed)
        torch.manual_seed(2000 + seed)
        res = net(ep)
        from utils.loss_utils import Loss

        total = Loss()(res)["loss"]
        params = list(net.named_parameters())
        grads = torch.autograd.grad(total, [p for _, p in params], allow_unused=True)
    finally:
        ref_picopose.aug_gtM_noise, torch.Tensor.float = orig_aug, orig_float
        torch.set_default_dtype(orig_dtype)
    assert total.dtype == torch.float64
    out = {"meta": np.array([B, seed, wseed], dtype=np.int64), "total_loss": total.detach().numpy(),
           "total_loss_f32_reference": f32["total_loss"], "kink_band": np.float64(KINK_BAND)}
    out.update(kinks)
    print("ReLU kinks of the fusion blocks: %d sites, %d pre-activations within %g of zero" % (len(kinks) // 2, sum(len(v) for k, v in kinks.items() if k.startswith("kink_idx/")), KINK_BAND))
    print("total loss f64 %.9f   fp32 reference %.9f" % (float(total), float(f32["total_loss"])))
    for k in [k for k in res if "loss" in k]:
        out[k] = res[k].detach().numpy()
    worst = (0.0, "")
    for (n, p), g in zip(params, grads):
        out[f"grad3f64used/{n}"] = np.bool_(g is not None)
        assert bool(out[f"grad3f64used/{n}"]) == bool(f32[f"grad3used/{n}"]), n
        g = torch.zeros_like(p) if g is None else g
        flat = g.detach().reshape(-1)
        stride = max(1, -(-flat.numel() // (GRAD_SAMPLES // 32)))
        out[f"grad3f64/{n}"] = flat[::stride].numpy()
        out[f"grad3f64norm/{n}"] = np.float64(flat.norm())
        ref32 = f32[f"grad3/{n}"]
        assert ref32.shape == out[f"grad3f64/{n}"].shape, n
        d = float(np.abs(ref32 - out[f"grad3f64/{n}"]).max()) / max(float(np.abs(out[f"grad3f64/{n}"]).max()), 1e-30)
        if bool(f32[f"grad3used/{n}"]) and d > worst[0]:
            worst = (d, n)
    print("fp32 reference against float64, worst tensor: %.3g of its maximum (%s)" % worst)
    np.savez_compressed(os.path.join(OUT, "train_grads_f64.npz"), **out)
    print("training-gradient fixture written: train_grads_f64")


GENERATORS = {"stage1": gen_stage1, "geometry": gen_geometry, "nets": gen_nets, "e2e": gen_e2e,
              "e2e_calibrated": gen_e2e_calibrated, "vit_wide": gen_vit_wide, "state_dict": gen_state_dict,
              "preprocess": gen_preprocess, "run_test": gen_run_test, "train_forward": gen_train_forward, "train_forward_edge": gen_train_forward_edge, "e2e_calibrated_vitl": gen_e2e_calibrated_vitl,
              "train_grads": gen_train_grads, "train_grads_dup": gen_train_grads_dup,
              "train_grads_f64": gen_train_grads_f64}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(4)
    for name, fn in GENERATORS.items():
        if a.only is None or a.only == name:
            fn()
