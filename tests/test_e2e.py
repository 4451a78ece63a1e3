"""End to end: Net.forward (eval) — oracle (CPU) and HIP model (GPU) against the REFERENCE Net's outputs
(tests/golden/e2e.npz).  Inputs/weights are regenerated from the stored seeds."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from netcfg import HEADS, TAKE, make_end_points, small_cfg  # noqa: E402

from oracle import nets as on  # noqa: E402
from oracle.weights import seeded_state_dict  # noqa: E402

gpu = pytest.mark.gpu
CASES = ["b1n4", "b2n3"]


def _load(golden_dir, tag):
    z = np.load(os.path.join(golden_dir, "e2e.npz"))
    B, N, hyp, seed, wseed = (int(v) for v in z[f"{tag}/meta"])
    ref = [{k.split("/", 2)[2]: z[k] for k in z.files if k.startswith(f"{tag}/h{h}/")} for h in range(hyp)]
    return z, B, N, hyp, seed, wseed, ref


def _keypoint_mismatch_is_explained(got_tar, got_src, ref_tar, ref_src, flow, cert, rel=2e-5):
    """Keypoint lists are discontinuous in (flow, logit): entries may differ only where the oracle's logit is
    within the float tolerance of 0 or a target coordinate is within it of an integer / of the bounds.  The
    tolerance is rel * max(1, max|tensor|) with rel = 2e-5 — about ten times the deviation measured between the engine's two
    arithmetic modes on the bench inputs (flow <= 7e-5 px at max|flow| 60, logits <= 5e-5: bench.py exact_mode) — instead of
    round 2's 5e-4, which declared 12 % of all slots "fragile"."""
    bad = (got_tar != ref_tar).any(-1) | (got_src != ref_src).any(-1)       # (B, H*W), k = w*H + h
    if not bad.any():
        return 0
    B, _, H, W = flow.shape
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    tx, ty = flow[:, 0] + xs, flow[:, 1] + ys
    frac = np.minimum(np.abs(tx - np.round(tx)), np.abs(ty - np.round(ty)))
    tol_f, tol_c = rel * max(1.0, float(np.abs(flow).max())), rel * max(1.0, float(np.abs(cert).max()))
    fragile = (np.abs(cert[:, 0]) < tol_c) | (frac < tol_f)                 # (B,H,W)
    fragile_k = fragile.transpose(0, 2, 1).reshape(B, H * W)
    assert not (bad & ~fragile_k).any(), int((bad & ~fragile_k).sum())
    return int(bad.sum())


@pytest.mark.parametrize("tag", CASES)
def test_oracle_forward_vs_reference(golden_dir, tag):
    from picopose_amd.picopose import Net

    torch.set_num_threads(8)
    z, B, N, hyp, seed, wseed, ref = _load(golden_dir, tag)
    sd = seeded_state_dict(Net(small_cfg()).state_dict(), wseed)
    fe = lambda x: on.vit_features(sd, x, HEADS, TAKE)  # noqa: E731
    ep = make_end_points(B, N, seed, feature_fn=fe, tem_pose=torch.from_numpy(z[f"{tag}/tem_pose_all"]))
    assert np.abs(ep["template_feature"][:, :, ::16, 3, 5].numpy() - z[f"{tag}/template_feature_probe"]).max() < 1e-4
    outs, aux = on.net_forward_test(sd, ep, hyp, HEADS, TAKE)
    for h in range(hyp):
        assert np.array_equal(outs[h]["tem_pose"].numpy(), ref[h]["tem_pose"])        # same templates picked
        assert np.abs(outs[h]["pred_poses"].numpy() - ref[h]["pred_poses"]).max() <= 1e-4
        _keypoint_mismatch_is_explained(outs[h]["pred_tar_pts"].numpy(), outs[h]["pred_src_pts"].numpy(),
                                        ref[h]["pred_tar_pts"], ref[h]["pred_src_pts"], aux["flow"][h].numpy(),
                                        aux["cert"][h].numpy(), rel=2e-5)


@gpu
@pytest.mark.parametrize("tag", CASES)
def test_hip_forward_vs_reference(golden_dir, tag):
    from picopose_amd.picopose import Net

    torch.set_num_threads(8)
    z, B, N, hyp, seed, wseed, ref = _load(golden_dir, tag)
    net = Net(small_cfg())
    sd = seeded_state_dict(net.state_dict(), wseed)
    net.load_state_dict(sd)
    net = net.cuda().eval()
    ep = make_end_points(B, N, seed, tem_pose=torch.from_numpy(z[f"{tag}/tem_pose_all"]))
    dev = {k: v.cuda() for k, v in ep.items()}
    # the bank is precomputed with the model's own feature extractor, exactly as run_test.py:120-134 does
    dev["template_feature"] = torch.stack([net.feature_extractor(dev["tem_rgb"][b])[-1] for b in range(B)])
    assert np.abs(dev["template_feature"][:, :, ::16, 3, 5].cpu().numpy() - z[f"{tag}/template_feature_probe"]).max() < 2e-3
    outs = net(dev, hyp)
    assert isinstance(outs, list) and len(outs) == hyp
    total_bad = 0
    for h in range(hyp):
        o = {k: v.cpu().numpy() for k, v in outs[h].items()}
        assert set(o) == set(ref[h]) and all(o[k].shape == ref[h][k].shape and o[k].dtype == ref[h][k].dtype for k in o)
        assert np.array_equal(o["tem_pose"], ref[h]["tem_pose"])                        # template ids: exact
        assert np.array_equal(o["tar_pts_2d"], ref[h]["tar_pts_2d"]) and np.array_equal(o["src_pts_3d"], ref[h]["src_pts_3d"])
        assert np.abs(o["pred_poses"] - ref[h]["pred_poses"]).max() <= 1e-4             # north_star tolerance
        # Key-point lists on THESE fixtures (plain random weights: flows of +-1500 px, logits of +-100 — nearly every slot is the -1
        # padding, and fp32 reassociation moves the flows by 1e-4 of their range) are compared slot by slot WITHOUT a margin argument:
        # >= 99.8 % of the 4096 slots bit-equal.  The bar that means something — every mismatch explained by a flow within 2e-5 of a
        # pixel boundary or a logit within 2e-5 of zero — is carried by the calibrated fixtures (test_hip_forward_vs_reference_calibrated).
        same = (o["pred_tar_pts"] == ref[h]["pred_tar_pts"]).all(-1) & (o["pred_src_pts"] == ref[h]["pred_src_pts"]).all(-1)
        total_bad += int((~same).sum())
    assert total_bad <= 0.002 * hyp * B * 4096, total_bad


@gpu
@pytest.mark.parametrize("batched", [True, False])
def test_extended_template_cache_gives_identical_outputs(batched):
    """a forward that reads the precomputed template-side DPT maps (and the bank's last-level
    features) returns bit-for-bit what recomputing the selected templates' ViT + DPT head returns."""
    from picopose_amd.picopose import Net

    B, N, hyp = 2, 5, 3
    net = Net(small_cfg())
    net.load_state_dict(seeded_state_dict(net.state_dict(), 21))
    net = net.cuda().eval()
    net.batch_hypotheses = batched
    dev = {k: v.cuda() for k, v in make_end_points(B, N, 33).items()}
    banks = [net.precompute_templates(dev["tem_rgb"][b], chunk=3) for b in range(B)]   # one "object" per crop
    dev["template_feature"] = torch.stack([bk["feature"] for bk in banks])
    ref = net(dev, hyp)
    dev["template_cache"] = {"obj_index": torch.arange(B, device="cuda"),
                             "dpt": [torch.stack([bk["dpt"][k] for bk in banks]) for k in range(3)]}
    got = net(dev, hyp)
    for h in range(hyp):
        assert set(got[h]) == set(ref[h])
        for k in ref[h]:
            assert torch.equal(got[h][k], ref[h][k]), (h, k)
    # the bank feature of precompute_templates is the reference's bank (run_test.py:130-131)
    assert torch.equal(banks[0]["feature"], net.feature_extractor(dev["tem_rgb"][0])[-1])


@gpu
def test_query_prefetch_inside_the_template_pass_keeps_every_bit():
    """Net.forward(ep, hyp, next_real_rgb=...) runs the NEXT batch's query ViT as extra rows of this batch's template-side ViT pass and
    the next call picks the stashed levels up: every output of both calls equals the plain forward's, bit for bit — also when the next
    batch has another size, when the stash does not belong to the batch that follows (ignored), and after a change of the arithmetic mode."""
    from picopose_amd import ops
    from picopose_amd.picopose import Net

    net = Net(small_cfg())
    net.load_state_dict(seeded_state_dict(net.state_dict(), 9))
    net = net.cuda().eval()
    hyp, N = 3, 5
    eps = []
    for B, seed in ((2, 61), (3, 62), (2, 63)):
        d = {k: v.cuda() for k, v in make_end_points(B, N, seed).items()}
        d["template_feature"] = torch.stack([net.feature_extractor(d["tem_rgb"][b])[-1] for b in range(B)])
        eps.append(d)
    plain = [net(d, hyp) for d in eps]
    assert getattr(net, "_query_stash", None) is None

    def same(a, b):
        for h in range(hyp):
            assert set(a[h]) == set(b[h])
            for k in a[h]:
                assert torch.equal(a[h][k], b[h][k]), (h, k)

    got0 = net(eps[0], hyp, next_real_rgb=eps[1]["real_rgb"])
    assert net._query_stash is not None and net._query_stash[1][0].shape[0] == 3 * 257
    got1 = net(eps[1], hyp, next_real_rgb=eps[2]["real_rgb"])          # uses the stash, leaves one for batch 2
    got2 = net(eps[2], hyp)                                            # uses the stash, leaves none
    assert net._query_stash is None
    same(got0, plain[0]); same(got1, plain[1]); same(got2, plain[2])   # noqa: E702
    # a stash that does not belong to the batch that follows is ignored
    net(eps[0], hyp, next_real_rgb=eps[1]["real_rgb"])
    same(net(eps[2], hyp), plain[2])
    # a DISCARDED look-ahead cannot be mistaken for a later batch of the same shape: the stash holds the look-ahead tensor itself, so its
    # storage stays allocated and a fresh batch cannot receive its address (stale hit -> the OTHER batch's query features)
    look = eps[1]["real_rgb"].clone()
    net(eps[0], hyp, next_real_rgb=look)
    addr = look.data_ptr()
    assert net._query_stash[3] is look
    del look                                                      # the serving loop drops the request ...
    fresh = dict(eps[1], real_rgb=torch.randn_like(eps[1]["real_rgb"]))    # ... and a new batch of the same shape arrives
    assert fresh["real_rgb"].data_ptr() != addr
    want = net(fresh, hyp)
    net(eps[0], hyp, next_real_rgb=eps[1]["real_rgb"].clone())    # (again a discarded look-ahead in front of `fresh`)
    same(net(fresh, hyp), want)
    # the stash owns its rows: it does not keep the previous batch's level buffers alive
    net(eps[0], hyp, next_real_rgb=eps[1]["real_rgb"])
    assert net._query_stash[1][0]._base is None
    # ... and so is one computed in another arithmetic mode
    net(eps[0], hyp, next_real_rgb=eps[1]["real_rgb"])
    old = ops.PRECISION
    ops.PRECISION = "f32"
    try:
        exact_plain = net(eps[1], hyp)
        assert net._query_stash is None
        net(eps[0], hyp, next_real_rgb=eps[1]["real_rgb"])
        same(net(eps[1], hyp), exact_plain)
    finally:
        ops.PRECISION = old


@gpu
def test_infer_image_walks_instances_like_run_test(monkeypatch):
    """pipeline.infer_image (run_test.py:141-188): instance mini-batches, templates gathered per object, hypotheses
    sorted by inlier ratio, t in millimetres — with and without the extended template bank."""
    from picopose_amd import ops
    from picopose_amd.picopose import Net
    from picopose_amd.pipeline import infer_batch, infer_image

    # (PLAIN seeded weights: the decoder's hidden maps reach 5e4 and leave the f16x3 operand range — DESIGN section 4, known since round 2;
    # the sticky saturation word would, rightly, refuse these poses.  This test is about the walk, not the numbers: reporting off.)
    monkeypatch.setattr(ops, "SATURATION_FLAG", False)

    n_obj, N, n_inst, hyp = 2, 4, 3, 2
    net = Net(small_cfg())
    net.load_state_dict(seeded_state_dict(net.state_dict(), 5))
    net = net.cuda().eval()
    tem = {k: v.cuda() for k, v in make_end_points(n_obj, N, 71).items() if k.startswith("tem_")}   # per-object banks
    banks = [net.precompute_templates(tem["tem_rgb"][o]) for o in range(n_obj)]
    tem["template_feature"] = torch.stack([b["feature"] for b in banks])
    inst = {k: v.cuda() for k, v in make_end_points(n_inst, 1, 72).items() if k.startswith("real_")}
    data = {k: v[None] for k, v in inst.items()}
    data["obj_idx"] = torch.tensor([[1, 0, 1]], device="cuda")
    data["score"] = torch.tensor([[0.9, 0.8, 0.7]], device="cuda")
    preds = infer_image(net, data, tem, hyp=hyp, bs=2)
    assert len(preds) == n_inst and all(len(p) == hyp for p in preds)
    for p in preds:
        assert p[0]["inliers_ratio"] >= p[1]["inliers_ratio"] and p[0]["R_stage_3"].shape == (9,) and p[0]["t_stage_3"].shape == (3,)
    # same numbers as one batch over all instances
    inputs = dict(inst)
    inputs.update({k: v[data["obj_idx"][0]] for k, v in tem.items()})
    for got, ref in zip(preds, infer_batch(net, inputs, hyp)):
        for g, r in zip(got, ref):
            assert np.allclose(g["R_stage_3"], np.asarray(r["R"]).reshape(9), atol=1e-6)
            assert np.allclose(g["t_stage_3"], np.asarray(r["t"]).reshape(3) * 1000, atol=1e-3)
    # the reference's strictly sequential walk (a host wait per mini-batch) returns the pipelined walk's results, bit for bit
    for got, ref in zip(infer_image(net, data, tem, hyp=hyp, bs=2, pipelined=False), preds):
        for g, r in zip(got, ref):
            assert np.array_equal(g["R_stage_3"], r["R_stage_3"]) and np.array_equal(g["t_stage_3"], r["t_stage_3"]) and g["inliers_ratio"] == r["inliers_ratio"]
    # the loader's NEXT image handed over (next_data): its first mini-batch's query ViT rides in this image's last forward — same bits for both
    inst2 = {k: v.cuda() for k, v in make_end_points(2, 1, 73).items() if k.startswith("real_")}
    data2 = {k: v[None] for k, v in inst2.items()}
    data2["obj_idx"] = torch.tensor([[0, 1]], device="cuda")
    data2["score"] = torch.tensor([[0.6, 0.5]], device="cuda")
    plain2 = infer_image(net, data2, tem, hyp=hyp, bs=2)
    ahead1 = infer_image(net, data, tem, hyp=hyp, bs=2, next_data=data2)
    assert net._query_stash is not None
    ahead2 = infer_image(net, data2, tem, hyp=hyp, bs=2)
    assert net._query_stash is None
    for a_, b_ in ((ahead1, preds), (ahead2, plain2)):
        for got, ref in zip(a_, b_):
            for g, r in zip(got, ref):
                assert np.array_equal(g["R_stage_3"], r["R_stage_3"]) and np.array_equal(g["t_stage_3"], r["t_stage_3"])
    # extended bank: identical poses
    tem["template_cache"] = {"dpt": [torch.stack([b["dpt"][k] for b in banks]) for k in range(3)]}
    for got, ref in zip(infer_image(net, data, tem, hyp=hyp, bs=2), preds):
        for g, r in zip(got, ref):
            assert np.array_equal(g["R_stage_3"], r["R_stage_3"]) and np.array_equal(g["t_stage_3"], r["t_stage_3"])


# ---- calibrated heads + dome geometry: the network -> key-points -> PnP chain on realistic occupancy ---------------------
CAL_CASES = ["vits_b2n4", "vitb_b2n6", "vitl_b2n4"]   # the last: ViT-L/14, the backbone of configs[4] / config/base.yaml
VIT_CFG = {"dinov2_vits14": (384, 6, [[0, 2], [3, 5], [6, 8], [9, 11]]), "dinov2_vitb14": (768, 12, [[0, 2], [3, 5], [6, 8], [9, 11]]),
           "dinov2_vitl14": (1024, 16, [[0, 5], [6, 11], [12, 17], [18, 23]])}


def _vit_cfg(vit):
    import types

    ns = types.SimpleNamespace
    C, _, idx = VIT_CFG[vit]
    return ns(hypothesis=5, stage1=ns(vit_type=vit, pretrained=False, interaction_indexes=idx), stage2=ns(in_channel=256, hidden_dim=256),
              stage3=ns(nclass=1, in_channels=C, use_bn=True, out_channels=[256, 512, 1024, 1024], num_levels=3, radius=4))


def _load_cal(golden_dir, tag):
    from oracle.weights import apply_head_calibration

    z = np.load(os.path.join(golden_dir, "e2e_calibrated.npz"))
    B, N, hyp, seed, wseed = (int(v) for v in z[f"{tag}/meta"])
    vit = str(z[f"{tag}/vit"])
    cal = {"flow": [tuple(r) for r in z[f"{tag}/cal_flow"]], "cert": [tuple(r) for r in z[f"{tag}/cal_cert"]],
           "proj_bn": float(z[f"{tag}/cal_proj_bn"]),
           "affine": {h: (float(z[f"{tag}/cal_affine_{h}"][0]), tuple(z[f"{tag}/cal_affine_{h}"][1:]))
                      for h in ("translation", "scale", "inplane")}}
    ref = [{k.split("/", 2)[2]: z[k] for k in z.files if k.startswith(f"{tag}/h{h}/")} for h in range(hyp)]
    weights = lambda template: apply_head_calibration(seeded_state_dict(template, wseed), cal)  # noqa: E731
    return z, B, N, hyp, seed, vit, ref, weights


def _valid(pts):
    return (pts[..., 0] >= 0).sum(-1)


@pytest.mark.parametrize("tag", CAL_CASES)
def test_oracle_forward_vs_reference_calibrated(golden_dir, tag):
    from picopose_amd.picopose import Net

    torch.set_num_threads(8)
    z, B, N, hyp, seed, vit, ref, weights = _load_cal(golden_dir, tag)
    _, heads, idx = VIT_CFG[vit]
    take = [b[-1] for b in idx]
    sd = weights(Net(_vit_cfg(vit)).state_dict())
    fe = lambda x: on.vit_features(sd, x, heads, take)  # noqa: E731
    ep = make_end_points(B, N, seed, feature_fn=fe, tem_pose=torch.from_numpy(z[f"{tag}/tem_pose_all"]), dome=True)
    assert np.abs(ep["template_feature"][:, :, ::16, 3, 5].numpy() - z[f"{tag}/template_feature_probe"]).max() < 1e-4
    outs, aux = on.net_forward_test(sd, ep, hyp, heads, take)
    for h in range(hyp):
        assert _valid(ref[h]["pred_tar_pts"]).min() >= 1000                           # the fixture carries data
        assert np.array_equal(outs[h]["tem_pose"].numpy(), ref[h]["tem_pose"])        # same templates picked
        assert np.abs(outs[h]["pred_poses"].numpy() - ref[h]["pred_poses"]).max() <= 1e-4
        bad = _keypoint_mismatch_is_explained(outs[h]["pred_tar_pts"].numpy(), outs[h]["pred_src_pts"].numpy(),
                                              ref[h]["pred_tar_pts"], ref[h]["pred_src_pts"], aux["flow"][h].numpy(),
                                              aux["cert"][h].numpy(), rel=2e-5)
        assert bad <= 4 * B


def _hip_calibrated_forward(golden_dir, tag, match_mode=None):
    from picopose_amd.picopose import Net

    z, B, N, hyp, seed, vit, ref, weights = _load_cal(golden_dir, tag)
    _, heads, idx = VIT_CFG[vit]
    net = Net(_vit_cfg(vit))
    sd = weights(net.state_dict())
    net.load_state_dict(sd)
    net = net.cuda().eval()
    net.match_mode = match_mode
    ep = make_end_points(B, N, seed, tem_pose=torch.from_numpy(z[f"{tag}/tem_pose_all"]), dome=True)
    dev = {k: v.cuda() for k, v in ep.items()}
    dev["template_feature"] = torch.stack([net.feature_extractor(dev["tem_rgb"][b])[-1] for b in range(B)])
    net.keep_stage3 = True
    from picopose_amd import ops

    ops.CHECK_SATURATION = True       # every f16x3 operand buffer is verified against the fp16 clamp (raises on a hit)
    try:
        outs = net(dev, hyp)
    finally:
        ops.CHECK_SATURATION = False
    fl, ce = net.last_stage3                              # NHWC, hypothesis-major (hyp*B, 64, 64, c)
    flow = fl.permute(0, 3, 1, 2).reshape(hyp, B, 2, 64, 64).cpu().numpy()
    cert = ce.permute(0, 3, 1, 2).reshape(hyp, B, 1, 64, 64).cpu().numpy()
    return z, B, N, hyp, ref, ep, dev, outs, flow, cert


def _check_hip_vs_reference(z, tag, B, hyp, ref, outs, flow, cert):
    total_bad = 0
    for h in range(hyp):
        o = {k: v.cpu().numpy() for k, v in outs[h].items()}
        assert np.array_equal(o["tem_pose"], ref[h]["tem_pose"])                        # template ids: exact
        assert np.abs(o["pred_poses"] - ref[h]["pred_poses"]).max() <= 1e-4             # north_star tolerance
        assert o["pred_tar_pts"].dtype == np.int64 and o["pred_tar_pts"].shape == ref[h]["pred_tar_pts"].shape
        # every slot that differs from the REFERENCE's list lies on a threshold of OUR flow / certainty maps within the
        # float tolerance stated for the offset tensors (5e-4 * max|tensor|): logit ~ 0, coordinate ~ integer
        total_bad += _keypoint_mismatch_is_explained(o["pred_tar_pts"], o["pred_src_pts"], ref[h]["pred_tar_pts"],
                                                     ref[h]["pred_src_pts"], flow[h], cert[h])
        assert _valid(o["pred_tar_pts"]).min() >= 1000
    assert total_bad <= 0.002 * hyp * B * 4096, total_bad
    return total_bad


# a20, branch by branch (see the test below): candidate poses of the kernel's refit against the LAPACK oracle's same-index candidates
PNP_BRANCH_TOL = 1e-6      # |dR|, |dt| (metres) per candidate
PNP_TIE_REL = 1e-9         # the kept index must be equal unless the oracle's two best mean reprojection errors are this close


@gpu
@pytest.mark.parametrize("tag", CAL_CASES)
def test_hip_forward_vs_reference_calibrated(golden_dir, tag):
    """~2500-3200 valid key-points per hypothesis (fixture from the reference Net): template ids exact, stage-2 poses
    1e-4, key-point lists bit-equal except threshold cases; then PnP/RANSAC on them."""
    from picopose_amd.pipeline import pnp_for_outputs
    from oracle import pnp as opnp

    z, B, N, hyp, ref, ep, dev, outs, flow, cert = _hip_calibrated_forward(golden_dir, tag)
    _check_hip_vs_reference(z, tag, B, hyp, ref, outs, flow, cert)
    rot, tvec, ratio, ok, npts = pnp_for_outputs(outs, dev["real_K"], return_npts=True)
    assert npts.min() >= 1000 and ok.all()
    # (a) like for like: the HIP PnP kernel fed with the REFERENCE's key-point lists against the CPU oracle of PnP/RANSAC on the
    # same lists (same sampling sequence, problem id h*B+b) in its solver-INDEPENDENT form: solver="lapack" (eigh / svd / lstsq —
    # nothing of the kernel's Jacobi sequences).  On the round-3 synthetic object (8 cm of relief on the optical axis, the
    # dataset's layout of real_pts2d, tests/netcfg.py) nearly every correspondence is an inlier of the true motion, so the
    # RANSAC winner does not depend on how a 5-point sample's 2-dimensional null space is resolved: the consensus sets are
    # IDENTICAL.  The refit on that set is EPnP's choice among three candidate poses (beta initialisations with N = 1 / 2 / 3
    # null-space vectors), by least mean reprojection error.  Round 3 compared only the chosen poses (0.6-3.2 mm apart where the
    # two solvers kept different candidates of nearly equal error) under a 5 mm bar that a real arithmetic difference in one
    # branch would have passed.  Now: pp_pnp_ransac_debug returns all three candidates and the comparison is
    # BRANCH BY BRANCH — every candidate pose within PNP_BRANCH_TOL of the oracle's same-index candidate, the kept index equal
    # whenever the oracle's two best errors are more than PNP_TIE_REL apart, and the returned pose = the kept candidate;
    # (b) the chain: PnP on the HIP net's own lists (a few threshold slots may differ): the same consensus within 2 %.
    from picopose_amd.pipeline import pnp_inputs
    from picopose_amd.utils.pose_recovery import refit_branches

    ref_outs = [dict(outs[h], pred_tar_pts=torch.from_numpy(ref[h]["pred_tar_pts"]).cuda(),
                     pred_src_pts=torch.from_numpy(ref[h]["pred_src_pts"]).cuda()) for h in range(hyp)]
    rrot, rtvec, rratio, rok, rnpts = pnp_for_outputs(ref_outs, dev["real_K"], return_npts=True)
    brot, btvec, bratio, bok, branches = refit_branches(*pnp_inputs(ref_outs, dev["real_K"]))
    assert np.array_equal(brot.reshape(rrot.shape), rrot) and np.array_equal(btvec.reshape(rtvec.shape), rtvec)   # same kernel, same bits
    worst = {"dt": 0.0, "dR": 0.0, "derr": 0.0, "chosen_dt": 0.0}
    kept_differs, ties = 0, 0
    for h in range(hyp):
        for b in range(B):
            t2 = ep["real_pts2d"][b].permute(2, 1, 0).numpy()
            sel = int(np.argmax([np.array_equal(ep["tem_pose"][b, n].numpy(), ref[h]["tem_pose"][b]) for n in range(N)]))
            s3 = ep["tem_pts3d"][b, sel].permute(2, 0, 1).numpy()
            orot, otvec, oratio, ook, obr = opnp.pose_recovery_ransac_pnp(t2, s3, ep["real_K"][b].numpy(), ref[h]["tem_pose"][b],
                                                                          ref[h]["pred_tar_pts"][b], ref[h]["pred_src_pts"][b],
                                                                          prob=h * B + b, solver="lapack", return_branches=True)
            n = int(rnpts[h, b])
            assert ook and rok[h, b] and n == int(_valid(ref[h]["pred_tar_pts"][b:b + 1])[0])
            assert rratio[h, b] > 0.8, (h, b, rratio[h, b])
            assert round(oratio * n) == round(rratio[h, b] * n), (h, b, oratio * n, rratio[h, b] * n)      # identical consensus
            cand, kept = branches[h * B + b]
            assert len(obr) == 3 and kept in (0, 1, 2)
            for a in range(3):
                (ke, kR, kt), (oe, oR, ot) = cand[a], obr[a]
                assert np.isfinite(ke) == np.isfinite(oe), (h, b, a, ke, oe)
                if not np.isfinite(ke):
                    continue
                worst["dt"] = max(worst["dt"], float(np.abs(kt - ot).max()))
                worst["dR"] = max(worst["dR"], float(np.abs(kR - oR).max()))
                worst["derr"] = max(worst["derr"], abs(ke - oe))
                assert np.abs(kt - ot).max() <= PNP_BRANCH_TOL and np.abs(kR - oR).max() <= PNP_BRANCH_TOL, (h, b, a, kt, ot)
                assert abs(ke - oe) <= PNP_BRANCH_TOL * 1e3, (h, b, a, ke, oe)        # px: |d err| <= f * |d pose| / z
            oerr = sorted(e for e, _, _ in obr)
            okept = int(np.argmin([e for e, _, _ in obr]))
            tie = oerr[1] - oerr[0] <= PNP_TIE_REL * oerr[0]
            ties += tie
            kept_differs += kept != okept
            assert kept == okept or tie, (h, b, kept, okept, oerr)
            # the returned pose is the kept candidate, and the oracle's candidate of THAT index is the same pose
            assert np.array_equal(rrot[h, b], cand[kept][1]) and np.array_equal(rtvec[h, b, :, 0], cand[kept][2])
            worst["chosen_dt"] = max(worst["chosen_dt"], float(np.abs(otvec - rtvec[h, b]).max()))
            assert abs(ratio[h, b] - rratio[h, b]) < 0.02
            # and it is the pose stage 2 predicted, refined: same ballpark
            assert np.abs(tvec[h, b, :, 0] - ref[h]["pred_poses"][b, :3, 3]).max() < 0.05
    print(f"PnP {tag}: HIP kernel vs solver-independent oracle on the reference's key-points, {hyp * B} problems: consensus identical; "
          f"per beta branch max |dt| {worst['dt']:.2e} m, max |dR| {worst['dR']:.2e}, max |d err| {worst['derr']:.2e} px; kept branch "
          f"differs in {kept_differs} problems ({ties} with the oracle's two best errors within {PNP_TIE_REL:g} relative); chosen poses "
          f"max |dt| {worst['chosen_dt']:.2e} m; inlier ratios {rratio.min():.3f}..{rratio.max():.3f}")


@gpu
@pytest.mark.parametrize("tag", CAL_CASES)
def test_hip_forward_vs_reference_calibrated_exact_mode(golden_dir, tag):
    """The same comparison in `--mode exact`: fp32 MFMA in every kernel (ops.PRECISION = "f32") and the exact-fp32 stage 1."""
    from picopose_amd import ops

    old = ops.PRECISION
    ops.PRECISION = "f32"
    try:
        z, B, N, hyp, ref, ep, dev, outs, flow, cert = _hip_calibrated_forward(golden_dir, tag, match_mode="exact")
        _check_hip_vs_reference(z, tag, B, hyp, ref, outs, flow, cert)
    finally:
        ops.PRECISION = old


@gpu
@pytest.mark.parametrize("tag", ["vits_b2n4", "vitb_b2n6", "vitl_b2n4"])
def test_hip_forward_vs_reference_calibrated_f16_mode(golden_dir, tag):
    """`ops.PRECISION = "f16"` (bench.py --mode fp16): plain fp16 operands, ONE MFMA per product, fp32 accumulation — the
    arithmetic BASELINE configs[4] names (ViT-L/14 "fp16").  Operands carry 11 bits instead of the f16x3 engine's 22, so the
    comparison with the reference's fp32 outputs uses fp16-grade bars, stated here from the measured deviations
    (profiles/r03/f16_mode_deviation.txt): same templates picked (stage 1 keeps its exact re-evaluation of near-ties, the
    bank is computed in the same mode), stage-2 poses within 1e-3 (measured <= 3e-4), at least 99 % (measured >= 99.49 %) of the 4096 key-point slots per (crop,
    hypothesis) bit-equal to the reference's."""
    from picopose_amd import ops

    old = ops.PRECISION
    ops.PRECISION = "f16"
    try:
        z, B, N, hyp, ref, ep, dev, outs, flow, cert = _hip_calibrated_forward(golden_dir, tag)
    finally:
        ops.PRECISION = old
    from oracle import nets as on  # noqa: F401  (the reference's own maps are not in the fixture: slots are compared directly)

    agree, pose_err = [], 0.0
    for h in range(hyp):
        o = {k: v.cpu().numpy() for k, v in outs[h].items()}
        assert np.array_equal(o["tem_pose"], ref[h]["tem_pose"])                        # template ids: exact
        pose_err = max(pose_err, float(np.abs(o["pred_poses"] - ref[h]["pred_poses"]).max()))
        same = (o["pred_tar_pts"] == ref[h]["pred_tar_pts"]).all(-1) & (o["pred_src_pts"] == ref[h]["pred_src_pts"]).all(-1)
        agree.append(float(same.mean()))
        assert _valid(o["pred_tar_pts"]).min() >= 1000
    print(f"f16 mode {tag}: stage-2 pose max abs err {pose_err:.2e}, key-point slot agreement min {min(agree):.4f} mean {np.mean(agree):.4f}")
    assert pose_err <= 1e-3
    assert min(agree) >= 0.99


@gpu
@pytest.mark.parametrize("factor", [30.0, 300.0, 3000.0])
def test_outlier_channels_of_a_trained_vit_f16x3_against_exact_mode(golden_dir, factor):
    """Trained DINOv2 carries a few residual-stream channels 10^2 .. 10^3 x the rest (block.py:104-106 adds ls.gamma * branch,
    layer_scale.py:27-28); every fixture here has seeded random weights.  This drives the engines with that kind of distribution: in EVERY
    ViT block 4 random channels of ls1.gamma and ls2.gamma are multiplied by `factor` (the calibrated ViT-B case otherwise), and the f16x3
    forward is compared with the strict-fp32 forward of the same network:
      * the ViT itself (LayerNorm, qkv, attention, MLP on rows whose channels span 3 decades): the four token levels within 2e-4 of
        their maximum, the same templates picked, stage-2 poses within 1e-4 — at every factor;
      * the rest of the path: >= 99.9 % of the key-point slots bit-equal — OR the sticky saturation word is raised (picopose_amd/ops.py).
        The decoder of these fixtures is calibrated for features of unit scale: with 300 x outliers its correlation volume (products of two
        DPT maps that grew with the ViT features) leaves the fp16 range of the operand format in the 1x1 layer behind the lookup
        (raft_decoder.py:147-153) — what must then happen is an ERROR with the poses, never a silently clipped answer."""
    from picopose_amd import _lib, ops
    from picopose_amd.picopose import Net
    from picopose_amd.pipeline import pnp_for_outputs

    tag = "vitb_b2n6"
    z, B, N, hyp, seed, vit, ref, weights = _load_cal(golden_dir, tag)
    net = Net(_vit_cfg(vit))
    sd = weights(net.state_dict())
    g = torch.Generator().manual_seed(1234)
    touched = 0
    for k in sd:
        if k.endswith(("ls1.gamma", "ls2.gamma")):
            ch = torch.randperm(sd[k].numel(), generator=g)[:4]
            sd[k] = sd[k].clone()
            sd[k][ch] *= factor
            touched += 1
    assert touched == 24
    net.load_state_dict(sd)
    net = net.cuda().eval()
    ep = make_end_points(B, N, seed, tem_pose=torch.from_numpy(z[f"{tag}/tem_pose_all"]), dome=True)
    dev = {k: v.cuda() for k, v in ep.items()}
    assert ops.saturation_word() is not None
    ops.saturation_raised()
    res, toks, raised, old = {}, {}, {}, ops.PRECISION
    try:
        for mode, match in (("f32", "exact"), ("f16x3", None)):
            ops.PRECISION, net.match_mode = mode, match
            with torch.no_grad():
                toks[mode] = [t.cpu() for t in net.feature_extractor.forward_tokens(dev["real_rgb"])[0]]
            dev["template_feature"] = torch.stack([net.feature_extractor(dev["tem_rgb"][b])[-1] for b in range(B)])
            assert not ops.saturation_raised(), f"{mode}: the ViT itself left the operand range"
            outs = net(dev, hyp)
            res[mode] = [{k: v.cpu() for k, v in o.items()} for o in outs]
            try:
                pnp_for_outputs(outs, dev["real_K"])          # the poses' copy carries the saturation word
                raised[mode] = False
            except _lib.PicoPoseHipError as e:
                assert "saturated" in str(e), e
                raised[mode] = True
    finally:
        ops.PRECISION, net.match_mode = old, None
    assert not raised["f32"] and not ops.saturation_raised()    # (a raised word was reset by the error path)
    top = max(float(t.abs().max()) for t in toks["f32"])
    spread = float(toks["f32"][-1].abs().amax(dim=(0, 1)).max() / toks["f32"][-1].abs().amax(dim=(0, 1)).median())
    for lv, (a, b) in enumerate(zip(toks["f16x3"], toks["f32"])):
        assert float((a - b).abs().max()) <= 2e-4 * float(b.abs().max()), (lv, float((a - b).abs().max()), float(b.abs().max()))
    agree = []
    for h in range(hyp):
        a, b = res["f16x3"][h], res["f32"][h]
        assert torch.equal(a["tem_pose"], b["tem_pose"])
        assert float((a["pred_poses"] - b["pred_poses"]).abs().max()) <= 1e-4
        same = (a["pred_tar_pts"] == b["pred_tar_pts"]).all(-1) & (a["pred_src_pts"] == b["pred_src_pts"]).all(-1)
        agree.append(float(same.float().mean()))
    print(f"outlier factor {factor:g}: max |token| {top:.3g}, largest / median channel of the last level {spread:.0f} x; stage 3 in f16x3: "
          + ("saturation word RAISED with the poses" if raised["f16x3"] else f"in range, key-point slots equal min {min(agree):.5f}"))
    if not raised["f16x3"]:
        assert min(agree) >= 0.999
    else:
        assert factor >= 300.0


@gpu
def test_sticky_saturation_word_is_set_by_every_producer_and_raised_with_the_poses(monkeypatch):
    """The default guard against clamped operands (no debugging switch): each kind of producer kernel sets the registered device word
    when a value leaves the fp16 range of the operand format, `saturation_raised()` reports and resets it, and a healthy call leaves it
    clear."""
    from picopose_amd import _lib, ops

    monkeypatch.setattr(ops, "PRECISION", "f16x3")
    assert ops.saturation_word() is not None
    ops.saturation_raised()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(512, 256, generator=g).cuda()
    w = (torch.randn(256, 256, generator=g) / 16).cuda()
    big = x.clone()
    big[7, 9] = 3.0e4                                          # 4 x = 1.2e5 > 65504
    img, bigimg = x.view(2, 16, 16, 256), big.view(2, 16, 16, 256)
    wc = ops.pack_conv_weight((torch.randn(256, 256, 3, 3, generator=g) / 48).cuda())
    ln_w, ln_b = torch.ones(256).cuda(), torch.zeros(256).cuda()
    cases = {
        "split pass": lambda t: ops.split_activation(t, 1, 512, 256, 0, 256),
        "GEMM epilogue (operand output)": lambda t: ops.linear(ops.Split(ops.split_activation(x, 1, 512, 256, 0, 256)), w * (1.0 if t is x else 4000.0),
                                                                 out_split=True),
        "LayerNorm (operand output)": lambda t: ops.layernorm(x, ln_w * (1.0 if t is x else 1.0e4), ln_b, 1e-6, out_split=True),
        "bilinear resize (operand output)": lambda t: ops.resize_bilinear(t.view(2, 16, 16, 256), 32, 32, out_split=True),
        "Winograd convolution (operand output)": lambda t: ops.conv2d(ops.split_image(img), wc * (1.0 if t is x else 4000.0), None, 3, pad=1, out_split=True, wino=True),
        "convolution epilogue (operand output)": lambda t: ops.conv2d(ops.split_image(img), wc * (1.0 if t is x else 4000.0), None, 3, pad=1, out_split=True),
        # attention: 2 images x 4 heads x 64 (qkv = 768 columns of a (2 * 64, 768) matrix); a huge V entry reaches the output operand
        "attention (operand output, fp32 qkv)": lambda t: ops.attention(qkv_of(t), 2, 64, 4, 64, out_split=True),
        "attention (operand output, operand qkv)": lambda t: ops.attention(ops.Split(ops.split_activation(qkv_of(t) * (1.0 if t is x else 0.25), 1, 128, 768, 0, 768)),
                                                                             2, 64, 4, 64, out_split=True) if t is x else ops.attention(
            ops.Split(ops.split_activation(qkv_ok_big_v, 1, 128, 768, 0, 768)), 2, 64, 4, 64, out_split=True),
        "warp (operand columns)": lambda t: ops.warp(t.view(2, 16, 16, 256), flow0, hl_into=(warp_tgt, 0)),
    }
    qkv_base = torch.randn(128, 768, generator=g).cuda()

    def qkv_of(t):
        q_ = qkv_base.clone()
        if t is not x:
            q_[:, 512:] *= 1.0e4                    # the values V: the soft-max mixes them into every output row
        return q_

    qkv_ok_big_v = qkv_base.clone()
    qkv_ok_big_v[:, 512:] *= 1.2e3                  # V in range as an operand (|4 v| < 65504) ... and so are its convex combinations: NOT reported
    flow0 = torch.zeros(2, 16, 16, 2, device="cuda")
    warp_tgt = ops.Split.empty(2 * 16 * 16, 256, "cuda")
    with torch.no_grad():
        for name, fn in cases.items():
            fn(x)
            assert not ops.saturation_raised(), f"{name}: a healthy input set the word"
            fn(big)
            if name == "attention (operand output, operand qkv)":
                assert not ops.saturation_raised(), "values inside the operand range set the word"
                continue
            assert ops.saturation_raised(), f"{name}: a value beyond the fp16 range did not set the word"
            assert not ops.saturation_raised(), "the word was not reset"
        # a map in range as an operand (|4 x| < 65504) whose Winograd transform is not (|B^T d B| / 16 up to 6.25 |x|): reported by the transform
        hot = torch.zeros(2, 16, 16, 256)
        sgn = torch.tensor([1.0, 0, -1, 0, 1, 0])               # the signs of B^T's first row (4, 0, -5, 0, 1, 0): amplification 10 per axis
        hot[0, 3:9, 3:9, 3] = 1.5e4 * sgn[:, None] * sgn[None, :]    # tile (1, 1) reads rows / columns 3 .. 8
        xs = ops.split_image(hot.cuda())
        assert not ops.saturation_raised()
        ops.winograd_shared(xs, cout=256)
        assert ops.saturation_raised()
    # the pipeline reads the word with the poses and raises
    w_ = ops.saturation_word()
    w_.fill_(1)
    from picopose_amd.utils.pose_recovery import _check_sat_row, _with_sat_row
    packed = _with_sat_row(torch.zeros(3, 15, dtype=torch.float64, device="cuda"))
    assert packed.shape == (4, 15)
    with pytest.raises(_lib.PicoPoseHipError, match="saturated"):
        _check_sat_row(packed.cpu().numpy(), 3)
    assert not ops.saturation_raised()


@gpu
@pytest.mark.parametrize("cfg", ["4", "5", "6", "7", "8"])
def test_hip_forward_vitb_with_pinned_persistent_kernels(golden_dir, monkeypatch, cfg):
    """The ViT-B net-vs-reference comparison with every pre-split GEMM / conv forced onto the persistent 256x128 (cfg 4),
    256x256 (cfg 5), row-shared 3x3 (cfg 6) and two- / three-workgroups-per-CU (cfg 7 / 8, dense layers) kernels — the kernels the
    headline bench runs — instead of the autotuner's pick."""
    monkeypatch.setenv("PP_GEMM_FORCE_CFG", cfg)
    z, B, N, hyp, ref, ep, dev, outs, flow, cert = _hip_calibrated_forward(golden_dir, "vitb_b2n6")
    _check_hip_vs_reference(z, "vitb_b2n6", B, hyp, ref, outs, flow, cert)


@gpu
@pytest.mark.parametrize("vit,B,N,half_bank", [("dinov2_vitb14", 32, 162, False), ("dinov2_vitl14", 64, 64, True)],
                         ids=["configs2_b32_n162_vitb", "configs4_share_b64_n64_vitl_f16bank"])
def test_full_size_forward_is_batch_independent(vit, B, N, half_bank):
    """BASELINE configs[2] at full size (32 crops x 162 templates, ViT-B/14, hyp 5, the bench's inputs and calibrated
    weights) and ONE RANK'S SHARE of configs[4] (64 crops x 64 of the 512 templates, ViT-L/14, feature bank stored fp16 —
    bench.py's `full_b64_n64_vitl`) through a size-independent property: every crop's outputs — template ids, stage-2 poses,
    key-point lists — are bit for bit what the same crop gives in a batch of 3.  Each layer then runs with other GEMM shapes,
    other tile configurations (the autotuner's pick per shape, the persistent / row-shared kernels at M = 655 360 against the
    small kernels at M = 61 440) and other tile tails, so any dependence of a row on its neighbours or on the kernel shows."""
    import bench
    from picopose_amd.picopose import Net

    net = Net(bench.make_cfg(vit))
    bench.seeded_weights(net, 4, vit)
    net = net.cuda().eval()
    ep = bench.make_end_points(B, N, "cuda", 100)
    fe = net.feature_extractor
    with torch.no_grad():
        ep["template_feature"] = torch.stack([torch.cat([fe(ep["tem_rgb"][b, s:s + 54])[-1] for s in range(0, N, 54)]) for b in range(B)])
    if half_bank:
        ep["template_feature"] = ep["template_feature"].half()
    full = net(ep, 5)
    assert min(int((o["pred_tar_pts"][..., 0] >= 0).sum(1).min()) for o in full) >= 1000
    pick = [3, 17, B - 1]
    sub = {k: v[pick].contiguous() for k, v in ep.items()}
    small = net(sub, 5)
    for h in range(5):
        for key in full[h]:
            assert torch.equal(small[h][key], full[h][key][pick]), (h, key)


@gpu
def test_configs4_share_in_fp16_engine_mode_is_batch_independent_and_agrees_with_f16x3():
    """BASELINE configs[4] names "ViT-L/14 fp16": ONE RANK'S SHARE of it at full size (64 crops x 64 of the 512 templates, feature
    bank stored fp16, hyp 5 — bench.py's `full_b64_n64_vitl`) in the engine's fp16 mode (`ops.PRECISION = "f16"`, bench.py
    --mode fp16: plain fp16 operands, one MFMA per product, fp32 accumulation).  Round 3 ran this size only in f16x3 mode and the
    fp16 engine mode only on the small ViT-L fixture (VERDICT r03 missing #3).  Two size-independent properties:
      * batch independence IN fp16 MODE: three crops give bit for bit what they give inside the batch of 64 (other GEMM shapes,
        other tiles, other tails — the TERMS = 1 kernels accumulate in one order whatever the configuration);
      * agreement with the f16x3 engine on the same inputs within the fp16-grade bars stated for the reference comparison
        (test_hip_forward_vs_reference_calibrated_f16_mode; measured values in profiles/r04/f16_mode_deviation.txt): the same
        templates wherever the two modes' stage-1 scores are not within 1e-3 of a tie, stage-2 poses within 1e-3 (measured
        5.9e-4), the key-point slots of a (crop, hypothesis) bit-equal to >= 97.5 % in the worst of the 320 pairs (measured 98.4 %)
        and >= 99 % on average (measured 99.3 %) — fp16 operands carry 11 bits, the slots flip where a certainty logit or a
        coordinate sits within ~1e-3 of its threshold."""
    import bench
    from picopose_amd import ops
    from picopose_amd.picopose import Net

    vit, B, N = "dinov2_vitl14", 64, 64
    net = Net(bench.make_cfg(vit))
    bench.seeded_weights(net, 4, vit)
    net = net.cuda().eval()
    ep = bench.make_end_points(B, N, "cuda", 100)
    fe = net.feature_extractor
    outs = {}
    old = ops.PRECISION
    try:
        for mode in ("f16", "f16x3"):
            ops.PRECISION = mode
            with torch.no_grad():      # the bank is computed in the mode under test, as bench.py --mode does
                ep["template_feature"] = torch.stack([fe(ep["tem_rgb"][b])[-1] for b in range(B)]).half()
            outs[mode] = net(ep, 5)
            if mode == "f16":
                pick = [3, 17, B - 1]
                sub = {k: v[pick].contiguous() for k, v in ep.items()}
                small = net(sub, 5)
                for h in range(5):
                    for key in small[h]:
                        assert torch.equal(small[h][key], outs[mode][h][key][pick]), (h, key)
    finally:
        ops.PRECISION = old
    a, b = outs["f16"], outs["f16x3"]
    assert min(int((o["pred_tar_pts"][..., 0] >= 0).sum(1).min()) for o in a) >= 1000
    same_tem = torch.stack([(a[h]["tem_pose"] == b[h]["tem_pose"]).flatten(1).all(1) for h in range(5)])        # (5, B)
    pose_err, agree = 0.0, []
    for h in range(5):
        m = same_tem[h]
        pose_err = max(pose_err, float((a[h]["pred_poses"][m] - b[h]["pred_poses"][m]).abs().max()))
        same = (a[h]["pred_tar_pts"][m] == b[h]["pred_tar_pts"][m]).all(-1) & (a[h]["pred_src_pts"][m] == b[h]["pred_src_pts"][m]).all(-1)
        agree.append(same.float().mean(1))
    agree = torch.cat(agree)
    print(f"configs[4] share, f16 vs f16x3 engine: same template in {int(same_tem.sum())} of {same_tem.numel()} (crop, hypothesis) pairs; on those: "
          f"stage-2 pose max abs diff {pose_err:.2e}, key-point slot agreement min {float(agree.min()):.4f} mean {float(agree.mean()):.4f}")
    # 64 random templates per crop have stage-1 scores a few 1e-4 apart: the two modes' banks differ by ~1e-3 of a feature, so the
    # top-5 ORDER may differ for near-tied templates — most pairs must still coincide
    assert float(same_tem.float().mean()) >= 0.8
    assert pose_err <= 1e-3
    assert float(agree.min()) >= 0.975 and float(agree.mean()) >= 0.99
