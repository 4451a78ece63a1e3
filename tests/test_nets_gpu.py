"""GPU parity of the network modules (HIP engine) against the CPU oracle, seeded weights shared by name."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from netcfg import HEADS, TAKE, small_cfg  # noqa: E402

from oracle import nets as on  # noqa: E402
from oracle.weights import seeded_state_dict  # noqa: E402

gpu = pytest.mark.gpu


def _rel(a, b):
    a, b = a.cpu(), b.cpu()
    return float((a - b).abs().max()) / max(1e-6, float(b.abs().max()))


@gpu
def test_feature_extractor_matches_oracle():
    from picopose_amd.model.stage1 import FeatureExtractor

    cfg = small_cfg()
    fe = FeatureExtractor(cfg.stage1)
    sd = seeded_state_dict(fe.state_dict(), 11)
    fe.load_state_dict(sd)
    fe = fe.cuda().eval()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 3, 224, 224, generator=g)
    ref = on.vit_features({"feature_extractor.dinov2." + k[len("dinov2."):]: v for k, v in sd.items()}, x, HEADS, TAKE)
    got = fe(x.cuda())
    assert len(got) == 4 and got[0].shape == (2, 384, 16, 16)
    for r, o in zip(ref, got):
        assert _rel(o, r) <= 2e-4, _rel(o, r)  # fp32 both sides; 12 blocks of reassociation noise


def _golden(golden_dir):
    import numpy as np

    z = np.load(os.path.join(golden_dir, "nets.npz"))
    return z, {k: torch.from_numpy(z[k]).cuda() for k in z.files if z[k].dtype == np.float32}


def _close(a, ref, tol):
    import numpy as np

    err = float(np.abs(a.cpu().numpy() - ref).max())
    assert err <= tol * max(1.0, float(np.abs(ref).max())), err


@gpu
def test_feature_extractor_vs_reference_golden(golden_dir):
    from picopose_amd.model.stage1 import FeatureExtractor

    z, t = _golden(golden_dir)
    fe = FeatureExtractor(small_cfg().stage1)
    fe.load_state_dict(seeded_state_dict(fe.state_dict(), int(z["vit/seed"])))
    feats = fe.cuda().eval()(t["vit/x"])
    _close(feats[-1], z["vit/feat_last"], 2e-4)
    _close(torch.stack([f[0, :, 3, 5] for f in feats]), z["vit/feat_probe"], 2e-4)


@gpu
def test_affine_regressor_vs_reference_golden(golden_dir):
    from picopose_amd.model.stage2 import AffineRegressor

    z, t = _golden(golden_dir)
    ar = AffineRegressor(small_cfg().stage2)
    ar.load_state_dict(seeded_state_dict(ar.state_dict(), int(z["aff/seed"])))
    tr, sc, ip = ar.cuda().eval()(t["aff/sim"])
    assert tr.shape == (3, 2) and sc.shape == (3,) and ip.shape == (3, 2)
    _close(tr, z["aff/translation"], 1e-4)
    _close(sc, z["aff/scale"], 1e-4)
    _close(ip, z["aff/inplane"], 1e-4)


@gpu
def test_corr_lookup_vs_reference_golden(golden_dir):
    from picopose_amd import ops

    z, t = _golden(golden_dir)
    out = ops.corr_lookup(ops.to_nhwc(t["corr/f1"]), ops.to_nhwc(t["corr/f2"]), ops.to_nhwc(t["corr/flow"]), 3, 2)
    _close(ops.to_nchw(out), z["corr/out"], 2e-5)


@gpu
def test_lookup_and_warp_with_shared_query_maps_and_strided_input():
    """The hypothesis-major batch keeps the query maps un-tiled (image b reads map b % Bq) and the template map
    is a channel slice of the decoder input buffer: bit-identical to the tiled / contiguous call."""
    from picopose_amd import ops

    g = torch.Generator(device="cuda").manual_seed(3)
    Bq, hyp, H, W, C = 3, 4, 16, 16, 64
    B = Bq * hyp
    f1 = torch.randn(B, H, W, C, device="cuda", generator=g)
    fq = torch.randn(Bq, H, W, C, device="cuda", generator=g)
    flow = torch.randn(B, H, W, 2, device="cuda", generator=g) * 3
    wide = torch.zeros(B, H, W, 160, device="cuda")
    wide[..., 32:96] = f1
    ref = ops.corr_lookup(f1, fq.repeat(hyp, 1, 1, 1), flow, 2, 2, c_pad=56)
    assert torch.equal(ops.corr_lookup(wide[..., 32:96], fq, flow, 2, 2, c_pad=56), ref)
    assert torch.equal(ops.warp(fq, flow), ops.warp(fq.repeat(hyp, 1, 1, 1), flow))
    out = torch.zeros(B, H, W, 160, device="cuda")
    ops.warp(fq, flow, out=out[..., 96:160])
    assert torch.equal(out[..., 96:160], ops.warp(fq.repeat(hyp, 1, 1, 1), flow)) and not out[..., :96].any()


@gpu
def test_offset_regressor_vs_reference_golden(golden_dir):
    from picopose_amd.model.stage3 import OffsetRegressor

    z, t = _golden(golden_dir)
    orr = OffsetRegressor(small_cfg().stage3)
    orr.load_state_dict(seeded_state_dict(orr.state_dict(), int(z["s3/seed"])))
    orr = orr.cuda().eval()
    ft, fr = [t[f"s3/ft{i}"] for i in range(4)], [t[f"s3/fr{i}"] for i in range(4)]
    dt = orr.dpt_head(ft)
    assert [tuple(d.shape) for d in dt] == [(1, 256, 16, 16), (1, 256, 32, 32), (1, 256, 64, 64)]
    _close(dt[0], z["s3/dpt_t_path4"], 2e-4)
    _close(dt[1][0, :, ::8, ::8], z["s3/dpt_t_path3_probe"], 2e-4)
    _close(dt[2][0, :, ::16, ::16], z["s3/dpt_t_path2_probe"], 2e-4)
    fl, ce = orr(ft, fr, t["s3/init_flow"], t["s3/init_cert"])
    assert [tuple(f.shape) for f in fl] == [(1, 2, 16, 16), (1, 2, 32, 32), (1, 2, 64, 64)]
    for i in range(3):
        # offsets in pixels / certainty logits: stated tolerance 5e-4 relative to the tensor's max (fp32 both sides)
        _close(fl[i], z[f"s3/flow{i}"], 5e-4)
        _close(ce[i], z[f"s3/cert{i}"], 5e-4)


@gpu
@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_dead_layer1_branch_does_not_reach_any_output(golden_dir, monkeypatch, precision):
    """dpt.py:252-272 computes layer_1 = resize_layers[0](projects[0](x0)) and layer_1_rn = layer1_rn(layer_1) but reads only
    `layer_1_rn.shape[2:]` (refinenet1 is commented out): the build does not compute the three layers.  With them computed
    (stage3.COMPUTE_DEAD_LAYER1) every output is bit for bit the same — and the same as with their weights replaced by NaN."""
    from picopose_amd import ops
    from picopose_amd.model import stage3

    z, t = _golden(golden_dir)
    monkeypatch.setattr(ops, "PRECISION", precision)
    outs = []
    for dead, poison in ((False, False), (True, False), (False, True)):
        monkeypatch.setattr(stage3, "COMPUTE_DEAD_LAYER1", dead)
        orr = stage3.OffsetRegressor(small_cfg().stage3)
        sd = seeded_state_dict(orr.state_dict(), int(z["s3/seed"]))
        if poison:
            for k in sd:
                if k.startswith(("dpt_head.projects.0.", "dpt_head.resize_layers.0.", "dpt_head.scratch.layer1_rn.")):
                    sd[k] = torch.full_like(sd[k], float("nan"))
        orr.load_state_dict(sd)
        orr = orr.cuda().eval()
        ft, fr = [t[f"s3/ft{i}"] for i in range(4)], [t[f"s3/fr{i}"] for i in range(4)]
        fl, ce = orr(ft, fr, t["s3/init_flow"], t["s3/init_cert"])
        outs.append([x.cpu() for x in orr.dpt_head(ft) + fl + ce])
    for o in outs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(outs[0], o))
    assert all(torch.isfinite(a).all() for a in outs[0])


@gpu
@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_out_conv_before_the_interpolation_is_the_same_map(golden_dir, monkeypatch, precision):
    """FeatureFusionBlock (dpt.py:150-155) is out_conv(interpolate(x)): a 1x1 convolution after a bilinear interpolation whose weights sum
    to one per pixel — interpolate(out_conv(x)) is the same map, and the build runs the convolution first, on a quarter of the pixels
    (stage3.OUT_CONV_FIRST).  Against the reference's order: equal up to the rounding of the reordered sums — measured 0.5 - 1.3e-6 of
    the path maps' max in both arithmetic modes (stated 3e-6) and, carried through the flow decoder of these seeded weights, up to 8e-6
    of the flows' / certainties' max (stated 3e-5); the reference fixture bars of test_offset_regressor_vs_reference_golden hold in
    either order (that test runs the default)."""
    from picopose_amd import ops
    from picopose_amd.model import stage3

    z, t = _golden(golden_dir)
    monkeypatch.setattr(ops, "PRECISION", precision)
    outs = []
    for first in (True, False):
        monkeypatch.setattr(stage3, "OUT_CONV_FIRST", first)
        orr = stage3.OffsetRegressor(small_cfg().stage3)
        orr.load_state_dict(seeded_state_dict(orr.state_dict(), int(z["s3/seed"])))
        orr = orr.cuda().eval()
        ft, fr = [t[f"s3/ft{i}"] for i in range(4)], [t[f"s3/fr{i}"] for i in range(4)]
        fl, ce = orr(ft, fr, t["s3/init_flow"], t["s3/init_cert"])
        outs.append([x.cpu() for x in orr.dpt_head(ft) + fl + ce])
    assert not all(torch.equal(a, b) for a, b in zip(*outs)), "both runs took the same order"
    dev = [(a - b).abs().max().item() / a.abs().max().item() for a, b in zip(*outs)]
    print(f"out_conv first vs reference order [{precision}]: max |diff| / max per output (3 path maps, 3 flows, 3 certainties):", " ".join(f"{d:.1e}" for d in dev))
    assert max(dev[:3]) <= 3e-6 and max(dev[3:]) <= 3e-5, dev
    _close(outs[0][0], z["s3/dpt_t_path4"], 2e-4)
    _close(outs[1][0], z["s3/dpt_t_path4"], 2e-4)


@gpu
@pytest.mark.parametrize("precision", ["f16x3", "f16"])
def test_fused_xhead_first_layers_equal_two_launches(golden_dir, monkeypatch, precision):
    """The flow and certainty heads' first layers (conv 3x3 640 -> 512 each, flow_decoder.py:58-72) run as ONE launch with the
    filters concatenated along N, their successors read channel slices of the shared hidden operand (stage3.FUSE_XHEADS): same
    flows and certainties as two launches.  Bit for bit when the two filters' operand scales (a power of two per weight tensor,
    from its max) coincide with the concatenated tensor's — the case for these weights and for any pair of equally initialised /
    trained heads; otherwise equal up to fp16 subnormals of the lo terms (stated bound 1e-6 of the tensor's max)."""
    from picopose_amd import ops
    from picopose_amd.model import stage3

    z, t = _golden(golden_dir)
    monkeypatch.setattr(ops, "PRECISION", precision)
    monkeypatch.setattr(ops, "WINOGRAD4", False)     # (the fused launch is a variant of the DIRECT convolutions: like for like)
    outs = []
    for fuse in (True, False):
        monkeypatch.setattr(stage3, "FUSE_XHEADS", fuse)
        orr = stage3.OffsetRegressor(small_cfg().stage3)
        orr.load_state_dict(seeded_state_dict(orr.state_dict(), int(z["s3/seed"])))
        orr = orr.cuda().eval()
        g = torch.Generator().manual_seed(5)
        ft = [0.5 * torch.randn(3, 384, 16, 16, generator=g).cuda() for _ in range(4)]
        fr = [0.5 * torch.randn(3, 384, 16, 16, generator=g).cuda() for _ in range(4)]
        fl, ce = orr(ft, fr, (0.5 + torch.randn(3, 2, 16, 16, generator=g)).cuda(), (torch.rand(3, 1, 16, 16, generator=g) > 0.3).float().cuda())
        outs.append([x.cpu() for x in fl + ce])
    for a, b in zip(*outs):
        assert float((a - b).abs().max()) <= 1e-6 * float(b.abs().max())
    print("fused vs two launches, bit-equal tensors:", sum(int(torch.equal(a, b)) for a, b in zip(*outs)), "of", len(outs[0]))


@gpu
@pytest.mark.parametrize("vit", ["dinov2_vitb14", "dinov2_vitl14"])
@pytest.mark.parametrize("force", [None, "4", "5", "7", "8"])
def test_feature_extractor_vs_reference_golden_vitb_vitl(golden_dir, monkeypatch, vit, force):
    """FeatureExtractor at the widths of BASELINE configs[2] (ViT-B/14) and config/base.yaml (ViT-L/14) against the
    reference's outputs (tests/golden/vit_wide.npz) — with the autotuner's kernels, and with every pre-split GEMM pinned
    to the persistent 256x128 (4) / 256x256 (5) / two-per-CU (7) kernels the headline bench runs."""
    import numpy as np
    import types

    from picopose_amd.model.stage1 import FeatureExtractor

    if force is not None:
        monkeypatch.setenv("PP_GEMM_FORCE_CFG", force)
    C, idx = {"dinov2_vitb14": (768, [[0, 2], [3, 5], [6, 8], [9, 11]]), "dinov2_vitl14": (1024, [[0, 5], [6, 11], [12, 17], [18, 23]])}[vit]
    z = np.load(os.path.join(golden_dir, "vit_wide.npz"))
    wseed, xseed = (int(v) for v in z[f"{vit}/seeds"])
    fe = FeatureExtractor(types.SimpleNamespace(vit_type=vit, pretrained=False, interaction_indexes=idx))
    fe.load_state_dict(seeded_state_dict(fe.state_dict(), wseed))
    fe = fe.cuda().eval()
    x = torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(xseed))
    feats = fe(x.cuda())
    assert len(feats) == 4 and feats[0].shape == (1, C, 16, 16)
    for l, f in enumerate(feats):          # tolerance: 2e-4 * max|level| as for ViT-S (24 blocks of fp32 reassociation at ViT-L)
        amax = float(z[f"{vit}/absmax"][l])
        assert float(np.abs(f[0, :, 3, 5].cpu().numpy() - z[f"{vit}/pixel_probe"][l]).max()) <= 2e-4 * amax
        assert float(np.abs(f[0, ::32].cpu().numpy() - z[f"{vit}/channel_probe"][l]).max()) <= 2e-4 * amax


@gpu
@pytest.mark.parametrize("H,levels,scale", [(16, 1, 1.0), (32, 2, 3.0), (64, 3, 0.7), (64, 3, 40.0), (24, 2, 2.0)])
def test_tiled_corr_lookup_equals_the_lane_per_position_kernel(monkeypatch, H, levels, scale):
    """The matrix-core correlation lookup (8x8 pixel tiles x 16x16 regions, f16x3) against the exact-fp32
    lane-per-position kernel it replaced (PP_CORR_TILED=0) and against the CPU oracle: smooth flows (one region pass),
    noisy flows (several passes per tile), flows far outside the image (no pass at all), a query map shared by several
    images of the batch, an output with padded channels."""
    from picopose_amd import ops

    g = torch.Generator().manual_seed(H * 10 + levels)
    B, C = 4, 64
    f1 = torch.randn(B, H, H, C, generator=g).cuda()
    f2 = torch.randn(2, H, H, C, generator=g).cuda()                     # image b reads f2[b % 2]
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(H, dtype=torch.float32), indexing="ij")
    smooth = torch.stack([0.15 * xx - 0.1 * yy + 1.3, 0.1 * xx + 0.2 * yy - 2.1], dim=-1)[None].repeat(B, 1, 1, 1)
    flow = (smooth + scale * torch.randn(B, H, H, 2, generator=g)).cuda()
    pad = -(-(levels * 25) // 8) * 8
    tiled = ops.corr_lookup(f1, f2, flow, levels, 2, c_pad=pad)
    monkeypatch.setenv("PP_CORR_TILED", "0")
    exact = ops.corr_lookup(f1, f2, flow, levels, 2, c_pad=pad)
    monkeypatch.delenv("PP_CORR_TILED")
    ref = on.corr_lookup(f1.cpu().permute(0, 3, 1, 2), f2.cpu().repeat(2, 1, 1, 1).permute(0, 3, 1, 2), flow.cpu().permute(0, 3, 1, 2), levels, 2)
    ref = ref.permute(0, 2, 3, 1)
    n = levels * 25
    assert float((exact[..., :n].cpu() - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
    assert float((tiled[..., :n].cpu() - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
    assert torch.equal(tiled[..., n:], torch.zeros_like(tiled[..., n:]))
    if H % 8 == 0:
        # the same lookup fed with the maps as hl operands (f1 as a column block of a wider operand, as the flow decoder
        # holds it; the pooled levels split from the fp32 pools): identical bits — a value's split does not depend on who makes it
        wide = ops.Split.empty(B * H * H, 96, f1.device)
        wide.hl.zero_()
        ops.split_activation(f1, B, H * H, C, H * H * C, C, into=(wide, 32))
        f2s = ops.Split(ops.split_activation(f2, 2, H * H, C, H * H * C, C))
        from_operands = ops.corr_lookup(f1, f2, flow, levels, 2, c_pad=pad, f1_hl=(wide, 32), f2_hl=f2s)
        assert torch.equal(from_operands, tiled)


@gpu
@pytest.mark.parametrize("H,levels,scale", [(16, 1, 1.0), (32, 2, 3.0), (64, 3, 0.7), (64, 3, 40.0)])
def test_fp16_mode_tiled_corr_lookup_with_one_term_products(monkeypatch, H, levels, scale):
    """ops.PRECISION = "f16" (bench.py --mode fp16): the tiled correlation lookup on plain fp16 operands, one MFMA per product (round 6;
    it ran the three-term form in that mode too).  Against the CPU oracle on the SAME fp16-rounded features — what the mode promises:
    exact products of 11-bit operands, fp32 sums (bar 2e-5 of the max, the f16x3 kernel's own) — and against the oracle on the
    unrounded features at the mode's operand rounding (2e-3)."""
    from picopose_amd import ops

    monkeypatch.setattr(ops, "PRECISION", "f16")
    g = torch.Generator().manual_seed(H * 10 + levels + 2)
    B, C = 4, 64
    f1 = torch.randn(B, H, H, C, generator=g).cuda()
    f2 = torch.randn(2, H, H, C, generator=g).cuda()
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(H, dtype=torch.float32), indexing="ij")
    smooth = torch.stack([0.15 * xx - 0.1 * yy + 1.3, 0.1 * xx + 0.2 * yy - 2.1], dim=-1)[None].repeat(B, 1, 1, 1)
    flow = (smooth + scale * torch.randn(B, H, H, 2, generator=g)).cuda()
    pad = -(-(levels * 25) // 8) * 8
    one = ops.corr_lookup(f1, f2, flow, levels, 2, c_pad=pad)
    monkeypatch.setattr(ops, "PRECISION", "f16x3")
    three = ops.corr_lookup(f1, f2, flow, levels, 2, c_pad=pad)
    assert not torch.equal(one, three), "both runs took the three-term form"
    n = levels * 25

    def oracle(a, b):
        # (the pyramid levels are average pools of the fp32 map; the kernel rounds each level's values as it stages them)
        return on.corr_lookup(a.cpu().permute(0, 3, 1, 2), b.cpu().repeat(2, 1, 1, 1).permute(0, 3, 1, 2), flow.cpu().permute(0, 3, 1, 2), levels, 2).permute(0, 2, 3, 1)

    ref = oracle(f1, f2)
    top = max(1.0, float(ref.abs().max()))
    assert float((one[..., :n].cpu() - ref).abs().max()) <= 2e-3 * top
    if levels == 1:   # one level: the staged values are exactly the fp16 roundings of the inputs
        r16 = oracle(f1.half().float(), f2.half().float())
        assert float((one[..., :n].cpu() - r16).abs().max()) <= 2e-5 * top
    assert torch.equal(one[..., n:], torch.zeros_like(one[..., n:]))


@gpu
@pytest.mark.parametrize("H,levels,scale", [(16, 1, 1.0), (32, 2, 3.0), (64, 3, 0.7), (64, 3, 40.0)])
def test_exact_mode_tiled_corr_lookup_on_fp32_matrix_cores(monkeypatch, H, levels, scale):
    """ops.PRECISION = "f32" (bench.py --mode exact): the tiled correlation lookup with fp32 chunks and v_mfma_f32_32x32x2_f32 (round 5)
    against the lane-per-position fp32 kernel it replaces in that mode (PP_CORR_TILED=0) and the CPU oracle — smooth, noisy and
    out-of-image flows, a shared query map, padded output channels.  Both are fp32 fma chains over the channels (in different orders)."""
    from picopose_amd import ops

    monkeypatch.setattr(ops, "PRECISION", "f32")
    g = torch.Generator().manual_seed(H * 10 + levels + 1)
    B, C = 4, 64
    f1 = torch.randn(B, H, H, C, generator=g).cuda()
    f2 = torch.randn(2, H, H, C, generator=g).cuda()
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(H, dtype=torch.float32), indexing="ij")
    smooth = torch.stack([0.15 * xx - 0.1 * yy + 1.3, 0.1 * xx + 0.2 * yy - 2.1], dim=-1)[None].repeat(B, 1, 1, 1)
    flow = (smooth + scale * torch.randn(B, H, H, 2, generator=g)).cuda()
    pad = -(-(levels * 25) // 8) * 8
    tiled = ops.corr_lookup(f1, f2, flow, levels, 2, c_pad=pad)
    monkeypatch.setenv("PP_CORR_TILED", "0")
    lane = ops.corr_lookup(f1, f2, flow, levels, 2, c_pad=pad)
    monkeypatch.delenv("PP_CORR_TILED")
    ref = on.corr_lookup(f1.cpu().permute(0, 3, 1, 2), f2.cpu().repeat(2, 1, 1, 1).permute(0, 3, 1, 2), flow.cpu().permute(0, 3, 1, 2), levels, 2)
    ref = ref.permute(0, 2, 3, 1)
    n = levels * 25
    bar = 2e-5 * max(1.0, float(ref.abs().max()))
    assert float((tiled[..., :n].cpu() - ref).abs().max()) <= bar
    assert float((tiled[..., :n] - lane[..., :n]).abs().max()) <= bar
    assert torch.equal(tiled[..., n:], torch.zeros_like(tiled[..., n:]))
