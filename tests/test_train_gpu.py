"""GPU: the training forward (Net.forward in train mode = model/picopose.py:114-137, forward values) against the REFERENCE's
own run (tests/golden/train_forward.npz) and, piece by piece, against the fixture-pinned oracle (oracle/train.py)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from netcfg import make_train_end_points, small_cfg  # noqa: E402
from test_train_oracle import CASES, LOSS_KEYS, load_train_fixture, patch_coords  # noqa: E402

gpu = pytest.mark.gpu


def _cuda(ep):
    return {k: v.cuda() for k, v in ep.items()}


@gpu
@pytest.mark.parametrize("name", CASES)
def test_keypoint_sampler_matches_the_reference(golden_dir, name):
    """Integer pixel coordinates of every key-point.  The per-point arithmetic is fp32 in both, but a GEMM's summation
    order is the library's: a projection that lands within an ulp of an integer may truncate to the neighbouring pixel.
    Stated tolerance: <= 0.1 % of the entries differ, each by one pixel or by validity."""
    from picopose_amd.picopose import Net

    z, ep, _, _ = load_train_fixture(golden_dir, name)
    kp = Net(small_cfg()).compute_keypoint_data(_cuda(ep))
    for k in ("src_pts", "tar_pts"):
        got, ref = kp[k].cpu(), patch_coords(z[f"kp_{k}_px"])
        assert got.shape == ref.shape == (ep["real_rgb"].shape[0], 4096, 2)
        differ = (got != ref).any(dim=-1)
        print(k, "entries differing from the reference:", int(differ.sum()), "of", differ.numel())
        assert differ.float().mean() <= 1e-3, (k, int(differ.sum()))
        d = (got[differ] - ref[differ]).abs() * 3.5
        assert all(bool((row <= 1.01).all()) or bool((a == -1).all()) or bool((b == -1).all())
                   for row, a, b in zip(d, got[differ], ref[differ]))
    assert int((kp["src_pts"][..., 0] != -1).sum()) > 1000


@gpu
def test_keypoint_sampler_agrees_with_the_oracle_on_other_geometry():
    from oracle import train as ot
    from picopose_amd.picopose import Net

    ep = make_train_end_points(3, 77)
    ep["real_mask"][1] = 0                       # a pair without any correspondence
    ref = ot.keypoint_data({k: v.clone() for k, v in ep.items()})
    got = Net(small_cfg()).compute_keypoint_data(_cuda(ep))
    for k in ("src_pts", "tar_pts"):
        differ = (got[k].cpu() != ref[k]).any(dim=-1)
        assert differ.float().mean() <= 1e-3, (k, int(differ.sum()))
        assert bool((got[k][1] == -1).all())


@gpu
@pytest.mark.parametrize("rows,C,relu", [(2 * 16 * 16, 256, True), (3 * 64 * 64, 256, False), (77, 64, True)])
def test_batchnorm_training_mode(rows, C, relu):
    from picopose_amd import ops
    from picopose_amd.model.common import bn_p

    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, C, generator=g) * 3 + 5            # a mean well away from zero: E[x^2] - E[x]^2 in fp32 would lose digits
    r1, r2 = torch.randn(rows, C, generator=g), torch.randn(rows, C, generator=g)
    bn = bn_p(C)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(C, generator=g))
        bn.running_mean.copy_(torch.randn(C, generator=g))
        bn.running_var.copy_(torch.rand(C, generator=g) + 0.5)
    rm, rv = bn.running_mean.clone(), bn.running_var.clone()
    ref = F.batch_norm(x.t()[None], rm, rv, bn.weight, bn.bias, True, 0.1, 1e-5)[0].t()
    ref = (F.relu(ref) if relu else ref) + r1 + r2
    bn = bn.cuda()
    with torch.no_grad():
        got = ops.batchnorm_train(x.cuda().reshape(1, 1, rows, C), bn, relu=relu, residual=r1.cuda().reshape(1, 1, rows, C),
                                  residual2=r2.cuda().reshape(1, 1, rows, C)).reshape(rows, C).cpu()
    assert (got - ref).abs().max() <= 2e-5 * ref.abs().max()
    assert (bn.running_mean.cpu() - rm).abs().max() <= 1e-6 * rm.abs().max()
    assert (bn.running_var.cpu() - rv).abs().max() <= 1e-5 * rv.abs().max()
    assert int(bn.num_batches_tracked) == 1


@gpu
def test_loss_kernels_agree_with_the_oracle():
    from oracle import train as ot
    from picopose_amd.utils.loss_utils import compute_stage_one_loss, compute_stage_three_loss, compute_stage_two_loss

    g = torch.Generator().manual_seed(3)
    ep = make_train_end_points(2, 9)
    kp = ot.keypoint_data({k: v.clone() for k, v in ep.items()})
    fs, ft = torch.randn(2, 96, 16, 16, generator=g), torch.randn(2, 96, 16, 16, generator=g)
    ref = ot.stage_one_loss(fs, ft, kp["src_pts"], kp["tar_pts"])
    got = compute_stage_one_loss(fs.cuda(), ft.cuda(), kp["src_pts"].cuda(), kp["tar_pts"].cuda())
    assert abs(float(got) - float(ref)) <= 1e-5 * float(ref)
    flows = [torch.randn(2, 2, s, s, generator=g) * 5 for s in (16, 32, 64)]
    certs = [torch.randn(2, 1, s, s, generator=g) * 3 for s in (16, 32, 64)]
    ref3 = ot.stage_three_loss(flows, certs, kp["tar_pts"])
    out = compute_stage_three_loss({}, [f.cuda() for f in flows], [c.cuda() for c in certs], kp["tar_pts"].cuda())
    for i, (lf, lc) in enumerate(ref3):
        assert abs(float(out[f"loss_flow{i}"]) - float(lf)) <= 1e-5 * float(lf)
        assert abs(float(out[f"loss_certainty{i}"]) - float(lc)) <= 1e-5 * float(lc)
    t, s, ip = torch.randn(2, 2, generator=g), torch.rand(2, generator=g) + 0.5, F.normalize(torch.randn(2, 2, generator=g), dim=1)
    ref2 = ot.stage_two_loss(ep, t, s, ip)
    got2 = compute_stage_two_loss(_cuda(ep), t.cuda(), s.cuda(), ip.cuda())
    for a, b in zip(got2, ref2):
        assert abs(float(a) - float(b)) <= 1e-5 * max(1.0, abs(float(b)))


@gpu
@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("engine_precision", ["f16x3", "f32"])
def test_training_forward_matches_the_reference_run(golden_dir, engine_precision, monkeypatch, name):
    """Net(train mode)(end_points) vs the reference's forward_train + Loss on the same batch, weights and noisy affines:
    every loss within 1e-3 relative (fp32 networks on another summation order; the flow losses sum |flow - gt| over ~2k
    pixels), the BatchNorm running buffers within 1e-4, the eval packing rebuilt afterwards."""
    from picopose_amd import ops
    from picopose_amd.picopose import Net
    from picopose_amd.utils.loss_utils import Loss

    monkeypatch.setattr(ops, "PRECISION", engine_precision)
    z, ep, weights, _ = load_train_fixture(golden_dir, name)
    net = Net(small_cfg())
    net.load_state_dict(weights(net.state_dict()))
    net = net.cuda().train()
    res = net.forward_train(_cuda(ep), pred_Ms=torch.from_numpy(z["pred_Ms"]).cuda())
    for k in LOSS_KEYS:
        assert abs(float(res[k]) - float(z[k])) <= 1e-3 * max(1.0, abs(float(z[k]))), (k, float(res[k]), float(z[k]))
    tot = Loss()(res)
    assert abs(float(tot["loss"]) - float(z["total_loss"])) <= 1e-3 * float(z["total_loss"])
    assert all(res[k].requires_grad for k in LOSS_KEYS)       # the full backward is live by default
    # the forward-only training step (no autograd) takes the fused inference kernels for the slice: same losses to 1e-5
    bn_state = {k: v.clone() for k, v in net.state_dict().items()}
    with torch.no_grad():
        plain = net.forward_train(_cuda(ep), pred_Ms=torch.from_numpy(z["pred_Ms"]).cuda())
    net.load_state_dict(bn_state)                             # (that second step moved the running buffers again)
    for k in LOSS_KEYS:
        assert not plain[k].requires_grad
        assert abs(float(plain[k]) - float(res[k].detach())) <= 2e-5 * max(1.0, abs(float(res[k].detach()))), (k, float(plain[k]), float(res[k].detach()))
    sd = net.state_dict()
    for key in z.files:
        if key.startswith("bn/"):
            got, ref = sd[key[3:]].cpu().numpy(), z[key]
            assert np.abs(got - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()), key
    # the eval packing folds the running buffers this step moved: the next eval call must re-fold them — the trained-on
    # module and a fresh one loaded with its state_dict give the same DPT maps
    feats = [torch.randn(2, 16, 16, 384, generator=torch.Generator().manual_seed(i)).cuda() for i in range(4)]
    fresh = Net(small_cfg())
    fresh.load_state_dict(net.state_dict())
    fresh = fresh.cuda().eval()
    net.eval()
    with torch.no_grad():
        a = net.offset_regressor.dpt_head.forward_nhwc(feats)
        b = fresh.offset_regressor.dpt_head.forward_nhwc(feats)
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    net.train()
    # `model(end_points)` dispatches on self.training, drawing its own noisy affines
    np.random.seed(0)
    torch.manual_seed(0)
    again = net(_cuda(ep))
    assert all(torch.isfinite(again[k]) for k in LOSS_KEYS)


# ---- the first backward slice (picopose_amd/autograd.py) against the REFERENCE's own autograd ------------------------------------
def _load_grad_fixture(golden_dir, name="train_grads"):
    from netcfg import train_kwargs
    from oracle.weights import apply_head_calibration, seeded_state_dict

    z = np.load(os.path.join(golden_dir, name + ".npz"))
    B, seed, wseed = (int(v) for v in z["meta"])
    cal = {"flow": [tuple(r) for r in z["cal_flow"]], "cert": [tuple(r) for r in z["cal_cert"]], "proj_bn": float(z["cal_proj_bn"]),
           "affine": {h: (float(z[f"cal_affine_{h}"][0]), tuple(z[f"cal_affine_{h}"][1:])) for h in ("translation", "scale", "inplane")}}
    ep = make_train_end_points(B, seed, poses=(torch.from_numpy(z["real_pose"]), torch.from_numpy(z["tem_pose"])), **train_kwargs(name))
    return z, ep, (lambda template: apply_head_calibration(seeded_state_dict(template, wseed), cal))


@gpu
@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_backward_slice_matches_the_reference_autograd(golden_dir, precision):
    """Scope "slice1": `Loss()(net(end_points))["loss"].backward()` on the HIP model fills the `.grad` of exactly the slice's parameters (every
    parameter of affine_regressor from the stage-2 losses, every parameter of the last ViT block from the InfoNCE loss) and they
    equal the reference's own autograd gradients (tests/golden/train_grads.npz: torch.autograd.grad on the reference Net, CPU)
    within 3e-4 x max|grad| per tensor on both engines (measured 8.1e-5 / 9.0e-5: profiles/r03/backward_slice.txt); parameters outside the slice get none."""
    from picopose_amd import ops
    from picopose_amd.picopose import Net
    from picopose_amd.utils.loss_utils import Loss

    z, ep, weights = _load_grad_fixture(golden_dir)
    old = ops.PRECISION
    ops.PRECISION = precision
    try:
        net = Net(small_cfg())
        net.load_state_dict(weights(net.state_dict()))
        net = net.cuda().train()
        net.train_backward = "slice1"
        res = net(_cuda(ep))
        for k in ("loss_info", "loss_2d_trans", "loss_scale", "loss_inplane"):
            assert abs(float(res[k]) - float(z[k])) <= 1e-3 * max(1.0, abs(float(z[k]))), (k, float(res[k]), float(z[k]))
            assert res[k].requires_grad
        assert not res["loss_flow0"].requires_grad            # (outside the slice: forward values)
        Loss()(res)["loss"].backward()
    finally:
        ops.PRECISION = old
    last = len(net.feature_extractor.dinov2.blocks) - 1
    worst, n_checked, report = 0.0, 0, []
    TOL_GRAD = 3e-4     # measured: 8.1e-5 (f16x3, backward products on range-normalised operands) / 9.0e-5 (f32)
    for name, p in net.named_parameters():
        key = f"grad/{name}"
        in_slice = name.startswith("affine_regressor.") or name.startswith(f"feature_extractor.dinov2.blocks.{last}.")
        if not in_slice:
            assert p.grad is None, name
            continue
        assert key in z.files and p.grad is not None, name
        ref = torch.from_numpy(z[key])
        flat = p.grad.detach().reshape(-1).cpu()
        stride = max(1, -(-flat.numel() // 65536))
        got = flat[::stride]
        assert got.shape == ref.shape, (name, got.shape, ref.shape)
        scale = float(ref.abs().max())
        err = float((got - ref).abs().max())
        rel_norm = abs(float(flat.double().norm()) - float(z[f"gradnorm/{name}"])) / max(float(z[f"gradnorm/{name}"]), 1e-30)
        worst = max(worst, err / max(scale, 1e-30))
        report.append((err / max(scale, 1e-30), rel_norm, name))
        n_checked += 1
    report.sort(reverse=True)
    print(f"backward slice [{precision}]: {n_checked} parameter tensors, worst max|err| / max|grad| = {worst:.2e}; worst five:",
          [(f"{a:.1e}", f"{b:.1e}", n.split(".", 2)[-1]) for a, b, n in report[:5]])
    assert n_checked == 43     # 29 tensors of the affine regressor + 14 of the last ViT block
    assert worst <= TOL_GRAD, report[:3]
    assert max(b for _, b, _ in report) <= TOL_GRAD


@gpu
@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_wide_backward_matches_the_reference_autograd(golden_dir, precision):
    """Scope "vit+stage2": the stage-1 and stage-2 losses train what the reference trains with them.  The `.grad` of
    EVERY dinov2 parameter the path uses (12 blocks, patch embedding, cls token, position embedding through its bicubic
    resampling) equals the reference's autograd of loss_info + loss_2d_trans + loss_scale + loss_inplane (fixture keys grad2/...:
    InfoNCE directly, the stage-2 losses through the similarity volume), the affine regressor's equals the stage-2 gradients
    (keys grad/...); dinov2.norm / mask_token (unused by the path) and the stage-3 modules get none.  Bars: max|err| / max|grad|
    per tensor and the relative error of its L2 norm (measured: profiles/r03/backward_slice.txt)."""
    from picopose_amd import ops
    from picopose_amd.picopose import Net
    from picopose_amd.utils.loss_utils import Loss

    z, ep, weights = _load_grad_fixture(golden_dir)
    old = ops.PRECISION
    ops.PRECISION = precision
    try:
        net = Net(small_cfg())
        net.load_state_dict(weights(net.state_dict()))
        net = net.cuda().train()
        net.train_backward = "vit+stage2"
        res = net(_cuda(ep))
        for k in ("loss_info", "loss_2d_trans", "loss_scale", "loss_inplane"):
            assert abs(float(res[k]) - float(z[k])) <= 1e-3 * max(1.0, abs(float(z[k]))), (k, float(res[k]), float(z[k]))
        Loss()(res)["loss"].backward()
    finally:
        ops.PRECISION = old
    TOL = 3e-4     # measured 8.8e-5 (f16x3) / 8.4e-5 (f32), norms 8.6e-5 / 8.2e-5
    report, n_checked = [], 0
    for name, p in net.named_parameters():
        if name.startswith("affine_regressor."):
            key, nkey = f"grad/{name}", f"gradnorm/{name}"
        elif name.startswith("feature_extractor.dinov2."):
            if not bool(z[f"grad2used/{name}"]):
                assert p.grad is None, name                     # norm.*, mask_token: not on the path
                continue
            key, nkey = f"grad2/{name}", f"grad2norm/{name}"
        else:
            assert p.grad is None, name                         # stage 3: forward-only
            continue
        assert p.grad is not None, name
        ref = torch.from_numpy(z[key])
        flat = p.grad.detach().reshape(-1).cpu()
        stride = max(1, -(-flat.numel() // (65536 if key.startswith("grad/") else 4096)))
        got = flat[::stride]
        assert got.shape == ref.shape, (name, got.shape, ref.shape)
        scale = max(float(ref.abs().max()), 1e-30)
        rel_norm = abs(float(flat.double().norm()) - float(z[nkey])) / max(float(z[nkey]), 1e-30)
        report.append((float((got - ref).abs().max()) / scale, rel_norm, name))
        n_checked += 1
    report.sort(reverse=True)
    print(f"wide backward [{precision}]: {n_checked} parameter tensors, worst max|err| / max|grad| = {report[0][0]:.2e}, worst norm error "
          f"{max(b for _, b, _ in report):.2e}; worst eight:", [(f"{a:.1e}", f"{b:.1e}", n.replace("feature_extractor.dinov2.", "")) for a, b, n in report[:8]])
    assert n_checked == 29 + 12 * 14 + 4     # affine regressor, 12 blocks, patch_embed.proj.{weight,bias} + cls_token + pos_embed
    assert report[0][0] <= TOL, report[:3]
    assert max(b for _, b, _ in report) <= TOL


@gpu
@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_infonce_backward_with_repeated_real_cells_matches_the_reference_autograd(golden_dir, precision):
    """the InfoNCE rows of the REAL tokens are re-projected key-points quantised to the 16x16 feature grid
    (utils/loss_utils.py:156-165), so several key-points can share a cell and the backward of the reference's `gather`
    (utils/torch_utils.py:257-283) is a scatter-ADD.  The fixtures of round 3 (real crop = the larger view) have no repeated row,
    so an `index_copy_` passed them.  tests/golden/train_grads_dup.npz = the reference's autograd of loss_info on a batch whose
    real crop is the SMALLER view (24 / 34 of 121 / 123 rows repeat an earlier cell): d(loss_info) / d(every dinov2 parameter);
    and the scatter is reproducible bit for bit (fixed summation order, no atomics)."""
    from picopose_amd import ops
    from picopose_amd.picopose import Net

    z, ep, weights = _load_grad_fixture(golden_dir, "train_grads_dup")
    assert int(z["infonce_repeats"].min()) > 10
    old = ops.PRECISION
    ops.PRECISION = precision
    grads = []
    try:
        for rep in range(2):
            net = Net(small_cfg())
            net.load_state_dict(weights(net.state_dict()))
            net = net.cuda().train()
            net.train_backward = "vit+stage2"
            res = net.forward_train(_cuda(ep), pred_Ms=torch.from_numpy(z["pred_Ms"]).cuda())
            assert abs(float(res["loss_info"]) - float(z["loss_info"])) <= 1e-3 * float(z["loss_info"])
            res["loss_info"].backward()
            grads.append({n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None})
    finally:
        ops.PRECISION = old
    assert all(torch.equal(grads[0][n], grads[1][n]) for n in grads[0]) and grads[0].keys() == grads[1].keys()
    report = []
    for name, p in net.named_parameters():
        if not name.startswith("feature_extractor.dinov2."):
            assert p.grad is None, name
            continue
        if not bool(z[f"gradiused/{name}"]):
            assert p.grad is None, name
            continue
        ref = torch.from_numpy(z[f"gradi/{name}"])
        flat = p.grad.detach().reshape(-1).cpu()
        got = flat[::max(1, -(-flat.numel() // 4096))]
        assert got.shape == ref.shape, name
        nref = float(z[f"gradinorm/{name}"])
        report.append((float((got - ref).abs().max()) / max(float(ref.abs().max()), 1e-30), abs(float(flat.double().norm()) - nref) / max(nref, 1e-30), name))
    report.sort(reverse=True)
    print(f"InfoNCE backward with repeated rows [{precision}]: {len(report)} tensors, worst max|err| / max|grad| = {report[0][0]:.2e}, worst norm "
          f"error {max(b for _, b, _ in report):.2e}")
    assert len(report) == 12 * 14 + 4
    assert report[0][0] <= 3e-4 and max(b for _, b, _ in report) <= 3e-4, report[:3]


@gpu
def test_scatter_add_rows_sums_repeated_destinations_in_a_fixed_order():
    from picopose_amd import _lib

    g = torch.Generator().manual_seed(5)
    n, C, R = 700, 384, 90
    src = torch.randn(n, C, generator=g)
    idx = torch.randint(0, R, (n,), generator=g)
    idx[:5] = 89
    ref = torch.zeros(R, C, dtype=torch.float64).index_add_(0, idx, src.double())
    outs = []
    for _ in range(2):
        dst = torch.zeros(R, C, device="cuda")
        _lib.check(_lib.lib().pp_scatter_add_rows(src.cuda().data_ptr(), idx.cuda().data_ptr(), n, C, dst.data_ptr(), _lib.stream_ptr()), "pp_scatter_add_rows")
        outs.append(dst.cpu())
    assert torch.equal(outs[0], outs[1])
    assert float((outs[0].double() - ref).abs().max()) <= 1e-5
    # sequential fp32 sums in ascending source order, exactly
    seq = torch.zeros(R, C)
    for i in range(n):
        seq[idx[i]] += src[i]
    assert torch.equal(outs[0], seq)


@gpu
def test_deterministic_option_repeats_bit_for_bit(golden_dir, monkeypatch):
    """ADVICE r03 / VERDICT r03 #7: with autograd.DETERMINISTIC the feature-warp and correlation-lookup adjoints accumulate their
    scatters in 64-bit fixed point with integer atomics (order-independent), and every other reduction of the backward has a fixed
    order: two full backward passes give the SAME BITS in every one of the 338 gradient tensors, and they agree with the default
    (fp32-atomic) gradients within the scatter kernels' stated 5e-4 of a tensor's largest entry."""
    from picopose_amd import autograd as ag
    from picopose_amd.picopose import Net
    from picopose_amd.utils.loss_utils import Loss

    z, ep, weights = _load_grad_fixture(golden_dir)
    pred_Ms = torch.from_numpy(z["pred_Ms"]).cuda()

    def grads(det):
        monkeypatch.setattr(ag, "DETERMINISTIC", det)
        net = Net(small_cfg())
        net.load_state_dict(weights(net.state_dict()))
        net = net.cuda().train()
        Loss()(net.forward_train(_cuda(ep), pred_Ms=pred_Ms))["loss"].backward()
        return {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}

    a, b, c = grads(True), grads(True), grads(False)
    assert a.keys() == b.keys() == c.keys() and len(a) >= 330
    differ = [n for n in a if not torch.equal(a[n], b[n])]
    assert not differ, differ[:5]
    # (the bias tensors in front of a training-mode BatchNorm have an analytically zero gradient: rounding noise on both sides — left
    # out by the reference's own norm, as in test_full_backward_matches_the_reference_autograd)
    live = [n for n in a if float(z[f"grad3norm/{n}"]) > 1e-6]
    assert len(live) >= 300
    worst = max(float((a[n] - c[n]).abs().max()) / float(c[n].abs().max()) for n in live)
    print(f"deterministic option: {len(a)} gradient tensors bit-equal over two runs; vs fp32-atomic gradients max|diff| / max|grad| = {worst:.2e}")
    assert worst <= 5e-4


@gpu
def test_full_training_steps_lower_the_total_loss(golden_dir):
    """run_train.py:109-130 with the default scope ("full"): forward_train (the reference run's noisy affines every step) -> Loss ->
    backward -> allreduce_gradients -> SGD over everything that received a gradient, four times on one batch: the total loss falls
    monotonically, every parameter the reference trains moves (the bias tensors with an analytically zero gradient move by their
    rounding noise or not at all), the untrained ones (dinov2.norm, mask_token, the DPT head's dead layers) stay bit-identical."""
    from picopose_amd.dist import allreduce_gradients
    from picopose_amd.picopose import Net
    from picopose_amd.utils.loss_utils import Loss

    z, ep, weights = _load_grad_fixture(golden_dir)
    net = Net(small_cfg())
    net.load_state_dict(weights(net.state_dict()))
    net = net.cuda().train()
    before = {k: v.detach().clone() for k, v in net.named_parameters()}
    ep, pred_Ms = _cuda(ep), torch.from_numpy(z["pred_Ms"]).cuda()
    totals, opt = [], None
    for step in range(4):
        total = Loss()(net.forward_train(dict(ep), pred_Ms=pred_Ms))["loss"]
        totals.append(float(total.detach()))
        total.backward()
        trained = [p for p in net.parameters() if p.grad is not None]
        if opt is None:
            opt = torch.optim.SGD(trained, lr=1e-5)
        allreduce_gradients(trained)
        opt.step()
        opt.zero_grad(set_to_none=True)
    print("total loss over four full SGD steps:", [round(v, 5) for v in totals])
    assert abs(totals[0] - float(z["total_loss"])) <= 1e-3 * float(z["total_loss"])
    assert all(b < a for a, b in zip(totals, totals[1:])), totals
    moved = 0
    for name, p in net.named_parameters():
        same = torch.equal(p.detach(), before[name])
        if not bool(z[f"grad3used/{name}"]):
            assert same, name
        else:
            moved += int(not same)
    assert moved >= 280, moved     # (of 338: a step of 1e-5 x a gradient of 1e-6 is below the last bit of some weights)


@gpu
def test_sgd_steps_on_the_slice_lower_its_losses(golden_dir):
    """The optimiser loop of run_train.py:109-130 on the slice: forward_train -> Loss -> backward -> allreduce_gradients (a no-op
    at world size 1, called as a trainer would) -> SGD step over the parameters that received a gradient, five times on one batch.
    Each of the slice's own losses (InfoNCE and the three stage-2 terms) must fall monotonically, every
    updated parameter must move, every other parameter must stay bit-identical, and the weight re-pack must follow the in-place
    updates (the second forward sees the new weights: ADVICE r02 on the pack cache)."""
    from picopose_amd.dist import allreduce_gradients
    from picopose_amd.picopose import Net
    from picopose_amd.utils.loss_utils import Loss

    z, ep, weights = _load_grad_fixture(golden_dir)
    net = Net(small_cfg())
    net.load_state_dict(weights(net.state_dict()))
    net = net.cuda().train()
    net.train_backward = "vit+stage2"
    before = {k: v.detach().clone() for k, v in net.named_parameters()}
    ep = _cuda(ep)
    keys = ("loss_info", "loss_2d_trans", "loss_scale", "loss_inplane")
    rows, opt = [], None
    for step in range(5):
        res = net(dict(ep))
        total = Loss()(res)
        rows.append([float(res[k].detach()) for k in keys])
        total["loss"].backward()
        trained = [p for p in net.parameters() if p.grad is not None]
        if opt is None:
            opt = torch.optim.SGD(trained, lr=3e-5)           # (tools/sgd_sweep.py: the L1 translation loss oscillates from 1e-4 up)
        allreduce_gradients(trained)
        opt.step()
        opt.zero_grad(set_to_none=True)
    print("slice losses over five SGD steps:", [[round(v, 5) for v in r] for r in rows])
    for j, k in enumerate(keys):
        col = [r[j] for r in rows]
        assert all(b <= a for a, b in zip(col, col[1:])) and col[-1] < col[0], (k, col)
    for name, p in net.named_parameters():
        in_slice = name.startswith("affine_regressor.") or (name.startswith("feature_extractor.dinov2.") and ".norm." not in name
                                                            and "dinov2.norm" not in name and "mask_token" not in name)
        same = torch.equal(p.detach(), before[name])
        assert same != in_slice, (name, in_slice, same)


@gpu
def test_backward_kernels_against_torch_autograd():
    """The row-wise adjoints of csrc/pp_backward.hip one by one against torch's autograd on CPU (fp32)."""
    from picopose_amd import autograd as ag

    g = torch.Generator().manual_seed(3)

    def check(fn_hip, fn_ref, *shapes, tol=2e-4):
        xs = [torch.randn(*s, generator=g) for s in shapes]
        a = [x.clone().cuda().requires_grad_(True) for x in xs]
        b = [x.clone().requires_grad_(True) for x in xs]
        ya, yb = fn_hip(*a), fn_ref(*b)
        w = torch.randn(*yb.shape, generator=g)
        (ya * w.cuda()).sum().backward()
        (yb * w).sum().backward()
        assert float((ya.detach().cpu() - yb.detach()).abs().max()) <= tol * max(1.0, float(yb.abs().max()))
        for u, v in zip(a, b):
            assert float((u.grad.cpu() - v.grad).abs().max()) <= tol * max(1.0, float(v.grad.abs().max())), (fn_ref, u.shape)

    for act in (None, "relu", "gelu", "leaky01", "tanh"):
        f = {None: lambda t: t, "relu": F.relu, "gelu": F.gelu, "leaky01": lambda t: F.leaky_relu(t, 0.1), "tanh": torch.tanh}[act]
        check(lambda x, w, b: ag.linear(x, w, b, act), lambda x, w, b: f(F.linear(x, w, b)), (70, 96), (40, 96), (40,))
    check(lambda x, w, b: ag.layernorm(x, w, b, 1e-6), lambda x, w, b: F.layer_norm(x, (384,), w, b, 1e-6), (50, 384), (384,), (384,))
    check(lambda t, gm, r: ag._ScaleResidual.apply(t, gm, r), lambda t, gm, r: r + gm * t, (33, 64), (64,), (33, 64))
    check(lambda x, w, b: ag._GroupNormRelu.apply(x, w, b, 32, True),
          lambda x, w, b: F.relu(F.group_norm(x.permute(0, 3, 1, 2), 32, w, b, 1e-5)).permute(0, 2, 3, 1), (2, 8, 8, 256), (256,), (256,))
    check(lambda x: ag._NormalizeRows.apply(x, 1e-12), lambda x: F.normalize(x, dim=1), (9, 2))
    B, T, heads, hd = 2, 37, 3, 64

    def ref_attn(qkv):
        q, k, v = qkv.view(B, T, 3, heads, hd).permute(2, 0, 3, 1, 4)
        return F.scaled_dot_product_attention(q, k, v).permute(0, 2, 1, 3).reshape(B * T, heads * hd)

    check(lambda qkv: ag._Attention.apply(qkv, B, T, heads, hd), ref_attn, (B * T, 3 * heads * hd))
    check(lambda x, w: ag.linear(ag._Im2col.apply(x, 3, 2, 1), w.permute(0, 2, 3, 1).reshape(w.shape[0], -1)).view(2, 4, 4, 24),
          lambda x, w: F.conv2d(x.permute(0, 3, 1, 2), w, None, stride=2, padding=1).permute(0, 2, 3, 1), (2, 8, 8, 16), (24, 16, 3, 3))


@gpu
@pytest.mark.parametrize("B,T,heads,qk_gain,dout_gain", [(2, 257, 12, 1.0, 1.0), (3, 37, 3, 1.0, 1e-7), (1, 1, 2, 1.0, 1.0), (2, 64, 6, 4.0, 30.0),
                                                         (2, 33, 1, 0.05, 1.0), (1, 300, 2, 2.5, 1e-3)])
def test_fused_attention_forward_and_adjoint_against_float64(B, T, heads, qk_gain, dout_gain):
    """pp_attention_train / pp_attention_backward (csrc/pp_attn_bwd.hip: scores and probabilities recomputed, nothing T x T stored) against
    torch's float64 autograd of layers/attention.py:49-62: ragged T (one query tile with one row at 257), peaked soft-maxes (qk_gain 4:
    rows with one probability ~1 and the rest down to 1e-30 — the per-tile ranges of dS), tiny and large output gradients.
    Stated tolerance 1e-5 of each tensor's max (the f16x3 products' 2^-22 over T-long sums); two runs are bit-equal (no atomics)."""
    from picopose_amd import autograd as ag
    from picopose_amd import ops

    hd = 64
    assert ag.FUSED_ATTENTION and ops.PRECISION == "f16x3"
    g = torch.Generator().manual_seed(100 + T)
    qkv = torch.randn(B * T, 3 * heads * hd, generator=g)
    qkv[:, :2 * heads * hd] *= qk_gain
    w = torch.randn(B * T, heads * hd, generator=g) * dout_gain
    w[::3] *= 1e-3                                              # rows of very different gradient size

    def ref(x):
        q, k, v = x.view(B, T, 3, heads, hd).permute(2, 0, 3, 1, 4)
        return (torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5, dim=-1) @ v).permute(0, 2, 1, 3).reshape(B * T, heads * hd)

    xr = qkv.double().requires_grad_(True)
    yr = ref(xr)
    (yr * w.double()).sum().backward()
    runs = []
    for _ in range(2):
        x = qkv.clone().cuda().requires_grad_(True)
        y = ag._Attention.apply(x, B, T, heads, hd)
        (y * w.cuda()).sum().backward()
        runs.append((y.detach().cpu(), x.grad.cpu()))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    y, dx = runs[0]
    assert torch.isfinite(dx).all()
    ey = float((y.double() - yr.detach()).abs().max() / yr.detach().abs().max())
    print(f"T={T}: out {ey:.2e}")
    assert ey <= 1e-5
    C = heads * hd
    for i, nm in enumerate("qkv"):
        a, b = dx[:, i * C:(i + 1) * C].double(), xr.grad[:, i * C:(i + 1) * C]
        den = b.abs().max() if float(b.abs().max()) > 0 else xr.grad.abs().max()      # (T = 1: dq = dk = 0 exactly)
        e = float((a - b).abs().max() / den)
        print(f"  d{nm} {e:.2e} of max {float(b.abs().max()):.3e}")
        assert e <= 1e-5, (nm, e)


@gpu
@pytest.mark.parametrize("R,M,N", [(8224, 768, 768), (8224, 3072, 768), (32768, 256, 2304), (16384, 192, 2304), (4104, 256, 256), (8224, 768, 200), (8224, 192, 200)])
def test_weight_gradient_products_with_k_slices_against_float64(R, M, N):
    """dW = dz^T x (autograd._mm_tn) when the output has few tiles and K = the rows of the batch is long: the K slices run as extra tile rows
    of ONE engine launch (PpGemmDesc.ksplit, K padded with zeros to a multiple of the slices) and are added in index order.  Against float64
    (1e-5 of the result's max: 2^-22 products over K-long sums), equal to the unsliced launch within fp32 summation order, and the same
    bits on every run.  (16384, 192, 2304): M is not a multiple of 256, the product is taken transposed; (8224, 192, 200): no slices possible."""
    from picopose_amd import autograd as ag
    from picopose_amd import ops

    g = torch.Generator().manual_seed(R + M)
    a = (torch.randn(R, M, generator=g) * 1e-4).cuda()            # a gradient: range-normalised on the way in
    b = torch.randn(R, N, generator=g).cuda()
    swap = M % 256 != 0 and N % 256 == 0
    S, kp = ops.ksplit_choice(N, M, R) if swap else ops.ksplit_choice(M, N, R)
    print(f"R={R} M={M} N={N}: S={S} K padded to {kp}")
    assert (S > 1) == (M % 256 == 0 or N % 256 == 0)
    want = a.double().t() @ b.double()
    got = ag._mm_tn(a, b, rb=False)
    assert torch.equal(got, ag._mm_tn(a, b, rb=False))
    e = float((got.double() - want).abs().max() / want.abs().max())
    print(f"  vs float64 {e:.2e}")
    assert e <= 1e-5
    try:
        ops.KSPLIT = False
        plain = ag._mm_tn(a, b, rb=False)
    finally:
        ops.KSPLIT = True
    assert float((got - plain).abs().max() / want.abs().max()) <= 5e-6


@gpu
@pytest.mark.parametrize("B,H,W,C,k", [(2, 16, 16, 64, 3), (3, 8, 24, 40, 3), (1, 16, 8, 128, 7), (2, 8, 8, 36, 1)])
def test_k_major_im2col_operand_equals_the_split_of_the_unfolded_map(B, H, W, C, k):
    """pp_im2col_t_operand (the B operand of a convolution's weight gradient: rows (tap, channel), K = the pixels of the batch, written
    in the engine's operand format) against torch's unfold of the same map, transposed and split by pp_split_transpose_t: bit-equal."""
    from picopose_amd import _lib, ops

    g = torch.Generator().manual_seed(B * H + C)
    x = torch.randn(B, H, W, C, generator=g).cuda()
    rows = B * H * W
    got = ops.Split.empty(k * k * C, rows, x.device)
    _lib.check(_lib.lib().pp_im2col_t_operand(x.data_ptr(), B, H, W, C, k, 1, k // 2, got.hl.data_ptr(), got.terms, _lib.stream_ptr()), "pp_im2col_t_operand")
    col = F.unfold(x.permute(0, 3, 1, 2), k, padding=k // 2)                    # (B, C k k, H W), channel-major rows
    col = col.view(B, C, k * k, H * W).permute(0, 3, 2, 1).reshape(rows, k * k * C).contiguous()   # (pixels, (tap, channel))
    want = ops.split_transposed(col)
    assert torch.equal(got.hl, want.hl)


@gpu
def test_stage3_adjoint_kernels_against_torch_autograd():
    """The adjoints of csrc/pp_backward3.hip one by one against torch's autograd on CPU (fp32): BatchNorm in training mode (+ReLU),
    bilinear resize (align_corners), ConvTranspose(kernel = stride), the feature warp, the fused correlation pyramid + lookup (against
    the oracle's materialised pyramid + grid_sample), the flow / certainty losses.  Scatter kernels (warp, lookup) use fp32 atomics:
    5e-4 of the largest gradient; the rest 2e-4."""
    from oracle import nets as onets
    from oracle import train as otrain
    from picopose_amd import autograd as ag

    g = torch.Generator().manual_seed(5)

    def check(fn_hip, fn_ref, tensors, tol=2e-4, nout=1):
        a = [x.clone().cuda().requires_grad_(True) for x in tensors]
        b = [x.clone().requires_grad_(True) for x in tensors]
        ya, yb = fn_hip(*a), fn_ref(*b)
        ya, yb = (ya, yb) if nout > 1 else ((ya,), (yb,))
        la = lb = 0.0
        for u, v in zip(ya, yb):
            assert float((u.detach().cpu() - v.detach()).abs().max()) <= tol * max(1.0, float(v.abs().max())), fn_ref
            w = torch.randn(tuple(v.shape), generator=g)
            la, lb = la + (u * w.cuda()).sum(), lb + (v * w).sum()
        la.backward()
        lb.backward()
        for u, v in zip(a, b):
            err, scale = float((u.grad.cpu() - v.grad).abs().max()), max(1e-3, float(v.grad.abs().max()))
            assert err <= tol * scale, (fn_ref, tuple(u.shape), err, scale)

    rn = lambda *s: torch.randn(*s, generator=g)   # noqa: E731
    # BatchNorm (training) with and without the ReLU
    for relu in (False, True):
        bn = torch.nn.BatchNorm2d(16)
        hb = type("BN", (), {})()
        hb.weight, hb.bias = None, None
        hb.running_mean, hb.running_var = torch.zeros(16).cuda(), torch.ones(16).cuda()
        hb.num_batches_tracked = torch.zeros((), dtype=torch.long).cuda()

        def hip_bn(x, w, b, relu=relu, hb=hb):
            hb.weight, hb.bias = w, b
            return ag._BatchNormTrain.apply(x, w, b, hb, relu)

        def ref_bn(x, w, b, relu=relu):
            y = F.batch_norm(x.permute(0, 3, 1, 2), None, None, w, b, True, 0.1, 1e-5).permute(0, 2, 3, 1)
            return F.relu(y) if relu else y

        check(hip_bn, ref_bn, [rn(2, 9, 7, 16) * 2 + 0.5, rn(16), rn(16)])
    # bilinear resize, align_corners = True
    for (H, W, Ho, Wo, mul) in ((8, 8, 16, 16, 2.0), (5, 7, 9, 11, 1.0), (16, 16, 32, 32, 1.0)):
        check(lambda x, Ho=Ho, Wo=Wo, mul=mul: ag.resize(x, Ho, Wo, mul),
              lambda x, Ho=Ho, Wo=Wo, mul=mul: mul * F.interpolate(x.permute(0, 3, 1, 2), size=(Ho, Wo), mode="bilinear", align_corners=True).permute(0, 2, 3, 1),
              [rn(2, H, W, 8)])
    # ConvTranspose2d(kernel = stride)
    for r in (2, 4):
        check(lambda x, w, b, r=r: ag._ConvTranspose.apply(x, w, b, r),
              lambda x, w, b, r=r: F.conv_transpose2d(x.permute(0, 3, 1, 2), w, b, stride=r).permute(0, 2, 3, 1), [rn(2, 4, 4, 16), rn(16, 24, r, r), rn(24)])
    # feature warp (flows reach outside the image)
    B, H, W, C = 2, 8, 8, 16

    def ref_warp(feat, flow):
        coords = (onets._pixel_grid(B, H, W) + flow.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
        return onets._sample(feat.permute(0, 3, 1, 2), coords).permute(0, 2, 3, 1)

    check(lambda feat, flow: ag._Warp.apply(feat, flow), ref_warp, [rn(B, H, W, C), rn(B, H, W, 2) * 2.5], tol=5e-4)
    # correlation pyramid + lookup
    for levels, r in ((1, 2), (3, 2)):
        Hc = 8
        ncorr = levels * (2 * r + 1) ** 2
        cp = -(-ncorr // 8) * 8

        def ref_lookup(f1, f2, flow, levels=levels, r=r):
            return onets.corr_lookup(f1.permute(0, 3, 1, 2), f2.permute(0, 3, 1, 2), flow.permute(0, 3, 1, 2), levels, r).permute(0, 2, 3, 1)

        check(lambda f1, f2, flow, levels=levels, r=r, cp=cp, ncorr=ncorr: ag._CorrLookup.apply(f1, f2, flow, levels, r, cp)[..., :ncorr],
              ref_lookup, [rn(2, Hc, Hc, 32), rn(2, Hc, Hc, 32), rn(2, Hc, Hc, 2) * 1.7], tol=5e-4)
    # flow / certainty losses of one level (64 x 64 key-point grid -> 16 x 16 maps)
    tp = torch.rand(2, 4096, 2, generator=g) * 60
    tp[torch.rand(2, 4096, generator=g) < 0.4] = -1.0

    def ref_loss(flow, cert):
        (lf, lc), = otrain.stage_three_loss([flow.permute(0, 3, 1, 2)], [cert.permute(0, 3, 1, 2)], tp)
        return lf, lc

    check(lambda flow, cert: ag.flow_level_losses(flow, cert, tp.cuda()), ref_loss, [rn(2, 16, 16, 2) * 3, rn(2, 16, 16, 1)], nout=2)


@gpu
@pytest.mark.parametrize("H,C,flow_sigma", [(16, 128, 0.3), (16, 256, 1.5), (8, 128, 6.0), (12, 128, 1.0)])
def test_correlation_lookup_adjoint_by_patches_against_float64_and_the_per_pixel_kernel(H, C, flow_sigma, monkeypatch):
    """The df2 half of the correlation-lookup adjoint scattered by 4 x 4 pixel patches (corr_lookup_scatter_kernel: taken when H, W are
    multiples of 4 and C of 128 — the flow decoder's maps) against torch's float64 autograd of the oracle's materialised pyramid +
    grid_sample, and against the per-pixel kernel it replaces (PP_CORR_SCATTER_PER_PIXEL=1).  Smooth flows (every position inside the
    patch's frame), rough ones (sigma 1.5: some outside — the direct-atomic fallback) and flows that tear the patch apart and leave the
    map (sigma 6).  1e-5 of the largest gradient (measured 1.3e-6: fp32 sums); both kernels agree to 2e-6 (measured 2.8e-7).  With the
    deterministic option the patch kernel repeats bit for bit."""
    from oracle import nets as onets
    from picopose_amd import autograd as ag

    levels, r = 3, 2
    ncorr = levels * (2 * r + 1) ** 2
    cp = -(-ncorr // 8) * 8
    g = torch.Generator().manual_seed(H + C)
    f1, f2 = torch.randn(2, H, H, C, generator=g), torch.randn(2, H, H, C, generator=g)
    flow = torch.randn(2, H, H, 2, generator=g) * flow_sigma + torch.tensor([0.7, -1.2])
    w = torch.randn(2, H, H, ncorr, generator=g)

    def run_hip():
        a = [t.clone().cuda().requires_grad_(True) for t in (f1, f2, flow)]
        y = ag._CorrLookup.apply(*a, levels, r, cp)[..., :ncorr]
        (y * w.cuda()).sum().backward()
        return [t.grad.cpu() for t in a]

    b = [t.double().requires_grad_(True) for t in (f1, f2, flow)]
    yb = onets.corr_lookup(b[0].permute(0, 3, 1, 2), b[1].permute(0, 3, 1, 2), b[2].permute(0, 3, 1, 2), levels, r).permute(0, 2, 3, 1)
    (yb * w.double()).sum().backward()
    got = run_hip()
    monkeypatch.setenv("PP_CORR_SCATTER_PER_PIXEL", "1")
    old = run_hip()
    monkeypatch.delenv("PP_CORR_SCATTER_PER_PIXEL")
    for name, u, o, v in zip(("df1", "df2", "dflow"), got, old, b):
        scale = float(v.grad.abs().max())
        e, d = float((u.double() - v.grad).abs().max()) / scale, float((u - o).abs().max()) / scale
        print(f"H={H} C={C} sigma={flow_sigma}: {name} vs float64 {e:.1e}, vs per-pixel kernel {d:.1e}")
        assert e <= 1e-5 and d <= 2e-6, (name, e, d)
    monkeypatch.setattr(ag, "DETERMINISTIC", True)
    d1, d2 = run_hip(), run_hip()
    assert all(torch.equal(x, y) for x, y in zip(d1, d2))
    assert float((d1[1].double() - b[1].grad).abs().max()) <= 1e-5 * float(b[1].grad.abs().max())


@gpu
@pytest.mark.parametrize("B,H,Cin,Cout,k", [(2, 32, 64, 256, 3), (2, 32, 256, 192, 3), (1, 64, 32, 126, 3), (2, 16, 16, 512, 7)])
def test_convolution_backward_with_k_slices_against_float64(B, H, Cin, Cout, k):
    """_Conv2d's three directions at sizes whose weight gradient takes the K slices (Cout a multiple of 256), the transposed product
    (Cout = 192, k k Cin a multiple of 256), the 126-channel layer padded to 128 in its backward, and a 7 x 7 kernel — against torch's
    float64 autograd of F.conv2d: 1e-5 of each gradient's maximum (measured 3.5e-6 at worst)."""
    from picopose_amd import autograd as ag

    g = torch.Generator().manual_seed(Cin + Cout)
    x, wt, bias = torch.randn(B, H, H, Cin, generator=g), torch.randn(Cout, Cin, k, k, generator=g) * (Cin * k * k) ** -0.5, torch.randn(Cout, generator=g)
    up = torch.randn(B, H, H, Cout, generator=g) * 1e-3
    a = [t.clone().cuda().requires_grad_(True) for t in (x, wt, bias)]
    ya = ag._Conv2d.apply(a[0], a[1], a[2], k, None)
    (ya * up.cuda()).sum().backward()
    b = [t.double().requires_grad_(True) for t in (x, wt, bias)]
    yb = F.conv2d(b[0].permute(0, 3, 1, 2), b[1], b[2], padding=k // 2).permute(0, 2, 3, 1)
    (yb * up.double()).sum().backward()
    assert float((ya.detach().cpu().double() - yb.detach()).abs().max()) <= 1e-5 * float(yb.detach().abs().max())
    for name, u, v in zip(("dx", "dw", "db"), a, b):
        e = float((u.grad.cpu().double() - v.grad).abs().max() / v.grad.abs().max())
        print(f"Cin={Cin} Cout={Cout} k={k}: {name} {e:.1e}")
        assert e <= 1e-5, (name, e)


@gpu
@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_full_backward_matches_the_reference_autograd(golden_dir, precision):
    """Scope "full" (the default): `Loss()(net(end_points))["loss"].backward()` is the reference's training step.  With the noisy
    affines the reference run drew, all ten losses equal the reference's and the `.grad` of EVERY parameter the reference trains
    (ViT, affine regressor, DPT head, flow decoder: fixture keys grad3/..., torch.autograd.grad of the reference's total loss)
    agrees: max|err| / max|grad| per tensor and the relative error of its L2 norm (bars below; measured: profiles/r03/
    backward_slice.txt).  Parameters the reference leaves without a gradient (dinov2.norm, mask_token, the DPT head's dead
    output convolutions / refinenet1) keep grad None here too."""
    from picopose_amd import ops
    from picopose_amd.picopose import Net
    from picopose_amd.utils.loss_utils import Loss

    z, ep, weights = _load_grad_fixture(golden_dir)
    old = ops.PRECISION
    ops.PRECISION = precision
    try:
        net = Net(small_cfg())
        net.load_state_dict(weights(net.state_dict()))
        net = net.cuda().train()
        assert net.train_backward is True
        res = net.forward_train(_cuda(ep), pred_Ms=torch.from_numpy(z["pred_Ms"]).cuda())
        for k in LOSS_KEYS:
            assert abs(float(res[k].detach()) - float(z[k])) <= 1e-3 * max(1.0, abs(float(z[k]))), (k, float(res[k]), float(z[k]))
            assert res[k].requires_grad, k
        tot = Loss()(res)["loss"]
        assert abs(float(tot.detach()) - float(z["total_loss"])) <= 1e-3 * float(z["total_loss"])
        tot.backward()
    finally:
        ops.PRECISION = old
    # Bars: max|err| / max|grad| <= 3e-3 per tensor against the reference's fp32 autograd, the tensor's L2 norm within 3e-4.
    # The DPT head's FUSION BLOCKS (scratch.refinenet*: convolutions between batch-statistics BatchNorms on 2 x 16 x 16 .. 64 x 64 samples,
    # gradients of 3e-7 .. 7e-5 against the model's largest 3.6) carried a restated 1e-2 bar in round 4 (3.9e-3 measured with the fused
    # attention forward).  Round 5 settles them against a FLOAT64 evaluation of the reference (train_grads_f64.npz, oracle/gen_golden.py
    # gen_train_grads_f64): every tensor's distance to float64 is reported beside the fp32 reference's own, and the fusion blocks are
    # held to F64_BAR of their maximum against float64 — the arbiter — instead of to a looser bar against another fp32 evaluation.
    z64 = np.load(os.path.join(golden_dir, "train_grads_f64.npz"))
    TOL, F64_BAR = 3e-3, 5e-3
    report, n_checked, n_zero = [], 0, 0
    gmax = max(float(z[k]) for k in z.files if k.startswith("grad3norm/"))
    for name, p in net.named_parameters():
        if not bool(z[f"grad3used/{name}"]):
            assert p.grad is None, name
            continue
        assert p.grad is not None, name
        ref = torch.from_numpy(z[f"grad3/{name}"])
        flat = p.grad.detach().reshape(-1).cpu()
        stride = max(1, -(-flat.numel() // 2048))
        got = flat[::stride]
        assert got.shape == ref.shape, (name, got.shape, ref.shape)
        scale = max(float(ref.abs().max()), 1e-30)
        nref = float(z[f"grad3norm/{name}"])
        # analytically ZERO gradients: a per-channel constant in front of a training-mode BatchNorm with only linear layers in
        # between (conv biases before their BatchNorm; bn2.bias / out_conv.bias of the last fusion block: resize and 1x1
        # convolutions keep constants, the flow decoder's projection BatchNorm removes them).  The reference holds rounding noise
        # there (1e-4 of the sibling weight's gradient or less); so must this build, and a relative comparison means nothing.
        sib = name[:-4] + "weight" if name.endswith(".bias") else None
        if sib is not None and f"grad3norm/{sib}" in z.files and nref < 1e-4 * float(z[f"grad3norm/{sib}"]):
            bar = max(1e-3 * float(z[f"grad3norm/{sib}"]), 1e-6 * gmax)     # (gmax: the largest gradient norm of the model)
            assert float(flat.double().norm()) < bar, (name, float(flat.double().norm()), bar)
            n_zero += 1
            continue
        ref64 = torch.from_numpy(z64[f"grad3f64/{name}"])
        d_hip64 = float((got.double() - ref64).abs().max()) / scale
        d_ref64 = float((ref.double() - ref64).abs().max()) / scale
        report.append((float((got - ref).abs().max()) / scale, abs(float(flat.double().norm()) - nref) / max(nref, 1e-30), name, ".scratch.refinenet" in name,
                       d_hip64, d_ref64))
        n_checked += 1
    report.sort(reverse=True)
    print(f"({n_zero} bias tensors with an analytically zero gradient: noise below 1e-3 of their weight's gradient / 1e-6 of the largest gradient on both sides)")
    short = lambda n: n.replace("feature_extractor.dinov2.", "vit.").replace("offset_regressor.", "")   # noqa: E731
    print(f"full backward [{precision}]: {n_checked} parameter tensors, worst max|err| / max|grad| = {report[0][0]:.2e}, worst norm error "
          f"{max(r[1] for r in report):.2e}; worst eight:", [(f"{r[0]:.1e}", f"{r[1]:.1e}", short(r[2]), "fusion block" if r[3] else "") for r in report[:8]])
    by64 = sorted(report, key=lambda r: -r[4])
    print(f"against float64 [{precision}]: worst |HIP - f64| / max|grad| = {by64[0][4]:.2e} (the fp32 reference's own worst: {max(r[5] for r in report):.2e}); "
          "worst eight (HIP-f64, ref32-f64, tensor):", [(f"{r[4]:.1e}", f"{r[5]:.1e}", short(r[2])) for r in by64[:8]])
    assert n_checked + n_zero == 338 and n_zero <= 40
    assert all(r[0] <= TOL for r in report if not r[3]), [r for r in report if r[0] > TOL and not r[3]][:3]
    assert all(r[4] <= F64_BAR for r in report), [r for r in by64 if r[4] > F64_BAR][:3]       # every tensor, the fusion blocks included
    assert max(r[1] for r in report) <= 3e-4     # (the L2 norms of the gradient tensors)


class _PinKinks(torch.autograd.Function):
    """TEST-ONLY: y = x with the listed elements (NHWC-flat indices) overwritten by the float64 forward's values; the gradient passes through.
    It puts every pre-activation float64 holds within 5e-4 of a ReLU's kink on float64's SIDE of it (the values move by < 1e-3 of the map's
    rms): the gradient of a ReLU network is discontinuous exactly there, so no bar against float64 tighter than one element's own
    gradient can be asked of ANY fp32 forward without it."""

    @staticmethod
    def forward(ctx, x, idx, val):
        y = x.clone()
        y.view(-1)[idx] = val
        return y

    @staticmethod
    def backward(ctx, dy):
        return dy, None, None


@gpu
@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_fusion_block_gradients_against_float64_with_the_relu_kinks_pinned(golden_dir, precision, monkeypatch):
    """What limits `test_full_backward_matches_the_reference_autograd` to 5e-3 against float64 on the DPT fusion blocks, isolated
    (profiles/r05/grad_f64.txt, tools/grad_nodes_hip.py): ONE pre-activation of 524 288 in a block — float64 holds it at 1.6e-5 on a map of
    rms 8.9, this build's fp32 forward 1e-5 on the other side of zero (the reference's own fp32 forward flips another one) — whose gradient
    (1.5e-9) is then wholly different, on parameter gradients whose sums cancel to 3e-7.  Here the same step runs with the pre-activations
    that float64 holds within 5e-4 of a ReLU kink of the fusion blocks (train_grads_f64.npz kink_idx / kink_val, 1 313 of 9.4 M elements)
    pinned to float64's values, and the relation VERDICT r04 asked for is asserted on every fusion-block tensor:
        |HIP - f64| <= 2 |reference-fp32 - f64| + KINK_FLOOR max|grad|,
    KINK_FLOOR = 1e-5 on the fp32 engine (measured: worst tensor 1.6e-5 against the reference's own 1.4e-5 — 4.7e-3 without the pins; the
    floor is used by tensors the reference holds to 1e-6: 6.4e-6 measured) and 3e-4 on the f16x3 engine (22-bit operands: 1.2e-4 measured,
    3.9e-3 without the pins)."""
    from picopose_amd import autograd as A
    from picopose_amd import ops
    from picopose_amd.picopose import Net
    from picopose_amd.utils.loss_utils import Loss

    z, ep, weights = _load_grad_fixture(golden_dir)
    z64 = np.load(os.path.join(golden_dir, "train_grads_f64.npz"))
    net = Net(small_cfg())
    net.load_state_dict(weights(net.state_dict()))
    net = net.cuda().train()
    mods = {id(m): n for n, m in net.named_modules()}
    calls, pinned, moved = {}, [0], [0]

    def pin(t, site):
        idx = torch.from_numpy(z64[f"kink_idx/{site}"]).cuda()
        val = torch.from_numpy(z64[f"kink_val/{site}"]).float().cuda()
        if idx.numel() == 0:
            return t
        assert idx.max() < t.numel()
        pinned[0] += idx.numel()
        moved[0] += int(((t.detach().reshape(-1)[idx] > 0) != (val > 0)).sum())
        return _PinKinks.apply(t, idx, val)

    def rcu(u, x, extra=None):                    # picopose_amd.autograd._rcu with the two ReLU inputs pinned (and bn1's ReLU un-fused for it)
        name = mods[id(u)]
        c = calls.get(name, 0)
        calls[name] = c + 1
        x = pin(x, f"{name}/{c}")
        h = A.conv2d(A._Act.apply(x, "relu"), u.conv1.weight, u.conv1.bias, 3, pad=1)
        h = A._Act.apply(pin(A.batchnorm_train(h, u.bn1, relu=False), f"{name}.bn1/{c}"), "relu")
        h = A.conv2d(h, u.conv2.weight, u.conv2.bias, 3, pad=1)
        h = A.add(A.batchnorm_train(h, u.bn2), x)
        return h if extra is None else A.add(h, extra)

    monkeypatch.setattr(A, "_rcu", rcu)
    monkeypatch.setattr(ops, "PRECISION", precision)
    res = net.forward_train(_cuda(ep), pred_Ms=torch.from_numpy(z["pred_Ms"]).cuda())
    Loss()(res)["loss"].backward()
    assert pinned[0] == sum(len(z64[k]) for k in z64.files if k.startswith("kink_idx/")) and len(calls) == 5
    KINK_FLOOR, report = {"f32": 1e-5, "f16x3": 3e-4}[precision], []
    for name, p in net.named_parameters():
        if ".scratch.refinenet" not in name or not bool(z[f"grad3used/{name}"]):
            continue
        ref, ref64 = torch.from_numpy(z[f"grad3/{name}"]), torch.from_numpy(z64[f"grad3f64/{name}"])
        sib = name[:-4] + "weight" if name.endswith(".bias") else None
        if sib is not None and float(z[f"grad3norm/{name}"]) < 1e-4 * float(z[f"grad3norm/{sib}"]):
            continue                                                  # (analytically zero gradients: see the test above)
        flat = p.grad.detach().reshape(-1).cpu()
        got = flat[:: max(1, -(-flat.numel() // 2048))]
        scale = float(ref64.abs().max())
        report.append((float((got.double() - ref64).abs().max()) / scale, float((ref.double() - ref64).abs().max()) / scale, name))
    report.sort(reverse=True)
    print(f"fusion blocks, kinks pinned [{precision}]: {pinned[0]} elements pinned, {moved[0]} of them were on the other side of zero; {len(report)} tensors; "
          "worst (HIP-f64, ref32-f64, tensor):", [(f"{r[0]:.1e}", f"{r[1]:.1e}", r[2].replace("offset_regressor.dpt_head.scratch.", "")) for r in report[:6]])
    bad = [r for r in report if r[0] > 2 * r[1] + KINK_FLOOR]
    assert not bad, bad[:4]


@gpu
def test_full_backward_directional_derivative_at_vitb_width(monkeypatch):
    """A size-independent property of the whole backward at the configs[2] width (ViT-B/14, 4 pairs), against an INDEPENDENT
    evaluator: the forward-only training step (fused inference kernels, no autograd).  With g = the gradient of the total loss
    from the autograd graph and d = g / |g|, the central difference (L(theta + eps d) - L(theta - eps d)) / (2 eps) of the
    forward-only loss equals |g| — checked on the fp32 engine (the difference of two losses of ~19 needs their last digits) to 2 %."""
    import types

    from netcfg import make_train_end_points
    from picopose_amd import ops
    from picopose_amd.picopose import Net
    from picopose_amd.utils.loss_utils import Loss
    from picopose_amd.utils.seeding import calibrated_state_dict

    monkeypatch.setattr(ops, "PRECISION", "f32")
    ns = types.SimpleNamespace
    cfg = ns(hypothesis=5, stage1=ns(vit_type="dinov2_vitb14", pretrained=False, interaction_indexes=[[0, 2], [3, 5], [6, 8], [9, 11]]),
             stage2=ns(in_channel=256, hidden_dim=256),
             stage3=ns(nclass=1, in_channels=768, use_bn=True, out_channels=[256, 512, 1024, 1024], num_levels=3, radius=4))
    net = Net(cfg)
    net.load_state_dict(calibrated_state_dict(net.state_dict(), 4, "dinov2_vitb14"))
    net = net.cuda().train()
    ep = {k: v.cuda() for k, v in make_train_end_points(4, 21).items()}
    from picopose_amd.utils.augment import aug_gtM_noise

    np.random.seed(5)
    torch.manual_seed(5)
    pred_Ms = aug_gtM_noise(ep)
    state = {k: v.clone() for k, v in net.state_dict().items()}

    def loss_plain():
        net.load_state_dict(state)                      # (the BatchNorm buffers: every evaluation starts from the same ones)
        with torch.no_grad():
            return float(Loss()(net.forward_train(dict(ep), pred_Ms=pred_Ms))["loss"])

    net.load_state_dict(state)
    total = Loss()(net.forward_train(dict(ep), pred_Ms=pred_Ms))["loss"]
    total.backward()
    params = [(n, p) for n, p in net.named_parameters() if p.grad is not None]
    gnorm = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for _, p in params)))
    l0 = loss_plain()
    assert abs(l0 - float(total.detach())) <= 2e-5 * abs(l0)
    eps = 0.02 / gnorm                                  # a first-order change of 0.02 in a loss of ~20
    base = {n: state[n].clone() for n, _ in params}
    vals = []
    for sign in (+1.0, -1.0):
        for n, p in params:
            state[n] = base[n] + sign * eps * p.grad / gnorm
        vals.append(loss_plain())
    for n, _ in params:
        state[n] = base[n]
    fd = (vals[0] - vals[1]) / (2 * eps)
    print(f"directional derivative at ViT-B width: |grad| = {gnorm:.5f}, central difference of the forward-only loss = {fd:.5f} "
          f"(L = {l0:.5f}, L+ = {vals[0]:.5f}, L- = {vals[1]:.5f})")
    assert abs(fd - gnorm) <= 2e-2 * gnorm, (fd, gnorm)
