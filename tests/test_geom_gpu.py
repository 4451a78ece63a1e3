"""GPU parity of the stage-2/3 glue kernels (through the C ABI) against the reference's own
outputs (tests/golden, generated from /root/reference) and against the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import geometry as og
from oracle import matching as om

gpu = pytest.mark.gpu


@pytest.fixture(scope="module")
def geo(golden_dir):
    z = np.load(os.path.join(golden_dir, "geometry.npz"))
    return z, {k: torch.from_numpy(z[k]).cuda() for k in z.files}


@gpu
def test_calc_pred_Ms_and_pose_recovery(geo):
    from picopose_amd.utils.pose_recovery import pose_recovery_2d_prediction
    from picopose_amd.utils.torch_utils import calc_pred_Ms

    z, t = geo
    Ms = calc_pred_Ms(t["scale"], t["inplane"], t["trans"], t["tem_pose"], t["K"], t["tem_M"])
    # entries are pixel-scale (hundreds): 1e-4 absolute is ~3 ulp
    assert np.abs(Ms.cpu().numpy() - z["pred_Ms"]).max() <= 1e-4
    poses = pose_recovery_2d_prediction(t["query_M"], t["K"], t["pred_Ms"], t["K"], t["tem_M"], t["tem_pose"])
    assert np.abs(poses.cpu().numpy() - z["pred_poses"]).max() <= 1e-4  # north_star: pose within 1e-4
    assert np.abs(poses.cpu().numpy() - z["pred_poses"]).max() <= 2e-6  # (in fact a few ulp)


@gpu
def test_contract_check_opt_in(geo, monkeypatch):
    """The reference's inverse_affine asserts a crop affine (torch_utils.py:100-101).  Default here: the caller's contract
    (no host sync); PP_CHECK_CONTRACTS=1 restores the AssertionError."""
    from picopose_amd.utils.pose_recovery import pose_recovery_2d_prediction

    z, t = geo
    bad = t["query_M"].clone()
    bad[0, 0, 1] = 0.5                                           # a shear: not a crop affine
    args = lambda qM: (qM, t["K"], t["pred_Ms"], t["K"], t["tem_M"], t["tem_pose"])  # noqa: E731
    pose_recovery_2d_prediction(*args(bad))                      # default: not checked
    monkeypatch.setenv("PP_CHECK_CONTRACTS", "1")
    pose_recovery_2d_prediction(*args(t["query_M"]))             # a valid crop affine passes
    with pytest.raises(AssertionError):
        pose_recovery_2d_prediction(*args(bad))
    uneven = t["query_M"].clone()
    uneven[1, 1, 1] *= 1.5
    with pytest.raises(AssertionError):
        pose_recovery_2d_prediction(*args(uneven))


@gpu
def test_init_correspondences(geo):
    from picopose_amd.utils.correspondence import compute_init_correspondences

    z, t = geo
    f, c = compute_init_correspondences(t["pred_Ms"], t["mask"])
    assert np.array_equal(c.cpu().numpy(), z["init_cert"])
    assert np.abs(f.cpu().numpy() - z["init_flow"]).max() <= 1e-4
    fi, ci = compute_init_correspondences(torch.eye(3).repeat(2, 1, 1).cuda(), torch.ones(2, 224, 224).cuda())
    assert np.array_equal(fi.cpu().numpy(), z["init_flow_identity"])  # exactly 0.5 everywhere
    with pytest.raises(AssertionError):
        compute_init_correspondences(t["pred_Ms"], torch.ones(6, 224, 200).cuda())


@gpu
def test_stage3_correspondences_bit_exact(geo):
    from picopose_amd.utils.correspondence import compute_stage3_correspondences

    z, t = geo
    tar, src = compute_stage3_correspondences(t["flow"], t["cert"])
    assert tar.dtype == torch.int64 and tar.shape == (6, 4096, 2)
    assert np.array_equal(tar.cpu().numpy(), z["tar_pts"])
    assert np.array_equal(src.cpu().numpy(), z["src_pts"])
    # seeded random inputs against the oracle, incl. a non-square map
    g = torch.Generator().manual_seed(9)
    flow = 4 * torch.randn(3, 2, 32, 32, generator=g)
    cert = torch.randn(3, 1, 32, 32, generator=g)
    rt, rs = og.compute_stage3_correspondences(flow, cert, threshold=0.3)
    gt, gs = compute_stage3_correspondences(flow.cuda(), cert.cuda(), threshold=0.3)
    assert torch.equal(gt.cpu(), rt) and torch.equal(gs.cpu(), rs)


@gpu
def test_gather_valid(geo):
    from picopose_amd.utils.torch_utils import gather

    z, t = geo
    g2 = gather(t["feat2"], t["tar_pts"][:1])
    g3 = gather(t["feat3"], t["src_pts"][:1])
    assert np.array_equal(g2.cpu().numpy(), z["gather2"])  # pure indexing: bit-exact
    assert np.array_equal(g3.cpu().numpy(), z["gather3"])
    assert gather(t["feat3"], t["src_pts"][2:3]).shape == (0, 3)


@gpu
def test_similarity_volume(golden_dir):
    from picopose_amd.utils.matching import matching_features_similarity

    z = np.load(os.path.join(golden_dir, "stage2_similarity.npz"))
    out = matching_features_similarity(torch.from_numpy(z["random/src"]).cuda(), torch.from_numpy(z["random/tar"]).cuda(),
                                       torch.from_numpy(z["random/src_mask"]).cuda(), None)
    assert out.shape == (2, 256, 16, 16)
    assert np.abs(out.cpu().numpy() - z["random/out"]).max() <= 2e-6
    assert np.array_equal(out.cpu().numpy() == 0, z["random/out"] == 0) or \
        np.abs(z["random/out"][(out.cpu().numpy() == 0) != (z["random/out"] == 0)]).max() <= 2e-6
    g = torch.Generator().manual_seed(2)
    for B, C in [(1, 384), (3, 768), (2, 1024)]:
        src, tar = torch.randn(B, C, 16, 16, generator=g), torch.randn(B, C, 16, 16, generator=g)
        m = (torch.rand(B, 224, 224, generator=g) < 0.5).float()
        ref = om.matching_features_similarity(src, tar, m, None)
        got = matching_features_similarity(src.cuda(), tar.cuda(), m.cuda(), None).cpu()
        assert (got - ref).abs().max().item() <= 2e-6


@gpu
def test_gather_rows_equals_indexing():
    from picopose_amd import ops

    g = torch.Generator().manual_seed(3)
    src = torch.randn(6, 5, 3, 8, 4, generator=g).cuda()          # (B, N, ...) -> rows of 96 floats
    idx = torch.tensor([29, 0, 7, 7, 13], device="cuda")
    assert torch.equal(ops.gather_rows(src.flatten(0, 1), idx), src.flatten(0, 1)[idx])
    small = torch.randn(30, 3, 3, generator=g).cuda()              # 9 floats per row: torch indexing path
    assert torch.equal(ops.gather_rows(small, idx), small[idx])
