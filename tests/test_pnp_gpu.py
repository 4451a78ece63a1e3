"""Known-answer tests of the GPU PnP/RANSAC (utils/pose_recovery.py:68-105).  cv2 is not available, so
parity with OpenCV is unpinned: correctness is defined on synthetic correspondences."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
gpu = pytest.mark.gpu


def _problem(rng, n_pts, n_out=0, noise=0.0):
    """One problem in the function's input layout (tests/pnp_problems.py), as a dict of per-problem arrays."""
    from pnp_problems import make_batch

    b = make_batch(rng, 1, n_pts, (n_out / n_pts) if n_pts else 0.0, noise)
    return {k: (v[0] if isinstance(v, np.ndarray) else v) for k, v in b.items()}


def _run(problems):
    from picopose_amd.utils.pose_recovery import pose_recovery_ransac_pnp_batched

    st = lambda k, dt=None: torch.from_numpy(np.stack([p[k] for p in problems])).cuda()  # noqa: E731
    return pose_recovery_ransac_pnp_batched(st("tar2d"), st("src3d"), st("K"), st("pose"), st("tar_pts"), st("src_pts"))


@gpu
def test_noise_free_recovers_ground_truth_pose():
    rng = np.random.default_rng(0)
    probs = [_problem(rng, n) for n in (6, 20, 200, 1500, 4096)]
    rot, tvec, ratio, ok = _run(probs)
    for i, p in enumerate(probs):
        assert ok[i] and rot.dtype == np.float64 and tvec.shape[1:] == (3, 1)
        assert np.abs(rot[i] - p["R"]).max() < 1e-4, (i, np.abs(rot[i] - p["R"]).max())   # float32 inputs
        assert np.abs(tvec[i, :, 0] - p["t"]).max() < 1e-4
        assert abs(np.linalg.det(rot[i]) - 1.0) < 1e-9 and np.abs(rot[i] @ rot[i].T - np.eye(3)).max() < 1e-9
        assert ratio[i] == 1.0


@gpu
def test_planted_outliers_are_rejected():
    rng = np.random.default_rng(1)
    probs = [_problem(rng, 400, n_out=120), _problem(rng, 2000, n_out=900), _problem(rng, 60, n_out=20, noise=0.2)]
    rot, tvec, ratio, ok = _run(probs)
    for i, p in enumerate(probs):
        n = 400 if i == 0 else (2000 if i == 1 else 60)
        assert ok[i]
        # random outliers land within 2 px of their true projection with probability ~1e-4
        assert abs(ratio[i] * n - p["n_in"]) <= (1 if i < 2 else 3), (ratio[i] * n, p["n_in"])
        tol = 1e-4 if i < 2 else 5e-3
        assert np.abs(rot[i] - p["R"]).max() < tol and np.abs(tvec[i, :, 0] - p["t"]).max() < tol


@gpu
def test_failure_outputs_and_single_problem_api():
    from picopose_amd.utils.pose_recovery import pose_recovery_ransac_pnp

    rng = np.random.default_rng(2)
    few = _problem(rng, 3)       # fewer correspondences than a minimal sample -> the reference's except branch
    none = _problem(rng, 0)
    rot, tvec, ratio, ok = _run([few, none])
    for i in range(2):
        assert not ok[i] and ratio[i] == 0.0
        assert np.array_equal(rot[i], np.eye(3)) and np.array_equal(tvec[i], np.array([[0.0], [0.0], [1.0]]))
    p = _problem(rng, 300, n_out=50)
    t = lambda k: torch.from_numpy(p[k]).cuda()  # noqa: E731
    r, tv, ra, success = pose_recovery_ransac_pnp(t("tar2d"), t("src3d"), t("K"), t("pose"), t("tar_pts"), t("src_pts"))
    assert success and isinstance(ra, float) and r.shape == (3, 3) and tv.shape == (3, 1)
    assert np.abs(r - p["R"]).max() < 1e-4 and abs(ra * 300 - p["n_in"]) <= 1


@gpu
def test_hip_kernel_agrees_with_the_cpu_oracle():
    """Same sampling hash, same published algorithm (oracle/pnp.py): the HIP kernel and the numpy restatement
    agree to solver tolerance (cyclic Jacobi vs LAPACK) on clean, noisy and outlier-ridden problems."""
    from oracle import pnp as op

    rng = np.random.default_rng(7)
    probs = [_problem(rng, 300), _problem(rng, 500, n_out=150, noise=0.3), _problem(rng, 64, n_out=16, noise=0.5),
             _problem(rng, 2500, n_out=1200, noise=0.2), _problem(rng, 9), _problem(rng, 4)]
    rot, tvec, ratio, ok = _run(probs)
    for i, p in enumerate(probs):
        r, t, ra, success = op.pose_recovery_ransac_pnp(p["tar2d"], p["src3d"], p["K"], p["pose"], p["tar_pts"],
                                                        p["src_pts"], prob=i)
        assert bool(ok[i]) == success, i
        if not success:
            assert np.array_equal(rot[i], r) and np.array_equal(tvec[i], t) and ratio[i] == ra
            continue
        n = int((p["tar_pts"][:, 0] != -1).sum())
        # stated tolerance: a 5-point sample leaves M^T M with a 2-dimensional null space, whose basis is the eigensolver's
        # choice (one-sided Jacobi here, LAPACK there): some of the 150 hypotheses differ, and on noisy data the winner
        # may be another model of nearly the same consensus
        assert abs(ratio[i] - ra) * n <= max(2, 0.02 * n), (i, ratio[i] * n, ra * n)
        # clean data: solver tolerance.  Noisy data: the two refits run on inlier sets that may differ by a point
        # or two, which moves the pose by about noise / (f * sqrt(n)) * depth — loosest for the 64-point case
        tol = (1e-6, 2e-3, 1e-2, 2e-3, 1e-6)[i]
        assert np.abs(rot[i] - r).max() < tol and np.abs(tvec[i] - t).max() < tol, (i, np.abs(rot[i] - r).max())
        # with the kernel's own eigen-solvers restated in the oracle (solver="kernel") the degenerate null space of a
        # minimal sample is resolved the same way: the same hypotheses, the same consensus (a point or two on a threshold)
        r, t, ra, success = op.pose_recovery_ransac_pnp(p["tar2d"], p["src3d"], p["K"], p["pose"], p["tar_pts"], p["src_pts"], prob=i,
                                                        solver="kernel")
        assert success and abs(ratio[i] - ra) * n <= 2, (i, ratio[i] * n, ra * n)
        assert np.abs(rot[i] - r).max() < tol and np.abs(tvec[i] - t).max() < tol


@gpu
def test_statistical_characterisation_against_ground_truth_and_oracle():
    """cv2.solvePnPRansac cannot be pinned here, so the kernel is characterised statistically: 200 seeded
    problems per regime (noise 0 / 0.3 / 1 px x outliers 0 / 30 / 60 % x n = 8 / 64 / 512 / 4096), success rate and
    error quantiles against the planted pose, the CPU oracle on a sub-sample.  Thresholds: the 1000-problem table of
    profiles/r02/pnp_stats.md with a margin, and what RANSAC theory allows — 150 five-point samples contain an
    all-inlier one with probability 1 - (1 - w^5)^150 = 0.786 at inlier fraction w = 0.4 (1.0 at w >= 0.7)."""
    import pnp_stats          # tests/pnp_stats.py (it runs the CPU oracle beside the kernel: test infrastructure)

    table = pnp_stats.run(problems=200, n_oracle=3, log=lambda s: None)
    assert len(table) == 36
    for r in table:
        n, out, noise, tag = r["n"], r["outliers"], r["noise_px"], (r["n"], r["outliers"], r["noise_px"])
        if r["n_inliers"] < 5:                                   # n = 8 with 60 % outliers: 3 true correspondences
            assert r["returned_success"] <= 0.02, tag            # (5 uniform outliers agreeing within 2 px: never, in practice)
            continue
        if out <= 0.3:
            assert r["returned_success"] >= 0.99, (tag, r)
            assert r["pose_found"] >= (0.93 if n == 8 or noise == 1.0 else 0.98), (tag, r)
            if noise == 0.0:
                assert r["rot_deg_p95"] < 1e-3 and r["trans_rel_p95"] < 1e-5, (tag, r)
                # exact data: kernel and oracle find the same consensus and the same pose
                assert r["oracle_vs_gpu_inlier_count_maxdiff"] == 0 and r["oracle_vs_gpu_rot_maxdiff"] < 1e-5, (tag, r)
            else:
                assert r["rot_deg_p95"] < 4.0 * noise / np.sqrt(r["n_inliers"]) * 3.2 + 0.3 * noise, (tag, r)
        else:                                                    # 60 % outliers: bounded by the sampling probability
            assert 0.55 <= r["returned_success"] <= 1.0, (tag, r)
            if noise <= 0.3:
                assert r["pose_found"] >= 0.55 and r["rot_deg_p50"] < 0.3, (tag, r)
        assert r["oracle_success_agrees"] >= 0.66, (tag, r)


@gpu
def test_refit_branches_agree_with_the_lapack_oracle_branch_by_branch():
    """pp_pnp_ransac_debug: the three candidate poses of the final EPnP refit (beta initialisations with N = 1 / 2 / 3 null-space
    vectors) against the solver-independent oracle's (numpy eigh / svd / lstsq) candidates of the same index, on problems whose
    consensus set both implementations agree on (clean data with planted outliers, mild noise without outliers).  A difference
    between the RETURNED poses of two correct EPnP implementations can be a different choice between candidates of nearly equal
    error; a difference inside one branch would be an arithmetic defect — this test separates the two."""
    from oracle import pnp as op
    from picopose_amd.utils.pose_recovery import refit_branches

    rng = np.random.default_rng(11)
    probs = [_problem(rng, 300), _problem(rng, 800, n_out=240), _problem(rng, 3000, noise=0.3), _problem(rng, 4096, n_out=400, noise=0.0),
             _problem(rng, 1200, noise=0.2), _problem(rng, 3)]
    st = lambda k: torch.from_numpy(np.stack([p[k] for p in probs])).cuda()  # noqa: E731
    rot, tvec, ratio, ok, branches = refit_branches(st("tar2d"), st("src3d"), st("K"), st("pose"), st("tar_pts"), st("src_pts"))
    compared = 0
    for i, p in enumerate(probs):
        r, t, ra, success, obr = op.pose_recovery_ransac_pnp(p["tar2d"], p["src3d"], p["K"], p["pose"], p["tar_pts"], p["src_pts"], prob=i,
                                                             return_branches=True)
        cand, kept = branches[i]
        if not success:
            assert not ok[i] and kept == -1 and obr == [] and all(not np.isfinite(e) for e, _, _ in cand)
            continue
        n = int((p["tar_pts"][:, 0] != -1).sum())
        assert round(ratio[i] * n) == round(ra * n), (i, ratio[i] * n, ra * n)       # same consensus: the refits see the same points
        for a in range(3):
            (ke, kR, kt), (oe, oR, ot) = cand[a], obr[a]
            assert np.isfinite(ke) == np.isfinite(oe), (i, a)
            if np.isfinite(ke):
                assert np.abs(kR - oR).max() <= 1e-6 and np.abs(kt - ot).max() <= 1e-6 and abs(ke - oe) <= 1e-6 * max(1.0, oe), (i, a, kt, ot)
                compared += 1
        errs = [e for e, _, _ in obr]
        if sorted(errs)[1] - min(errs) > 1e-9 * min(errs) + 1e-12:
            assert kept == int(np.argmin(errs)), (i, kept, errs)
        assert np.array_equal(rot[i], cand[kept][1]) and np.array_equal(tvec[i, :, 0], cand[kept][2])
    assert compared >= 12
