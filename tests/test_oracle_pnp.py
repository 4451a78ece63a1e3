"""The PnP/RANSAC oracle (oracle/pnp.py) against known answers.  cv2 is absent, so the
oracle is pinned the same way as the HIP kernel: exact synthetic correspondences must give back the pose."""
import numpy as np

from oracle import pnp as op
from test_pnp_gpu import _problem


def _solve(p, prob=0):
    return op.pose_recovery_ransac_pnp(p["tar2d"], p["src3d"], p["K"], p["pose"], p["tar_pts"], p["src_pts"], prob=prob)


def test_noise_free_recovers_ground_truth_pose():
    rng = np.random.default_rng(0)
    for i, n in enumerate((6, 20, 200, 1500)):
        p = _problem(rng, n)
        rot, tvec, ratio, ok = _solve(p, i)
        assert ok and ratio == 1.0 and rot.shape == (3, 3) and tvec.shape == (3, 1)
        assert np.abs(rot - p["R"]).max() < 1e-4 and np.abs(tvec[:, 0] - p["t"]).max() < 1e-4   # float32 inputs
        assert abs(np.linalg.det(rot) - 1.0) < 1e-9


def test_planted_outliers_are_rejected():
    rng = np.random.default_rng(1)
    p = _problem(rng, 400, n_out=120)
    rot, tvec, ratio, ok = _solve(p)
    assert ok and abs(ratio * 400 - p["n_in"]) <= 1
    assert np.abs(rot - p["R"]).max() < 1e-4 and np.abs(tvec[:, 0] - p["t"]).max() < 1e-4


def test_failure_branch_and_gather_order():
    rng = np.random.default_rng(2)
    for n in (0, 3):       # fewer correspondences than a minimal sample: the reference's except branch (:99-105)
        rot, tvec, ratio, ok = _solve(_problem(rng, n))
        assert not ok and ratio == 0.0 and np.array_equal(rot, np.eye(3)) and np.array_equal(tvec, [[0.0], [0.0], [1.0]])
    p = _problem(rng, 50)
    p3, p2 = op.gather_valid(p["tar2d"], p["src3d"], p["pose"], p["tar_pts"], p["src_pts"])
    slots = np.nonzero(p["tar_pts"][:, 0] != -1)[0]      # list order is preserved (torch_utils.py:257-284)
    assert len(p3) == 50
    for r in (0, 17, 49):
        tx, ty = p["tar_pts"][slots[r]]
        assert np.array_equal(p2[r], p["tar2d"][:, ty, tx])
    assert len(set(op.sample_indices(3, 7, 50))) == op.SAMPLE
