"""Synthetic PnP problems in the layout utils/pose_recovery.py:68-105 consumes (shared by tests/test_pnp_gpu.py and
tests/pnp_stats.py): a template-camera-frame 3-D map (3,64,64), an original-image 2-D map (2,64,64) and the two
-1-padded (4096,2) int64 [x, y] key-point lists of compute_stage3_correspondences, plus the planted ground truth."""
import numpy as np

K0 = np.array([[572.4114, 0, 325.2611], [0, 573.57043, 242.04899], [0, 0, 1.0]])
H = W = 64


def random_rotations(rng, n):
    q, _ = np.linalg.qr(rng.standard_normal((n, 3, 3)))
    return q * np.sign(np.linalg.det(q))[:, None, None]


def make_batch(rng, P, n_pts, outlier_frac=0.0, noise=0.0):
    """P problems with n_pts correspondences each, round(outlier_frac*n_pts) of them uniform-random image points, Gaussian
    pixel noise of std `noise` on the others -> dict of stacked arrays + ground truth R (P,3,3), t (P,3), n_in."""
    n_out = int(round(outlier_frac * n_pts))
    R_tem, t_tem = random_rotations(rng, P), np.tile(np.array([0.02, -0.01, 0.8]), (P, 1))
    R_gt, t_gt = random_rotations(rng, P), np.array([0.05, -0.03, 0.9]) + 0.05 * rng.standard_normal((P, 3))
    obj = (rng.random((P, n_pts, 3)) - 0.5) * 0.2
    cam_tem = np.einsum("pnk,pjk->pnj", obj, R_tem) + t_tem[:, None]
    cam_gt = np.einsum("pnk,pjk->pnj", obj, R_gt) + t_gt[:, None]
    uv = np.einsum("pnk,jk->pnj", cam_gt / cam_gt[..., 2:], K0)[..., :2] + noise * rng.standard_normal((P, n_pts, 2))
    if n_out:
        uv[:, :n_out] = rng.random((P, n_out, 2)) * np.array([640, 480])        # (cells/slots below are random anyway)
    perm = lambda: np.argsort(rng.random((P, H * W)), axis=1)[:, :n_pts]        # noqa: E731  distinct cells per problem
    cells, tcells, slots = perm(), perm(), np.sort(perm(), axis=1)
    src3d = np.zeros((P, 3, H * W), np.float32)
    tar2d = np.zeros((P, 2, H * W), np.float32)
    tar_pts = -np.ones((P, H * W, 2), np.int64)
    src_pts = -np.ones((P, H * W, 2), np.int64)
    rows = np.arange(P)[:, None]
    for c in range(3):
        src3d[rows, c, cells] = cam_tem[..., c]
    for c in range(2):
        tar2d[rows, c, tcells] = uv[..., c]
    src_pts[rows, slots] = np.stack([cells % W, cells // W], axis=-1)
    tar_pts[rows, slots] = np.stack([tcells % W, tcells // W], axis=-1)
    pose = np.tile(np.eye(4, dtype=np.float32), (P, 1, 1))
    pose[:, :3, :3], pose[:, :3, 3] = R_tem, t_tem
    return dict(tar2d=tar2d.reshape(P, 2, H, W), src3d=src3d.reshape(P, 3, H, W), K=np.tile(K0.astype(np.float32), (P, 1, 1)),
                pose=pose, tar_pts=tar_pts, src_pts=src_pts, R=R_gt, t=t_gt, n_in=n_pts - n_out)


def pose_errors(rot, tvec, R_gt, t_gt):
    """-> (rotation error in degrees (P,), relative translation error |dt|/|t| (P,))."""
    tr = np.einsum("pij,pij->p", rot, R_gt)
    ang = np.degrees(np.arccos(np.clip((tr - 1) / 2, -1, 1)))
    return ang, np.linalg.norm(tvec.reshape(-1, 3) - t_gt, axis=1) / np.linalg.norm(t_gt, axis=1)
