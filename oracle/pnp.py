"""CPU oracle of the PnP/RANSAC pose recovery — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
path (picopose_amd/) never does.

Reference call site: utils/pose_recovery.py:68-105 — `gather` of the 2-D / 3-D maps at the selected
keypoints (utils/torch_utils.py:257-284), object-frame points `(X - t_tem) @ R_tem` (:84), then
`cv2.solvePnPRansac(..., reprojectionError=2, iterationsCount=150, flags=cv2.SOLVEPNP_EPNP)` (:93-95) and
`cv2.Rodrigues` (:96); the except branch returns (I, [0,0,1]^T, 0.0, False) (:99-105).

The arithmetic lives in a third-party dependency that is neither vendored under /root/reference nor installed
here: opencv-python==4.9.0.80 (requirements.txt:3).  PARITY UNPINNED against cv2: this file restates the
published algorithm — RANSAC over 5-point minimal samples with EPnP as the model solver (Lepetit,
Moreno-Noguer, Fua, "EPnP: an accurate O(n) solution to the PnP problem", IJCV 2009: 4 control points from
the PCA of the model points, barycentric coordinates, the 12x12 null space of M^T M, the three beta
initialisations + 5 Gauss-Newton steps on the 6 control-point distances, absolute orientation by SVD, the
solution of least reprojection error), a 2 px reprojection test and an EPnP refit on the inliers — with this
build's own sampling sequence (a counter-based hash instead of cv::RNG, all 150 iterations instead of the
99 %-confidence early exit).  It is pinned by known answers (tests/test_pnp_gpu.py, tests/test_oracle_pnp.py):
exact synthetic correspondences recover the ground-truth pose, planted outliers are rejected.

Written with numpy (fp64 like OpenCV's solver; eigen-decompositions by LAPACK where the HIP kernel uses
cyclic Jacobi), so agreement with the HIP kernel is to solver tolerance, not bit-wise.
"""
import numpy as np

SAMPLE = 5     # minimal sample of solvePnPRansac for SOLVEPNP_EPNP
MAXP = 4096    # 64 x 64 keypoint slots (utils/correspondence.py:28-59)
_M32 = 0xFFFFFFFF


def hash32(x):
    """The sampling hash of csrc/pp_pnp.hip (lowbias32)."""
    x &= _M32
    x ^= x >> 16
    x = (x * 0x7FEB352D) & _M32
    x ^= x >> 15
    x = (x * 0x846CA68B) & _M32
    x ^= x >> 16
    return x


def sample_indices(prob, h, npts):
    """5 distinct indices of hypothesis h of problem `prob` (same sequence as the HIP kernel)."""
    s = hash32(((0x9E3779B9 * (prob + 1)) & _M32) ^ ((h * 7919 + 17) & _M32))
    idx = []
    while len(idx) < SAMPLE:
        s = hash32((s + 0x6D2B79F5) & _M32)
        c = s % npts
        if c not in idx:
            idx.append(c)
    return idx


def gather_valid(tar_pts_2d, src_pts_3d, tem_pose, tar_pts, src_pts):
    """utils/torch_utils.py:257-284 at pose_recovery.py:76-77 + the object-frame transform of :84, in float32.
    tar_pts_2d (2,H,W), src_pts_3d (3,H,W), tem_pose (4,4), tar_pts/src_pts (N,2) int64 [x, y], -1 = empty slot."""
    t2 = np.asarray(tar_pts_2d, np.float32)
    s3 = np.asarray(src_pts_3d, np.float32)
    P = np.asarray(tem_pose, np.float32)
    tp, sp = np.asarray(tar_pts, np.int64), np.asarray(src_pts, np.int64)
    ok = (tp[:, 0] != -1) & (tp[:, 1] != -1) & (sp[:, 0] != -1) & (sp[:, 1] != -1)
    tp, sp = tp[ok][:MAXP], sp[ok][:MAXP]
    p2 = t2[:, tp[:, 1], tp[:, 0]].T.astype(np.float32)                     # (n,2)
    d = s3[:, sp[:, 1], sp[:, 0]].T - P[:3, 3][None]                          # (n,3) float32
    R = P[:3, :3]
    # component j = sum_i d_i R[i][j], accumulated left to right in float32 like the kernel
    p3 = (d[:, 0:1] * R[0][None] + d[:, 1:2] * R[1][None]) + d[:, 2:3] * R[2][None]
    return p3.astype(np.float32), p2


def _lstsq(A, b):
    return np.linalg.lstsq(A, b, rcond=None)[0]


def _horn(H):
    """Proper rotation closest to U V^T of H = sum pc pw^T (Kabsch)."""
    U, _, Vt = np.linalg.svd(H)
    D = np.diag([1.0, 1.0, np.sign(np.linalg.det(U @ Vt)) or 1.0])
    return U @ D @ Vt


def _eig3_jacobi(A):
    """Cyclic Jacobi on a symmetric 3x3 matrix in the kernel's rotation order (csrc/pp_pnp.hip eig3_sym): eigenvalues
    ascending, eigenvectors as rows.  Used with solver="kernel" so that eigenvector SIGNS (which fix the control points'
    parametrisation) are the kernel's, not LAPACK's."""
    A = np.array(A, np.float64)
    V = np.eye(3)
    for _ in range(30):
        off = A[0, 1] ** 2 + A[0, 2] ** 2 + A[1, 2] ** 2
        diag = A[0, 0] ** 2 + A[1, 1] ** 2 + A[2, 2] ** 2
        if off <= 1e-30 * (diag + 1e-300):
            break
        for p, q in ((0, 1), (0, 2), (1, 2)):
            apq = A[p, q]
            if abs(apq) < 1e-300:
                continue
            theta = (A[q, q] - A[p, p]) / (2.0 * apq)
            t = (1.0 if theta >= 0 else -1.0) / (abs(theta) + np.sqrt(theta * theta + 1.0))
            c = 1.0 / np.sqrt(t * t + 1.0)
            sn = t * c
            J = np.eye(3)
            J[p, p] = J[q, q] = c
            J[p, q], J[q, p] = sn, -sn
            A = J.T @ A @ J
            V = J.T @ V                                   # rows p, q of V: c v_p - s v_q, s v_p + c v_q
    w = np.diag(A).copy()
    order = sorted(range(3), key=lambda k: (w[k], k))
    return w[order], V[order]


def _round_robin(s):
    """The 6 disjoint column pairs (p < q) of round s of the kernel's 12-player tournament."""
    pairs = [(s, 11)]
    for k in range(11):
        if k != s:
            q = (2 * s - k) % 11
            if k < q:
                pairs.append((k, q))
    return pairs


def _hestenes12(G):
    """One-sided Jacobi on the columns of the symmetric PSD 12x12 G, restating csrc/pp_pnp.hip hestenes12 (same pair
    order, same convergence floor, incremental column norms; the kernel evaluates the rotation tangent in fp32, this
    restatement in fp64 — an angle difference of 1e-7).  Returns (lambda^2 per column, V) with V[:, k] the eigenvector."""
    g = np.array(G, np.float64)
    v = np.eye(12)
    for _ in range(14):
        alpha = (g * g).sum(axis=0)
        floor2 = 2.5e-29 * alpha.max()
        rotated = False
        for s in range(11):
            for p, q in _round_robin(s):
                gamma = float(g[:, p] @ g[:, q])
                ap, aq = alpha[p], alpha[q]
                if gamma * gamma > floor2 * (ap + aq):
                    rotated = True
                    zeta = (aq - ap) / (2.0 * gamma)
                    t = (1.0 if zeta >= 0 else -1.0) / (abs(zeta) + np.sqrt(1.0 + zeta * zeta))
                    if not (abs(t) <= 1.0):
                        t = 0.0
                    c = 1.0 / np.sqrt(1.0 + t * t)
                    sn = c * t
                    gp, gq, vp, vq = g[:, p].copy(), g[:, q].copy(), v[:, p].copy(), v[:, q].copy()
                    g[:, p], g[:, q] = c * gp - sn * gq, sn * gp + c * gq
                    v[:, p], v[:, q] = c * vp - sn * vq, sn * vp + c * vq
                    alpha[p] = c * c * ap - 2.0 * c * sn * gamma + sn * sn * aq
                    alpha[q] = sn * sn * ap + 2.0 * c * sn * gamma + c * c * aq
        if not rotated:
            break
    return (g * g).sum(axis=0), v


def _canonical_signs(v):
    """Columns of v (principal axes of the model points) with a solver-independent sign: the component of largest magnitude
    (lowest index among equals) is positive.  EPnP's estimate is NOT invariant to the side of the centroid a control point is
    put on — on noisy data the two choices give poses ~1 mm apart at equal reprojection error — and an eigen-solver's sign is an
    accident of its rotation sequence, so kernel (csrc/pp_pnp.hip) and oracle fix it the same way.  (OpenCV takes the sign its
    SVD returns; unknowable here.)"""
    v = np.array(v, np.float64)
    for k in range(v.shape[1]):
        i = int(np.argmax(np.abs(v[:, k])))
        if v[i, k] < 0:
            v[:, k] = -v[:, k]
    return v


def epnp(p3, p2, cam, solver="lapack", branches=None):
    """EPnP on the given points: returns (mean reprojection error, R (3,3), t (3,)); error = inf if no solution.
    branches: a list that receives, per beta initialisation (N = 1 / 2 / 3), the candidate (err, R, t) — err = inf where the
    branch gave no pose (what pp_pnp_ransac_debug returns for the refit).
    solver="lapack": numpy eigh (independent of the kernel); solver="kernel": the kernel's own eigen-solvers restated
    (_eig3_jacobi, _hestenes12), which fixes the basis of the degenerate null space of a minimal sample the same way."""
    fu, fv, uc, vc = cam
    p3 = np.asarray(p3, np.float64)
    p2 = np.asarray(p2, np.float64)
    n = len(p3)
    c0 = p3.mean(axis=0)
    q = p3 - c0
    if solver == "kernel":
        w, vr = _eig3_jacobi(q.T @ q)
        v = vr.T
    else:
        w, v = np.linalg.eigh(q.T @ q)                      # ascending
    v = _canonical_signs(v)
    cws = [c0]
    for k in range(3):                                       # largest first
        cws.append(c0 + np.sqrt(max(w[2 - k], 0.0) / n) * v[:, 2 - k])
    cws = np.array(cws)
    CC = (cws[1:] - cws[0]).T
    det = np.linalg.det(CC)
    cci = np.linalg.inv(CC) if abs(det) > 1e-300 else np.zeros((3, 3))
    a123 = q @ cci.T
    al = np.concatenate([1.0 - a123.sum(axis=1, keepdims=True), a123], axis=1)   # (n,4)
    M = np.zeros((2 * n, 12))
    for j in range(4):
        M[0::2, 3 * j] = al[:, j] * fu
        M[0::2, 3 * j + 2] = al[:, j] * (uc - p2[:, 0])
        M[1::2, 3 * j + 1] = al[:, j] * fv
        M[1::2, 3 * j + 2] = al[:, j] * (vc - p2[:, 1])
    if solver == "kernel":
        lam2, ev = _hestenes12(M.T @ M)
        order = sorted(range(12), key=lambda k: (lam2[k], k))
        vn = [ev[:, order[k]] for k in range(4)]
    else:
        _, ev = np.linalg.eigh(M.T @ M)
        vn = [ev[:, k] for k in range(4)]                    # null-space basis, smallest eigenvalue first
    pa, pb = [0, 0, 0, 1, 1, 2], [1, 2, 3, 2, 3, 3]
    dv = np.array([[vn[i][3 * pa[p]:3 * pa[p] + 3] - vn[i][3 * pb[p]:3 * pb[p] + 3] for p in range(6)] for i in range(4)])
    dot = lambda i, j: (dv[i] * dv[j]).sum(axis=1)           # noqa: E731  (6,)
    L = np.stack([dot(0, 0), 2 * dot(0, 1), dot(1, 1), 2 * dot(0, 2), 2 * dot(1, 2), dot(2, 2), 2 * dot(0, 3),
                  2 * dot(1, 3), 2 * dot(2, 3), dot(3, 3)], axis=1)
    rho = np.array([((cws[pa[p]] - cws[pb[p]]) ** 2).sum() for p in range(6)])
    best = (np.inf, np.eye(3), np.array([0.0, 0.0, 1.0]))
    for approx in range(3):
        b = np.zeros(4)
        with np.errstate(all="ignore"):
            if approx == 0:
                x = _lstsq(L[:, [0, 1, 3, 6]], rho)
                b[0] = np.sqrt(abs(x[0]))
                b[1:] = (-x[1:] if x[0] < 0 else x[1:]) / b[0]
            else:
                x = _lstsq(L[:, :3] if approx == 1 else L[:, :5], rho)
                if x[0] < 0:
                    b[0], b[1] = np.sqrt(-x[0]), (np.sqrt(-x[2]) if x[2] < 0 else 0.0)
                else:
                    b[0], b[1] = np.sqrt(x[0]), (np.sqrt(x[2]) if x[2] > 0 else 0.0)
                if x[1] < 0:
                    b[0] = -b[0]
                if approx == 2:
                    b[2] = x[3] / b[0] if abs(b[0]) > 1e-300 else 0.0
            if not np.all(np.isfinite(b)):
                if branches is not None:
                    branches.append((np.inf, np.eye(3), np.zeros(3)))
                continue
            for _ in range(5):                               # Gauss-Newton on the 6 distance constraints
                A = np.stack([2 * L[:, 0] * b[0] + L[:, 1] * b[1] + L[:, 3] * b[2] + L[:, 6] * b[3],
                              L[:, 1] * b[0] + 2 * L[:, 2] * b[1] + L[:, 4] * b[2] + L[:, 7] * b[3],
                              L[:, 3] * b[0] + L[:, 4] * b[1] + 2 * L[:, 5] * b[2] + L[:, 8] * b[3],
                              L[:, 6] * b[0] + L[:, 7] * b[1] + L[:, 8] * b[2] + 2 * L[:, 9] * b[3]], axis=1)
                bb = np.array([b[0] * b[0], b[0] * b[1], b[1] * b[1], b[0] * b[2], b[1] * b[2], b[2] * b[2], b[0] * b[3],
                               b[1] * b[3], b[2] * b[3], b[3] * b[3]])
                b = b + _lstsq(A, rho - L @ bb)
            ccs = sum(b[k] * vn[k] for k in range(4)).reshape(4, 3)
            pcs = al @ ccs
            if pcs[:, 2].sum() < 0:                          # solve_for_sign: mean depth must be positive
                ccs, pcs = -ccs, -pcs
            pc0, pw0 = pcs.mean(axis=0), p3.mean(axis=0)
            R = _horn((pcs - pc0).T @ (p3 - pw0))
            t = pc0 - R @ pw0
            cam_pts = p3 @ R.T + t
            du = uc + fu * cam_pts[:, 0] / cam_pts[:, 2] - p2[:, 0]
            dvv = vc + fv * cam_pts[:, 1] / cam_pts[:, 2] - p2[:, 1]
            err = np.sqrt(du * du + dvv * dvv).mean()
        if branches is not None:
            branches.append((err if np.isfinite(err) else np.inf, R, t))
        if np.isfinite(err) and err < best[0]:
            best = (err, R, t)
    return best


def _inliers(p3, p2, cam, R, t, th2):
    fu, fv, uc, vc = cam
    with np.errstate(all="ignore"):
        c = p3.astype(np.float64) @ R.T + t
        du = uc + fu * c[:, 0] / c[:, 2] - p2[:, 0]
        dv = vc + fv * c[:, 1] / c[:, 2] - p2[:, 1]
        return (du * du + dv * dv) <= th2                    # NaN is never an inlier


def pose_recovery_ransac_pnp(tar_pts_2d, src_pts_3d, K, tem_pose, tar_pts, src_pts, prob=0, iterations=150,
                             reproj_error=2.0, solver="lapack", return_branches=False):
    """utils/pose_recovery.py:68-105 -> (rot (3,3) f64, tvecs (3,1) f64, inliers_ratio float, success bool).
    `prob` is the problem's index in its batch (it seeds the sampling sequence, as in the batched HIP launch).
    return_branches: a fifth element, the refit's three beta-branch candidates [(err, R, t)] * 3 ([] on failure)."""
    fail = (np.eye(3), np.array([[0.0], [0.0], [1.0]]), 0.0, False)
    p3, p2 = gather_valid(tar_pts_2d, src_pts_3d, tem_pose, tar_pts, src_pts)
    n = len(p3)
    if return_branches:
        fail = fail + ([],)
    if n < SAMPLE:
        return fail
    K = np.asarray(K, np.float32).astype(np.float64)
    cam = (K[0, 0], K[1, 1], K[0, 2], K[1, 2])
    th2 = float(np.float32(reproj_error)) ** 2
    best_c, best = -1, None
    for h in range(min(iterations, 256)):                    # one hypothesis per thread of the 256-thread workgroup
        idx = sample_indices(prob, h, n)
        err, R, t = epnp(p3[idx], p2[idx], cam, solver)
        if not np.isfinite(err):
            R, t = np.zeros((3, 3)), np.zeros(3)
        c = int(_inliers(p3, p2, cam, R, t, th2).sum())
        if c > best_c:
            best_c, best = c, (R, t)
    if best_c < SAMPLE:
        return fail
    use = _inliers(p3, p2, cam, best[0], best[1], th2)
    branches = []
    err, R, t = epnp(p3[use], p2[use], cam, solver, branches)
    if not np.isfinite(err):
        R, t = best
    res = (R, t.reshape(3, 1), best_c / n, True)
    return res + (branches,) if return_branches else res


def reprojection_gap(tar_pts_2d, src_pts_3d, K, tem_pose, tar_pts, src_pts, pose_a, pose_b):
    """Median pixel distance between the projections of the problem's 3-D points under two poses (R, t) — the well-posed
    way to compare two PnP answers when the scene leaves part of the pose weakly constrained (a shallow object)."""
    p3, _ = gather_valid(tar_pts_2d, src_pts_3d, tem_pose, tar_pts, src_pts)
    K = np.asarray(K, np.float64)

    def proj(R, t):
        c = p3.astype(np.float64) @ np.asarray(R, np.float64).T + np.asarray(t, np.float64).reshape(1, 3)
        return np.stack([K[0, 2] + K[0, 0] * c[:, 0] / c[:, 2], K[1, 2] + K[1, 1] * c[:, 1] / c[:, 2]], axis=1)

    return float(np.median(np.linalg.norm(proj(*pose_a) - proj(*pose_b), axis=1)))
