"""Host mirror of reference utils/pose_recovery.py (HIP through the C ABI)."""
import os

import torch

from .. import _lib


def pose_recovery_2d_prediction(query_M, query_K, pred_Ms, template_K, template_Ms, template_poses):
    """Drop-in for reference utils/pose_recovery.py:9-65 -> pred_poses (B,4,4).

    The reference asserts, with two host syncs per call, that query_M is a crop affine (M01 = M10 = 0, M00 = M11:
    `inverse_affine`, torch_utils.py:100-101).  Here that is the caller's contract by default — the path stays
    sync-free — and PP_CHECK_CONTRACTS=1 in the environment restores the two asserts (same AssertionError)."""
    qM, qK, pM, tK, tM, tp = _lib.dev_f32(query_M, query_K, pred_Ms, template_K, template_Ms, template_poses)
    if os.environ.get("PP_CHECK_CONTRACTS") == "1":
        assert torch.all(qM[:, 0, 1] == 0) and torch.all(qM[:, 1, 0] == 0)          # torch_utils.py:100
        assert torch.all(qM[:, 0, 0] == qM[:, 1, 1])                                 # torch_utils.py:101
    B = qM.shape[0]
    out = torch.empty(B, 4, 4, dtype=torch.float32, device=qM.device)
    rc = _lib.lib().pp_pose_recovery_2d(qM.data_ptr(), qK.data_ptr(), pM.data_ptr(), tK.data_ptr(), tM.data_ptr(),
                                        tp.data_ptr(), B, out.data_ptr(), _lib.stream_ptr())
    _lib.check(rc, "pp_pose_recovery_2d")
    return out


def pnp_launch(tar_pts_2d, src_pts_3d, K, tem_pose, tar_pts, src_pts, iterations=150, reproj_error=2.0, branches=False):
    """Enqueue the batched PnP/RANSAC kernel; returns DEVICE tensors (rot (P,3,3) f64, tvec (P,3) f64, ratio (P) f64,
    ok (P) i32, npts (P) i32 = correspondences each problem received) without synchronising.  branches=True: a sixth tensor
    (P,40) f64, the refit's three beta-branch candidates [R, t, error] and the index of the one kept (pp_pnp_ransac_debug)."""
    t2, s3, Kd, pose = _lib.dev_f32(tar_pts_2d, src_pts_3d, K, tem_pose)
    tp, sp = tar_pts.contiguous().long(), src_pts.contiguous().long()
    P, _, H, W = t2.shape
    N = tp.shape[1]
    dev = t2.device
    rot = torch.empty(P, 3, 3, dtype=torch.float64, device=dev)
    tvec = torch.empty(P, 3, dtype=torch.float64, device=dev)
    ratio = torch.empty(P, dtype=torch.float64, device=dev)
    ok = torch.empty(P, dtype=torch.int32, device=dev)
    npts = torch.empty(P, dtype=torch.int32, device=dev)
    if branches:
        dbg = torch.empty(P, 40, dtype=torch.float64, device=dev)
        rc = _lib.lib().pp_pnp_ransac_debug(t2.data_ptr(), s3.data_ptr(), Kd.data_ptr(), pose.data_ptr(), tp.data_ptr(), sp.data_ptr(),
                                            P, H, W, N, int(iterations), float(reproj_error), rot.data_ptr(), tvec.data_ptr(),
                                            ratio.data_ptr(), ok.data_ptr(), npts.data_ptr(), dbg.data_ptr(), _lib.stream_ptr())
        _lib.check(rc, "pp_pnp_ransac_debug")
        return rot, tvec, ratio, ok, npts, dbg
    rc = _lib.lib().pp_pnp_ransac(t2.data_ptr(), s3.data_ptr(), Kd.data_ptr(), pose.data_ptr(), tp.data_ptr(), sp.data_ptr(),
                                  P, H, W, N, int(iterations), float(reproj_error), rot.data_ptr(), tvec.data_ptr(),
                                  ratio.data_ptr(), ok.data_ptr(), npts.data_ptr(), _lib.stream_ptr())
    _lib.check(rc, "pp_pnp_ransac")
    return rot, tvec, ratio, ok, npts


def refit_branches(tar_pts_2d, src_pts_3d, K, tem_pose, tar_pts, src_pts, iterations=150, reproj_error=2.0):
    """Parity instrument (tests): the batched kernel's result plus, per problem, the three beta-branch candidates of its final
    refit -> rot (P,3,3), tvec (P,3,1), ratio (P), ok (P), branches: list (P) of ([(err, R (3,3), t (3,))] * 3, kept index)."""
    rot, tvec, ratio, ok, npts, dbg = pnp_launch(tar_pts_2d, src_pts_3d, K, tem_pose, tar_pts, src_pts, iterations, reproj_error,
                                                 branches=True)
    d = dbg.cpu().numpy()
    out = []
    for p in range(d.shape[0]):
        cand = [(float(d[p, 13 * a + 12]) if d[p, 13 * a + 12] < 1e299 else float("inf"), d[p, 13 * a:13 * a + 9].reshape(3, 3).copy(),
                 d[p, 13 * a + 9:13 * a + 12].copy()) for a in range(3)]
        out.append((cand, int(d[p, 39])))
    return rot.cpu().numpy(), tvec.cpu().numpy()[:, :, None], ratio.cpu().numpy(), ok.cpu().numpy() != 0, out


def pose_recovery_ransac_pnp_batched(tar_pts_2d, src_pts_3d, K, tem_pose, tar_pts, src_pts, iterations=150,
                                     reproj_error=2.0, return_npts=False):
    """All (instance, hypothesis) problems of a batch in ONE launch and ONE device->host copy
    (the reference loops over them on the host with a sync each, run_test.py:168-184).

    tar_pts_2d (P,2,H,W), src_pts_3d (P,3,H,W), K (P,3,3), tem_pose (P,4,4), tar_pts/src_pts (P,N,2) int64
    -> rot (P,3,3) f64, tvec (P,3,1) f64, inliers_ratio (P) f64, success (P) bool   (numpy arrays)
    [+ npts (P) int32 with return_npts]."""
    rot, tvec, ratio, ok, npts = pnp_launch(tar_pts_2d, src_pts_3d, K, tem_pose, tar_pts, src_pts, iterations, reproj_error)
    P = rot.shape[0]
    # one packed device->host copy (P x 15 doubles + the saturation row) instead of four
    host = _with_sat_row(torch.cat([rot.reshape(P, 9), tvec, ratio[:, None], ok.double()[:, None], npts.double()[:, None]], dim=1)).cpu().numpy()
    host = _check_sat_row(host, P)
    res = (host[:, :9].reshape(P, 3, 3).copy(), host[:, 9:12].reshape(P, 3, 1).copy(), host[:, 12].copy(), host[:, 13] != 0)
    return res + (host[:, 14].astype("int32"),) if return_npts else res


def _with_sat_row(packed):
    """packed (P, 15) f64 on the device + ONE more row whose first entry is the sticky operand-saturation word (picopose_amd/ops.py): the
    host learns with the poses' own copy — no extra synchronisation — whether the forward that produced them clamped an operand."""
    from .. import ops

    w = ops.saturation_word(packed.device)
    if w is None:
        return packed
    row = torch.zeros(1, packed.shape[1], dtype=packed.dtype, device=packed.device)
    row[0, 0] = w[0]
    return torch.cat([packed, row])


def _check_sat_row(host, P):
    """host (P or P + 1, 15): strips the saturation row; raises (and resets the word) if it is set."""
    if host.shape[0] == P:
        return host
    if host[P, 0] != 0:
        from .. import ops

        ops.saturation_word().zero_()
        raise ops.saturation_error("a forward since the saturation word was last read (sticky: normally the forward whose poses were just read)")
    return host[:P]


class PnPHandle:
    """A batched PnP launch whose result is on its way to the host (pose_recovery_ransac_pnp_batched_async)."""
    __slots__ = ("host", "event", "P")

    def __init__(self, host, event, P):
        self.host, self.event, self.P = host, event, P

    def result(self, return_npts=False):
        """Wait for the copy and unpack: rot (P,3,3) f64, tvec (P,3,1) f64, inliers_ratio (P) f64, success (P) bool [+ npts]."""
        self.event.synchronize()
        host, P = _check_sat_row(self.host.numpy(), self.P), self.P
        res = (host[:, :9].reshape(P, 3, 3).copy(), host[:, 9:12].reshape(P, 3, 1).copy(), host[:, 12].copy(), host[:, 13] != 0)
        return res + (host[:, 14].astype("int32"),) if return_npts else res


def pose_recovery_ransac_pnp_batched_async(tar_pts_2d, src_pts_3d, K, tem_pose, tar_pts, src_pts, iterations=150, reproj_error=2.0,
                                           host=None, stream=None):
    """pose_recovery_ransac_pnp_batched without the host wait: the launch and ONE asynchronous device->host copy (P x 15 doubles
    into a pinned buffer, `host` to reuse one) are enqueued on the current stream; `.result()` of the returned handle waits for
    them.  A serving loop launches batch i + 1 before it reads batch i's poses, so the GPU never waits for the host.
    stream: a side torch.cuda.Stream for the PnP launch and the copy (it first waits for the current stream, i.e. for the forward
    that produced the inputs): the batch's PnP — one 512-thread workgroup per problem, latency-bound fp64 work on 160 of the
    256 CUs — then runs beside the NEXT batch's forward instead of in front of it."""
    inputs = (tar_pts_2d, src_pts_3d, K, tem_pose, tar_pts, src_pts)
    if stream is not None:
        stream.wait_stream(torch.cuda.current_stream())
        for t in inputs:
            t.record_stream(stream)      # (the caching allocator must not hand these blocks out while the side stream reads them)
    with torch.cuda.stream(stream if stream is not None else torch.cuda.current_stream()):
        rot, tvec, ratio, ok, npts = pnp_launch(*inputs, iterations, reproj_error)
        P = rot.shape[0]
        packed = _with_sat_row(torch.cat([rot.reshape(P, 9), tvec, ratio[:, None], ok.double()[:, None], npts.double()[:, None]], dim=1))
        if host is None or tuple(host.shape) != tuple(packed.shape):
            host = torch.empty(tuple(packed.shape), dtype=torch.float64, pin_memory=True)
        host.copy_(packed, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
    return PnPHandle(host, ev, P)


def pose_recovery_ransac_pnp(tar_pts_2d, src_pts_3d, K, tem_pose, tar_pts, src_pts):
    """Drop-in for reference utils/pose_recovery.py:68-105 (one instance/hypothesis):
    -> (rot ndarray(3,3), tvecs ndarray(3,1), inliers_ratio float, success bool); never raises for bad
    geometry — failure returns (I, [0,0,1]^T, 0.0, False) like the reference's except branch."""
    rot, tvec, ratio, ok = pose_recovery_ransac_pnp_batched(tar_pts_2d[None], src_pts_3d[None], K[None], tem_pose[None],
                                                           tar_pts[None], src_pts[None])
    return rot[0], tvec[0], float(ratio[0]), bool(ok[0])
