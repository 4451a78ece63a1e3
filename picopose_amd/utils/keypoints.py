"""Host mirror of the reference's utils/keypoints.py (training forward): the key-point sampler that
builds the ground-truth correspondences of a (template, real) pair.  The per-point geometry runs in one HIP call
(pp_train_keypoints); the small matrix inverses are torch calls, as in the reference."""
import torch

from .. import _lib


class KeypointInput:
    """utils/keypoints.py:13-21: the fields of one view the sampler reads."""

    def __init__(self, K, full_depth, mask, M, full_rgb=None, rgb=None):
        self.K, self.full_depth, self.mask, self.M, self.full_rgb, self.rgb = K, full_depth, mask, M, full_rgb, rgb


def inverse_affine(M):
    """utils/torch_utils.py:93-111: inverse of a crop affine (uniform scale, no skew — asserted like the reference)."""
    assert (M[:, 1, 0] == 0).all() and (M[:, 0, 1] == 0).all()
    assert (M[:, 0, 0] == M[:, 1, 1]).all(), f"M: {M}"
    s = M[:, 0, 0]
    inv = torch.eye(3, device=M.device, dtype=M.dtype).repeat(M.shape[0], 1, 1)
    inv[:, 0, 0] = 1 / s
    inv[:, 1, 1] = 1 / s
    inv[:, :2, 2] = -M[:, :2, 2] / s.unsqueeze(1)
    return inv


class KeyPointSampler:
    """utils/keypoints.py:94-205.  tar_size / patch_size are the reference's defaults (224 / 3.5: a 64 x 64 grid), which is
    what the kernel is built for."""

    def __init__(self, tar_size=224, patch_size=3.5):
        if (tar_size, patch_size) != (224, 3.5):
            raise _lib.PicoPoseHipError("the HIP key-point sampler is built for the reference's 224 / 3.5 grid")
        self.tar_size, self.patch_size = tar_size, patch_size

    def sample_pts(self, T_src2target, T_tar2source, src_data, tar_data):
        """-> {"src_pts", "tar_pts"}: (B,4096,2) fp32 patch coordinates, -1 where a key-point has no correspondence."""
        smask, tmask, sdepth, tdepth = _lib.dev_f32(src_data.mask, tar_data.mask, src_data.full_depth, tar_data.full_depth)
        sM, tM, sK, tK, Tst, Tts = _lib.dev_f32(src_data.M, tar_data.M, src_data.K, tar_data.K, T_src2target, T_tar2source)
        sMi, tMi, sKi, tKi = _lib.dev_f32(inverse_affine(sM), inverse_affine(tM), torch.inverse(sK).float(), torch.inverse(tK).float())
        B, mh, mw = smask.shape
        dh, dw = sdepth.shape[1:]
        assert tmask.shape == smask.shape and tdepth.shape == sdepth.shape
        L = _lib.lib()
        ws_bytes = L.pp_train_keypoints_workspace_bytes(B)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=smask.device)
        src = torch.empty(B, 4096, 2, dtype=torch.float32, device=smask.device)
        tar = torch.empty_like(src)
        p = lambda t: t.data_ptr()  # noqa: E731
        rc = L.pp_train_keypoints(p(smask), p(tmask), mh, mw, p(sdepth), p(tdepth), dh, dw, p(sMi), p(tMi), p(sM), p(tM), p(sKi), p(tKi),
                                  p(sK), p(tK), p(Tst), p(Tts), B, p(src), p(tar), p(ws), ws_bytes, _lib.stream_ptr())
        _lib.check(rc, "pp_train_keypoints")
        return {"src_pts": src, "tar_pts": tar}
