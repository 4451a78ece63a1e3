"""Crop preprocessing: host bounding-box / affine logic against the oracle restatement of
utils/data_utils.py and provider/bop_test_dataset.py on CPU; the device resize against the oracle on GPU.
cv2 itself is not available: the interpolation parity with OpenCV is unpinned (oracle/preprocess.py header)."""
import numpy as np
import pytest

from oracle import preprocess as op

gpu = pytest.mark.gpu


def _case(seed, H=480, W=640):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    mask = np.zeros((H, W), np.uint8)
    y, x = int(rng.integers(0, H - 60)), int(rng.integers(0, W - 90))
    hh, ww = int(rng.integers(20, H - y)), int(rng.integers(20, W - x))
    mask[y:y + hh, x:x + ww] = rng.random((hh, ww)) < 0.8
    return img, mask, [x, y, ww, hh]


def test_bbox_logic_matches_oracle_and_known_answers():
    from picopose_amd.utils import preprocess as hp

    m = np.zeros((480, 640), np.uint8)
    m[100:220, 300:380] = 1
    assert hp.get_bbox(m) == op.get_bbox(m) == [100, 220, 280, 400]           # 120 x 80 box -> 120 x 120 around its centre
    m2 = np.zeros((480, 640), np.uint8)
    m2[0:50, 600:640] = 1                                                        # touches two borders: shifted back inside
    assert hp.get_bbox(m2) == op.get_bbox(m2) == [0, 50, 590, 640]
    assert hp.get_square_bbox([10, 400, 5, 30], (480, 640)) == op.get_square_bbox([10, 400, 5, 30], (480, 640))
    for s in range(5):
        _, mk, _ = _case(s)
        assert hp.get_bbox(mk) == op.get_bbox(mk)


def test_boxes_equal_the_reference_functions_outputs(golden_dir):
    """Product AND oracle against tests/golden/preprocess_boxes.npz = outputs of the reference's own get_bbox /
    get_square_bbox (utils/data_utils.py:131-196; oracle/gen_golden.py gen_preprocess), incl. boxes clamped at every border."""
    import os

    from picopose_amd.utils import preprocess as hp

    z = np.load(os.path.join(golden_dir, "preprocess_boxes.npz"))
    for m, ratio, want in zip(z["masks"], z["mask_ratio"], z["mask_boxes"]):
        assert hp.get_bbox(m, float(ratio)) == op.get_bbox(m, float(ratio)) == want.tolist()
    for box, size, ratio, want in zip(z["boxes"], z["sizes"], z["box_ratio"], z["square_boxes"]):
        args = ([int(v) for v in box], (int(size[0]), int(size[1])), float(ratio))
        assert hp.get_square_bbox(*args) == op.get_square_bbox(*args) == want.tolist()


def test_small_mask_branch_keeps_the_detection_box_like_the_reference():
    # bop_test_dataset.py:170-185: with <= minimum_n_point mask pixels the crop WINDOW comes from the detection box but
    # `bbox` (returned, and used by M_crop as [-bbox[2], -bbox[0]]) stays the detection's [x, y, w, h]
    img, mask, det = _case(2)
    mask[:] = 0
    mask[5, 5] = 1
    ref = op.crop_instance(img, mask, det)
    x, y, w, h = det
    y1, y2, x1, x2 = op.get_square_bbox([y, y + h, x, x + w], mask.shape)
    assert ref["bbox"] == det
    assert np.allclose(ref["M"], np.array([[224 / (y2 - y1), 0, -w * 224 / (y2 - y1)], [0, 224 / (x2 - x1), -x * 224 / (x2 - x1)],
                                           [0, 0, 1]], np.float32))


def test_oracle_resize_known_answers():
    ramp = np.arange(8, dtype=np.float64)[None, :, None].repeat(4, 0)
    up = op.resize_linear(ramp, 16)[0, :, 0]                                      # 2x upsampling of a ramp: pixel-centre alignment
    assert np.allclose(up[:4], [0.0, 0.25, 0.75, 1.25]) and up[-1] == 7.0
    assert np.array_equal(op.resize_nearest(np.arange(6)[None].repeat(6, 0), 3)[0], [0, 2, 4])
    assert np.allclose(op.resize_linear(np.full((5, 7, 3), 0.3), 224), 0.3)


@pytest.mark.parametrize("h,w", [(120, 120), (400, 400), (37, 53), (224, 224), (225, 223), (448, 448), (61, 61), (333, 333), (20, 20), (479, 479)])
def test_oracle_resize_second_opinion_torch_interpolate(h, w):
    """A second, independent implementation beside the restatement of OpenCV in oracle/preprocess.py (cv2 itself cannot be had
    here): torch.nn.functional.interpolate(mode="bilinear", align_corners=False) implements the same half-pixel definition as
    cv2.INTER_LINEAR — source coordinate (x + 0.5) * in / out - 0.5, clamped at the first pixel, second tap clamped at the last
    — and mode="nearest" the same floor(x * in / out) as cv2.INTER_NEAREST.  Crop shapes of get_instance
    (provider/bop_test_dataset.py:186-190): float images, up- and down-scaling, square and not.  In float64 the two agree to
    rounding (measured <= 4e-14) and the nearest maps are equal.  Where real OpenCV would still differ from both: its float
    path keeps the interpolation WEIGHTS in float32 (HResizeLinear / VResizeLinear coefficient buffers), a relative 6e-8 on a
    [0, 1] image, and its uint8 path (not used by the reference, which resizes rgb / 255.0) is 11-bit fixed point."""
    import torch
    import torch.nn.functional as F

    rng = np.random.default_rng(h * 1000 + w)
    img = rng.random((h, w, 3))
    ours = op.resize_linear(img, 224)
    theirs = F.interpolate(torch.from_numpy(img).permute(2, 0, 1)[None], size=(224, 224), mode="bilinear", align_corners=False)[0]
    assert np.abs(ours - theirs.permute(1, 2, 0).numpy()).max() <= 1e-12
    m = rng.integers(0, 2, (h, w)).astype(np.float64)
    near = F.interpolate(torch.from_numpy(m)[None, None], size=(224, 224), mode="nearest")[0, 0].numpy()
    assert np.array_equal(op.resize_nearest(m, 224), near)


@gpu
@pytest.mark.parametrize("seed,flag", [(0, False), (1, True), (2, False), (3, True)])
def test_crop_instance_device_vs_oracle(seed, flag):
    import torch

    from picopose_amd.utils import preprocess as hp

    img, mask, det = _case(seed)
    if seed == 2:
        mask[:] = 0
        mask[5, 5] = 1                                                            # too few points: detection-box fallback
    ref = op.crop_instance(img, mask, det, rgb_mask_flag=flag)
    got = hp.crop_instance(img, mask, det, rgb_mask_flag=flag)
    assert got["bbox"] == ref["bbox"]
    assert np.array_equal(got["M"].numpy(), ref["M"]) and np.array_equal(got["pts2d"].numpy(), ref["pts2d"])
    assert torch.equal(got["mask"].cpu(), torch.from_numpy(ref["mask"]))
    assert np.abs(got["rgb"].cpu().numpy() - ref["rgb"]).max() <= 1e-6            # double arithmetic on both sides


@gpu
@pytest.mark.parametrize("flag", [False, True])
def test_crop_template_device_vs_oracle(flag):
    from picopose_amd.utils import preprocess as hp

    rng = np.random.default_rng(11)
    H, W = 480, 640
    rgba = rng.integers(0, 256, (H, W, 4), dtype=np.uint8)
    alpha = np.zeros((H, W), np.uint8)
    alpha[120:300, 200:330] = 255
    alpha[110:120, 200:330] = 128                                   # anti-aliased rim: in the bbox / colour mask, not in the int mask
    rgba[..., 3] = alpha
    depth = (rng.random((H, W)) * 400 + 600).astype(np.float64) * (alpha > 0)
    K = np.array([[572.4114, 0, 320], [0, 573.57043, 240], [0, 0, 1.0]])
    pose = np.eye(4)
    pose[:3, 3] = [10.0, -20.0, 800.0]
    ref = op.crop_template(rgba, depth, K, pose, rgb_mask_flag=flag)
    got = hp.crop_template(rgba, depth, K, pose, rgb_mask_flag=flag)
    assert got["bbox"] == ref["bbox"] and np.array_equal(got["M"].numpy(), ref["M"])
    assert np.array_equal(got["mask"].cpu().numpy(), ref["mask"])
    assert np.abs(got["rgb"].cpu().numpy() - ref["rgb"]).max() <= 1e-6
    assert np.abs(got["pts3d"].cpu().numpy() - ref["pts3d"]).max() <= 1e-6
    assert np.allclose(got["pose"].numpy()[:3, 3], [0.01, -0.02, 0.8])
