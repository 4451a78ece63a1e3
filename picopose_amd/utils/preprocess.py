"""Crop preprocessing of a detection — mirror of provider/bop_test_dataset.py:146-207 (`BOPTestset.get_instance`) and
utils/data_utils.py:131-196, 231-250 on the device.

The bounding-box arithmetic, the crop affine `M`, the 64x64 lookup grid `pts2d`, the channel flip and the CLIP
normalisation give the reference's values (boxes pinned by reference-generated fixtures, tests/golden/preprocess_boxes.npz); the two `cv2.resize` calls run in `pp_crop_resize_normalize`
(OpenCV's published INTER_LINEAR / INTER_NEAREST definitions; cv2 is not available here to pin them bit-for-bit).
File IO (image / RLE decoding) stays with the caller."""
import ctypes

import numpy as np
import torch

from .. import _lib

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)       # bop_test_dataset.py:40
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)       # bop_test_dataset.py:41


def _fit_axis(lo, hi, side, limit):
    """A window of `side` (truncated to an even number of pixels) centred on the integer midpoint of [lo, hi), pushed back
    inside [0, limit]: first off the low border, then off the high one (it may then stick out below 0 again, as upstream)."""
    mid, half = int((lo + hi) / 2), int(side / 2)
    a, b = mid - half, mid + half
    if a < 0:
        a, b = 0, b - a
    if b > limit:
        a, b = a - (b - limit), limit
    return int(a), int(b)


def _square(rmin, rmax, cmin, cmax, img_width, img_length, size_ratio=1.0):
    """Square window of utils/data_utils.py:139-165 (= :170-196): side = the larger extent, capped by the smaller image
    dimension, times size_ratio; rows and columns are fitted independently.  Pinned by tests/golden/preprocess_boxes.npz
    (outputs of the reference functions)."""
    side = min(max(rmax - rmin, cmax - cmin), min(img_width, img_length)) * size_ratio
    return [*_fit_axis(rmin, rmax, side, img_width), *_fit_axis(cmin, cmax, side, img_length)]


def get_bbox(label, size_ratio=1.0):
    """utils/data_utils.py:131-165: square box around the non-zero pixels of a (H, W) mask -> [rmin, rmax, cmin, cmax]."""
    label = np.asarray(label)
    rows, cols = np.any(label, axis=1), np.any(label, axis=0)
    rmin, rmax = np.where(rows)[0][[0, -1]]
    cmin, cmax = np.where(cols)[0][[0, -1]]
    return _square(int(rmin), int(rmax) + 1, int(cmin), int(cmax) + 1, label.shape[0], label.shape[1], size_ratio)


def get_square_bbox(bbox, img_size, size_ratio=1.0):
    """utils/data_utils.py:167-196."""
    return _square(bbox[0], bbox[1], bbox[2], bbox[3], img_size[0], img_size[1], size_ratio)


def crop_instance(image_u8, mask_u8, det_bbox_xywh, img_size=224, pts_size=64, minimum_n_point=8, rgb_mask_flag=False,
                  device="cuda"):
    """One detection (bop_test_dataset.py:162-207): image (H,W,3) uint8 ndarray as loaded, full-frame binary mask (H,W)
    ndarray, detection box [x, y, w, h] -> dict(rgb (3,S,S) f32 cuda, mask (S,S) f32 cuda, bbox, M (3,3) f32 tensor,
    pts2d (P,P,2) f64 tensor) — the per-instance entries `real_rgb`, `real_mask`, `real_M`, `real_pts2d` of end_points."""
    mask_u8 = np.ascontiguousarray(mask_u8, dtype=np.uint8)
    image_u8 = np.ascontiguousarray(np.asarray(image_u8)[..., :3], dtype=np.uint8)
    h, w = mask_u8.shape
    assert image_u8.shape[:2] == (h, w)
    if np.sum(mask_u8) > minimum_n_point:
        bbox = get_bbox(mask_u8)
        y1, y2, x1, x2 = bbox
    else:
        # bop_test_dataset.py:172-173: the crop window comes from the detection box, but `bbox` — which feeds M_crop and
        # is returned — stays the detection's [x, y, w, h] (an upstream quirk: M then translates by (-w, -x)); kept for
        # parity with the reference's outputs on such instances
        b = bbox = list(det_bbox_xywh)
        y1, y2, x1, x2 = get_square_bbox([b[1], b[1] + b[3], b[0], b[0] + b[2]], (h, w))
    img_d = torch.from_numpy(image_u8).to(device)
    msk_d = torch.from_numpy(mask_u8).to(device)
    rgb = torch.empty(3, img_size, img_size, dtype=torch.float32, device=device)
    mask = torch.empty(img_size, img_size, dtype=torch.float32, device=device)
    mean, std = (ctypes.c_double * 3)(*CLIP_MEAN), (ctypes.c_double * 3)(*CLIP_STD)
    _lib.check(_lib.lib().pp_crop_resize_normalize(img_d.data_ptr(), h, w, msk_d.data_ptr(), y1, y2, x1, x2, img_size,
                                                   int(rgb_mask_flag), mean, std, rgb.data_ptr(), mask.data_ptr(),
                                                   _lib.stream_ptr()), "pp_crop_resize_normalize")
    M_crop = np.array([[1, 0, -bbox[2]], [0, 1, -bbox[0]], [0, 0, 1]], dtype=np.float32)
    M_resize = np.array([[img_size / (y2 - y1), 0, 0], [0, img_size / (x2 - x1), 0], [0, 0, 1]], dtype=np.float32)
    M = M_resize @ M_crop
    patch = img_size / pts_size                                  # utils/torch_utils.py:287-295 (y first, as there)
    x = np.arange(0, img_size, patch, dtype=np.float32) + patch / 2
    yy, xx = np.meshgrid(x, x, indexing="ij")
    pts = np.concatenate((np.stack([yy, xx], axis=2), np.ones((pts_size, pts_size, 1))), axis=2)
    p = np.linalg.inv(M) @ pts.reshape(-1, 3).transpose(1, 0)
    pts2d = (p[:2] / p[2:]).transpose(1, 0).reshape(pts_size, pts_size, 2)
    return {"rgb": rgb, "mask": mask, "bbox": bbox, "M": torch.from_numpy(M), "pts2d": torch.from_numpy(pts2d)}


def crop_template(rgba_u8, depth_mm, K, object_pose_mm, img_size=224, pts_size=64, rgb_mask_flag=False, device="cuda"):
    """One template view (bop_test_dataset.py:210-264, `_get_template`): rendered RGBA (H,W,4) uint8, depth (H,W) in mm,
    template intrinsics K (3,3), object pose (4,4) with t in mm -> dict(rgb (3,S,S), mask (S,S), pts3d (P,P,3) metres
    [cuda f32], bbox, M (3,3) f32, K, pose (t in metres)) — the per-template entries `tem_*` of end_points."""
    rgba_u8 = np.asarray(rgba_u8)
    mask = (rgba_u8[..., 3] / 255).astype(np.float32)                   # :222
    y1, y2, x1, x2 = bbox = get_bbox(mask)                              # :223
    mask_int = np.ascontiguousarray(mask.astype(int).astype(np.uint8))  # :242 `.astype(int)`: only alpha == 255 survives
    rgb_u8 = np.ascontiguousarray(rgba_u8[..., :3], dtype=np.uint8)
    h, w = mask.shape
    img_d, msk_d = torch.from_numpy(rgb_u8).to(device), torch.from_numpy(mask_int).to(device)
    # (the reference masks the colours with `mask > 0`, i.e. alpha > 0, not with the integer mask: data for that test)
    msk_rgb_d = torch.from_numpy(np.ascontiguousarray((mask > 0).astype(np.uint8))).to(device) if rgb_mask_flag else msk_d
    rgb = torch.empty(3, img_size, img_size, dtype=torch.float32, device=device)
    mout = torch.empty(img_size, img_size, dtype=torch.float32, device=device)
    mean, std = (ctypes.c_double * 3)(*CLIP_MEAN), (ctypes.c_double * 3)(*CLIP_STD)
    L = _lib.lib()
    _lib.check(L.pp_crop_resize_normalize(img_d.data_ptr(), h, w, msk_rgb_d.data_ptr(), y1, y2, x1, x2, img_size, int(rgb_mask_flag),
                                          mean, std, rgb.data_ptr(), None, _lib.stream_ptr()), "pp_crop_resize_normalize")
    dummy = torch.empty(3, img_size, img_size, dtype=torch.float32, device=device)
    _lib.check(L.pp_crop_resize_normalize(img_d.data_ptr(), h, w, msk_d.data_ptr(), y1, y2, x1, x2, img_size, 0, mean, std,
                                          dummy.data_ptr(), mout.data_ptr(), _lib.stream_ptr()), "pp_crop_resize_normalize")
    depth_d = torch.from_numpy(np.ascontiguousarray(np.asarray(depth_mm) / 1000.0, dtype=np.float32)).to(device)   # :227
    pts = torch.empty(pts_size, pts_size, 3, dtype=torch.float32, device=device)
    K = np.asarray(K, dtype=np.float64)
    _lib.check(L.pp_depth_points_nearest(depth_d.data_ptr(), h, w, y1, y2, x1, x2, pts_size, float(K[0, 0]), float(K[1, 1]),
                                         float(K[0, 2]), float(K[1, 2]), pts.data_ptr(), _lib.stream_ptr()),
               "pp_depth_points_nearest")
    pose = np.array(object_pose_mm, dtype=np.float64)
    pose[:3, 3] = pose[:3, 3] / 1000.0                                  # :246
    M_crop = np.array([[1, 0, -bbox[2]], [0, 1, -bbox[0]], [0, 0, 1]], dtype=np.float32)
    M_resize = np.array([[img_size / (y2 - y1), 0, 0], [0, img_size / (x2 - x1), 0], [0, 0, 1]], dtype=np.float32)
    return {"rgb": rgb, "mask": mout, "pts3d": pts, "bbox": bbox, "M": torch.from_numpy(M_resize @ M_crop),
            "K": torch.from_numpy(K.astype(np.float32)), "pose": torch.from_numpy(pose.astype(np.float32))}
