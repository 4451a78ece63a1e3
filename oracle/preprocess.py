"""CPU restatement of the crop preprocessing of provider/bop_test_dataset.py:146-207 (TEST INFRASTRUCTURE).

numpy only.  `cv2.resize` (opencv-python 4.9.0.80, requirements.txt:3) is not installed here and not vendored, so the
two interpolation modes are restated from OpenCV's published definition — parity with cv2 is UNPINNED:
  INTER_LINEAR on float input: source coordinate (x + 0.5) * scale - 0.5, floor/fraction, a coordinate left of the
      first pixel takes the first pixel, one at or past the last pixel takes the last pixel; separable weights;
  INTER_NEAREST: source index min(floor(x * scale), n - 1).
Everything else (bounding boxes, crop affine M, the 64 x 64 lookup grid, channel order, CLIP normalisation) follows
the reference line by line: utils/data_utils.py:131-196 (get_bbox, get_square_bbox), :231-250 (get_bop_image),
provider/bop_test_dataset.py:39-41 (ToTensor + Normalize), :179-196 (M, pts2d), utils/torch_utils.py:287-295."""
import numpy as np

CLIP_MEAN = np.array([0.48145466, 0.4578275, 0.40821073])
CLIP_STD = np.array([0.26862954, 0.26130258, 0.27577711])


def _square(rmin, rmax, cmin, cmax, img_width, img_length, size_ratio=1.0):
    # data_utils.py:139-165 / 170-196 (shared tail of get_bbox and get_square_bbox)
    r_b, c_b = rmax - rmin, cmax - cmin
    b = min(max(r_b, c_b), min(img_width, img_length)) * size_ratio
    center = [int((rmin + rmax) / 2), int((cmin + cmax) / 2)]
    rmin, rmax = center[0] - int(b / 2), center[0] + int(b / 2)
    cmin, cmax = center[1] - int(b / 2), center[1] + int(b / 2)
    if rmin < 0:
        rmax += -rmin
        rmin = 0
    if cmin < 0:
        cmax += -cmin
        cmin = 0
    if rmax > img_width:
        rmin -= rmax - img_width
        rmax = img_width
    if cmax > img_length:
        cmin -= cmax - img_length
        cmax = img_length
    return [int(rmin), int(rmax), int(cmin), int(cmax)]


def get_bbox(label, size_ratio=1.0):
    """data_utils.py:131-165: square box around the non-zero pixels of a mask -> [rmin, rmax, cmin, cmax]."""
    img_width, img_length = label.shape
    rows, cols = np.any(label, axis=1), np.any(label, axis=0)
    rmin, rmax = np.where(rows)[0][[0, -1]]
    cmin, cmax = np.where(cols)[0][[0, -1]]
    return _square(int(rmin), int(rmax) + 1, int(cmin), int(cmax) + 1, img_width, img_length, size_ratio)


def get_square_bbox(bbox, img_size, size_ratio=1.0):
    """data_utils.py:167-196."""
    return _square(bbox[0], bbox[1], bbox[2], bbox[3], img_size[0], img_size[1], size_ratio)


def resize_linear(img, S):
    """(h, w, c) float -> (S, S, c): cv2.resize(..., (S, S), interpolation=cv2.INTER_LINEAR) on a float image."""
    h, w = img.shape[:2]

    def taps(n):
        f = (np.arange(S, dtype=np.float64) + 0.5) * (n / S) - 0.5
        i0 = np.floor(f).astype(np.int64)
        fr = f - i0
        lo = i0 < 0
        i0[lo], fr[lo] = 0, 0.0
        hi = i0 >= n - 1
        i0[hi], fr[hi] = n - 1, 0.0
        return i0, np.minimum(i0 + 1, n - 1), fr

    y0, y1, fy = taps(h)
    x0, x1, fx = taps(w)
    img = img.astype(np.float64)
    top = img[y0][:, x0] * (1 - fx)[None, :, None] + img[y0][:, x1] * fx[None, :, None]
    bot = img[y1][:, x0] * (1 - fx)[None, :, None] + img[y1][:, x1] * fx[None, :, None]
    return top * (1 - fy)[:, None, None] + bot * fy[:, None, None]


def resize_nearest(img, S):
    """(h, w) -> (S, S): cv2.INTER_NEAREST."""
    h, w = img.shape[:2]
    yi = np.minimum(np.floor(np.arange(S) * (h / S)).astype(np.int64), h - 1)
    xi = np.minimum(np.floor(np.arange(S) * (w / S)).astype(np.int64), w - 1)
    return img[yi][:, xi]


def crop_instance(image_rgb_u8, mask_u8, det_bbox_xywh, img_size=224, pts_size=64, minimum_n_point=8, rgb_mask_flag=False):
    """bop_test_dataset.py:162-207 for one detection: full image (H, W, 3) uint8 as imageio loads it, full-frame binary
    mask (H, W), detection box [x, y, w, h] -> dict(rgb (3,S,S) f32, mask (S,S) f32, bbox, M (3,3) f32, pts2d (P,P,2) f64)."""
    h, w = mask_u8.shape
    if np.sum(mask_u8) > minimum_n_point:
        bbox = get_bbox(mask_u8)                                                        # :170-171
        y1, y2, x1, x2 = bbox
    else:   # :172-173 — `bbox` is NOT reassigned here: it stays the detection's [x, y, w, h] and feeds M_crop below
        b = bbox = list(det_bbox_xywh)
        y1, y2, x1, x2 = get_square_bbox([b[1], b[1] + b[3], b[0], b[0] + b[2]], (h, w))
    m = mask_u8[y1:y2, x1:x2]
    rgb = image_rgb_u8.astype(np.uint8)[..., ::-1][y1:y2, x1:x2, :3] / 255.0         # data_utils.py:245 (channel flip)
    if rgb_mask_flag:
        rgb = rgb * (m[:, :, None] > 0).astype(np.uint8)
    rgb = resize_linear(rgb, img_size)
    mask = resize_nearest(m.astype(np.int64), img_size)
    rgb = ((rgb.transpose(2, 0, 1) - CLIP_MEAN[:, None, None]) / CLIP_STD[:, None, None]).astype(np.float32)  # ToTensor + Normalize
    M_crop = np.array([[1, 0, -bbox[2]], [0, 1, -bbox[0]], [0, 0, 1]], dtype=np.float32)
    M_resize = np.array([[img_size / (y2 - y1), 0, 0], [0, img_size / (x2 - x1), 0], [0, 0, 1]], dtype=np.float32)
    M = M_resize @ M_crop
    patch = img_size / pts_size                                                          # torch_utils.py:287-295
    x = np.arange(0, img_size, patch, dtype=np.float32) + patch / 2
    yy, xx = np.meshgrid(x, x, indexing="ij")
    pts = np.concatenate((np.stack([yy, xx], axis=2), np.ones((pts_size, pts_size, 1))), axis=2)
    p = np.linalg.inv(M) @ pts.reshape(-1, 3).transpose(1, 0)
    pts2d = (p[:2] / p[2:]).transpose(1, 0).reshape(pts_size, pts_size, 2)
    return {"rgb": rgb, "mask": mask.astype(np.float32), "bbox": bbox, "M": M, "pts2d": pts2d}


def crop_template(rgba_u8, depth_mm, K, object_pose_mm, img_size=224, pts_size=64, rgb_mask_flag=False):
    """bop_test_dataset.py:210-264 (`_get_template`) for one rendered view; file loading left to the caller."""
    rgb = rgba_u8[..., :3]
    mask = (rgba_u8[..., 3] / 255).astype(np.float32)
    y1, y2, x1, x2 = bbox = get_bbox(mask)
    mask = mask[y1:y2, x1:x2]
    depth = depth_mm / 1000.0
    # data_utils.py:97-115 on the crop
    ys, xs = np.mgrid[y1:y2, x1:x2]
    pt2 = depth[y1:y2, x1:x2].astype(np.float32)
    pt0 = (xs.astype(np.float32) - np.float32(K[0, 2])) * pt2 / np.float32(K[0, 0])
    pt1 = (ys.astype(np.float32) - np.float32(K[1, 2])) * pt2 / np.float32(K[1, 1])
    cloud = np.stack([pt0, pt1, pt2]).transpose((1, 2, 0))
    pts = resize_nearest(cloud, pts_size)
    rgbf = rgb[..., ::-1][y1:y2, x1:x2, :] / 255.0
    if rgb_mask_flag:
        rgbf = rgbf * (mask[:, :, None] > 0).astype(np.uint8)
    rgbf = resize_linear(rgbf, img_size)
    mask_r = resize_nearest(mask.astype(int), img_size)
    rgbn = ((rgbf.transpose(2, 0, 1) - CLIP_MEAN[:, None, None]) / CLIP_STD[:, None, None]).astype(np.float32)
    pose = np.array(object_pose_mm, dtype=np.float64)
    pose[:3, 3] = pose[:3, 3] / 1000.0
    M_crop = np.array([[1, 0, -bbox[2]], [0, 1, -bbox[0]], [0, 0, 1]], dtype=np.float32)
    M_resize = np.array([[img_size / (y2 - y1), 0, 0], [0, img_size / (x2 - x1), 0], [0, 0, 1]], dtype=np.float32)
    return {"rgb": rgbn, "mask": mask_r.astype(np.float32), "pts3d": pts.astype(np.float32), "bbox": bbox,
            "M": M_resize @ M_crop, "pose": pose.astype(np.float32)}
