"""Restatement of the reference's mask post-processing.

TEST INFRASTRUCTURE.  Each function cites the measurement.py lines it follows; cv2 calls are
restated in numpy (cv2.resize INTER_NEAREST, cv2.bitwise_or, cv2.moments on a binary image).
"""
import numpy as np


def roi_keep(xyxy, h, w, roi=(10, 300, 1270, 760)):
    """measurement.py:224-229,251-260: int-truncate the box, keep iff its centre is inside
    the ROI clamped to the frame.  xyxy f32 [N,4] -> bool [N], int boxes [N,4]."""
    x_min = max(0, min(int(roi[0]), w - 1))
    y_min = max(0, min(int(roi[1]), h - 1))
    x_max = max(0, min(int(roi[2]), w - 1))
    y_max = max(0, min(int(roi[3]), h - 1))
    ib = np.trunc(np.asarray(xyxy, dtype=np.float64)).astype(np.int64)   # python int() truncates
    if not (x_min < x_max and y_min < y_max):
        return np.ones(len(ib), bool), ib
    cx = 0.5 * (ib[:, 0] + ib[:, 2])
    cy = 0.5 * (ib[:, 1] + ib[:, 3])
    keep = (x_min <= cx) & (cx <= x_max) & (y_min <= cy) & (cy <= y_max)
    return keep, ib


def resize_nearest(arr, w, h):
    """cv2.resize(arr, (w,h), interpolation=cv2.INTER_NEAREST) (measurement.py:79): OpenCV's
    resizeNN takes src = min(floor(dst * ifx), src_size-1) with ifx = 1./(dst_size/src_size)
    in double, no half-pixel offset."""
    sh, sw = arr.shape
    ify, ifx = 1.0 / (h / sh), 1.0 / (w / sw)
    ys = np.minimum(np.floor(np.arange(h) * ify).astype(np.int64), sh - 1)
    xs = np.minimum(np.floor(np.arange(w) * ifx).astype(np.int64), sw - 1)
    return arr[ys][:, xs]


def instance_bitmap(mask_f, h, w):
    """measurement.py:70-86: masks.data[idx] -> (nearest-resized to frame) > 0 -> u8, None if empty."""
    arr = np.asarray(mask_f)
    if arr.shape != (h, w):
        arr = resize_nearest(arr, w, h)
    m = (arr > 0).astype(np.uint8)
    return m if np.count_nonzero(m) > 0 else None


def combine_masks(mask_list, h, w):
    """measurement.py:160-168."""
    if not mask_list:
        return None
    out = np.zeros((h, w), np.uint8)
    for m in mask_list:
        if m is not None and m.shape == (h, w):
            out |= m.astype(np.uint8)
    return out


def lower_envelope(mask):
    """measurement.py:170-185: per column the largest y with mask>0, else -1."""
    h, w = mask.shape
    env = np.full((w,), -1, dtype=np.int64)
    rev = mask[::-1, :] > 0
    has = rev.any(axis=0)
    idx = np.argmax(rev, axis=0)
    env[has] = h - 1 - idx[has]
    return env


def stitch_stats(mask, box):
    """measurement.py:302-323: (cx, cy, px_width, left_px, right_px) for one stitch;
    `mask` may be None; `box` = int (x1,y1,x2,y2)."""
    x1, y1, x2, y2 = box
    if mask is not None and mask.sum() > 0:
        ys, xs = np.nonzero(mask > 0)
        m00 = float(len(xs))                      # cv2.moments of a 0/1 image
        cx, cy = float(xs.sum() / m00), float(ys.sum() / m00)
        left, right = float(xs.min()), float(xs.max())
        return cx, cy, right - left, left, right
    return float((x1 + x2) / 2), float((y1 + y2) / 2), float(x2 - x1), float(x1), float(x2)
