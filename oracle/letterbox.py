"""Restatement of Ultralytics LetterBox + to-tensor preprocessing.

TEST INFRASTRUCTURE -- PARITY UNPINNED: cv2 is not installed here, so the u8
INTER_LINEAR resize restates OpenCV 4.x's generic fixed-point path
(imgproc/resize.cpp: HResizeLinear<uchar,int,short,2048> + VResizeLinear<uchar,int,short,
FixedPtCast<int,uchar,22>>) and cannot be cross-checked against the library.
Reference call site: measurement.py:208-210 (imgsz=960 on a 1280x960 frame -> 736x960).
"""
import numpy as np

COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS


def letterbox_geometry(H0, W0, imgsz, auto=True, stride=32, scaleup=True):
    """-> dict(r, new_w, new_h, top, bottom, left, right, H, W).  `imgsz` int or (h, w)."""
    new_shape = (imgsz, imgsz) if isinstance(imgsz, int) else tuple(imgsz)
    r = min(new_shape[0] / H0, new_shape[1] / W0)
    if not scaleup:
        r = min(r, 1.0)
    new_w, new_h = int(round(W0 * r)), int(round(H0 * r))
    dw, dh = new_shape[1] - new_w, new_shape[0] - new_h
    if auto:
        dw, dh = dw % stride, dh % stride
    dw /= 2
    dh /= 2
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    return dict(r=r, new_w=new_w, new_h=new_h, top=top, bottom=bottom, left=left, right=right,
                H=new_h + top + bottom, W=new_w + left + right)


def _linear_tables(ssize, dsize, horizontal):
    """OpenCV resize(): per-destination source index + 11-bit fixed-point weights.
    fx is computed in double, cast to float, floored.  The horizontal pass clamps
    (sx<0 -> sx=0,fx=0; sx>=ssize-1 -> sx=ssize-1,fx=0); the vertical pass keeps the
    weights and clips the two ROW indices to [0, ssize-1] instead."""
    scale = 1.0 / (dsize / ssize)              # OpenCV: scale_x = 1./inv_scale_x
    idx = np.empty(dsize, np.int64)
    a0 = np.empty(dsize, np.int32)
    a1 = np.empty(dsize, np.int32)
    for d in range(dsize):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = int(np.floor(f))
        f = np.float32(f - np.float32(s))
        if horizontal:
            if s < 0:
                s, f = 0, np.float32(0)
            if s >= ssize - 1:
                s, f = ssize - 1, np.float32(0)
        idx[d] = s
        # saturate_cast<short>(float * 2048): cvRound = round half to even
        a0[d] = int(np.rint(np.float32(np.float32(1) - f) * np.float32(COEF_SCALE)))
        a1[d] = int(np.rint(f * np.float32(COEF_SCALE)))
    return idx, a0, a1


def resize_linear_u8(img, new_w, new_h):
    """cv2.resize(img, (new_w,new_h), interpolation=cv2.INTER_LINEAR) for uint8 HxWxC.
    OpenCV (imgproc/resize.cpp, hal::resize) swaps in INTER_AREA when both scales are exactly 2
    (`interpolation == INTER_LINEAR && is_area_fast && iscale_x == 2 && iscale_y == 2`): the u8 fast path is the
    rounded 2x2 box mean (a + b + c + d + 2) >> 2 (ResizeAreaFastVec_SIMD_8u: v_rshr_pack<2>).  A 1280x1280 frame at
    imgsz=640 takes that path."""
    img = np.asarray(img, dtype=np.uint8)
    H0, W0 = img.shape[:2]
    if W0 == 2 * new_w and H0 == 2 * new_h:
        s = img.astype(np.int32)
        return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    xi, xa0, xa1 = _linear_tables(W0, new_w, True)
    yi, ya0, ya1 = _linear_tables(H0, new_h, False)
    src = img.astype(np.int32)
    x1 = np.minimum(xi + 1, W0 - 1)
    # horizontal pass: int rows scaled by 2^11
    rows = src[:, xi] * xa0[None, :, None] + src[:, x1] * xa1[None, :, None]
    r0 = np.clip(yi, 0, H0 - 1)
    r1 = np.clip(yi + 1, 0, H0 - 1)
    s0 = rows[r0] >> 4
    s1 = rows[r1] >> 4
    out = (((ya0[:, None, None] * s0) >> 16) + ((ya1[:, None, None] * s1) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def letterbox(img, imgsz, auto=True, stride=32):
    """uint8 [H0,W0,3] -> uint8 [H,W,3] padded with 114 (cv2.copyMakeBorder BORDER_CONSTANT)."""
    H0, W0 = img.shape[:2]
    g = letterbox_geometry(H0, W0, imgsz, auto, stride)
    if (W0, H0) != (g["new_w"], g["new_h"]):
        img = resize_linear_u8(img, g["new_w"], g["new_h"])
    out = np.full((g["H"], g["W"], 3), 114, np.uint8)
    out[g["top"]:g["top"] + g["new_h"], g["left"]:g["left"] + g["new_w"]] = img
    return out, g
