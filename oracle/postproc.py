"""Restatement of Ultralytics' post-processing: non_max_suppression (incl. torchvision.ops.nms
CPU semantics), process_mask, crop_mask, scale_boxes, clip_boxes.

TEST INFRASTRUCTURE -- PARITY UNPINNED (see oracle/__init__.py).  SURVEY.md section 8 rows
U6-U8; reached by the reference through measurement.py:208-210 with
conf=0.20, iou=0.25, max_det=200 (config.py:71-73).
"""
import numpy as np
import torch
import torch.nn.functional as F

MAX_WH = 7680.0     # Ultralytics: class offset for batched NMS
MAX_NMS = 30000     # Ultralytics: cap on boxes entering torchvision.ops.nms


def nms_torchvision(boxes, scores, iou_thres):
    """torchvision.ops.nms CPU kernel semantics, fp32: stable sort by score descending,
    suppress j when inter/(area_i+area_j-inter) > iou_thres; returns kept indices in
    decreasing-score order."""
    boxes = np.asarray(boxes, dtype=np.float32)
    scores = np.asarray(scores, dtype=np.float32)
    n = boxes.shape[0]
    if n == 0:
        return np.zeros((0,), dtype=np.int64)
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    order = np.argsort(-scores, kind="stable")
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    thr = float(iou_thres)   # the C++ kernel takes a double and compares float ovr > double thr
    for _i in range(n):
        i = order[_i]
        if suppressed[i]:
            continue
        keep.append(i)
        rest = order[_i + 1:]
        xx1 = np.maximum(x1[i], x1[rest])
        yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest])
        yy2 = np.minimum(y2[i], y2[rest])
        w = np.maximum(np.float32(0), xx2 - xx1)
        h = np.maximum(np.float32(0), yy2 - yy1)
        inter = w * h
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = inter / (areas[i] + areas[rest] - inter)
        suppressed[rest[ovr.astype(np.float64) > thr]] = True
    return np.asarray(keep, dtype=np.int64)


def xywh2xyxy(x):
    y = np.empty_like(x)
    xy = x[..., :2]
    wh = x[..., 2:4] / np.float32(2)
    y[..., :2] = xy - wh
    y[..., 2:4] = xy + wh
    return y


def non_max_suppression(pred, conf_thres=0.25, iou_thres=0.45, max_det=300, nc=80, agnostic=False):
    """pred: f32 [B,4+nc+nm,A] -> list of f32 [n_i, 6+nm] rows [x1,y1,x2,y2,conf,cls,coeffs...],
    conf-descending (no wall-clock time_limit: that early exit is nondeterministic)."""
    pred = np.asarray(pred, dtype=np.float32)
    B, no, A = pred.shape
    nm = no - 4 - nc
    mi = 4 + nc
    out = []
    conf_t = np.float32(conf_thres)
    for b in range(B):
        x = pred[b].T.copy()                       # [A, no]
        xc = x[:, 4:mi].max(1) > conf_t
        x[:, :4] = xywh2xyxy(x[:, :4])
        x = x[xc]
        if x.shape[0] == 0:
            out.append(np.zeros((0, 6 + nm), np.float32))
            continue
        box, cls, mask = x[:, :4], x[:, 4:mi], x[:, mi:]
        j = cls.argmax(1)                          # first maximal index, as torch.max
        conf = cls[np.arange(cls.shape[0]), j]
        x = np.concatenate((box, conf[:, None], j[:, None].astype(np.float32), mask), 1)
        x = x[conf > conf_t]
        if x.shape[0] > MAX_NMS:
            x = x[np.argsort(-x[:, 4], kind="stable")[:MAX_NMS]]
        c = x[:, 5:6] * np.float32(0.0 if agnostic else MAX_WH)
        keep = nms_torchvision(x[:, :4] + c, x[:, 4], iou_thres)[:max_det]
        out.append(x[keep])
    return out


def crop_mask(masks, boxes):
    """masks f32 [n,h,w], boxes f32 [n,4] in mask pixels: zero everything outside the box."""
    n, h, w = masks.shape
    x1, y1, x2, y2 = torch.chunk(boxes[:, :, None], 4, 1)
    r = torch.arange(w, dtype=x1.dtype)[None, None, :]
    c = torch.arange(h, dtype=x1.dtype)[None, :, None]
    return masks * ((r >= x1) * (r < x2) * (c >= y1) * (c < y2))


@torch.inference_mode()
def process_mask(proto, coeffs, boxes, shape, mode="logit"):
    """proto f32 [nm,mh,mw], coeffs f32 [n,nm], boxes f32 [n,4] in letterboxed pixels,
    shape=(H,W) of the letterboxed input -> f32 0/1 masks [n,H,W].
    mode="logit"  : current Ultralytics -- crop, bilinear upsample of the logits, > 0.0
    mode="sigmoid": Ultralytics 8.0.x  -- sigmoid, crop, bilinear upsample, > 0.5"""
    proto = torch.as_tensor(proto, dtype=torch.float32)
    coeffs = torch.as_tensor(coeffs, dtype=torch.float32)
    boxes = torch.as_tensor(boxes, dtype=torch.float32)
    c, mh, mw = proto.shape
    ih, iw = shape
    masks = coeffs @ proto.view(c, -1)
    if mode == "sigmoid":
        masks = masks.sigmoid()
    masks = masks.view(-1, mh, mw)
    wr, hr = mw / iw, mh / ih
    db = boxes.clone()
    db[:, 0] *= wr
    db[:, 2] *= wr
    db[:, 3] *= hr
    db[:, 1] *= hr
    masks = crop_mask(masks, db)
    masks = F.interpolate(masks[None], (ih, iw), mode="bilinear", align_corners=False)[0]
    return masks.gt_(0.5 if mode == "sigmoid" else 0.0)


def scale_boxes(img1_shape, boxes, img0_shape):
    """Letterboxed (img1) -> original (img0) pixel coords, clipped.  boxes f32 [n,4] xyxy."""
    boxes = np.array(boxes, dtype=np.float32, copy=True)
    gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
    pad = (round((img1_shape[1] - img0_shape[1] * gain) / 2 - 0.1),
           round((img1_shape[0] - img0_shape[0] * gain) / 2 - 0.1))
    boxes[:, 0] -= np.float32(pad[0])
    boxes[:, 2] -= np.float32(pad[0])
    boxes[:, 1] -= np.float32(pad[1])
    boxes[:, 3] -= np.float32(pad[1])
    boxes[:, :4] /= np.float32(gain)
    boxes[:, 0] = boxes[:, 0].clip(0, img0_shape[1])
    boxes[:, 2] = boxes[:, 2].clip(0, img0_shape[1])
    boxes[:, 1] = boxes[:, 1].clip(0, img0_shape[0])
    boxes[:, 3] = boxes[:, 3].clip(0, img0_shape[0])
    return boxes
