"""End-to-end parity checker: the oracle's whole predict pipeline on a few frames, and the comparison of an engine's
detections + masks with it (north_star: mask IoU >= 0.999, |d box| < 1e-3).

TEST INFRASTRUCTURE -- PARITY UNPINNED (see oracle/__init__.py).  Used by tests/, __graft_entry__.smoke() and bench.py
(outside the timed region, as the checker only).  Restates what `model.predict(...)` returns at measurement.py:208-211:
rows [x1,y1,x2,y2,conf,cls] conf-descending + one 0/1 mask per row at the letterboxed size.
"""
import numpy as np

from .model import OracleModel
from .postproc import non_max_suppression, process_mask

IOU_GATE = 0.999        # north_star: mask IoU per instance
BOX_NORM_GATE = 1e-3    # north_star: |d box|, read as normalised by max(H, W) (SURVEY 8d also reports pixels)


def oracle_predict(blob, frames, nc, conf, iou, max_det, mode="fp32", mask_mode="logit", model=None):
    """frames u8 [B,H,W,3] already letterboxed -> list of (det f32 [n,6+nm], masks u8 [n,H,W])."""
    B, H, W, _ = frames.shape
    om = model or OracleModel(blob, H, W, mode)
    pred, proto = om.forward_u8(frames, swap_rb=True)
    dets = non_max_suppression(pred.numpy(), conf, iou, max_det, nc=nc)
    out = []
    for b, d in enumerate(dets):
        if len(d):
            m = process_mask(proto[b], d[:, 6:], d[:, :4], (H, W), mask_mode).numpy().astype(np.uint8)
        else:
            m = np.zeros((0, H, W), np.uint8)
        out.append((d, m))
    return out


def _iou(a, b):
    a, b = a > 0, b > 0
    u = np.logical_or(a, b).sum()
    return 1.0 if u == 0 else float(np.logical_and(a, b).sum() / u)


def compare(got, want, H, W, match_px=4.0):
    """got / want: lists (one entry per frame) of (det [n,6+nm], masks [n,H,W]).  A wanted instance is MATCHED by the
    unused got-instance of the same class whose box is nearest (max |d coordinate| < match_px).  Returns the figures the
    north_star gates are written in, over all matched instances, and whether the kept sets agree row for row."""
    n_want = n_got = n_matched = 0
    box_px = conf_d = 0.0
    ious = []
    same_order = True
    for (gd, gm), (wd, wm) in zip(got, want):
        n_want += len(wd); n_got += len(gd)
        used = np.zeros(len(gd), bool)
        for i in range(len(wd)):
            if len(gd) == 0:
                same_order = False
                continue
            d = np.abs(gd[:, :4] - wd[i, :4]).max(1)
            d[used | (gd[:, 5] != wd[i, 5])] = np.inf
            j = int(d.argmin())
            if not d[j] < match_px:
                same_order = False
                continue
            used[j] = True
            same_order &= (j == i)
            n_matched += 1
            box_px = max(box_px, float(d[j]))
            conf_d = max(conf_d, float(abs(gd[j, 4] - wd[i, 4])))
            ious.append(_iou(gm[j], wm[i]))
    kept_equal = bool(same_order and n_want == n_got == n_matched)
    res = dict(n_instances=n_want, n_engine=n_got, n_matched=n_matched, kept_set_equal=kept_equal,
               box_px_max=round(box_px, 6), box_norm_max=round(box_px / max(H, W), 8), conf_abs_max=round(conf_d, 6),
               mask_iou_min=round(min(ious), 6) if ious else None,
               mask_iou_p1=round(float(np.percentile(ious, 1)), 6) if ious else None,
               mask_iou_mean=round(float(np.mean(ious)), 6) if ious else None)
    res["meets_north_star"] = bool(kept_equal and ious and res["mask_iou_min"] >= IOU_GATE and res["box_norm_max"] < BOX_NORM_GATE)
    return res


def engine_predict(eng, frames_dev, conf, iou, max_det, mask_mode="logit"):
    """The engine's pipeline on device frames u8 [B,H,W,3] -> the same structure as oracle_predict (host arrays)."""
    # the same entry-point pair the benchmark's timed loop uses: vti_forward_scored -> vti_nms_scored
    best = eng.alloc_best(frames_dev.shape[0], frames_dev.device)
    pred, proto = eng.forward(frames_dev, True, best=best)
    dets, counts = eng.nms(pred, conf, iou, max_det, best=best)
    masks, offsets = eng.masks(dets, counts, proto, mask_mode, "u8")
    cnt, off = counts.cpu().tolist(), offsets.cpu().tolist()
    out = []
    for b in range(frames_dev.shape[0]):
        out.append((dets[b, :cnt[b]].cpu().numpy(), masks[off[b]:off[b] + cnt[b]].cpu().numpy()))
    return out
