"""Oracle checks on CPU: hand-computed fixtures (tests/golden/hand_cases.json) and the committed
regression vectors (tests/golden/oracle_regression.npz, made by tests/golden/make_golden.py).

The reference itself has no tests or fixtures and cannot be imported here, so these
pin the RESTATEMENT: parity with the reference stays "unpinned" (DESIGN.md)."""
import json
import os

import numpy as np
import torch

from oracle import consumer as oc
from oracle.letterbox import letterbox, letterbox_geometry, resize_linear_u8
from oracle.postproc import nms_torchvision, non_max_suppression, process_mask, scale_boxes

HERE = os.path.dirname(os.path.abspath(__file__))
HAND = json.load(open(os.path.join(HERE, "golden", "hand_cases.json")))
REG = np.load(os.path.join(HERE, "golden", "oracle_regression.npz"))


def _pred_from_boxes(boxes, scores, classes, nc, nm):
    A = len(boxes)
    pred = np.zeros((1, 4 + nc + nm, A), np.float32)
    b = np.asarray(boxes, np.float32)
    pred[0, 0] = (b[:, 0] + b[:, 2]) / 2
    pred[0, 1] = (b[:, 1] + b[:, 3]) / 2
    pred[0, 2] = b[:, 2] - b[:, 0]
    pred[0, 3] = b[:, 3] - b[:, 1]
    for a, (s, c) in enumerate(zip(scores, classes)):
        pred[0, 4 + c, a] = s
    pred[0, 4 + nc:] = np.arange(nm * A, dtype=np.float32).reshape(nm, A)
    return pred


def test_nms_hand_case():
    h = HAND["nms"]
    pred = _pred_from_boxes(h["boxes_xyxy"], h["scores"], h["classes"], nc=2, nm=2)
    out = non_max_suppression(pred, h["conf"], h["iou"], 300, nc=2)[0]
    exp = h["keep_class_aware"]
    assert out.shape == (len(exp), 8)
    assert np.allclose(out[:, :4], np.asarray(h["boxes_xyxy"], np.float32)[exp])
    assert np.allclose(out[:, 4], np.asarray(h["scores"], np.float32)[exp])
    assert out[:, 5].tolist() == [h["classes"][i] for i in exp]
    assert np.array_equal(out[:, 6:], pred[0, 6:, exp])          # coefficients ride along
    out = non_max_suppression(pred, h["conf"], h["iou"], 300, nc=2, agnostic=True)[0]
    assert np.allclose(out[:, 4], np.asarray(h["scores"], np.float32)[h["keep_agnostic"]])
    assert len(non_max_suppression(pred, h["conf"], h["iou"], 2, nc=2)[0]) == 2     # max_det
    assert len(non_max_suppression(pred, 0.95, h["iou"], 300, nc=2)[0]) == 0        # nothing above conf


def test_nms_ties_keep_input_order_and_strict_threshold():
    boxes = np.array([[0, 0, 10, 10], [100, 100, 110, 110], [0, 0, 10, 10]], np.float32)
    keep = nms_torchvision(boxes, np.array([0.5, 0.5, 0.5], np.float32), 0.5)
    assert keep.tolist() == [0, 1]                      # stable: first of the tied duplicates wins
    # IoU exactly == threshold is NOT suppressed (strict >): inter 1, union 2 -> IoU 0.5 exactly
    b = np.array([[0, 0, 2, 1], [0, 0, 1, 1]], np.float32)
    assert nms_torchvision(b, np.array([0.9, 0.8], np.float32), 0.5).tolist() == [0, 1]
    assert nms_torchvision(b, np.array([0.9, 0.8], np.float32), 0.49).tolist() == [0]


def test_letterbox_hand_cases():
    h = HAND["letterbox"]
    g = letterbox_geometry(*h["frame_hw"], h["imgsz"])
    for k, v in h["expect"].items():
        assert g[k] == v, k
    assert letterbox_geometry(640, 640, 640)["top"] == 0 and letterbox_geometry(640, 640, 640)["H"] == 640
    r = HAND["resize_2x2_to_4x4"]
    src = np.asarray(r["src"], np.uint8)[:, :, None]
    dst = resize_linear_u8(src, 4, 4)[:, :, 0]
    assert dst[0].tolist() == r["dst_row0"]
    img = np.full((960, 1280, 3), 7, np.uint8)
    out, _ = letterbox(img, 960)
    assert out.shape == (736, 960, 3) and (out[:8] == 114).all() and (out[-8:] == 114).all() and (out[8:-8] == 7).all()


def test_process_mask_hand_case():
    h = HAND["process_mask"]
    H, W = h["H"], h["W"]
    proto = np.zeros((4, H // 4, W // 4), np.float32)
    proto[1] = 1.0
    coeff = np.array([[0, 1, 0, 0]], np.float32)
    box = np.array([h["box_xyxy"]], np.float32)
    for mode, key in (("logit", "logit_rows"), ("sigmoid", "sigmoid_rows")):
        m = process_mask(proto, coeff, box, (H, W), mode)[0].numpy()
        lo, hi = h[key]
        exp = np.zeros((H, W), np.float32)
        exp[lo:hi, lo:hi] = 1
        assert np.array_equal(m, exp), mode


def test_scale_boxes_and_consumer_hand_cases():
    out = scale_boxes((736, 960), np.array([[0, 8, 960, 728]], np.float32), (960, 1280))
    assert np.allclose(out, [[0, 0, 1280, 960]])          # pad 8 rows removed, /0.75
    c = HAND["consumer"]
    m = np.asarray(c["mask"], np.uint8)
    assert oc.lower_envelope(m).tolist() == c["envelope"]
    cx, cy, pw, left, right = oc.stitch_stats(m, (0, 0, 5, 3))
    assert (cx, cy) == (c["m10"] / c["m00"], c["m01"] / c["m00"]) and (left, right, pw) == (c["min_col"], c["max_col"], 3.0)
    assert oc.stitch_stats(None, (2, 4, 10, 8)) == (6.0, 6.0, 8.0, 2.0, 10.0)     # fall back to the box
    assert oc.instance_bitmap(np.zeros((4, 4), np.float32), 8, 8) is None
    big = oc.instance_bitmap(m.astype(np.float32), 8, 12)                         # nearest x2
    assert big.shape == (8, 12) and big.sum() == 4 * m.sum() and big[2, 2] == 1 and big[0, 0] == 0
    assert np.array_equal(oc.combine_masks([m, None, m], 4, 6), m)
    keep, ib = oc.roi_keep(np.array([[100.9, 400.2, 200.7, 500.9], [0, 0, 50, 50]], np.float32), 960, 1280)
    assert keep.tolist() == [True, False] and ib[0].tolist() == [100, 400, 200, 500]


def test_regression_vectors():
    """The committed oracle outputs are reproduced bit for bit (guards the oracle against drift)."""
    lb, _ = letterbox(REG["lb_frame"], 96)
    assert np.array_equal(lb, REG["lb_out"])
    det = non_max_suppression(REG["nms_pred"], 0.25, 0.45, 20, nc=3)
    for b, d in enumerate(det):
        assert np.array_equal(d, REG[f"nms_det{b}"])
        assert (np.diff(d[:, 4]) <= 0).all()            # conf-descending
    for mode in ("logit", "sigmoid"):
        m = process_mask(REG["pm_proto"], REG["pm_coeff"], REG["pm_boxes"], (160, 160), mode).numpy().astype(np.uint8)
        assert np.array_equal(m, REG[f"pm_{mode}"])
    assert np.array_equal(oc.resize_nearest(REG["nn_in"], 40, 31), REG["nn_out"])
    assert np.array_equal(oc.lower_envelope(REG["nn_out"].astype(np.uint8)), REG["env"])
    sb = scale_boxes((736, 960), np.array([[10.5, 20.25, 950.0, 730.0], [-5, 3, 400, 800]], np.float32), (960, 1280))
    assert np.array_equal(sb, REG["sb_out"])
