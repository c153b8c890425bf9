"""Generates tests/golden/oracle_regression.npz: seeded inputs -> oracle outputs, so that a change to
the oracle itself is noticed.  (The reference cannot be run here -- ultralytics/cv2/torchvision are not
installed and its weights are absent -- so these vectors pin the RESTATEMENT, not the reference:
"parity unpinned", see DESIGN.md.)   Run:  python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.letterbox import letterbox           # noqa: E402
from oracle.postproc import non_max_suppression, process_mask, scale_boxes  # noqa: E402
from oracle.consumer import resize_nearest, lower_envelope  # noqa: E402


def synth_pred(rng, B, nc, nm, A, n_inst=6, dup=3):
    """Planted instances + jittered duplicates + low background."""
    pred = np.zeros((B, 4 + nc + nm, A), np.float32)
    pred[:, 4:4 + nc] = rng.uniform(0, 0.05, (B, nc, A)).astype(np.float32)
    pred[:, :2] = rng.uniform(20, 140, (B, 2, A)).astype(np.float32)
    pred[:, 2:4] = rng.uniform(4, 30, (B, 2, A)).astype(np.float32)
    pred[:, 4 + nc:] = rng.standard_normal((B, nm, A)).astype(np.float32)
    for b in range(B):
        slots = rng.choice(A, n_inst * dup, replace=False)
        for i in range(n_inst):
            cx, cy = rng.uniform(30, 130, 2)
            w, h = rng.uniform(12, 40, 2)
            c = int(rng.integers(nc))
            for d in range(dup):
                a = slots[i * dup + d]
                pred[b, :4, a] = [cx + rng.normal(0, 0.6), cy + rng.normal(0, 0.6), w + rng.normal(0, 0.6), h + rng.normal(0, 0.6)]
                pred[b, 4 + c, a] = rng.uniform(0.5, 0.9)
    return pred


def main():
    rng = np.random.Generator(np.random.PCG64(2024))
    out = {}
    frame = rng.integers(0, 256, (48, 64, 3), dtype=np.uint8)
    lb, g = letterbox(frame, 96)
    out["lb_frame"], out["lb_out"] = frame, lb
    pred = synth_pred(rng, 2, 3, 4, 160)
    det = non_max_suppression(pred, 0.25, 0.45, 20, nc=3)
    out["nms_pred"] = pred
    for b, d in enumerate(det):
        out[f"nms_det{b}"] = d
    proto = rng.standard_normal((4, 40, 40)).astype(np.float32)
    boxes = det[0][:, :4]
    out["pm_proto"], out["pm_coeff"], out["pm_boxes"] = proto, det[0][:, 6:], boxes
    out["pm_logit"] = process_mask(proto, det[0][:, 6:], boxes, (160, 160), "logit").numpy().astype(np.uint8)
    out["pm_sigmoid"] = process_mask(proto, det[0][:, 6:], boxes, (160, 160), "sigmoid").numpy().astype(np.uint8)
    out["sb_out"] = scale_boxes((736, 960), np.array([[10.5, 20.25, 950.0, 730.0], [-5, 3, 400, 800]], np.float32), (960, 1280))
    m = (rng.uniform(size=(23, 30)) > 0.6).astype(np.float32)
    out["nn_in"], out["nn_out"] = m, resize_nearest(m, 40, 31)
    out["env"] = lower_envelope(out["nn_out"].astype(np.uint8))
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_regression.npz"), **out)
    print("wrote oracle_regression.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
