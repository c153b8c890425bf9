#!/usr/bin/env python3
"""Diagnostic (GPU): the body of tests/test_gpu_predict.py::test_drop_empty_masks_flag_and_output_reuse in a loop inside ONE
process, reporting WHICH comparison differs when one does (the test failed once in a combined run and passed alone).

    python tests/dev/flake_probe.py [iterations]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

import vti_amd  # noqa: E402
from gpu_util import frames_u8  # noqa: E402
from test_gpu_predict import _calibrated_model  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    fr = frames_u8(2, 640, 640, seed=41)
    bad = 0
    for it in range(iters):
        # some unrelated allocations of varying size between rounds, so that torch.empty() hands out recycled (dirty) blocks
        junk = [torch.randint(0, 255, (int(1e6) * (1 + (it + k) % 7),), dtype=torch.uint8, device="cuda") for k in range(4)]
        del junk
        base = _calibrated_model(vti_amd, 80, "h2", fr[0], 640, 0.25, target=300)
        plain = base.predict(fr, conf=0.25, iou=0.7)
        snap = [(r.boxes.data.clone(), r.masks.data_u8.clone()) for r in plain]
        dm = vti_amd.YOLO(base._blob, dtype="h2", max_batch=2, drop_empty_masks=True)
        drop = dm.predict(fr, conf=0.25, iou=0.7)
        again = base.predict(fr[::-1].copy(), conf=0.25, iou=0.7)
        msgs = []
        for b, (r, (bx, mk), d) in enumerate(zip(plain, snap, drop)):
            if not torch.equal(r.boxes.data, bx):
                msgs.append(f"frame {b}: earlier Results' boxes changed")
            if not torch.equal(r.masks.data_u8, mk):
                msgs.append(f"frame {b}: earlier Results' masks changed")
            keep = mk.flatten(1).any(1).bool()
            if len(d.boxes) != int(keep.sum()):
                msgs.append(f"frame {b}: drop kept {len(d.boxes)} rows, non-empty masks {int(keep.sum())} of {len(keep)}")
            elif not torch.equal(d.boxes.data, bx[keep]):
                msgs.append(f"frame {b}: drop boxes differ (max |d| {(d.boxes.data - bx[keep]).abs().max().item():.3e})")
            elif not torch.equal(d.masks.data_u8, mk[keep]):
                msgs.append(f"frame {b}: drop masks differ in {int((d.masks.data_u8 != mk[keep]).flatten(1).any(1).sum())} instances")
        if again[1].boxes.data.shape != snap[0][0].shape or not torch.equal(again[1].boxes.data, snap[0][0]):
            msgs.append(f"again[1] boxes {tuple(again[1].boxes.data.shape)} vs snap[0] {tuple(snap[0][0].shape)}")
        if msgs:
            bad += 1
            print(f"iter {it}: " + "; ".join(msgs), flush=True)
    print(f"{bad} of {iters} iterations differed", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
