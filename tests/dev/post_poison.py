#!/usr/bin/env python3
"""Diagnostic (GPU): the whole predict pipeline (vti_predict + vti_mask_stats_bits) into output sets whose EVERY buffer was
pre-filled with a poison byte, N times, compared with the first run -- dets rows below the count, counts, offsets, xyxy, the live
mask slots and m00.  A post-processing kernel that reads memory nobody wrote (or races) shows up as a mismatch.

    python tests/dev/post_poison.py [--dtype h2] [--batch 2] [--runs 40] [--target 300]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

import vti_amd  # noqa: E402
from gpu_util import frames_u8  # noqa: E402
from test_gpu_predict import _calibrated_model  # noqa: E402


def live(o, B):
    cnt, off = o["counts"].cpu().tolist(), o["offsets"].cpu().tolist()
    return dict(counts=cnt, offsets=off,
                dets=[o["dets"][b, :cnt[b]].clone() for b in range(B)],
                xyxy=[o["xyxy"][b, :cnt[b]].clone() for b in range(B)],
                masks=o["masks"][:off[B]].clone())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="h2")
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--runs", type=int, default=40)
    ap.add_argument("--target", type=int, default=300)
    args = ap.parse_args()
    B = args.batch
    fr = frames_u8(B, 640, 640, seed=41)
    model = _calibrated_model(vti_amd, 80, args.dtype, fr[0], 640, 0.25, target=args.target)
    eng = model._engine(640, 640, B)
    x = torch.from_numpy(fr).cuda()
    ref = None
    bad = 0
    for r in range(args.runs):
        fill = (0x00, 0xFF, 0x7C, 0xFB, 0x3C)[r % 5]
        o = eng.alloc_outputs(B, 300, B * 300, "bits", x.device)
        for v in o.values():
            v.view(torch.uint8).fill_(fill) if v.is_contiguous() else v.fill_(float("nan"))
        eng.predict_into(x, o, 0.25, 0.7, 300, False, True, "logit", "bits")
        stats = eng.mask_stats_bits(o["masks"], 640, 640, offsets=o["offsets"])
        torch.cuda.synchronize()
        cur = live(o, B)
        cur["m00"] = stats[:cur["offsets"][B], 0].clone()
        cur["dead_m00"] = int(stats[cur["offsets"][B]:, 0].abs().sum())
        if ref is None:
            ref = cur
            print(f"reference run: counts {cur['counts']}, empty masks {int((cur['m00'] == 0).sum())}", flush=True)
            continue
        msgs = []
        if cur["counts"] != ref["counts"] or cur["offsets"] != ref["offsets"]:
            msgs.append(f"counts {cur['counts']} vs {ref['counts']}, offsets {cur['offsets']} vs {ref['offsets']}")
        else:
            for b in range(B):
                if not torch.equal(cur["dets"][b], ref["dets"][b]):
                    msgs.append(f"frame {b}: dets differ")
                if not torch.equal(cur["xyxy"][b], ref["xyxy"][b]):
                    msgs.append(f"frame {b}: xyxy differ")
            if not torch.equal(cur["masks"], ref["masks"]):
                d = (cur["masks"] != ref["masks"]).flatten(1).any(1).nonzero().flatten().tolist()
                msgs.append(f"mask slots differ: {d[:10]}")
            if not torch.equal(cur["m00"], ref["m00"]):
                msgs.append("m00 differ")
        if cur["dead_m00"]:
            msgs.append("dead slots report a non-empty mask")
        if msgs:
            bad += 1
            print(f"run {r} (fill 0x{fill:02X}): " + "; ".join(msgs), flush=True)
    print(f"{args.dtype} B={B}: {bad} of {args.runs - 1} poisoned runs differed from the first", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
