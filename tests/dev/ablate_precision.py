#!/usr/bin/env python3
"""CPU-only ablation (no GPU minutes): which tensors of the YOLOv8n-seg forward must carry more than fp16's 11 bits for the
whole predict pipeline to meet the north-star gate (mask IoU >= 0.999 per instance, |d box| < 1e-3 normalised, same kept
set) against the fp32 oracle?  Uses the oracle's rounding hook (oracle/model.py: q()) with a per-layer policy.

    python tests/dev/ablate_precision.py [--frames 8] [--out profiles/r03_precision_ablation.txt]

Policies round (a) the stored output of a conv and (b) its weights.  "h2" is the split-fp16 pair the h2 engine stores:
v*S = hi + lo with hi = fp16(v*S), lo = fp16(v*S - hi) (22-23 significant bits).  "mN" keeps N mantissa bits.
TEST/TOOLING ONLY: imports oracle/.
"""
import argparse
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle.model import OracleModel  # noqa: E402
from oracle import parity as op  # noqa: E402
from oracle.postproc import non_max_suppression, process_mask  # noqa: E402

CONF, IOU, MAX_DET = 0.25, 0.7, 300


def r_fp16(t):
    return t.half().float()


def r_h2(t, scale=16.0):
    s = t * scale
    hi = s.half().float()
    lo = (s - hi).half().float()
    return (hi + lo) / scale


def r_bits(n):
    def f(t):
        m, e = torch.frexp(t)
        k = float(1 << n)
        return torch.ldexp(torch.round(m * k) / k, e)
    return f


ROUND = {"fp32": lambda t: t, "fp16": r_fp16, "h2": r_h2}
for n in (12, 14, 16, 18, 20):
    ROUND[f"m{n}"] = r_bits(n)


class PolicyModel(OracleModel):
    """OracleModel with a per-conv rounding policy: policy(name) -> (activation rounding, weight rounding) names."""

    def __init__(self, blob, H, W, policy):
        super().__init__(blob, H, W, "fp32")
        self.policy = policy
        self.inp_round = ROUND[policy("input")[0]]
        for name, (w, b, k, s, kind) in list(self.p.items()):
            self.p[name] = (ROUND[policy(name)[1]](w), b, k, s, kind)
        self._cur = None

    def q(self, t):
        if self._cur is None:
            return self.inp_round(t)
        return ROUND[self.policy(self._cur)[0]](t)

    def conv(self, x, name, res=None, out_fp32=False):
        self._cur = name
        try:
            return super().conv(x, name, res, out_fp32)
        finally:
            self._cur = None


def is_head(name):
    return name.startswith("model.22.")


POLICIES = {
    # name: (description, policy)
    "fp16 everywhere (the fp16 engine)": lambda n: ("fp16", "fp16"),
    "fp16 weights, fp32 activations": lambda n: ("fp32", "fp16"),
    "fp16 activations, fp32 weights": lambda n: ("fp16", "fp32"),
    "fp16 trunk (model.0-21), fp32 head (model.22.*)": lambda n: ("fp32", "fp32") if is_head(n) else ("fp16", "fp16"),
    "fp32 trunk, fp16 head": lambda n: ("fp16", "fp16") if is_head(n) else ("fp32", "fp32"),
    "fp16 backbone (model.0-9) only": lambda n: ("fp16", "fp16") if n != "input" and not is_head(n) and int(n.split(".")[1]) <= 9 else ("fp32", "fp32"),
    "fp16 stem+layer1+C2f-2 only (model.0-2)": lambda n: ("fp16", "fp16") if n == "input" or (not is_head(n) and int(n.split(".")[1]) <= 2) else ("fp32", "fp32"),
    "fp16 class towers only (model.22.cv3.*)": lambda n: ("fp16", "fp16") if n.startswith("model.22.cv3.") else ("fp32", "fp32"),
    "12 mantissa bits everywhere": lambda n: ("m12", "m12"),
    "14 mantissa bits everywhere": lambda n: ("m14", "m14"),
    "16 mantissa bits everywhere": lambda n: ("m16", "m16"),
    "18 mantissa bits everywhere": lambda n: ("m18", "m18"),
    "20 mantissa bits everywhere": lambda n: ("m20", "m20"),
    "h2 split-fp16 pairs everywhere (the h2 engine)": lambda n: ("h2", "h2"),
    "h2 activations, fp16 weights": lambda n: ("h2", "fp16"),
}


def predict(model, frames, nc):
    B, H, W, _ = frames.shape
    pred, proto = model.forward_u8(frames, swap_rb=True)
    dets = non_max_suppression(pred.numpy(), CONF, IOU, MAX_DET, nc=nc)
    out = []
    for b, d in enumerate(dets):
        m = process_mask(proto[b], d[:, 6:], d[:, :4], (H, W), "logit").numpy().astype(np.uint8) if len(d) else np.zeros((0, H, W), np.uint8)
        out.append((d, m))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--out", default="")
    ap.add_argument("--only", default="", help="substring filter on policy names")
    args = ap.parse_args()
    import vti_amd
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    H = W = 640
    nc = 80
    eng = vti_amd.Engine("n", nc, H=H, W=W, max_batch=1, dtype="fp32")       # host-side plan only: the conv table
    frames = np.random.Generator(np.random.PCG64(1234)).integers(0, 256, (args.frames, H, W, 3), dtype=np.uint8)
    # class prior calibrated as bench.py does, but on the CPU oracle: ~60 anchors per frame clear conf
    blob0 = vti_amd.random_weights(eng, seed=1, cls_bias=0.0)
    pred, _ = OracleModel(blob0, H, W, "fp32").forward_u8(frames[:4])
    p = pred[:, 4:4 + nc].amax(1).flatten().clamp(1e-7, 1 - 1e-7)
    kth = torch.topk(torch.log(p / (1 - p)), 60 * 4).values[-1].item()
    bias = float(math.log(CONF / (1 - CONF)) - kth)
    blob = vti_amd.random_weights(eng, seed=1, cls_bias=bias)
    ref = OracleModel(blob, H, W, "fp32")
    t0 = time.time()
    want = predict(ref, frames, nc)
    lines = [f"# precision ablation on the CPU oracle: {args.frames} frames 640x640, YOLOv8n-seg nc=80, seeded random weights (cls bias {bias:.3f}), "
             f"conf {CONF} iou {IOU}; {sum(len(d) for d, _ in want)} instances; gate: kept set equal, IoU min >= 0.999, |d box|/640 < 1e-3",
             f"{'policy':58s} {'kept':>9s} {'same':>5s} {'box_px':>8s} {'box_norm':>9s} {'conf':>8s} {'iou_min':>8s} {'iou_p1':>8s} {'iou_mean':>9s} gate"]
    print("\n".join(lines), flush=True)
    print(f"# reference predict: {time.time() - t0:.1f} s", flush=True)
    for name, pol in POLICIES.items():
        if args.only and args.only not in name:
            continue
        got = predict(PolicyModel(blob, H, W, pol), frames, nc)
        r = op.compare(got, want, H, W)
        ln = (f"{name:58s} {r['n_engine']:4d}/{r['n_instances']:<4d} {str(r['kept_set_equal']):>5s} {r['box_px_max']:8.4f} {r['box_norm_max']:9.2e} "
              f"{r['conf_abs_max']:8.1e} {r['mask_iou_min']:8.5f} {r.get('mask_iou_p1', float('nan')):8.5f} {r['mask_iou_mean']:9.6f} {'PASS' if r['meets_north_star'] else 'fail'}")
        lines.append(ln)
        print(ln, flush=True)
    if args.out:
        with open(os.path.join(ROOT, args.out), "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
