#!/usr/bin/env python3
"""Run-to-run determinism of the forward (GPU): the same frames through the same engine N times, every output compared BIT FOR BIT
with the first run's -- a missing barrier / event shows up here as a rare mismatch long before a tolerance test notices.  A second
stream keeps the chip busy with copies during half of the runs to perturb the timing.

    python tools/determinism.py [--dtype h2] [--batch 64] [--runs 60]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import vti_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="h2")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--runs", type=int, default=60)
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--poison", type=int, default=0,
                    help="N > 0: instead, build N fresh engines whose workspace and outputs are recycled blocks pre-filled with 0xFF / 0x7C / "
                         "0x00 bytes (NaN patterns in every storage type) and compare each one's first forward with a clean engine's: "
                         "a result that depends on memory the forward never wrote shows up as a mismatch")
    args = ap.parse_args()
    B, H, W, nc = args.batch, args.size, args.size, 80
    eng = vti_amd.Engine("n", nc, H=H, W=W, max_batch=B, dtype=args.dtype)
    eng.load_weights(vti_amd.random_weights(eng, 1, cls_bias=-3.0), 0)
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    x = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, device="cuda", generator=g)
    best = eng.alloc_best(B, x.device)
    pred0, proto0 = eng.forward(x, True, best=best)
    torch.cuda.synchronize()
    pred0, proto0, best0 = pred0.clone(), proto0.clone(), best.clone()
    if args.poison:
        blob = vti_amd.random_weights(eng, 1, cls_bias=-3.0)
        bad = 0
        for r in range(args.poison):
            fill = (0xFF, 0x7C, 0x00, 0xFB)[r % 4]
            blocks = [torch.full((int(eng.workspace_bytes * f) + 4096,), fill, dtype=torch.uint8, device="cuda") for f in (1.0, 0.5, 0.25, 0.1)]
            blocks += [torch.full((n,), fill, dtype=torch.uint8, device="cuda") for n in (pred0.numel() * 4, proto0.numel() * proto0.element_size(), best0.numel() * 4)]
            torch.cuda.synchronize()
            del blocks                                  # back to torch's cache, dirty: the next allocations of these sizes take them
            e2 = vti_amd.Engine("n", nc, H=H, W=W, max_batch=B, dtype=args.dtype)
            e2.load_weights(blob, 0)
            b2 = e2.alloc_best(B, x.device)
            pred, proto = e2.forward(x, True, best=b2)
            torch.cuda.synchronize()
            ok = (torch.equal(pred, pred0), bool((proto.view(torch.uint8) == proto0.view(torch.uint8)).all()), torch.equal(b2, best0))
            if not all(ok):
                bad += 1
                msg = f"poison 0x{fill:02X} engine {r}: pred equal {ok[0]} proto equal {ok[1]} best equal {ok[2]}"
                if not ok[0]:
                    idx = (pred != pred0).nonzero()
                    msg += f"; pred differs at {idx.shape[0]} places, first {idx[0].tolist()}, rows {sorted(set(idx[:, 1].tolist()))[:10]}, nan {int(pred.isnan().sum())}"
                if not ok[1]:
                    msg += f"; proto nan {int(proto.float().isnan().sum())}"
                print(msg, flush=True)
            del e2, pred, proto, b2
        print(f"{args.dtype} B={B} {H}x{W}: {bad} of {args.poison} poisoned fresh engines differed from the clean one", flush=True)
        return 1 if bad else 0
    noise_stream = torch.cuda.Stream()
    big = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    big2 = torch.empty_like(big)
    bad = 0
    for r in range(args.runs):
        if r & 1:
            with torch.cuda.stream(noise_stream):
                for _ in range(3):
                    big2.copy_(big, non_blocking=True)
        pred, proto = eng.forward(x, True, best=best)
        torch.cuda.synchronize()
        ok = torch.equal(pred, pred0), torch.equal(proto.view(torch.int32) if proto.dtype == torch.float32 else proto.view(torch.int16),
                                                   proto0.view(torch.int32) if proto0.dtype == torch.float32 else proto0.view(torch.int16)), torch.equal(best, best0)
        if not all(ok):
            bad += 1
            dp = (pred != pred0)
            msg = f"run {r}: pred equal {ok[0]} proto equal {ok[1]} best equal {ok[2]}"
            if not ok[0]:
                idx = dp.nonzero()
                msg += f"; pred differs at {idx.shape[0]} places, first {idx[0].tolist()}, frames {sorted(set(idx[:, 0].tolist()))[:8]}, rows {sorted(set(idx[:, 1].tolist()))[:8]}"
            if not ok[1]:
                d = (proto.float() != proto0.float()).nonzero()
                msg += f"; proto differs at {d.shape[0]} places, first {d[0].tolist()}"
            print(msg, flush=True)
    print(f"{args.dtype} B={B} {H}x{W}: {bad} of {args.runs} runs differed from the first", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
