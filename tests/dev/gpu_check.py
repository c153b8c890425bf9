"""Developer script (GPU box): layer-by-layer parity of the HIP forward vs the CPU oracle, then a
rough timing.  Not part of the product; the judged checks live in tests/ and bench.py."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import vti_amd
from oracle.model import OracleModel


def check(dtype, B=2, scale="n", nc=80, H=640, W=640, cls_bias=-2.0):
    eng = vti_amd.Engine(scale, nc, H=H, W=W, max_batch=B, dtype=dtype)
    blob = vti_amd.random_weights(eng, seed=1, cls_bias=cls_bias)
    eng.load_weights(blob, 0)
    rng = np.random.Generator(np.random.PCG64(0))
    frames = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    x = torch.from_numpy(frames).cuda()
    pred, proto = eng.forward(x, swap_rb=True)
    torch.cuda.synchronize()
    om = OracleModel(blob, H, W, mode=dtype)
    t0 = time.time()
    opred, oproto = om.forward_u8(frames, swap_rb=True, record=True)
    print(f"[{dtype}] oracle forward {time.time()-t0:.2f}s")
    worst = 0
    for i, t in enumerate(eng.conv_table()):
        try:
            got = eng.debug_conv_output(i, B).cpu()
        except vti_amd.VtiError:
            print(f"  {i:2d} {t['name']:26s} (fused into the next conv; not materialised)")
            continue
        ref = om.taps[t["name"]]
        err = (got - ref).abs().max().item()
        den = ref.abs().max().item() + 1e-9
        flag = "" if err / den < (2e-2 if dtype == "fp16" else 1e-4) else "   <<<<<<"
        worst = max(worst, err / den)
        print(f"  {i:2d} {t['name']:26s} max|d|={err:.3e} ref_max={den:.3e} rel={err/den:.2e} mean|ref|={ref.abs().mean():.3f}{flag}")
    e_pred = (pred.cpu() - opred).abs()
    print(f"[{dtype}] pred max|d| box={e_pred[:, :4].max():.3e} cls={e_pred[:, 4:4+nc].max():.3e} mc={e_pred[:, 4+nc:].max():.3e}")
    e_proto = (proto.float().cpu().permute(0, 3, 1, 2) - oproto).abs().max().item()
    print(f"[{dtype}] proto max|d|={e_proto:.3e}; worst layer rel={worst:.2e}")
    ncand = (opred[:, 4:4+nc].amax(1) > 0.25).sum(1)
    print(f"[{dtype}] candidates/frame at conf .25: {ncand.tolist()}")
    return eng, blob


def timing(B=64, dtype="fp16"):
    eng = vti_amd.Engine("n", 80, H=640, W=640, max_batch=B, dtype=dtype)
    eng.load_weights(vti_amd.random_weights(eng, 1), 0)
    x = torch.randint(0, 256, (B, 640, 640, 3), dtype=torch.uint8, device="cuda")
    pred, proto = eng.forward(x)
    torch.cuda.synchronize()
    for _ in range(3):
        eng.forward(x, pred=pred, proto=proto)
    torch.cuda.synchronize()
    t0 = time.time()
    n = 10
    for _ in range(n):
        eng.forward(x, pred=pred, proto=proto)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / n
    fl = 2 * eng.macs_per_frame * B
    print(f"[timing {dtype} B={B}] {dt*1e3:.3f} ms/forward  {B/dt:.0f} fps  {fl/dt/1e12:.1f} TFLOP/s ({fl/dt/2.517e15*100:.2f}% of MFMA peak)")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("all", "fp32"):
        check("fp32")
    if what in ("all", "fp16"):
        check("fp16")
    if what in ("all", "time"):
        timing(64, "fp16")
        timing(1, "fp16")
        timing(16, "fp32")
