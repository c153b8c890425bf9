"""First look at the h2 engine on the GPU: per-layer error vs the fp32 oracle, forward time, parity gate."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import vti_amd
from oracle.model import OracleModel
from oracle import parity as op

B, H, W, nc = 2, 640, 640, 80
eng = vti_amd.Engine("n", nc, H=H, W=W, max_batch=B, dtype="h2")
blob = vti_amd.random_weights(eng, seed=1, cls_bias=-6.9)
eng.load_weights(blob, 0)
fr = np.random.Generator(np.random.PCG64(3)).integers(0, 256, (B, H, W, 3), dtype=np.uint8)
x = torch.from_numpy(fr).cuda()
pred, proto = eng.forward(x)
torch.cuda.synchronize()
om = OracleModel(blob, H, W, "fp32")
opred, oproto = om.forward_u8(fr, record=True)
worst = 0
for i, t in enumerate(eng.conv_table()):
    try:
        got = eng.debug_conv_output(i, B).cpu()
    except vti_amd.VtiError:
        continue
    ref = om.taps[t["name"]]
    err = (got - ref).abs().max().item() / max(ref.abs().max().item(), 1.0)
    worst = max(worst, err)
    print(f"{t['name']:28s} rel err {err:.2e}")
print("worst layer rel err", worst)
e = (pred.cpu() - opred).abs()
print("pred box err px", e[:, :4].max().item(), "cls", e[:, 4:84].max().item(), "mc", e[:, 84:].max().item())
print("proto err", (proto.float().cpu().permute(0, 3, 1, 2) - oproto).abs().max().item())
got = op.engine_predict(eng, x, 0.25, 0.7, 300)
want = op.oracle_predict(blob, fr, nc, 0.25, 0.7, 300, mode="fp32")
print(op.compare(got, want, H, W))
for dt in ("h2", "fp16", "fp32"):
    e64 = vti_amd.Engine("n", nc, H=H, W=W, max_batch=64, dtype=dt)
    e64.load_weights(blob, 0)
    x64 = torch.randint(0, 256, (64, H, W, 3), dtype=torch.uint8, device="cuda")
    o = e64.alloc_outputs(64, 300, 64 * 64, "bits")
    for _ in range(30):
        e64.forward(x64, True, pred=o["pred"], proto=o["proto"], best=o["best"])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 30
    for _ in range(n):
        e64.forward(x64, True, pred=o["pred"], proto=o["proto"], best=o["best"])
    torch.cuda.synchronize()
    print(dt, "forward bs=64 ms", (time.perf_counter() - t0) / n * 1e3)
