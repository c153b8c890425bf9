import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch, vti_amd
from gpu_util import ref_conv
def run(c1, c2, H, W, wn=0, nrep=0, th=0, B=3):
    rng = np.random.default_rng(c1 * 1000 + c2)
    x = rng.standard_normal((B, H, W, c1)).astype(np.float32)
    w = (rng.standard_normal((c2, c1, 3, 3)) / np.sqrt(9 * c1)).astype(np.float32)
    b = rng.standard_normal(c2).astype(np.float32) * 0.5
    out, _, cfg = vti_amd.debug_conv2d(torch.from_numpy(x).half().cuda(), w, b, 3, 1, 0, "fp16", waves_n=wn, nrep=nrep, tile=(th, 20 if th else 0))
    ref = ref_conv(x, w, b, 3, 1, 0, "fp16").half().float()
    d = (out.float().cpu() - ref).abs()
    bad = (d > 2e-3 * ref.abs().max()).nonzero()
    print(c1, c2, H, W, cfg, "err", d.max().item(), "nbad", len(bad), "first bad", bad[:3].tolist(), "chan set", sorted(set(bad[:, 3].tolist()))[:12], "rows", sorted(set(bad[:, 1].tolist()))[:12], "cols", sorted(set(bad[:, 2].tolist()))[:24])
run(80, 80, 24, 40)
run(80, 80, 24, 40, 1, 5, 8)
run(80, 80, 16, 20, 1, 5, 16)
run(96, 64, 16, 20, 1, 4, 16)
run(96, 80, 16, 20, 1, 5, 16)
run(80, 64, 16, 20, 1, 4, 16)
run(64, 80, 16, 20, 1, 5, 16)
run(160, 80, 16, 20, 1, 5, 16)
run(128, 80, 40, 40, 2, 3, 8, B=2)
run(128, 80, 40, 40, 2, 3, 8, B=3)
run(80, 80, 24, 40, 2, 1, 4)
run(64, 80, 80, 80, 0, 0, 0, B=2)
