"""Developer tool: the persistent 1x1 kernel (conv1_pk) against the per-tile kernel on the 1x1 shapes of YOLOv8n-seg, bs=64."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, vti_amd

SHAPES = [(32, 32, 160), (48, 32, 160), (64, 64, 80), (128, 64, 80), (128, 128, 40), (256, 128, 40), (256, 256, 20), (384, 256, 20),
          (256, 128, 20), (512, 256, 20), (384, 128, 40), (192, 128, 40), (192, 64, 80), (96, 64, 80)]
flt = sys.argv[1] if len(sys.argv) > 1 else ""
B = 64
rng = np.random.default_rng(0)
for c1, c2, hw in SHAPES:
    name = f"{c1}-{c2}_{hw}"
    if flt and flt not in name:
        continue
    x = torch.randn((B, hw, hw, c1), device="cuda").half()
    w = (rng.standard_normal((c2, c1, 1, 1)) / np.sqrt(c1)).astype(np.float32)
    b = np.zeros(c2, np.float32)
    nt = -(-c2 // 16)
    res = []
    for wn in (1, 2, 4):
        for nrep in (1, 2, 3, 4, 5):
            bn = wn * nrep
            gy = -(-nt // bn)
            if nt / (gy * bn) < 0.74:
                continue
            for nwm in (1, 2, 4):
                if nwm * wn > 4:
                    continue
                try:
                    _, ms, cfg = vti_amd.debug_conv2d(x, w, b, 1, 1, 0, "fp16", c1=c1, tile=(nwm, 80), waves_n=wn, nrep=nrep, iters=8)
                    if cfg["pk"]:
                        res.append((ms * 1e3, nwm, wn, nrep, cfg["lds"]))
                except Exception:
                    pass
    res.sort()
    _, ms0, cfg0 = vti_amd.debug_conv2d(x, w, b, 1, 1, 0, "fp16", c1=c1, iters=8)
    os.environ["VTI_NO_PK1"] = "1"
    _, ms1, cfg1 = vti_amd.debug_conv2d(x, w, b, 1, 1, 0, "fp16", c1=c1, iters=8)
    del os.environ["VTI_NO_PK1"]
    byts = B * hw * hw * (c1 + c2) * 2
    print(f"{name:14s} planner {ms0*1e3:6.1f}us m{cfg0['tile'][0]} wn{cfg0['waves_n']} n{cfg0['nrep']} pk{int(cfg0['pk'])} | per-tile {ms1*1e3:6.1f}us | best " +
          "  ".join(f"{r[0]:.1f}us m{r[1]} wn{r[2]} n{r[3]} {r[4]//1024}K" for r in res[:5]) +
          (f" | {byts/res[0][0]/1e6:.2f} TB/s" if res else ""), flush=True)
