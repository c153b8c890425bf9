"""Developer tool: sweep the persistent 3x3 kernel's geometries (tile rows, wave split, register tile) over the
3x3/s1 shapes of YOLOv8n-seg at bs=64, next to the per-tile kernel (VTI_NO_PK=1) and the planner's pick."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, vti_amd

SHAPES = [(16, 16, 160), (32, 32, 80), (64, 64, 40), (128, 128, 20), (64, 64, 80), (64, 80, 80), (80, 80, 80),
          (64, 32, 80), (128, 64, 40), (128, 80, 40), (128, 32, 40), (80, 80, 40), (32, 32, 40), (64, 64, 160),
          (256, 64, 20), (256, 80, 20), (256, 32, 20), (64, 64, 20), (80, 80, 20), (32, 32, 20)]
flt = sys.argv[1] if len(sys.argv) > 1 else ""
DT = sys.argv[2] if len(sys.argv) > 2 else "fp16"          # fp16 | h2 | fp32
B = 64
rng = np.random.default_rng(0)
for c1, c2, hw in SHAPES:
    name = f"{c1}-{c2}_{hw}"
    if flt and flt not in name:
        continue
    x = torch.randn((B, hw, hw, c1), device="cuda")
    x = vti_amd.h2_encode(x.cpu()).cuda() if DT == "h2" else x.half() if DT == "fp16" else x
    w = (rng.standard_normal((c2, c1, 3, 3)) / np.sqrt(c1 * 9)).astype(np.float32)
    b = np.zeros(c2, np.float32)
    nt = -(-c2 // 16)
    res = []
    for wn in (1, 2, 4):
        for nrep in (1, 2, 3, 4, 5):
            bn = wn * nrep
            gy = -(-nt // bn)
            if nt / (gy * bn) < 0.74:
                continue
            for th in (4, 8, 12, 16):
                if (th // 4) * wn > 4:
                    continue
                try:
                    _, ms, cfg = vti_amd.debug_conv2d(x, w, b, 3, 1, 0, DT, c1=c1, tile=(th, 20), waves_n=wn, nrep=nrep, iters=8)
                    if cfg["pk"]:
                        res.append((ms * 1e3, th, wn, nrep, cfg["lds"]))
                except Exception:
                    pass
    res.sort()
    _, ms0, cfg0 = vti_amd.debug_conv2d(x, w, b, 3, 1, 0, DT, c1=c1, iters=8)
    os.environ["VTI_NO_PK"] = "1"
    _, ms1, cfg1 = vti_amd.debug_conv2d(x, w, b, 3, 1, 0, DT, c1=c1, iters=8)
    del os.environ["VTI_NO_PK"]
    fl = 2 * hw * hw * c1 * c2 * 9 * B
    print(f"{name:14s} planner {ms0*1e3:6.1f}us th{cfg0['tile'][0]} wn{cfg0['waves_n']} n{cfg0['nrep']} pk{int(cfg0['pk'])} | per-tile {ms1*1e3:6.1f}us | best " +
          "  ".join(f"{r[0]:.1f}us th{r[1]} wn{r[2]} n{r[3]} {r[4]//1024}K" for r in res[:5]) +
          (f" | {fl/res[0][0]/1e6:.0f} TF/s" if res else ""), flush=True)
