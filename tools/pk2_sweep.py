"""Developer tool: sweep the stride-2 persistent kernel's geometries over YOLOv8n-seg's five stride-2 3x3 layers (bs=64).
   python tools/pk2_sweep.py [dtype=h2]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, vti_amd
DT = sys.argv[1] if len(sys.argv) > 1 else "h2"
SHAPES = [(32, 64, 160), (64, 128, 80), (128, 256, 40), (64, 64, 80), (128, 128, 40)]
B = 64
rng = np.random.default_rng(0)
for c1, c2, hw in SHAPES:
    x = torch.randn((B, hw, hw, c1), device="cuda")
    x = vti_amd.h2_encode(x.cpu()).cuda() if DT == "h2" else x.half() if DT == "fp16" else x
    w = (rng.standard_normal((c2, c1, 3, 3)) / np.sqrt(c1 * 9)).astype(np.float32)
    b = np.zeros(c2, np.float32)
    res = []
    for wn, nrep in ((4, 1), (2, 2), (1, 4), (1, 2), (2, 1)):
        for th in (4, 8, 12, 16):
            if (th // 4) * wn > 4:
                continue
            try:
                _, ms, cfg = vti_amd.debug_conv2d(x, w, b, 3, 2, 0, DT, c1=c1, tile=(th, 20), waves_n=wn, nrep=nrep, iters=8)
                if cfg["pk"]:
                    res.append((ms * 1e3, th, wn, nrep, cfg["lds"]))
            except Exception:
                pass
    res.sort()
    _, ms0, cfg0 = vti_amd.debug_conv2d(x, w, b, 3, 2, 0, DT, c1=c1, iters=8)
    print(f"{c1}-{c2}_{hw}s2  planner {ms0*1e3:6.1f}us th{cfg0['tile'][0]} wn{cfg0['waves_n']} n{cfg0['nrep']} pk{int(cfg0['pk'])} {cfg0['lds']//1024}K | best " +
          "  ".join(f"{r[0]:.1f}us th{r[1]} wn{r[2]} n{r[3]} {r[4]//1024}K" for r in res[:6]), flush=True)
