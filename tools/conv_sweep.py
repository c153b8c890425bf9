"""Developer tool: sweep launch geometries for the conv shapes of YOLOv8n-seg (bs=64) and report the best."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch, vti_amd
from conv_bench import SHAPES

flt = sys.argv[1] if len(sys.argv) > 1 else ""
B = 64
rng = np.random.default_rng(0)
TILES = [(16, 20), (8, 40), (4, 80), (2, 160), (20, 16), (10, 32), (8, 20), (4, 40), (2, 80), (1, 160), (10, 16), (5, 32),
         (4, 20), (2, 40), (1, 80), (5, 16), (8, 10), (1, 320), (2, 20), (1, 40)]
for name, (c1, c2, k, s, kind, H, W) in SHAPES.items():
    if flt and flt not in name:
        continue
    if c1 == 3:
        x = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, device="cuda")
    else:
        x = torch.randn((B, H, W, c1), device="cuda").half()
    w = (rng.standard_normal((c1, c2, k, k) if kind == 2 else (c2, c1, k, k)) / np.sqrt(c1 * k * k)).astype(np.float32)
    b = np.zeros(c2, np.float32)
    Ho, Wo = (H, W) if kind == 2 else ((H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1)
    nt = (4 * c2 if kind == 2 else c2) // 16
    res = []
    for wn in (1, 2, 4):
        for nrep in (1, 2, 3, 4, 5):
            bn = wn * nrep
            gy = -(-nt // bn)
            if nt / (gy * bn) < 0.74:
                continue
            bm = (4 // wn) * 80
            for th, tw in TILES:
                if th * tw > bm or th * tw < bm * 0.6 or Ho % th or Wo % tw:
                    continue
                try:
                    _, ms, cfg = vti_amd.debug_conv2d(x, w, b, k, s, kind, "fp16", c1=c1, tile=(th, tw), waves_n=wn, nrep=nrep, iters=6)
                    res.append((ms * 1e3, th, tw, wn, nrep, cfg["lds"]))
                except Exception as e:
                    pass
    res.sort()
    _, ms0, cfg0 = vti_amd.debug_conv2d(x, w, b, k, s, kind, "fp16", c1=c1, iters=6)
    print(f"{name:22s} planner {ms0*1e3:7.1f} us {cfg0['tile']} wn{cfg0['waves_n']} n{cfg0['nrep']} | best " +
          "  ".join(f"{r[0]:.1f}us {r[1]}x{r[2]} wn{r[3]} n{r[4]} lds{r[5]//1024}K" for r in res[:4]) + f" | worst {res[-1][0]:.1f} ({len(res)} cfgs)")
