#!/usr/bin/env python3
"""Forward (and the whole predict step) as direct stream launches vs one HIP-graph replay, same box, interleaved rounds.
    python tools/graph_vs_stream.py [dtype=h2] [batch=64]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vti_amd

dtype = sys.argv[1] if len(sys.argv) > 1 else "h2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
H = W = 640
eng = vti_amd.Engine("n", 80, H=H, W=W, max_batch=B, dtype=dtype)
eng.load_weights(vti_amd.random_weights(eng, 1, cls_bias=-4.6), 0)
g = torch.Generator(device="cuda"); g.manual_seed(1)
x = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, device="cuda", generator=g)
cap = B * 128
o = eng.alloc_outputs(B, 300, cap, "bits", x.device)
o["best"] = eng.alloc_best(B, x.device)


def fwd():
    eng.forward(x, True, pred=o["pred"], proto=o["proto"], best=o["best"])


def step():
    fwd()
    eng.nms(o["pred"], 0.25, 0.7, 300, False, dets=o["dets"], counts=o["counts"], best=o["best"])
    eng.masks(o["dets"], o["counts"], o["proto"], "logit", "bits", capacity=cap, masks=o["masks"], offsets=o["offsets"])
    eng.scale_boxes(o["dets"], o["counts"], H, W, xyxy=o["xyxy"])


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


s = torch.cuda.Stream()
graphs = {}
with torch.cuda.stream(s):
    for name, fn in (("forward", fwd), ("step", step)):
        fn(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            fn()
        graphs[name] = gr
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    for r in range(2):
        for name, fn in (("forward", fwd), ("step", step)):
            t_s = timeit(fn)
            t_g = timeit(graphs[name].replay)
            print(f"{dtype} B={B} {name:8s} stream {t_s:.4f} ms   graph {t_g:.4f} ms   ({100 * (t_g / t_s - 1):+.2f} %)", flush=True)
