"""Developer script: capture forward+NMS+masks+scale_boxes for a small batch into a HIP graph and time replays."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vti_amd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
eng = vti_amd.Engine("n", 80, H=640, W=640, max_batch=B, dtype="fp16")
eng.load_weights(vti_amd.random_weights(eng, 1, cls_bias=-6.0), 0)
x = torch.randint(0, 256, (B, 640, 640, 3), dtype=torch.uint8, device="cuda")
out = eng.alloc_outputs(B, 300, B * 64, "bits")
def run():
    eng.forward(x, True, pred=out["pred"], proto=out["proto"])
    eng.nms(out["pred"], 0.25, 0.7, 300, dets=out["dets"], counts=out["counts"])
    eng.masks(out["dets"], out["counts"], out["proto"], "logit", "bits", capacity=B * 64, masks=out["masks"], offsets=out["offsets"])
    eng.scale_boxes(out["dets"], out["counts"], 640, 640, xyxy=out["xyxy"])
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): run()
torch.cuda.synchronize()
ref = {k: v.clone() for k, v in out.items()}
def timeit(fn, n=200):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
with torch.cuda.stream(s):
    eager = timeit(run)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    run()
for k in ("pred", "dets", "counts", "masks"): out[k].zero_()
g.replay(); torch.cuda.synchronize()
ok = all(torch.equal(out[k], ref[k]) for k in ("pred", "proto", "dets", "counts", "masks", "xyxy"))
graph = timeit(g.replay)
print(f"B={B}: eager {eager:.3f} ms/predict, graph replay {graph:.3f} ms/predict ({B/graph*1e3:.0f} fps), outputs identical: {ok}, dets {out['counts'].tolist()[:4]}")
