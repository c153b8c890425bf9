import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vti_amd, time
sys.argv = ["bench.py"]
import bench
B = 64
eng = vti_amd.Engine("n", 80, H=640, W=640, max_batch=B, dtype="fp16")
gen = torch.Generator(device="cuda"); gen.manual_seed(1234)
frames = torch.randint(0, 256, (B, 640, 640, 3), dtype=torch.uint8, device="cuda", generator=gen)
blob, bias = bench.calibrated_weights(vti_amd, eng, frames, 0.25, target=60)
pred, proto = eng.forward(frames)
cand = (pred[:, 4:84].amax(1) > 0.25).sum(1)
dets, counts = eng.nms(pred, 0.25, 0.7, 300)
torch.cuda.synchronize()
print("candidates/frame: min", cand.min().item(), "median", cand.median().item(), "max", cand.max().item())
print("kept/frame: min", counts.min().item(), "median", counts.float().median().item(), "max", counts.max().item())
wh = (dets[..., 2:4] - dets[..., :2])[dets[..., 4] > 0]
print("box w/h mean", wh.mean(0).tolist(), "max", wh.max(0).values.tolist())
for name, fn in (("nms", lambda: eng.nms(pred, 0.25, 0.7, 300, dets=dets, counts=counts)),):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(20): fn()
    torch.cuda.synchronize(); print(name, (time.time() - t0) / 20 * 1e6, "us")
