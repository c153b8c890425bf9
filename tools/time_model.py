"""Developer script: forward time of one model configuration (scale, size, batch) with seeded random weights."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vti_amd
scale, H, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
dtype = sys.argv[4] if len(sys.argv) > 4 else "fp16"
eng = vti_amd.Engine(scale, 80, H=H, W=H, max_batch=B, dtype=dtype)
eng.load_weights(vti_amd.random_weights(eng, 1, gain=1.5 if scale == "m" else 1.7), 0)
x = torch.randint(0, 256, (B, H, H, 3), dtype=torch.uint8, device="cuda")
pred, proto = eng.forward(x)
torch.cuda.synchronize()
for _ in range(3):
    eng.forward(x, pred=pred, proto=proto)
torch.cuda.synchronize()
n = 10
t0 = time.time()
for _ in range(n):
    eng.forward(x, pred=pred, proto=proto)
torch.cuda.synchronize()
dt = (time.time() - t0) / n
fl = 2 * eng.macs_per_frame * B
print(f"[{scale} {H}x{H} B={B} {dtype}] {dt*1e3:.3f} ms/forward  {B/dt:.0f} fps  {fl/dt/1e12:.1f} TFLOP/s ({fl/dt/2.517e15*100:.2f}% of MFMA peak), "
      f"workspace {eng.workspace_bytes/2**30:.2f} GiB, launches {eng.num_launches}, finite {bool(torch.isfinite(pred).all())}")
