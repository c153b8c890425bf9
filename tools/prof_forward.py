"""Developer script: run N forwards at batch B for rocprofv3 --kernel-trace --stats."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vti_amd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dtype = sys.argv[2] if len(sys.argv) > 2 else "fp16"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 5
eng = vti_amd.Engine("n", 80, H=640, W=640, max_batch=B, dtype=dtype)
eng.load_weights(vti_amd.random_weights(eng, 1), 0)
x = torch.randint(0, 256, (B, 640, 640, 3), dtype=torch.uint8, device="cuda")
pred, proto = eng.forward(x)
torch.cuda.synchronize()
for _ in range(n):
    eng.forward(x, pred=pred, proto=proto)
torch.cuda.synchronize()
