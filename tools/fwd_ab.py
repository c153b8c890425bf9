"""Developer script: multi-stream forward time (bs=64) of one dtype under environment toggles, interleaved on one box:
   python tools/fwd_ab.py h2 "A=0" "VTI_P3_LANES=1" ...   (each arm is a child process with that environment)"""
import os, subprocess, sys
dt = sys.argv[1]
arms = sys.argv[2:] or ["A=0"]
code = r'''
import sys, time, torch
sys.path.insert(0, ".")
import vti_amd
dt = sys.argv[1]
e = vti_amd.Engine("n", 80, H=640, W=640, max_batch=64, dtype=dt)
e.load_weights(vti_amd.random_weights(e, 1), 0)
x = torch.randint(0, 256, (64, 640, 640, 3), dtype=torch.uint8, device="cuda")
o = e.alloc_outputs(64, 300, 4096, "bits")
t_end = time.time() + 1.0
while time.time() < t_end:
    for _ in range(8): e.forward(x, True, pred=o["pred"], proto=o["proto"], best=o["best"])
    torch.cuda.synchronize()
n = 40
t0 = time.perf_counter()
for _ in range(n): e.forward(x, True, pred=o["pred"], proto=o["proto"], best=o["best"])
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / n * 1e3:.4f}")
'''
for rnd in range(2):
    for arm in arms:
        env = dict(os.environ)
        k, v = arm.split("=", 1)
        env[k] = v
        out = subprocess.run([sys.executable, "-c", code, dt], env=env, capture_output=True, text=True)
        print(f"round {rnd} {arm:28s} forward {out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:]} ms", flush=True)
