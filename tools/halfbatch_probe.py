"""Developer probe: one forward over 64 frames vs two concurrent half-batch forwards (two contexts, two streams)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vti_amd
B = 64
x = torch.randint(0, 256, (B, 640, 640, 3), dtype=torch.uint8, device="cuda")
def mk(b):
    e = vti_amd.Engine("n", 80, H=640, W=640, max_batch=b, dtype="fp16")
    e.load_weights(vti_amd.random_weights(e, 1), 0)
    return e
full = mk(B)
halves = [mk(B // 2), mk(B // 2)]
quarters = [mk(B // 4) for _ in range(4)]
def bench(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
pf, qf = full.forward(x)
def run_full(): full.forward(x, pred=pf, proto=qf)
def split(engs):
    k = len(engs); per = B // k
    xs = [x[i * per:(i + 1) * per].contiguous() for i in range(k)]
    outs = [e.forward(xi) for e, xi in zip(engs, xs)]
    streams = [torch.cuda.Stream() for _ in range(k)]
    def run():
        cur = torch.cuda.current_stream()
        for s in streams: s.wait_stream(cur)
        for e, xi, o, s in zip(engs, xs, outs, streams):
            with torch.cuda.stream(s):
                e.forward(xi, pred=o[0], proto=o[1])
        for s in streams: cur.wait_stream(s)
    return run
print(f"full 64: {bench(run_full):.3f} ms")
print(f"2 x 32 concurrent: {bench(split(halves)):.3f} ms")
print(f"4 x 16 concurrent: {bench(split(quarters)):.3f} ms")
print(f"full 64 again: {bench(run_full):.3f} ms")
