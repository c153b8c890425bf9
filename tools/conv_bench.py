"""Developer micro-benchmark: time single conv shapes of YOLOv8n-seg (bs=64) through vti_debug_conv2d.
usage: conv_bench.py [name-filter] [iters] [th tw wn nrep]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, vti_amd

SHAPES = {  # name: (c1, c2, k, s, kind, H_in, W_in)
    "l1_16-32_s2": (16, 32, 3, 2, 0, 320, 320),
    "c2f2_m_16-16": (16, 16, 3, 1, 0, 160, 160),
    "c2f2_cv1_32-32": (32, 32, 1, 1, 0, 160, 160),
    "l3_32-64_s2": (32, 64, 3, 2, 0, 160, 160),
    "c2f4_m_32-32": (32, 32, 3, 1, 0, 80, 80),
    "c2f4_cv2_128-64": (128, 64, 1, 1, 0, 80, 80),
    "head_64-64_80": (64, 64, 3, 1, 0, 80, 80),
    "head_64-80_80": (64, 80, 3, 1, 0, 80, 80),
    "head_80-80_80": (80, 80, 3, 1, 0, 80, 80),
    "proto_cv2_64-64_160": (64, 64, 3, 1, 0, 160, 160),
    "proto_cv3_64-32_160": (64, 32, 1, 1, 0, 160, 160),
    "deconv_64-64": (64, 64, 2, 2, 2, 80, 80),
    "c2f6_m_64-64_40": (64, 64, 3, 1, 0, 40, 40),
    "c2f8_m_128-128_20": (128, 128, 3, 1, 0, 20, 20),
    "l7_128-256_s2": (128, 256, 3, 2, 0, 40, 40),
    "sppf_cv2_512-256": (512, 256, 1, 1, 0, 20, 20),
    "sppf_cv1_256-128_20": (256, 128, 1, 1, 0, 20, 20),
    "sppf_cv2_512-256_20": (512, 256, 1, 1, 0, 20, 20),
    "c2f8_cv1_256-256_20": (256, 256, 1, 1, 0, 20, 20),
    "c2f8_cv2_384-256_20": (384, 256, 1, 1, 0, 20, 20),
    "l5_64-128_s2": (64, 128, 3, 2, 0, 80, 80),
    "l16_64-64_s2": (64, 64, 3, 2, 0, 80, 80),
    "l19_128-128_s2": (128, 128, 3, 2, 0, 40, 40),
    "stem": (3, 16, 3, 2, 0, 640, 640),
}
def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    force = [int(v) for v in sys.argv[3:7]] if len(sys.argv) >= 7 else [0, 0, 0, 0]
    DT = os.environ.get("VTI_BENCH_DTYPE", "fp16")
    B = 64
    rng = np.random.default_rng(0)
    for name, (c1, c2, k, s, kind, H, W) in SHAPES.items():
        if flt and flt not in name:
            continue
        if c1 == 3:
            x = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, device="cuda")
        else:
            x = torch.randn((B, H, W, c1), device="cuda")
            x = vti_amd.h2_encode(x.cpu()).cuda() if DT == "h2" else x.half() if DT == "fp16" else x
        w = (rng.standard_normal((c1, c2, k, k) if kind == 2 else (c2, c1, k, k)) / np.sqrt(c1 * k * k)).astype(np.float32)
        b = np.zeros(c2, np.float32)
        out, ms, cfg = vti_amd.debug_conv2d(x, w, b, k, s, kind, DT, c1=c1, tile=(force[0], force[1]), waves_n=force[2], nrep=force[3], iters=iters)
        Ho, Wo = out.shape[1], out.shape[2]
        macs = (H * W if kind == 2 else Ho * Wo) * c1 * c2 * k * k * B
        byts = x.numel() * x.element_size() + out.numel() * out.element_size()
        print(f"{name:22s} {ms*1e3:8.1f} us  {2*macs/ms/1e9:7.1f} TF/s  {byts/ms/1e6:7.0f} GB/s (min-traffic)  cfg {cfg}")


if __name__ == "__main__":
    main()
