"""Developer script: series vs pipelined bench runs of one dtype on one box (python tools/bench_modes.py [dtype])."""
import json, subprocess, sys
dt = sys.argv[1] if len(sys.argv) > 1 else "h2"
for m in ([], ["--pipeline"]):
    out = subprocess.run([sys.executable, "bench.py", "--steps", "30", "--warmup", "5", "--no-cpu-baseline", "--no-fp32-line", "--no-fp16-line",
                          "--no-host-fed", "--parity-frames", "0", "--dtype", dt] + m, capture_output=True, text=True).stdout
    d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    r = d["roofline"]
    print(f"{' '.join(m) or 'series':12s} fps {d['value']:.0f} step {d['ms_per_step']:.4f} fwd(in-region) {r['avg_ms']:.4f} iso {r['isolated']['avg_ms']:.4f} "
          f"post {d['stage_ms']['nms+masks+scale_boxes']:.4f} frac {r['frac']}")
