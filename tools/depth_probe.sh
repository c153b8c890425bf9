#!/bin/bash
# Per-op single-stream durations under different LDS-DMA ring depths (VTI_PK_DEPTH cap 2/3/4), one box, one call:
#   bash tools/depth_probe.sh   -> gpurun_out/depth_<d>.txt
for d in 2 3 4; do
  rm -rf gpurun_out/_dp$d; mkdir -p gpurun_out/_dp$d
  VTI_PK_DEPTH=$d VTI_SINGLE_STREAM=1 VTI_LIST_OPS=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/_dp$d -- python3 tools/prof_forward.py 64 fp16 6 2> gpurun_out/_dp$d/ops.txt > /dev/null || exit 1
  python3 tools/op_times.py gpurun_out/_dp$d 64 fp16 > gpurun_out/depth_$d.txt || exit 1
done
python3 - <<'PY'
import re
t = {}
for d in (2, 3, 4):
    for l in open(f"gpurun_out/depth_{d}.txt"):
        m = re.match(r"\s*(\d+)\s+\d\s+(.{60})\s+([\d.]+)\s", l)
        if m: t.setdefault(int(m.group(1)), {"name": m.group(2).strip()})[d] = float(m.group(3))
        if l.startswith("sum"): break
tot = {2: 0, 3: 0, 4: 0}; best = 0
for i in sorted(t):
    r = t[i]
    if 2 not in r: continue
    for d in (2, 3, 4): tot[d] += r.get(d, 0)
    best += min(r.get(d, 1e9) for d in (2, 3, 4))
    flag = "" if abs(r[2] - min(r[3], r[4])) < 0.6 else ("  <<" if min(r[3], r[4]) < r[2] else "  >>")
    print(f"{i:2d} {r['name'][:58]:58s} {r[2]:7.1f} {r.get(3,0):7.1f} {r.get(4,0):7.1f}{flag}")
print("sum", tot, "best-per-op", round(best, 1))
PY
