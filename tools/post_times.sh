#!/bin/bash
# Per-kernel times of the post-processing stage under the bench (kernel trace; the forward's kernels are filtered out).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/_pt && mkdir -p gpurun_out/_pt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/_pt -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2> gpurun_out/_pt/err.txt || exit 1
python3 - <<'PY'
import csv, glob, collections
f = sorted(glob.glob("gpurun_out/_pt/**/*kernel_trace.csv", recursive=True))[-1]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "conv" in n or "stem" in n or "sppf" in n: continue
    agg[n[:60]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v = sorted(v)
    print(f"{k:62s} {len(v):5d} median {v[len(v)//2]/1e3:9.1f} us")
PY
