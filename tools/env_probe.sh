#!/bin/bash
# Per-op single-stream durations with and without environment toggles, one box, one call:
#   bash tools/env_probe.sh "VTI_NO_T512=1" ...   -> table of the ops whose time differs
arms=("A=default" "$@")
i=0
for arm in "${arms[@]}"; do
  rm -rf gpurun_out/_ep$i; mkdir -p gpurun_out/_ep$i
  env $arm VTI_SINGLE_STREAM=1 VTI_LIST_OPS=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/_ep$i -- python3 tools/prof_forward.py 64 fp16 6 2> gpurun_out/_ep$i/ops.txt > /dev/null || exit 1
  env $arm python3 tools/op_times.py gpurun_out/_ep$i 64 fp16 > gpurun_out/env_probe_$i.txt || exit 1
  i=$((i+1))
done
python3 - "${#arms[@]}" "${arms[@]}" <<'PY'
import re, sys
n = int(sys.argv[1]); arms = sys.argv[2:]
t = {}
for d in range(n):
    for l in open(f"gpurun_out/env_probe_{d}.txt"):
        m = re.match(r"\s*(\d+)\s+\d\s+(.{60})\s+([\d.]+)\s", l)
        if m: t.setdefault(m.group(2).strip(), {})[d] = float(m.group(3))
        if l.startswith("sum"): break
print("arms:", arms)
tot = [0.0] * n
for name, r in t.items():
    for d in range(n): tot[d] += r.get(d, 0)
    if max(r.values()) - min(r.values()) >= 1.0:
        print(f"{name[:58]:58s} " + " ".join(f"{r.get(d, 0):7.1f}" for d in range(n)))
print("sum", [round(x, 1) for x in tot])
PY
