#!/bin/bash
# usage: pmc_kernel.sh <outdir> <kernel-substring> <counters...>   (bench.py under rocprofv3 --pmc; per-dispatch rows by grid size)
out=$1; kern=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/$out
VTI_SINGLE_STREAM=1 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/$out -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/$out.log 2>&1
python3 - "$kern" "$out" <<'PY'
import csv, glob, collections, sys
kern, out = sys.argv[1], sys.argv[2]
f = glob.glob(f"gpurun_out/{out}/**/*counter_collection.csv", recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if kern in r["Kernel_Name"]]
agg = collections.defaultdict(list)
for r in rows: agg[(int(r["Grid_Size"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
for (g, k), v in sorted(agg.items()): print(f"grid {g:8d} {k:32s} mean {sum(v)/len(v):16.1f}  n={len(v)}")
PY
