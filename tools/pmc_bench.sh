#!/bin/bash
# usage: pmc_bench.sh <outdir> <kernel-substring> <counters...>
out=$1; kern=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/$out
rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/$out -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/$out.log 2>&1
python3 - "$kern" "$out" <<'PY'
import csv, glob, collections, sys
kern, out = sys.argv[1], sys.argv[2]
f = glob.glob(f"gpurun_out/{out}/**/*counter_collection.csv", recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if kern in r["Kernel_Name"]]
agg = collections.defaultdict(list)
for r in rows: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items(): print(f"{k:32s} mean {sum(v)/len(v):16.1f}  n={len(v)}")
PY
