#!/bin/bash
# usage: pmc_one.sh <outdir> <counters...> -- <conv_bench args>
out=$1; shift
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/$out
rocprofv3 --pmc "${ctrs[@]}" --output-format csv -d gpurun_out/$out -- python3 tools/conv_bench.py "$@" > gpurun_out/$out.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/$out/**/*counter_collection.csv", recursive=True)
if not f: print("no counter file"); raise SystemExit
rows = [r for r in csv.DictReader(open(f[0])) if "conv_kernel" in r["Kernel_Name"]]
agg = collections.defaultdict(list)
for r in rows: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items(): print(f"{k:32s} mean {sum(v)/len(v):14.1f}  n={len(v)}")
PY
