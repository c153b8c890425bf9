#!/bin/bash
# Effective shader clock of the forward's kernels with the conv3_pk stagger off / on, same box (MI355X_MICROARCH.md, DVFS give-back:
# clock ~ GRBM_GUI_ACTIVE / 8 / kernel wall time; the quotient reads high on short dispatches, so compare arms, not absolutes).
#   bash tools/clock_ab.sh   -> gpurun_out/clock_ab.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export VTI_SINGLE_STREAM=1
for s in 0 10 0 10; do
  rm -rf gpurun_out/_clk && mkdir -p gpurun_out/_clk
  VTI_PK_STAGGER=$s rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/_clk -- python3 tools/prof_forward.py 64 h2 12 > /dev/null 2> gpurun_out/_clk.err || exit 1
  python3 - $s <<'PY'
import csv, glob, collections, sys
s = sys.argv[1]
kt = glob.glob("gpurun_out/_clk/**/*kernel_trace.csv", recursive=True)[0]
cc = glob.glob("gpurun_out/_clk/**/*counter_collection.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
for r in csv.DictReader(open(cc)):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
    d = dur.get(r["Dispatch_Id"])
    if not d or "vti" not in d[0]: continue
    fam = "conv3_pk" if "conv3_pk" in d[0] else "conv1_pk" if "conv1_pk" in d[0] else "other vti conv" if "conv" in d[0] or "stem" in d[0] or "bneck" in d[0] else "rest"
    a = agg[fam]; a[0] += float(r["Counter_Value"]); a[1] += d[1]; a[2] += 1
tot = [sum(v[0] for v in agg.values()), sum(v[1] for v in agg.values())]
line = f"stagger {s:>2s}: " + "  ".join(f"{k} {v[0] / 8 / v[1]:.3f} GHz ({v[1] / 1e3 / 13:.0f} us/fwd)" for k, v in sorted(agg.items())) + f"  | all {tot[0] / 8 / tot[1]:.3f} GHz, {tot[1] / 1e6 / 13:.3f} ms/fwd"
print(line)
open("gpurun_out/clock_ab.txt", "a").write(line + "\n")
PY
done
