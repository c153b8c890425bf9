#!/bin/bash
# usage: pmc_forward.sh <kernel-substring> : SQ counters (two passes) for the launches of one kernel instantiation in a single-stream forward
kern=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export VTI_SINGLE_STREAM=1
for pass in 1 2; do
  rm -rf gpurun_out/_pmc$pass
  if [ $pass = 1 ]; then ctr="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
  else ctr="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU"; fi
  rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/_pmc$pass -- python3 tools/prof_forward.py 64 fp16 2 > /dev/null 2> gpurun_out/_pmc$pass.err || { tail -5 gpurun_out/_pmc$pass.err; exit 1; }
done
python3 - "$kern" <<'PY'
import csv, glob, collections, sys
kern = sys.argv[1]
for d in ("gpurun_out/_pmc1", "gpurun_out/_pmc2"):
    f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items(): print(f"{k:28s} mean/launch {sum(v)/len(v):16.0f}  launches {len(v)}")
PY
