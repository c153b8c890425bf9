#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 evidence for bench.py, copied to profiles/ by the caller.
#   1. --kernel-trace --stats of the bench command             -> <tag>_bench_kernel_stats.csv + family summary
#   2. --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes)  -> <tag>_hbm_traffic.json: HBM bytes per forward / per step by kernel family
#   3. single-stream per-launch roofline table                  -> <tag>_op_roofline.txt   (tools/op_times.py)
#   4. --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES (own pass) -> <tag>_mfma_busy.txt: matrix-pipe busy share per conv kernel
# The program is always directly after `--` (no env/bash hop under rocprofv3).
tag=${1:-r03}
DT=${2:-h2}                 # engine dtype of the profiled run (the bench headline's)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/profiles; mkdir -p $out; rm -rf gpurun_out/_p1 gpurun_out/_p2 gpurun_out/_p3 gpurun_out/_p4 gpurun_out/_ops
args="bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-line --no-fp16-line --no-host-fed --parity-frames 0 --preheat 0.2 --dtype $DT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/_p1 -- python3 $args > $out/${tag}_bench_under_rocprof.json 2> gpurun_out/_p1.err || exit 1
cp $(find gpurun_out/_p1 -name '*kernel_stats.csv' | head -1) $out/${tag}_bench_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/_p2 -- python3 $args > /dev/null 2> gpurun_out/_p2.err || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/_p3 -- python3 $args > /dev/null 2> gpurun_out/_p3.err || exit 1
export VTI_SINGLE_STREAM=1
mkdir -p gpurun_out/_ops
VTI_LIST_OPS=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/_ops -- python3 tools/prof_forward.py 64 $DT 5 2> gpurun_out/_ops/ops.txt > /dev/null || exit 1
python3 tools/op_times.py gpurun_out/_ops 64 $DT > $out/${tag}_op_roofline.txt || exit 1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/_p4 -- python3 tools/prof_forward.py 64 $DT 3 > /dev/null 2> gpurun_out/_p4.err || exit 1
unset VTI_SINGLE_STREAM
python3 - "$tag" "$DT" <<'PY'
import csv, glob, collections, json, re, sys
tag, DT = sys.argv[1], sys.argv[2]
out = "gpurun_out/profiles"
def fam(n):
    if any(k in n for k in ("conv_kernel", "stem_kernel", "conv3_pk", "conv1_pk", "stem_l1", "bneck_pk", "convfold_kernel")):
        return "conv family (conv3_pk + conv1_pk + bneck_pk + conv_kernel + stem_l1_kernel)"
    for k in ("decode_kernel", "masks_group_kernel", "masks_kernel", "nms_kernel", "nms_scan_kernel", "mask_plan_kernel", "mask_clear_kernel", "mask_offsets_kernel",
              "sppf_pool", "upsample2x", "scale_boxes", "letterbox"):
        if k in n: return k
    return None
def short(n):
    m = re.search(r"vti\d*(\w+?)I(DF16_|f|NS_4h2_tE)((?:L[ib]\d+E)*)", n)          # mangled names
    if m: return m.group(1) + "<" + {"DF16_": "h", "f": "f"}.get(m.group(2), "h2") + "," + ",".join(re.findall(r"L[ib](\d+)E", m.group(3))) + ">"
    m = re.search(r"vti::(\w+)<(?:vti::)?(\w+)((?:, [\w]+)*)>", n)                   # demangled names
    if m: return m.group(1) + "<" + {"_Float16": "h", "float": "f", "h2_t": "h2"}.get(m.group(2), m.group(2)) + m.group(3).replace(" ", "") + ">"
    return n[:40]
# ---- kernel trace: per-family totals over the run
f = glob.glob("gpurun_out/_p1/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = fam(r["Kernel_Name"])
    if k: agg[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
nfwd = len(agg["sppf_pool"])                # one SPPF pool launch per forward
lines = [f"rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline ... --dtype {DT}   ({nfwd} forwards incl. calibration/pre-heat/warm-up/isolated timing)",
         "NOTE: the forward runs its independent branches on side streams, so kernels overlap: per-kernel durations here are",
         "inflated by sharing the chip and their sum exceeds the wall time of a forward (bench.py reports that).",
         f"Non-overlapped per-launch durations with work and roofline fractions: {tag}_op_roofline.txt (VTI_SINGLE_STREAM=1).",
         f"{'kernel family':80s} {'calls':>7s} {'avg us':>10s} {'us/forward (post-processing: us/step)':>12s}"]
npost = len(agg["nms_kernel"])
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    per = nfwd if (k.startswith("conv family") or k in ("sppf_pool", "upsample2x", "decode_kernel")) else max(npost, 1)
    lines.append(f"{k:80s} {len(v):7d} {sum(v)/len(v)/1e3:10.1f} {sum(v)/per/1e3:12.1f}")
open(f"{out}/{tag}_bench_kernel_summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
# ---- PMC: HBM bytes per forward (bs=64).  FETCH_SIZE/WRITE_SIZE are in KiB; gfx950 reports half the bytes of wide coalesced
# reads (MI355X_MICROARCH.md, HBM), so fetch is doubled.
def pmc_sum(d, name):
    f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
    tot = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = fam(r["Kernel_Name"])
        if k and r["Counter_Name"] == name:
            tot[k] += float(r["Counter_Value"]); cnt[k] += 1
    return tot, cnt
ft, fc = pmc_sum("gpurun_out/_p2", "FETCH_SIZE")
wt, wc = pmc_sum("gpurun_out/_p3", "WRITE_SIZE")
nf = fc["sppf_pool"]; nw = wc["sppf_pool"]
npf = max(fc["nms_kernel"], 1); npw = max(wc["nms_kernel"], 1)
res = {"dtype": DT, "note": "HBM traffic per forward (conv family, sppf_pool) / per bench step (post-processing kernels), bs=64, from rocprofv3 PMC, separate passes; FETCH_SIZE x2 correction for gfx950 applied",
       "families": {}}
for k in ft:
    fwdk = k.startswith("conv family") or k in ("sppf_pool", "upsample2x", "decode_kernel")
    fetch = ft[k] / (nf if fwdk else npf) * 1024 * 2; write = wt.get(k, 0.0) / (nw if fwdk else npw) * 1024
    res["families"][k] = {"fetch_bytes": fetch, "write_bytes": write, "total_bytes": fetch + write}
json.dump(res, open(f"{out}/{tag}_hbm_traffic.json", "w"), indent=1)
print(json.dumps({k: round(v["total_bytes"] / 1e6, 1) for k, v in res["families"].items()}, indent=1))
# ---- PMC: matrix-pipe busy share per conv kernel (single stream)
f = glob.glob("gpurun_out/_p4/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    if fam(r["Kernel_Name"]) and fam(r["Kernel_Name"]).startswith("conv family"):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
L = ["rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -- python3 tools/prof_forward.py 64 " + DT + " 3   (VTI_SINGLE_STREAM=1; sums over all launches of a kernel instantiation)",
     "SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs, SQ_BUSY_CYCLES over the 32 shader engines (32 SIMDs each), so",
     "matrix-pipe busy share per SIMD = MFMA_BUSY / (32 x SQ_BUSY)  [last column]",
     f"{'kernel<dtype,template args>':44s} {'MFMA_BUSY':>14s} {'SQ_BUSY':>14s} {'ratio':>8s} {'busy/SIMD':>10s}"]
tm = tb = 0.0
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0)):
    m, b = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), v.get("SQ_BUSY_CYCLES", 0.0)
    tm += m; tb += b
    L.append(f"{k:44s} {m:14.0f} {b:14.0f} {m / b if b else 0:8.3f} {m / b / 32 if b else 0:10.3f}")
L.append(f"{'all conv kernels':44s} {tm:14.0f} {tb:14.0f} {tm / tb if tb else 0:8.3f} {tm / tb / 32 if tb else 0:10.3f}")
open(f"{out}/{tag}_mfma_busy.txt", "w").write("\n".join(L) + "\n")
print("\n".join(L[:12]))
PY
