#!/bin/bash
# Per kernel instantiation of a single-stream bs=64 forward + one post-processing step: where the SIMD cycles go (rocprofv3 PMC, two passes).
#   bash tools/sq_profile.sh [tag] [dtype=h2]  ->  gpurun_out/profiles/<tag>_sq_profile.txt
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/profiles; mkdir -p $out
export VTI_SINGLE_STREAM=1
args="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-line --no-host-fed --no-fp16-line --parity-frames 0 --preheat 0 --dtype ${2:-h2}"
rm -rf gpurun_out/_sq1 gpurun_out/_sq2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d gpurun_out/_sq1 -- python3 $args > /dev/null 2> gpurun_out/_sq1.err || exit 1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_MFMA --output-format csv -d gpurun_out/_sq2 -- python3 $args > /dev/null 2> gpurun_out/_sq2.err || exit 1
python3 - "$tag" <<'PY'
import csv, glob, collections, re, sys
tag = sys.argv[1]
def short(n):
    m = re.search(r"vti\d*(\w+?)I(DF16_|f|NS_4h2_tE)((?:L[ib]\d+E)*)", n)
    if m: return m.group(1) + "<" + {"DF16_": "h", "f": "f"}.get(m.group(2), "h2") + "," + ",".join(re.findall(r"L[ib](\d+)E", m.group(3))) + ">"
    m = re.search(r"vti::(\w+)<(?:vti::)?(\w+)((?:, [\w]+)*)>", n)                   # demangled names
    if m: return m.group(1) + "<" + {"_Float16": "h", "float": "f", "h2_t": "h2"}.get(m.group(2), m.group(2)) + m.group(3).replace(" ", "") + ">"
    m = re.search(r"vti::(\w+)", n)
    return m.group(1) if m else n[:40]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for d in ("gpurun_out/_sq1", "gpurun_out/_sq2"):
    f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if "vti" not in r["Kernel_Name"]: continue
        k = short(r["Kernel_Name"])
        name = r["Counter_Name"] + ("_2" if d.endswith("2") and r["Counter_Name"] == "SQ_BUSY_CYCLES" else "")
        acc[k][name] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_BUSY_CYCLES" and d.endswith("1"): cnt[k] += 1
L = ["rocprofv3 --pmc (two passes) -- python3 bench.py --steps 2 --warmup 1 ...   (VTI_SINGLE_STREAM=1, bs=64; sums over all launches of a kernel instantiation)",
     "kernel cycles per SIMD = SQ_BUSY_CYCLES / 32 (summed over the 32 shader engines); shares below are of those cycles:",
     "  valu = 4 * SQ_INSTS_VALU / 1024 SIMDs;  mfma = SQ_VALU_MFMA_BUSY_CYCLES / 1024;  lds = SQ_LDS_IDX_ACTIVE / 256 CUs (conf = bank-conflict share of it);",
     "  wait = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (share of resident wave time spent on s_waitcnt)",
     f"{'kernel<dtype,template args>':40s} {'launches':>8s} {'Mcyc/SIMD':>10s} {'valu':>6s} {'mfma':>6s} {'lds':>6s} {'conf':>6s} {'wait':>6s} {'salu/valu':>9s}"]
rows = []
for k, v in acc.items():
    b1, b2 = v.get("SQ_BUSY_CYCLES", 0.0), v.get("SQ_BUSY_CYCLES_2", 0.0)
    if not b1 or not b2: continue
    cyc1, cyc2 = b1 / 32, b2 / 32
    valu = 4 * v.get("SQ_INSTS_VALU", 0) / 1024 / cyc1
    mfma = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / cyc1
    lds = v.get("SQ_LDS_IDX_ACTIVE", 0) / 256 / cyc2
    conf = v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v.get("SQ_LDS_IDX_ACTIVE", 0), 1)
    wait = v.get("SQ_WAIT_INST_ANY", 0) / max(v.get("SQ_WAVE_CYCLES", 0), 1)
    sv = v.get("SQ_INSTS_SALU", 0) / max(v.get("SQ_INSTS_VALU", 0), 1) * (cyc1 / cyc2)
    rows.append((cyc1, f"{k:40s} {cnt[k]:8d} {cyc1/1e6:10.2f} {valu:6.2f} {mfma:6.2f} {lds:6.2f} {conf:6.2f} {wait:6.2f} {sv:9.2f}"))
L += [r for _, r in sorted(rows, key=lambda t: -t[0])]
open(f"gpurun_out/profiles/{tag}_sq_profile.txt", "w").write("\n".join(L) + "\n")
print("\n".join(L))
PY
