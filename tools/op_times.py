"""Per-launch roofline table of one forward (single stream, so durations are the kernels' own):
   VTI_SINGLE_STREAM=1 VTI_LIST_OPS=1 rocprofv3 --kernel-trace --output-format csv -d D -- python3 tools/prof_forward.py 64 fp16 5 2> D/ops.txt
   python3 tools/op_times.py D [batch] [dtype]
Columns: duration (median of the last forwards), algorithmic GFLOP (2 x MAC of the convs the launch computes, as SURVEY 8d counts them), TFLOP/s and
fraction of the dense MFMA peak, algorithmic bytes (input of the first conv + output of the last + shortcut, B frames) and TB/s, fraction of 8 TB/s.
The plan (MACs, shapes) comes from the library's host-side tables: no GPU needed for that part."""
import csv, glob, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dtype = sys.argv[3] if len(sys.argv) > 3 else "fp16"
PEAK_TF = 157.3 if dtype == "fp32" else 2517.0      # h2 runs on the fp16 matrix pipe (4 MFMAs where fp16 needs one)
es = 2 if dtype == "fp16" else 4
import vti_amd
eng = vti_amd.Engine("n", 80, H=640, W=640, max_batch=B, dtype=dtype)
rows = {t["name"]: t for t in eng.conv_table()}
ops = []
for l in open(d + "/ops.txt"):
    m = re.match(r"\[op\s+(\d+)\] lane (\d) (.*)", l)
    if m:
        i = int(m.group(1))
        if i == len(ops): ops.append((int(m.group(2)), m.group(3).strip()))
f = sorted(glob.glob(d + "/**/*_kernel_trace.csv", recursive=True))[-1]
tr = [r for r in csv.DictReader(open(f)) if "vti" in r["Kernel_Name"]]
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(ops)
nf = len(tr) // n
last = [tr[-n * k - n: len(tr) - n * k] for k in range(min(nf, 4))]


def work(name):
    """(GFLOP, algorithmic MB) of one launch from the conv names in its label."""
    names = re.findall(r"model\.[0-9a-z_.]+", name)
    convs = [rows[x] for x in names if x in rows]
    if not convs:
        return 0.0, 0.0
    gf = 2.0 * sum(c["macs"] for c in convs) * B / 1e9
    order = sorted(convs, key=lambda c: list(rows).index(c["name"]))
    first, lastc = order[0], order[-1]
    in_b = first["h_in"] * first["w_in"] * first["c1"] * (1 if first["c1"] == 3 else es)
    out_es = 4 if lastc["kind"] == 1 else es             # head outputs are fp32 rows of pred
    out_b = lastc["h_out"] * lastc["w_out"] * lastc["c2"] * out_es
    if re.search(r"\.m\.\d+\.cv2", lastc["name"]) and int(lastc["name"].split(".")[1]) in (2, 4, 6, 8):
        out_b += lastc["h_out"] * lastc["w_out"] * lastc["c2"] * es * (0 if len(order) > 1 else 1)    # shortcut read (from LDS when fused)
    return gf, (in_b + out_b) * B / 1e6


tot = tot_gf = 0
out = []
hdr = f"{'#':>2s} {'ln':>2s} {'op':58s} {'us':>7s} {'GFLOP':>7s} {'TF/s':>6s} {'%MFMA':>6s} {'MB':>7s} {'TB/s':>5s} {'%HBM':>5s}  kernel"
for i, (lane, name) in enumerate(ops):
    ds = sorted(int(fw[i]["End_Timestamp"]) - int(fw[i]["Start_Timestamp"]) for fw in last)
    med = ds[len(ds) // 2]
    tot += med
    gf, mb = work(name)
    tot_gf += gf
    us = med / 1e3
    tf = gf / us * 1e3 if us else 0.0                        # GFLOP / us = 1000 TFLOP/s
    tb = mb / us if us else 0.0                              # MB / us = TB/s
    kn = re.sub(r"^.*vti\d*", "", last[0][i]["Kernel_Name"])[:30]
    out.append((med, f"{i:2d} {lane:2d} {name[:58]:58s} {us:7.1f} {gf:7.2f} {tf:6.0f} {100 * tf / PEAK_TF:6.1f} {mb:7.1f} {tb:5.2f} {100 * tb / 8.0:5.1f}  {kn}"))
print(f"single-stream forward, B={B} {dtype}: per-launch durations (median of the last {len(last)} forwards), algorithmic work, rooflines (MFMA {PEAK_TF:.0f} TF/s, HBM 8.0 TB/s)")
print(hdr)
for _, l in out: print(l)
print(f"sum {tot/1e6:.3f} ms over {n} launches; {tot_gf:.1f} GFLOP -> {tot_gf / (tot / 1e3) * 1e3:.0f} TF/s single-stream ({100 * tot_gf / (tot / 1e3) * 1e3 / PEAK_TF:.1f} % of MFMA peak; the multi-stream forward bench.py times is shorter)")
print("--- top 12 by time")
print(hdr)
for _, l in sorted(out, reverse=True)[:12]: print(l)
