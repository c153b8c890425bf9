"""Label a single-stream kernel trace with the plan's op names.
   VTI_SINGLE_STREAM=1 VTI_LIST_OPS=1 rocprofv3 --kernel-trace --output-format csv -d D -- python3 tools/prof_forward.py 64 fp16 5 2> D/ops.txt
   python3 tools/op_times.py D"""
import csv, glob, re, sys
d = sys.argv[1]
ops = []
for l in open(d + "/ops.txt"):
    m = re.match(r"\[op\s+(\d+)\] lane (\d) (.*)", l)
    if m:
        i = int(m.group(1))
        if i == len(ops): ops.append((int(m.group(2)), m.group(3).strip()))
f = sorted(glob.glob(d + "/**/*_kernel_trace.csv", recursive=True))[-1]
rows = [r for r in csv.DictReader(open(f)) if "vti" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(ops)
nf = len(rows) // n
last = [rows[-n * k - n: len(rows) - n * k] for k in range(min(nf, 4))]
tot = 0
out = []
for i, (lane, name) in enumerate(ops):
    ds = sorted(int(fw[i]["End_Timestamp"]) - int(fw[i]["Start_Timestamp"]) for fw in last)
    med = ds[len(ds) // 2]
    tot += med
    kn = re.sub(r"^.*vti\d*", "", last[0][i]["Kernel_Name"])[:34]
    out.append((med, f"{i:2d} lane {lane} {name:44s} {med/1000:8.1f} us  {kn} lds {last[0][i]['LDS_Block_Size']}"))
for _, l in out: print(l)
print(f"sum {tot/1e6:.3f} ms over {n} launches")
print("--- top 15")
for _, l in sorted(out, reverse=True)[:15]: print(l)
