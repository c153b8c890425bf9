"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel durations of the last forward pass."""
import csv, re, sys, glob
path = sys.argv[1]
per = int(sys.argv[2]) if len(sys.argv) > 2 else 80
f = glob.glob(path + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "vti" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-per:]
def short(n):
    m = re.search(r"conv_kernelI(DF16_|f)Li(\d)ELi(\d)ELi(\d)ELi(\d)", n)
    if m: return f"conv<{'h' if m.group(1) != 'f' else 'f'},k{m.group(2)},s{m.group(3)},n{m.group(4)},m{m.group(5)}>"
    m = re.search(r"conv3_pkI(DF16_|f)Li(\d)ELi(\d)ELi(\d)E", n)
    if m: return f"pk3<{'h' if m.group(1) != 'f' else 'f'},n{m.group(2)},w{m.group(3)},f{m.group(4)}>"
    m = re.search(r"vti(?:::|\d+)(\w+?)(?:I|E|\()", n)
    return m.group(1) if m else n[:30]
tot = 0
for i, r in enumerate(last):
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tot += d
    print(f"{i:2d} {short(r['Kernel_Name']):24s} {d/1000:8.1f} us  wgs {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])}x{r['Grid_Size_Y']} lds {r['LDS_Block_Size']} vgpr {r['VGPR_Count']}")
span = int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])
print(f"sum {tot/1e6:.3f} ms   span {span/1e6:.3f} ms   kernels {len(last)}")
