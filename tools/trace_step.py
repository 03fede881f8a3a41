"""Per-launch listing of ONE replayed training step out of a rocprofv3 --kernel-trace csv (the launches between the last two
AdamW kernels): duration, grid, VGPRs, LDS, kernel name.   python tools/trace_step.py <kernel_trace.csv> [name filter ...]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"] and int(r["Grid_Size_X"]) > 100000]
step = rows[idx[-2] + 1: idx[-1] + 1]
flt = sys.argv[2:]
tot = 0.0
for r in step:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*", "", n)
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    if flt and not any(f in n for f in flt):
        continue
    wg = int(r["Workgroup_Size_X"])
    print(f"{d:7.1f}  grid {int(r['Grid_Size_X']) // wg:6d}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']} wg {wg:4d} v{r['VGPR_Count']:>3s} lds {r['LDS_Block_Size']:>6s}  {n[:100]}")
print(f"{len(step)} launches, {tot / 1e3:.3f} ms of kernel time")
