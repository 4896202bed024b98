"""Diagnostic: per (kernel, grid) launch durations of a rocprofv3 kernel trace (last `n` steps' worth is not separated:
all launches are pooled).   python tools/trace_shapes.py <kernel_trace.csv> [name filter]"""
import csv, sys
from collections import defaultdict
flt = sys.argv[2] if len(sys.argv) > 2 else "igemm_dma"
acc = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    if flt not in n:
        continue
    n = n[:n.find("(")] if "(" in n else n
    acc[(n, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))].append(
        (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = sorted(acc.items(), key=lambda kv: -sum(kv[1]))
print(f"{'kernel':44s} {'WGs':>8s} {'n':>5s} {'total ms':>9s} {'mean us':>8s} {'min':>7s} {'max':>7s}")
for (n, g, gy, gz), d in rows[:60]:
    print(f"{n:44s} {g:6d}x{gy}x{gz} {len(d):5d} {sum(d)/1e3:9.3f} {sum(d)/len(d):8.1f} {min(d):7.1f} {max(d):7.1f}")
