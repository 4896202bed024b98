"""Which kernels surround each __amd_rocclr_copyBuffer / at::native launch in a rocprofv3 --kernel-trace csv (diagnostic)."""
import collections
import csv
import glob
import sys

files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "")[:60]


ctx = collections.Counter()
for i, r in enumerate(rows):
    n = r["Kernel_Name"]
    if "rocclr_copyBuffer" in n or "at::native" in n:
        prev = short(rows[i - 1]["Kernel_Name"]) if i else "-"
        nxt = short(rows[i + 1]["Kernel_Name"]) if i + 1 < len(rows) else "-"
        ctx[(short(n)[:40], prev, nxt, r.get("Grid_Size", "?"))] += 1
print(len(rows), "dispatches")
for k, v in ctx.most_common(40):
    print(v, k)
