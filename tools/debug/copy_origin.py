"""Which HIP calls do the `__amd_rocclr_copyBuffer` / fill dispatches of a step come from?

    rocprofv3 --kernel-trace --hip-runtime-trace -f csv -d gpurun_out/copyorg -- python3 bench.py --steps 2 --warmup 1 \
        --no-cpu-baseline --no-roofline --no-graph
    python3 tools/debug/copy_origin.py gpurun_out/copyorg

Joins the kernel trace with the HIP API trace on the correlation id and prints, per runtime-internal kernel, the API
functions that dispatched it and the kernels that ran just before / after it on the same queue (to locate the call site).

Round 6 finding (phi = l, bs 8): 2 382 copyBuffer dispatches whether the run has 3 or 6 steps -- they are the H2D copies of the
model's construction (hipMemcpyWithStream 1 553, hipMemcpyAsync 829), none of them inside a step.
"""
import collections
import csv
import glob
import sys


def main(d):
    kt = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))
    at = sorted(glob.glob(d + "/**/*hip_api_trace.csv", recursive=True))
    if not kt:
        print("no kernel trace under", d)
        return
    api = {}
    if at:
        for r in csv.DictReader(open(at[0])):
            api[r["Correlation_Id"]] = r["Function"]
    rows = list(csv.DictReader(open(kt[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    by_queue = collections.defaultdict(list)
    for r in rows:
        by_queue[r["Queue_Id"]].append(r)
    print("kernels:", len(rows), " api records:", len(api))
    internal = [r for r in rows if "rocclr" in r["Kernel_Name"] or "FillFunctor" in r["Kernel_Name"]]
    print("runtime-internal dispatches:", len(internal))
    fn = collections.Counter((r["Kernel_Name"][:40], api.get(r["Correlation_Id"], "?")) for r in internal)
    for k, n in fn.most_common():
        print(f"  {n:6d}  {k[0]:40s} <- {k[1]}")
    ctx = collections.Counter()
    sizes = collections.Counter()
    for q, lst in by_queue.items():
        for i, r in enumerate(lst):
            if "rocclr" not in r["Kernel_Name"]:
                continue
            prev = next((lst[j]["Kernel_Name"] for j in range(i - 1, -1, -1) if "rocclr" not in lst[j]["Kernel_Name"]), "-")
            nxt = next((lst[j]["Kernel_Name"] for j in range(i + 1, len(lst)) if "rocclr" not in lst[j]["Kernel_Name"]), "-")
            ctx[(prev.split("(")[0][-60:], nxt.split("(")[0][-60:])] += 1
            sizes[(r.get("Grid_Size_X", r.get("Grid_Size")), r.get("Workgroup_Size_X", r.get("Workgroup_Size")))] += 1
    print("grid sizes:", sizes.most_common(12))
    print("context (previous kernel -> next kernel on the same queue):")
    for k, n in ctx.most_common(40):
        print(f"  {n:6d}  {k[0]}  ->  {k[1]}")


if __name__ == "__main__":
    main(sys.argv[1])
