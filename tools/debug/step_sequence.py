"""Ordered kernel list of ONE eager, single-stream step (diagnostic: which launch sequences sit on the chain).

    rocprofv3 --kernel-trace -f csv -d gpurun_out/seq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --serial --no-graph [--phi nano]
    python3 tools/debug/step_sequence.py gpurun_out/seq > gpurun_out/step_sequence.txt

The last step is found as the last occurrence of the first kernel of the forward (nchw_to_nhwc of the image)."""
import csv
import glob
import re
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", n)[:70]


def main(d):
    kt = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
    rows = list(csv.DictReader(open(kt)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "nchw_to_nhwc" in r["Kernel_Name"]]
    # the forward opens with two layout transposes (image, radar): first of the last pair
    first = starts[-2] if len(starts) >= 2 else 0
    # walk back over the weight-preparation launches that precede them in stream order
    step = rows[first:]
    t0 = int(step[0]["Start_Timestamp"])
    prev_end = t0
    print(f"# {len(step)} kernels from the last forward's first layout transpose to the end of the trace")
    print("#  idx   start_us   dur_us   gap_us  grid        kernel")
    for i, r in enumerate(step):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        g = r.get("Grid_Size_X", r.get("Grid_Size", "?"))
        w = r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))
        print(f"{i:5d} {(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f} {(s - prev_end) / 1e3:8.1f}  {g:>8s}/{w:<4s} {short(r['Kernel_Name'])}")
        prev_end = e


if __name__ == "__main__":
    main(sys.argv[1])
