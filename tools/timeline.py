"""Where in the replayed step is the device under-used (diagnostic; companion of tools/exposure.py).

    python tools/timeline.py <kernel_trace.csv> [bin_us=250]

For the last whole replayed step of a rocprofv3 kernel trace: one line per time bin with the average number of resident
kernels, the fraction of the bin with exactly one / no kernel resident, and the kernels that held most of the bin."""
import csv
import sys
from collections import defaultdict


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    i = n.find("(")
    n = n[:i] if i > 0 else n
    return n.replace("_kernel", "")


def main():
    path = sys.argv[1]
    bin_ns = int(float(sys.argv[2]) * 1000) if len(sys.argv) > 2 else 250_000
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if r[2].startswith(("planes_pack", "planes_split"))]
    steps = []
    for m in marks:
        if not steps or rows[m][0] - rows[steps[-1]][0] > 5_000_000:
            steps.append(m)
    rows = rows[steps[-2]:steps[-1]]
    t0 = rows[0][0]
    t1 = max(r[1] for r in rows)
    nb = (t1 - t0 + bin_ns - 1) // bin_ns
    busy = [defaultdict(float) for _ in range(nb)]
    ev = []
    for s, e, n in rows:
        ev.append((s, 1))
        ev.append((e, -1))
        b0, b1 = (s - t0) // bin_ns, (e - t0 - 1) // bin_ns
        for b in range(b0, min(b1, nb - 1) + 1):
            lo, hi = max(s, t0 + b * bin_ns), min(e, t0 + (b + 1) * bin_ns)
            if hi > lo:
                busy[b][n] += hi - lo
    ev.sort()
    one = [0.0] * nb
    none = [0.0] * nb
    live, last = 0, t0
    for t, d in ev:
        while last < t:
            b = (last - t0) // bin_ns
            if b >= nb:
                break
            hi = min(t, t0 + (b + 1) * bin_ns)
            if live == 1:
                one[b] += hi - last
            elif live == 0:
                none[b] += hi - last
            last = hi
        live += d
    print(f"step of {(t1 - t0) / 1e6:.3f} ms, {len(rows)} kernels, bins of {bin_ns / 1000:.0f} us: t(ms) avg-resident one% idle% | top kernels")
    for b in range(nb):
        tot = sum(busy[b].values())
        top = sorted(busy[b].items(), key=lambda kv: -kv[1])[:3]
        print(f"{b * bin_ns / 1e6:7.2f} {tot / bin_ns:5.2f} {100 * one[b] / bin_ns:4.0f} {100 * none[b] / bin_ns:4.0f} | "
              + "  ".join(f"{k[:34]} {v / bin_ns:.2f}" for k, v in top))


if __name__ == "__main__":
    main()
