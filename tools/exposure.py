"""Critical-path view of a rocprofv3 kernel trace of the replayed step (diagnostic).

    python tools/exposure.py <kernel_trace.csv> [n_steps_at_end=2]

For the last replayed steps: per kernel name, the time during which it was the ONLY kernel resident on the device
("exclusive": shortening it shortens the step one for one), the time it shared the device, launch count, and the idle
time of the device.  Kernel names are shortened to the text before the first '('."""
import csv
import sys
from collections import defaultdict


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    i = n.find("(")
    return n[:i] if i > 0 else n


def main():
    path = sys.argv[1]
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    # a step begins with the once-per-forward weight split (planes_pack_kernel / planes_split_kernel): take the last `nsteps`
    # whole steps (tracing stretches a step; host-side gaps are not a reliable boundary)
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    marks = [i for i, r in enumerate(rows) if r[2].startswith(("planes_pack_kernel", "planes_split_kernel"))]
    # (in eager runs the marker also fires at first-use registrations: keep marks that are >= 5 ms after the last KEPT one)
    steps = []
    for m in marks:
        if not steps or rows[m][0] - rows[steps[-1]][0] > 5_000_000:
            steps.append(m)
    if len(steps) > nsteps:
        # the last marker opens the final step, which ends with the trace: use the nsteps steps before it
        rows = rows[steps[-nsteps - 1]:steps[-1]]
    elif len(steps) >= 2:
        nsteps = len(steps) - 1
        rows = rows[steps[0]:steps[-1]]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    ev = []
    for s, e, n in rows:
        ev.append((s, 1, n))
        ev.append((e, -1, n))
    ev.sort(key=lambda x: (x[0], x[1]))
    live = defaultdict(int)
    excl, shared, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
    for s, e, n in rows:
        cnt[n] += 1
    last, idle, hist = t0, 0.0, defaultdict(float)
    nlive = 0
    for t, d, n in ev:
        dt = t - last
        if dt > 0:
            hist[min(nlive, 4)] += dt
            if nlive == 0:
                idle += dt
            elif nlive == 1:
                k = next(k for k, v in live.items() if v > 0)
                excl[k] += dt
            else:
                for k, v in live.items():
                    if v > 0:
                        shared[k] += dt
        live[n] += d
        nlive += d
        last = t
    span = (t1 - t0) / 1e6
    print(f"span {span:.3f} ms over {nsteps} step(s) -> {span / nsteps:.3f} ms/step; idle {idle / 1e6 / nsteps:.3f} ms/step; kernels {len(rows) // nsteps}/step")
    print("residency (ms/step):", {k: round(v / 1e6 / nsteps, 3) for k, v in sorted(hist.items())})
    print(f"{'kernel':60s} {'n/step':>7s} {'excl ms':>8s} {'shared ms':>9s}")
    for k in sorted(cnt, key=lambda k: -excl[k])[:45]:
        print(f"{k[:60]:60s} {cnt[k] / nsteps:7.1f} {excl[k] / 1e6 / nsteps:8.3f} {shared[k] / 1e6 / nsteps:9.3f}")


if __name__ == "__main__":
    main()
