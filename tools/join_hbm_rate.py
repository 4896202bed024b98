"""Joins the PMC HBM-traffic table with the serial rocprofv3 kernel stats: measured HBM GB/s per kernel.
    python tools/join_hbm_rate.py profiles/r01_hbm_traffic_pmc_phi-l_bs8_512.csv \
        profiles/r01_kernel_stats_phi-l_bs8_512_serial.csv > profiles/r01_hbm_rate_per_kernel_phi-l_bs8_512.csv"""
import csv
import re
import sys


def norm(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*$", "", name)


traffic = {r["kernel"]: r for r in csv.DictReader(open(sys.argv[1]))}
stats = {}
for r in csv.DictReader(open(sys.argv[2])):
    k = norm(r["Name"])
    c, t = stats.get(k, (0, 0.0))
    stats[k] = (c + int(r["Calls"]), t + float(r["TotalDurationNs"]))
rows = []
for k, tr in traffic.items():
    if k not in stats or k.startswith("__amd"):
        continue
    calls, tot = stats[k]
    avg_us = tot / calls / 1e3
    mb = float(tr["avg_HBM_MB"])
    rows.append((tot, k, calls, avg_us, mb, mb * 1.048576 / avg_us * 1e3))        # MiB per us -> GB/s
rows.sort(reverse=True)
print("kernel,launches,avg_duration_us,avg_HBM_MB,HBM_GB_per_s,fraction_of_8TBps")
for tot, k, calls, us, mb, gbs in rows:
    print(f"\"{k}\",{calls},{us:.1f},{mb:.2f},{gbs:.0f},{gbs / 8000:.3f}")
