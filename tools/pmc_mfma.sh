#!/bin/bash
# MFMA-pipe utilisation per kernel over one bench step: one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE)
# over `bench.py --steps 2 --warmup 1 --no-graph --serial`.  SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe busy cycles
# summed over SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X guide), so
#   utilisation = MFMA_BUSY / (GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs).
# usage (on the GPU box): tools/pmc_mfma.sh OUT.csv
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out/pmc_mfma
rm -rf $out; mkdir -p $out
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/p -- python3 bench.py --steps 2 --warmup 1 \
    --no-graph --serial --no-cpu-baseline --no-roofline > $out/p.log 2>&1
python3 - "$out" "$1" <<'PY'
import csv, glob, sys, collections, re
out, dst = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(f"{out}/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        name = re.sub(r"\(.*$", "", name)
        if name.startswith("__amd_rocclr"):        # model set-up copies, not the step
            continue
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cnt[name] += 1
rows = []
for k, d in agg.items():
    gui, busy = d.get("GRBM_GUI_ACTIVE", 0.0), d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    if gui <= 0:
        continue
    rows.append((gui, k, cnt[k], busy / (gui / 8.0 * 1024.0)))
rows.sort(reverse=True)
tot_gui = sum(r[0] for r in rows)
tot_busy = sum(d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for d in agg.values())
with open(dst, "w") as fo:
    fo.write("kernel,launches,share_of_gpu_active_cycles,mfma_pipe_utilisation\n")
    for gui, k, n, u in rows:
        fo.write(f"\"{k}\",{n},{gui / tot_gui:.4f},{u:.4f}\n")
    fo.write(f"\"ALL KERNELS\",{sum(cnt.values())},1.0000,{tot_busy / (tot_gui / 8.0 * 1024.0):.4f}\n")
print(open(dst).read()[:2500])
PY
