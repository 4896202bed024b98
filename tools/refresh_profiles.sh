#!/bin/bash
# Regenerates the evidence under profiles/ (run on the MI355X box; outputs land in gpurun_out/refresh/, copy them to
# profiles/ with the round prefix afterwards: tools/collect_profiles.sh rNN):  bash tools/refresh_profiles.sh rNN
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
R=${1:?round prefix, e.g. r04}
out=gpurun_out/refresh
rm -rf $out; mkdir -p $out
# ---- the other BASELINE configurations and comparison runs (one call, one box)
python3 bench.py --phi nano --no-cpu-baseline > $out/bench_phi-nano_bs8_512.json 2>/dev/null
python3 bench.py --no-cpu-baseline --dtype f32-mfma > $out/bench_phi-l_bs8_512_f32-mfma.json 2>/dev/null
python3 bench.py --no-cpu-baseline --dtype bf16 > $out/bench_phi-l_bs8_512_bf16.json 2>/dev/null
python3 bench.py --no-cpu-baseline --dtype bf16 --batch 16 > $out/bench_phi-l_bs16_512_bf16.json 2>/dev/null
python3 bench.py --no-cpu-baseline --dtype bf16 --batch 4 --size 1024 > $out/bench_phi-l_bs4_1024_bf16.json 2>/dev/null
python3 bench.py --no-cpu-baseline --batch 4 --size 1024 > $out/bench_phi-l_bs4_1024.json 2>/dev/null
python3 bench.py --no-cpu-baseline --detail 2> $out/per_shape_detail_phi-l_bs8_512.txt > /dev/null
python3 bench.py --no-cpu-baseline --detail --dtype bf16 --batch 16 2> $out/per_shape_detail_phi-l_bs16_512_bf16.txt > /dev/null
VRNET_BENCH_FORCE_DP=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 \
    bench.py --gpus 1 --no-cpu-baseline --no-roofline 2>/dev/null | grep metric > $out/line_dp1_rccl_3segments_phi-l_bs8_512.json
# ---- the issue ceiling of the x6 loop on THIS round's boxes (bench.py reads the newest round's file only; round-4 ADVICE: r04 had
# none and reported r03's number)
[ -x tools/micro/x6_peak.bin ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/micro/x6_peak.hip -o tools/micro/x6_peak.bin
(echo "# tools/micro/x6_peak.hip: register-resident x6 inner loop, no memory (mfma only / splits only / both)"; timeout 300 tools/micro/x6_peak.bin 2>&1 | grep -v amdgpu.ids) > $out/x6_issue_ceiling_micro.txt
cp $out/x6_issue_ceiling_micro.txt profiles/${R}_x6_issue_ceiling_micro.txt
# ---- isolated probe: the Cluster kernels alone
(echo "# python tools/cluster_probe.py  (Cluster kernels alone, phi = l, bs 8, 512 px)"; timeout 300 python3 tools/cluster_probe.py 2>&1 | grep -v amdgpu.ids) > $out/cluster_probe.txt
# ---- same-call A/B against the previous round's head (a worktree under .ab/base, built), when present
if [ -d .ab/base ]; then
  (echo "# tools/ab.sh 3: ms per step, the previous round's head (.ab/base) against this tree, alternating runs of one call"; bash tools/ab.sh 3 "previous-head|.ab/base|" "this-tree|.|" 2>&1) > $out/ab_vs_previous_head.txt
  (echo "# the schedule switches, one at a time (ms per step, same call)"
   for f in "" "--no-overlap-fusion" "--no-fused-fusion" "--early-wgrads 0" "--mlp-recompute off" ""; do
     ms=$(python3 bench.py --no-cpu-baseline --no-roofline $f 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])"); echo "[$f] $ms"; done) > $out/ab_schedule_switches.txt
fi
# ---- diagnostic build: launch-skipping ablations of the step
export VRNET_HIP_LIB=$PWD/asy-vrnet_amd/csrc/libvrnet_hip_tuning.so
[ asy-vrnet_amd/csrc/libvrnet_hip_tuning.so -nt asy-vrnet_amd/csrc/libvrnet_hip.so ] || echo "STALE diagnostic build (run make tuning after make)" >&2
tools/sweep_env.sh "" "VRNET_ABLATE=igemm" "VRNET_ABLATE=wgrad" "VRNET_ABLATE=igemm,wgrad" "VRNET_ABLATE=moments,affine" \
    "VRNET_ABLATE=cluster" "VRNET_ABLATE=misc,dwconv" "VRNET_ABLATE=igemm,wgrad,moments,affine,cluster,misc,dwconv" \
    "VRNET_ABLATE=igemm_small,wgrad_small" "VRNET_ABLATE=igemm_mid,wgrad_mid" "VRNET_ABLATE=igemm_big,wgrad_big" \
    "VRNET_SPLITK=0" "VRNET_BN_ZMASK=0" "" > $out/ablation_ms_per_step.txt 2>&1
FLAGS="--dtype bf16 --batch 16" tools/sweep_env.sh "" "VRNET_ABLATE=igemm" "VRNET_ABLATE=wgrad" "VRNET_ABLATE=igemm,wgrad" \
    "VRNET_ABLATE=moments,affine" "VRNET_ABLATE=cluster" "VRNET_ABLATE=misc,dwconv" > $out/ablation_ms_per_step_bf16_bs16.txt 2>&1
unset VRNET_HIP_LIB
# ---- rocprofv3: per-kernel totals (serial = the condition of bench.py's HIP-event measurement; hipgraph = the replayed step)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/graph -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > $out/graph.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/serial -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-roofline --serial --no-graph > $out/serial.log 2>&1
cp $(ls $out/graph/*/*kernel_stats.csv | head -1) $out/kernel_stats_phi-l_bs8_512_hipgraph.csv
cp $(ls $out/serial/*/*kernel_stats.csv | head -1) $out/kernel_stats_phi-l_bs8_512_serial.csv
python3 tools/exposure.py $(ls $out/graph/*/*kernel_trace.csv | head -1) 3 > $out/exposure_hipgraph.txt 2>&1
rm -rf $out/graph $out/serial
rocprofv3 --kernel-trace --stats --output-format csv -d $out/bf16 -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-roofline --serial --no-graph --dtype bf16 --batch 16 > $out/bf16.log 2>&1
cp $(ls $out/bf16/*/*kernel_stats.csv | head -1) $out/kernel_stats_phi-l_bs16_512_bf16_serial.csv
rm -rf $out/bf16
# ---- PMC passes (separate: FETCH_SIZE, WRITE_SIZE; then the matrix-pipe busy counter)
bash tools/pmc_traffic.sh $out/hbm_traffic_pmc_phi-l_bs8_512.csv > $out/pmc.log 2>&1
rm -rf gpurun_out/pmc_traffic
bash tools/pmc_mfma.sh $out/mfma_util_pmc_phi-l_bs8_512.csv > $out/pmc_mfma.log 2>&1
rm -rf gpurun_out/pmc_mfma
python3 tools/join_hbm_rate.py $out/hbm_traffic_pmc_phi-l_bs8_512.csv $out/kernel_stats_phi-l_bs8_512_serial.csv > $out/hbm_rate_per_kernel_phi-l_bs8_512.csv 2>/dev/null
# ---- the replayed step WITHOUT a tracer: device-clock stamps at the forks / chain ends / joins (a tracer delays the second chain)
(echo "# python tools/debug/section_stamps.py   (phi = l, bs 8, 512 px, fp32; device clock, us)"; python3 tools/debug/section_stamps.py 2>&1 | grep -v amdgpu.ids) > $out/section_timeline_untraced_phi-l_bs8_512.txt
(echo "# python tools/debug/section_stamps.py --bf16 --batch 16"; python3 tools/debug/section_stamps.py --bf16 --batch 16 2>&1 | grep -v amdgpu.ids) > $out/section_timeline_untraced_phi-l_bs16_512_bf16.txt
# ---- the headline line LAST, with the PMC traffic of THIS kernel source in place (bench.py checks the source hash), by the
# driver's exact command
cp $out/hbm_traffic_pmc_phi-l_bs8_512.csv profiles/${R}_hbm_traffic_pmc_phi-l_bs8_512.csv
cp $out/hbm_traffic_pmc_phi-l_bs8_512.csv.meta.json profiles/${R}_hbm_traffic_pmc_phi-l_bs8_512.csv.meta.json
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_phi-l_bs8_512.json 2> $out/bench_l.err
# the serial rocprofv3 summary of the SAME command line's instrumented replay conditions, for the roofline cross-check
# (tests/test_bench_contract.py: per-family FLOPs / rocprof duration within 5 % of the line's per-family rate)
ls -la $out
