#!/bin/bash
# Regenerates the evidence under profiles/ (run on the MI355X box; outputs land in gpurun_out/refresh/, copy them to
# profiles/ with the round prefix afterwards):  bash tools/refresh_profiles.sh
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
R=r03
out=gpurun_out/refresh
rm -rf $out; mkdir -p $out
python3 bench.py > $out/bench_phi-l_bs8_512.json 2> $out/bench_l.err
python3 bench.py --phi nano --no-cpu-baseline > $out/bench_phi-nano_bs8_512.json 2>/dev/null
python3 bench.py --no-cpu-baseline --dtype f32-mfma > $out/bench_phi-l_bs8_512_f32-mfma.json 2>/dev/null
python3 bench.py --no-cpu-baseline --no-fused-mlp > $out/bench_phi-l_bs8_512_no-fused-mlp.json 2>/dev/null
python3 bench.py --no-cpu-baseline --dtype bf16 > $out/bench_phi-l_bs8_512_bf16.json 2>/dev/null
python3 bench.py --no-cpu-baseline --dtype bf16 --batch 16 > $out/bench_phi-l_bs16_512_bf16.json 2>/dev/null
python3 bench.py --no-cpu-baseline --dtype bf16 --batch 4 --size 1024 > $out/bench_phi-l_bs4_1024_bf16.json 2>/dev/null
python3 bench.py --no-cpu-baseline --batch 4 --size 1024 > $out/bench_phi-l_bs4_1024.json 2>/dev/null
python3 bench.py --no-cpu-baseline --detail 2> $out/per_shape_detail_phi-l_bs8_512.txt > /dev/null
VRNET_BENCH_FORCE_DP=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 \
    bench.py --gpus 1 --no-cpu-baseline --no-roofline 2>/dev/null | grep metric > $out/line_dp1_rccl_3segments_phi-l_bs8_512.json
python3 tools/x6_probe.py > $out/x6_vs_fp32_mfma_gemm_probe.txt 2>/dev/null
python3 tools/x6_probe.py wgrad >> $out/x6_vs_fp32_mfma_gemm_probe.txt 2>/dev/null
# diagnostic build: fused-Mlp store / wait experiments and the launch-skipping ablations
export VRNET_HIP_LIB=$PWD/asy-vrnet_amd/csrc/libvrnet_hip_tuning.so
(for d in 0 1 2; do VRNET_MLP_DBG=$d python3 tools/mlp_probe.py 2>&1 | grep "^M"; done) > $out/mlp_fused_probe.txt
tools/sweep_env.sh "" "VRNET_ABLATE=igemm" "VRNET_ABLATE=wgrad" "VRNET_ABLATE=igemm,wgrad" "VRNET_ABLATE=moments,affine" \
    "VRNET_ABLATE=igemm,wgrad,moments,affine" > $out/ablation_ms_per_step.txt 2>&1
unset VRNET_HIP_LIB
rocprofv3 --kernel-trace --stats --output-format csv -d $out/graph -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > $out/graph.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/serial -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-roofline --serial --no-graph > $out/serial.log 2>&1
cp $(ls $out/graph/*/*kernel_stats.csv | head -1) $out/kernel_stats_phi-l_bs8_512_hipgraph.csv
cp $(ls $out/serial/*/*kernel_stats.csv | head -1) $out/kernel_stats_phi-l_bs8_512_serial.csv
rm -rf $out/graph $out/serial
bash tools/pmc_traffic.sh $out/hbm_traffic_pmc_phi-l_bs8_512.csv > $out/pmc.log 2>&1
rm -rf gpurun_out/pmc_traffic
bash tools/pmc_mfma.sh $out/mfma_util_pmc_phi-l_bs8_512.csv > $out/pmc_mfma.log 2>&1
rm -rf gpurun_out/pmc_mfma
python3 tools/join_hbm_rate.py $out/hbm_traffic_pmc_phi-l_bs8_512.csv $out/kernel_stats_phi-l_bs8_512_serial.csv > $out/hbm_rate_per_kernel_phi-l_bs8_512.csv 2>/dev/null
# the headline line last, with the PMC traffic of THIS kernel source in place (bench.py checks the source hash)
cp $out/hbm_traffic_pmc_phi-l_bs8_512.csv profiles/${R}_hbm_traffic_pmc_phi-l_bs8_512.csv
cp $out/hbm_traffic_pmc_phi-l_bs8_512.csv.meta.json profiles/${R}_hbm_traffic_pmc_phi-l_bs8_512.csv.meta.json
python3 bench.py > $out/bench_phi-l_bs8_512.json 2> $out/bench_l.err
(echo "# tools/micro/x6_overlap.hip: 24 MFMAs + 4 fragment splits per K16 step; clustered / software-pipelined, accumulators free / pinned to AGPRs"; timeout 120 tools/micro/x6_overlap.bin) > $out/x6_overlap_micro.txt 2>&1
(echo "# tools/micro/mfma_chain.hip: v_mfma_f32_32x32x16_bf16 issue rate vs number of independent accumulator chains"; timeout 120 tools/micro/mfma_chain.bin) > $out/mfma_chain_micro.txt 2>&1
(echo "# tools/micro/x6_peak.hip: register-resident x6 inner loop, no memory (mfma only / splits only / both)"; timeout 120 tools/micro/x6_peak.bin) > $out/x6_issue_ceiling_micro.txt 2>&1
# isolated probes and micro-benchmarks behind DESIGN 3.4 / 3.5
(echo "# python tools/cluster_probe.py  (Cluster kernels alone, phi = l, bs 8, 512 px)"; timeout 300 python3 tools/cluster_probe.py 2>&1 | grep -v amdgpu.ids) > $out/cluster_probe.txt
(echo "# python tools/planes_shape_probe.py B H W Cin Cout ...  (single conv launches alone: fp32 MFMA / x6 / x6 with pre-split weights)"; timeout 600 python3 tools/planes_shape_probe.py 8 32 32 320 1280 8 32 32 1280 320 8 32 32 256 320 8 32 32 320 512 8 16 16 512 2048 8 16 16 2048 512 8 16 16 512 512 8 64 64 128 256 8 64 64 256 1024 8 128 128 64 256 2>&1 | grep -v amdgpu.ids) > $out/planes_shape_probe.txt
(echo "# tools/micro/lds_fill.hip: per-CU rate of bringing L2-resident data into LDS (LDS-DMA / load + ds_write / load only)"; timeout 120 tools/micro/lds_fill.bin) > $out/lds_fill_micro.txt 2>&1
(echo "# tools/micro/wg_placement.hip: where the dispatcher puts the workgroups of an under-filled grid"; timeout 60 tools/micro/wg_placement.bin) > $out/wg_placement_micro.txt 2>&1
(echo "# tools/micro/dma_stream.hip: streaming a row-major fp32 matrix into LDS by LDS-DMA, by chunk size per row"; timeout 60 tools/micro/dma_stream.bin) > $out/dma_stream_micro.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/bf16 -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-roofline --serial --no-graph --dtype bf16 --batch 16 > $out/bf16.log 2>&1
cp $(ls $out/bf16/*/*kernel_stats.csv | head -1) $out/kernel_stats_phi-l_bs16_512_bf16_serial.csv
rm -rf $out/bf16
ls -la $out
