"""The bench line contract (driver + judge read it): checked on the committed round evidence and on bench.py's
argument defaults, without a GPU."""
import glob
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_lines_carry_every_field():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_*.json")))
    assert files
    for f in files:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                  "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
            assert k in d, (f, k)
        assert d["unit"] == "images/sec" and d["higher_is_better"] is True and d["scaling"] == "weak"
        assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
        assert abs(d["value"] - d["config"]["global_batch"] / d["ms_per_step"] * 1e3) < 0.01 * d["value"]
        r = d["roofline"]
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert k in r, (f, k)
        assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 2e-3
        if d["cpu_baseline"] is not None:
            for k in ("value", "unit", "cores", "kind", "sample"):
                assert k in d["cpu_baseline"], (f, k)
            assert d["cpu_baseline"]["kind"] in ("port", "reference")
    head = json.loads(open(os.path.join(ROOT, "profiles", "r01_bench_phi-l_bs8_512.json")).read().strip().splitlines()[-1])
    assert head["dtype"] == "f32" and head["n_gpus"] == 1 and head["cpu_baseline"] is not None
    assert "configs[1]" in head["config"]["workload"] and head["roofline"]["traffic"]
    head2 = json.loads(open(os.path.join(ROOT, "profiles", "r02_bench_phi-l_bs8_512.json")).read().strip().splitlines()[-1])
    assert head2["dtype"] == "f32" and head2["n_gpus"] == 1 and "configs[1]" in head2["config"]["workload"]
    cb = head2["cpu_baseline"]
    assert cb is not None and "bs=8" in cb["sample"] and "warm-up" in cb["sample"] and cb["cpu_model"] and cb["fwd_bs1_images_per_sec"] > 0
    fam = head2["roofline"]["families"]             # the roofline is priced per kernel family that actually ran
    assert any("x6" in k for k in fam) and abs(sum(v["share_of_flops"] for v in fam.values()) - 1.0) < 1e-2


def test_traffic_is_reported_only_for_the_measured_kernel_sources(tmp_path, monkeypatch):
    """roofline.traffic comes from a committed PMC table: it must be None as soon as the kernel sources differ from the
    ones the table was measured on."""
    import bench
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic_pmc_phi-l_bs8_512.csv")))
    meta = files[-1] + ".meta.json"
    assert os.path.exists(meta)
    same = json.load(open(meta))["kernel_source_hash"] == bench.kernel_source_hash()
    assert (bench.pmc_traffic_bytes("l") is not None) == same
    monkeypatch.setattr(bench, "kernel_source_hash", lambda: "0" * 16)
    assert bench.pmc_traffic_bytes("l") is None
    assert bench.pmc_traffic_bytes("nano") is None


def test_bench_defaults_are_the_headline_config():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for flag, default in (("--gpus", "1"), ("--phi", '"l"'), ("--batch", "8"), ("--size", "512"), ("--dtype", '"f32"')):
        m = re.search(r'add_argument\("%s"[^)]*default=([^,)]+)' % re.escape(flag), src)
        assert m and m.group(1).strip() == default, (flag, m and m.group(1))
    # nothing under /root/reference is read at run time; the oracle is imported only for the cpu_baseline leg
    assert "/root/reference" not in src


def test_bench_starts_its_ranks_as_a_child_before_anything_touches_hip():
    """`python bench.py --gpus N` without RANK in the environment must hand over to `python -m torch.distributed.run` as a CHILD
    process, with the loopback rendezvous the GPU box needs, BEFORE this process has imported the package (which loads the
    HIP library) or initialised the GPU: a process that has touched HIP must never start another program in its place."""
    import subprocess
    import sys
    probe = r'''
import sys, subprocess, json
sys.argv = ["bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1"]
seen = {}
def fake_run(cmd, *a, **k):
    import torch
    seen["cmd"] = cmd
    seen["pkg_loaded"] = any(m.startswith("asy_vrnet_amd") for m in sys.modules)
    seen["cuda_initialised"] = torch.cuda.is_initialized()
    class R: returncode = 7
    return R()
subprocess.run = fake_run
import os
os.environ.pop("RANK", None)
import bench
try:
    bench.main()
except SystemExit as e:
    seen["exit"] = e.code
print("PROBE" + json.dumps(seen))
'''
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, "-c", probe], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("PROBE")]
    assert line, out.stdout + out.stderr
    seen = json.loads(line[0][5:])
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nnodes=1" in cmd and "--nproc-per-node=2" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-6:] == ["--gpus", "2", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["exit"] == 7, "the child's return code is passed on"
    assert not seen["pkg_loaded"] and not seen["cuda_initialised"]


def test_roofline_families_follow_the_rocprof_summary_of_the_same_round():
    """The per-family rates in the committed headline line (HIP events, gated replay) against FLOPs / rocprofv3 serial
    duration of the same round's refresh call (profiles/rNN_kernel_stats_*_serial.csv), for every family that carries >= 5 % of
    the FLOPs: the line may read LOW by the dispatch gap + event packets an event pair adds (2.5-7 us per launch: 3-7 % on
    these 36-155 us launches; up to 9 % where split contractions put two launches into one timed call) and never high.  (Round 3's driver line read family 9 27 % low: the event pairs had included
    the host's launch latency; from round 4 the instrumented replay is issued behind a device-side gate.)"""
    import csv
    benches = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_phi-l_bs8_512.json")))
    line = json.loads(open(benches[-1]).read().strip().splitlines()[-1])
    rnd_ = os.path.basename(benches[-1]).split("_")[0]
    roof = line["roofline"]
    if "timing" not in roof:
        pytest.skip("no gated-replay line committed yet for the newest round")
    stats = os.path.join(ROOT, "profiles", f"{rnd_}_kernel_stats_phi-l_bs8_512_serial.csv")
    rows = list(csv.DictReader(open(stats)))
    fam_of = [("igemm_planes_kernel", "x6 with weights pre-split"), ("mlp_fused_kernel", "fused Mlp"),
              ("igemm_dma_kernel<0, 3, 2, 1, 6>", "x6: six exact"), ("igemm_dma_kernel<1, 3, 2, 1, 6>", "x6: six exact"),
              ("igemm_dma_kernel<0, 3, 2, 2, 6>", "x6: six exact"), ("igemm_dma_kernel<1, 3, 2, 2, 6>", "x6: six exact"),
              ("igemm_dma_kernel<0, 3, 1, 1, 0>", "fp32 MFMA, LDS-DMA"), ("igemm_dma_kernel<1, 3, 1, 1, 0>", "fp32 MFMA, LDS-DMA")]
    # the finishing launch of a split contraction belongs to the conv call the line timed (its time counts, it is no call)
    aux_of = [("igemm_splitk_finish_kernel<2, 1, 2, 2>", "x6: six exact"), ("igemm_splitk_finish_kernel<1, 2, 4, 1>", "x6 with weights pre-split")]
    total_gf = roof["avg_launch_gflop"] * roof["launches_per_step"]
    checked = 0
    for fam, v in roof["families"].items():
        if v["share_of_flops"] < 0.05 or not v["achieved"]:
            continue
        ns = calls = 0
        for r in rows:
            if any(k in r["Name"] and fam.startswith(f) for k, f in fam_of):
                ns += int(r["TotalDurationNs"])
                calls += int(r["Calls"])
            elif any(k in r["Name"] and fam.startswith(f) for k, f in aux_of):
                ns += int(r["TotalDurationNs"])
        assert calls and calls % v["launches_per_step"] == 0, (fam, calls, v["launches_per_step"])
        steps = calls // v["launches_per_step"]
        rate = v["share_of_flops"] * total_gf / (ns / steps * 1e-9) / 1e3          # TFLOP/s by rocprofv3
        us_prof = ns / calls / 1e3                                                  # average launch by rocprofv3
        us_line = us_prof * rate / v["achieved"]                                    # ... by the line's event pairs
        # (the two numbers come from two runs of the refresh call: the 150-190 us fused-Mlp launches, HBM-bound, repeat to a
        #  few per cent between runs, hence the relative terms beside the fixed 8 us)
        assert -0.5 - 0.03 * us_prof < us_line - us_prof < max(8.0, 0.07 * us_prof), \
            (fam, "rocprofv3", round(rate, 1), "line", v["achieved"], us_prof, us_line)
        # (a split contraction is two launches inside one timed call: one more dispatch gap that rocprofv3's durations do not
        #  contain -- 3.6 us on the 41 us average launch of the pre-split family from round 4 on)
        assert abs(rate - v["achieved"]) < 0.10 * rate, (fam, "rocprofv3", round(rate, 1), "line", v["achieved"])
        checked += 1
    assert checked >= 2
