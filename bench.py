"""Headline benchmark: images/sec forward+backward of the ASY-VRNet fusion hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--phi l|nano|...] [--batch 8] [--size 512]

N > 1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`
(one rank per GPU, RCCL).  A "step" is one forward+backward pass of EfficientVRNet over one
synthetic batch (BASELINE.json configs[1]: bs=8/GPU, 512x512 image + 4x512x512 radar, fp32, det+seg
heads), driven by the fixed scalar sum_k mean(det_k^2) + mean(seg^2).  Inputs are resident in HBM
before the timed region.  Rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline      dominant kernel class (implicit-GEMM conv fwd/dgrad on the fp32 MFMA): algorithmic FLOPs per
                launch / average launch duration, measured with HIP events on the launch stream in an
                instrumented, serialised replay of the timed steps (the timed region itself runs as one captured
                hipGraph with independent chains on concurrent streams, where an event pair would bracket several
                overlapping kernels); profiles/ holds the rocprofv3 summary of the same command.
  cpu_baseline  the CPU oracle (pure-torch restatement, kind "port") timed on the host cores, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32
BF16_MFMA_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md, dense bf16 MFMA (no sparsity)
X6_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 6.0   # fp32-equivalent FLOP/s of the x6 kernels: six bf16 MFMAs per fp32 product block
# matrix-pipe ceiling per kernel family (hip.last_kernel()): fp32-equivalent TFLOP/s
FAMILY_PEAK = {1: FP32_MFMA_PEAK_TFLOPS, 2: FP32_MFMA_PEAK_TFLOPS, 3: BF16_MFMA_PEAK_TFLOPS, 4: None, 5: None, 6: X6_PEAK_TFLOPS,
               7: X6_PEAK_TFLOPS, 8: BF16_MFMA_PEAK_TFLOPS, 9: X6_PEAK_TFLOPS, 10: X6_PEAK_TFLOPS, 11: BF16_MFMA_PEAK_TFLOPS,
               12: X6_PEAK_TFLOPS, 13: BF16_MFMA_PEAK_TFLOPS}
FAMILY_NAME = {1: "fp32 MFMA, register-staged", 2: "fp32 MFMA, LDS-DMA ring", 3: "bf16-rounded operands",
               4: "direct (tiny channel counts, no MFMA)", 5: "direct (narrow outputs over wide inputs, HBM streams, no MFMA)", 6: "x6: six exact bf16 x bf16 products per fp32 product",
               7: "fused Mlp (fc1 -> GELU -> fc2 in one kernel), x6, weights pre-split into bf16 planes",
               8: "fused Mlp, bf16-rounded operands",
               9: "x6 with weights pre-split into bf16 planes once per step (1x1 convs)",
               10: "plane GEMM, x6: both operands already bf16 planes in HBM (1x1 convs of the ClusterBlocks)",
               11: "plane GEMM on bf16 tensors (compute_dtype bf16)",
               12: "plane weight gradient, x6", 13: "plane weight gradient on bf16 tensors"}


def kernel_source_hash():
    """Hash of the kernel sources: ties a committed PMC table to the build it was measured on."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "asy-vrnet_amd", "csrc", "*.hip")) +
                    glob.glob(os.path.join(ROOT, "asy-vrnet_amd", "csrc", "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic_bytes(phi):
    """Average HBM bytes per igemm launch from the committed rocprofv3 PMC passes (profiles/; FETCH_SIZE doubled
    for gfx950 as the MI355X guide prescribes, separate --pmc passes).  The table carries the hash of the kernel
    sources it was measured on (<csv>.meta.json, written by tools/pmc_traffic.sh): None when no table exists or when
    the kernels have changed since -- a stale table is never reported."""
    import csv
    import glob
    if phi != "l":
        return None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic_pmc_phi-l_bs8_512.csv")))
    if not files:
        return None
    try:
        meta = json.load(open(files[-1] + ".meta.json"))
    except (OSError, ValueError):
        return None
    if meta.get("kernel_source_hash") != kernel_source_hash():
        return None
    n, tot = 0, 0.0
    for row in csv.DictReader(open(files[-1])):
        if row["kernel"].startswith(("igemm_kernel", "igemm_dma_kernel", "igemm_planes_kernel", "mlp_fused_kernel", "tiny::conv_kernel", "tiny::conv_fixed_kernel",
                                     "narrow::fwd_kernel", "narrow::dgrad_kernel")):
            k = int(row["launches"])
            n += k
            tot += k * float(row["avg_HBM_MB"]) * 1024 * 1024
    return round(tot / n) if n else None


def x6_issue_ceiling():
    """(rate, file) of the register-resident x6 loop with in-register operand splits (tools/micro/x6_peak.hip), read from
    the NEWEST round's committed micro-benchmark output only (an older round's file may describe other kernel sources);
    (None, None) when no such file exists."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_x6_issue_ceiling_micro.txt")))
    if not files:
        return None, None
    best = None
    for line in open(files[-1]):
        m = re.match(r"\s*2x2 both .*?:\s*[\d.]+ ms\s+([\d.]+) TF", line)
        if m:
            best = max(best or 0.0, float(m.group(1)))
    return best, "profiles/" + os.path.basename(files[-1])


def conv_flops(B, H, W, Cin, OH, OW, Cout, kh, kw, stride, pad, dil):
    """Algorithmic FLOPs of one conv launch (forward, data gradient or weight gradient alike): 2*B*OH*OW*Cout*Cin*kh*kw,
    the convention of torch's flop counter -- except for dilated convs, where only multiply-adds whose
    tap lands inside the image are counted (at 16x16 the d=12/18 ASPP branches have 5-8 of 9 taps entirely in the
    padding; the kernels skip them, and counting them would inflate the achieved rate)."""
    if dil == 1:
        return 2.0 * B * OH * OW * Cout * Cin * kh * kw
    ny = sum(sum(1 for oy in range(OH) if 0 <= oy * stride - pad + ky * dil < H) for ky in range(kh))
    nx = sum(sum(1 for ox in range(OW) if 0 <= ox * stride - pad + kx * dil < W) for kx in range(kw))
    return 2.0 * B * Cout * Cin * ny * nx


def loss_of(det, seg):
    """The fixed synthetic scalar that drives the backward pass: L = sum_k mean(det_k^2) + mean(seg^2).  On the GPU
    it is evaluated, with its gradient, by asy_vrnet_amd.losses.mean_square_loss (three launches; the same value and gradient
    as the eager torch expression below, tests/test_loss.py) -- the ~27 eager elementwise / reduce launches of the torch form
    sat between the forward and the backward pass, 0.25 ms of the timed step that is the harness's, not the path's (of the
    driver-to-driver change round 4 -> 5, 26.21 -> 23.71 ms, that 0.25 ms is this harness change; rounds 1-4 timed the torch
    form).  The CPU baseline (tensors on the host) and non-fp32 outputs (autocast) evaluate the torch form."""
    if seg.is_cuda and seg.dtype == torch.float32 and all(d.dtype == torch.float32 for d in det):
        from asy_vrnet_amd.losses import mean_square_loss
        return mean_square_loss(det, seg)
    return sum((d * d).mean() for d in det) + (seg * seg).mean()


def make_batches(n, batch, size, rank, device):
    out = []
    for s in range(n):
        g = torch.Generator().manual_seed(1234 + 1000 * rank + s)
        x = torch.randn((batch, 3, size, size), generator=g)
        r = torch.rand((batch, 4, size, size), generator=g)
        out.append((x.to(device), r.to(device)))
    return out


class ConvTimer:
    """Wraps hip.conv2d / hip.conv2d_wgrad with HIP events (torch.cuda.Event on the current = launch stream).
    A pair's elapsed time is the kernel plus the dispatch gap of a dependent launch and the event packets -- about 2.6 us per
    launch against rocprofv3's begin-to-end stamps on the same build (profiles/r04: 3-7 % on 36-155 us launches); it is NOT
    subtracted (a pair around an "empty" kernel measures 6.2 us, mostly that kernel): the rates in the line are that much
    conservative, and tests/test_bench_contract.py holds them to the rocprofv3 summary of the same round."""

    @staticmethod
    def dur(e0, e1):
        return e0.elapsed_time(e1)

    def __init__(self, hip):
        self.hip = hip
        self.rec = {"igemm": [], "wgrad": [], "cluster_fwd": [], "cluster_bwd": []}
        self.orig = (hip.conv2d, hip.conv2d_wgrad, hip.cluster_fwd, hip.cluster_bwd)
        self.orig_mlp = (hip.mlp_fwd, hip.mlp_bwd, hip.mlp_bwd_rc)
        self.orig_planes = (hip.gemm_planes, hip.wgrad_planes)

    def __enter__(self):
        hip, rec = self.hip, self.rec
        o_conv, o_wgrad, o_cf, o_cb = self.orig

        def conv2d(a, lda, w, bias, y, ldy, B, H, W, Cin, OH, OW, Cout, kh, kw, stride, pad, dil, *rest, **kw_):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            o_conv(a, lda, w, bias, y, ldy, B, H, W, Cin, OH, OW, Cout, kh, kw, stride, pad, dil, *rest, **kw_)
            e1.record()
            rec["igemm"].append((conv_flops(B, H, W, Cin, OH, OW, Cout, kh, kw, stride, pad, dil), e0, e1,
                                 f"M{B*OH*OW} N{Cout} K{Cin}x{kh}x{kw}" + (f"d{dil}" if dil > 1 else "") + f" k{hip.last_kernel()}",
                                 hip.last_kernel()))

        def conv2d_wgrad(x, ldx, dy, lddy, dw, db, rs, B, H, W, Cin, OH, OW, Cout, kh, kw, stride, pad, dil, *rest, **kw_):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            o_wgrad(x, ldx, dy, lddy, dw, db, rs, B, H, W, Cin, OH, OW, Cout, kh, kw, stride, pad, dil, *rest, **kw_)
            e1.record()
            rec["wgrad"].append((conv_flops(B, H, W, Cin, OH, OW, Cout, kh, kw, stride, pad, dil), e0, e1,
                                 f"M{B*OH*OW} N{Cout} K{Cin}x{kh}x{kw}" + (f"d{dil}" if dil > 1 else "") + f" k{hip.last_kernel()}",
                                 hip.last_kernel()))
        def cluster_fwd(f, v, ld, alpha, beta, out, ldo, idx, wgt, B, H, W, E, Dh, fold, **kw_):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            o_cf(f, v, ld, alpha, beta, out, ldo, idx, wgt, B, H, W, E, Dh, fold, **kw_)
            e1.record()
            rec["cluster_fwd"].append((3.0 * B * H * W * E * Dh * 4, e0, e1, f"{H}x{W} E{E} D{Dh} fold{fold}"))      # read f, v; write out

        def cluster_bwd(f, v, ld, alpha, beta, idx, dout, lddo, df, dv, lddf, da, db, acc, B, H, W, E, Dh, fold, **kw_):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = o_cb(f, v, ld, alpha, beta, idx, dout, lddo, df, dv, lddf, da, db, acc, B, H, W, E, Dh, fold, **kw_)
            e1.record()
            rec["cluster_bwd"].append((5.0 * B * H * W * E * Dh * 4, e0, e1, f"{H}x{W} E{E} D{Dh} fold{fold}"))      # read f, v, g; write df, dv
            return out                # (the workspace holding the (d alpha, d beta) partials when their reduction is deferred)
        hip.conv2d, hip.conv2d_wgrad, hip.cluster_fwd, hip.cluster_bwd = conv2d, conv2d_wgrad, cluster_fwd, cluster_bwd
        o_mf, o_mb, o_mbrc = self.orig_mlp

        # the fused Mlp launches belong to the same kernel class (dense conv forward / data gradient): one launch does
        # the work of two 1x1 convs, 2 * (2 * M * C * HID) FLOPs
        def mlp_fwd(*a):
            M, C, HID = a[-4], a[-3], a[-2]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            o_mf(*a)
            e1.record()
            rec["igemm"].append((4.0 * M * C * HID, e0, e1, f"mlpF M{M} C{C} H{HID} k{hip.last_kernel()}", hip.last_kernel()))

        def mlp_bwd(*a):
            M, C, HID = a[-4], a[-3], a[-2]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            o_mb(*a)
            e1.record()
            rec["igemm"].append((4.0 * M * C * HID, e0, e1, f"mlpB M{M} C{C} H{HID} k{hip.last_kernel()}", hip.last_kernel()))

        def mlp_bwd_rc(*a):       # (the recomputed first GEMM is not algorithmic work: the same 4 M C HID as mlp_bwd)
            M, C, HID = a[-4], a[-3], a[-2]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            o_mbrc(*a)
            e1.record()
            rec["igemm"].append((4.0 * M * C * HID, e0, e1, f"mlpBrc M{M} C{C} H{HID} k{hip.last_kernel()}", hip.last_kernel()))
        hip.mlp_fwd, hip.mlp_bwd, hip.mlp_bwd_rc = mlp_fwd, mlp_bwd, mlp_bwd_rc
        o_gp, o_wp = self.orig_planes

        # the plane GEMMs (csrc/pgemm.hip) are launches of the same two classes: a 1x1 conv's forward / data gradient, its
        # weight gradient
        def gemm_planes(a, b, M, N, K, *rest, **kw_):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            o_gp(a, b, M, N, K, *rest, **kw_)
            e1.record()
            rec["igemm"].append((2.0 * M * N * K, e0, e1, f"M{M} N{N} K{K}x1x1 k{hip.last_kernel()}", hip.last_kernel()))

        def wgrad_planes(x, dy, M, Cin, Cout, *rest, **kw_):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            o_wp(x, dy, M, Cin, Cout, *rest, **kw_)
            e1.record()
            rec["wgrad"].append((2.0 * M * Cin * Cout, e0, e1, f"M{M} N{Cout} K{Cin}x1x1 k{hip.last_kernel()}", hip.last_kernel()))
        hip.gemm_planes, hip.wgrad_planes = gemm_planes, wgrad_planes
        return self

    def __exit__(self, *a):
        self.hip.conv2d, self.hip.conv2d_wgrad, self.hip.cluster_fwd, self.hip.cluster_bwd = self.orig
        self.hip.mlp_fwd, self.hip.mlp_bwd, self.hip.mlp_bwd_rc = self.orig_mlp
        self.hip.gemm_planes, self.hip.wgrad_planes = self.orig_planes

    def summary(self, key):
        torch.cuda.synchronize()
        r = self.rec[key]
        flops = sum(x[0] for x in r)
        ms = sum(self.dur(x[1], x[2]) for x in r)
        return len(r), flops, ms

    def by_family(self, key):
        """{kernel family: [launches, flops, ms]} and the matrix-pipe-ideal time of the mix (ms)."""
        torch.cuda.synchronize()
        fam, ideal = {}, 0.0
        for x in self.rec[key]:
            a = fam.setdefault(x[4], [0, 0.0, 0.0])
            a[0] += 1
            a[1] += x[0]
            a[2] += self.dur(x[1], x[2])
            pk = FAMILY_PEAK.get(x[4])
            if pk:
                ideal += x[0] / (pk * 1e12) * 1e3
        return fam, ideal

    def detail(self, key, unit_scale, unit):
        """Per-shape breakdown (stderr aid for tuning): launches, total ms, achieved rate."""
        torch.cuda.synchronize()
        agg = {}
        for work, e0, e1, tag, *_ in self.rec[key]:
            a = agg.setdefault(tag, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += work
            a[2] += self.dur(e0, e1)
        for tag, (n, work, ms) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
            print(f"  {key:12s} {tag:28s} x{n:3d} {ms:8.3f} ms  {work / ms * 1e3 / unit_scale:9.1f} {unit}", file=sys.stderr)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(phi, size, batch, seed_sd):
    """the CPU oracle (the build's restatement of the reference, kind "port") forward+backward on the host
    cores at the benchmark's own batch (bs 8 by default), 1 warm-up + 2 timed iterations (a bounded sample: ~10 s per
    iteration for phi=l at bs 8), plus a bs=1 eval forward; threads = cores this process may use, capped at 32
    (measured in round 1: more threads make the oracle slower)."""
    from oracle import vrnet_oracle as O
    import asy_vrnet_amd as A
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(avail, 32)))
    m = A.EfficientVRNet(4, 9, phi, img_size=size)
    A.randomize_state_dict(m.state_dict(), seed=seed_sd)
    pn = {k for k, _ in m.named_parameters()}
    P = {k: (v.detach().clone().requires_grad_(k in pn and v.numel() > 0) if v.dtype.is_floating_point else v.clone())
         for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(1234)
    x, r = torch.randn((batch, 3, size, size), generator=g), torch.rand((batch, 4, size, size), generator=g)

    def fwd_bwd():
        for t in P.values():
            if t.dtype.is_floating_point:
                t.grad = None
        det, seg, _ = O.forward(P, x, r, phi, True)
        loss_of(det, seg).backward()
    t0 = time.perf_counter()
    fwd_bwd()                                   # warm-up (allocator, oneDNN primitive caches)
    warm = time.perf_counter() - t0
    iters, t0 = 0, time.perf_counter()
    while iters < 2 or (time.perf_counter() - t0 < 10.0 and iters < 8):
        fwd_bwd()
        iters += 1
    el = time.perf_counter() - t0
    with torch.no_grad():
        O.forward(P, x[:1], r[:1], phi, False)
        t1 = time.perf_counter()
        n1 = 0
        while n1 < 3:
            O.forward(P, x[:1], r[:1], phi, False)
            n1 += 1
        f1 = (time.perf_counter() - t1) / n1
    return {"value": round(iters * batch / el, 4), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "cpu_model": cpu_model(), "host_cores": os.cpu_count(),
            "fwd_bs1_images_per_sec": round(1.0 / f1, 3),
            "sample": f"CPU oracle (pure-torch restatement of the reference) fwd+bwd, phi={phi}, bs={batch}, "
                      f"{size}x{size}, 1 warm-up ({warm:.1f} s) + {iters} timed iteration(s) in {el:.1f} s; "
                      f"bs=1 eval forward {1e3 * f1:.0f} ms"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--phi", default="l")
    ap.add_argument("--batch", type=int, default=8, help="per-GPU batch")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=0, help="batch of the CPU baseline (default: --batch)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--detail", action="store_true", help="per-shape kernel breakdown on stderr (tuning aid)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f32-mfma", "bf16"],
                    help="f32 (headline metric, BASELINE configs[1]: fp32 tensors, fp32-accurate products -- six exact "
                         "bf16 x bf16 products per fp32 product on the bf16 MFMA where a kernel exists, fp32 MFMA elsewhere), "
                         "f32-mfma (fp32 MFMA only) or bf16-operand dense convs (configs[2..4])")
    ap.add_argument("--serial", action="store_true", help="one stream: no concurrent chains (diagnostic)")
    ap.add_argument("--pair", action="store_true",
                    help="image and radar chain of every backbone stage as ONE two-stream batch (one launch per layer) "
                         "instead of two chains on two forked streams")
    ap.add_argument("--no-pair", action="store_true", help="(default) two chains on two forked streams")
    ap.add_argument("--no-fused-mlp", action="store_true", help="Mlp as two conv launches (A/B aid)")
    ap.add_argument("--no-weight-planes", action="store_true", help="x6 kernels split the weights themselves (A/B aid)")
    ap.add_argument("--no-bn-colstats", action="store_true", help="BatchNorm statistics by a pass over z (A/B aid)")
    ap.add_argument("--no-overlap-fusion", action="store_true", help="RadarEnhanceByImage in front of both chains, as rounds 1-4 (A/B aid)")
    ap.add_argument("--early-wgrads", type=int, default=2, help="a section's deferred weight gradients start right behind it: 0 never, "
                    "1 every section (measured slower), 2 the last section only (default)")
    ap.add_argument("--no-fused-fusion", action="store_true", help="fusion blocks without the fused passes of csrc/fusion.hip (A/B aid)")
    ap.add_argument("--no-fused-upsample", action="store_true", help="A/B: CoCUpsample as conv -> BN apply -> upsample (three launches)")
    ap.add_argument("--mlp-recompute", default="auto", help="fused Mlp backward recomputes the pre-activation: auto (default), all, off")
    ap.add_argument("--diagnostic", action="store_true",
                    help="allow VRNET_* environment knobs and the diagnostic library build (tools/sweep_env.sh ablations); the "
                         "line is then marked `diagnostic`, its metric string says so, and it is not a measurement")
    ap.add_argument("--plane-gemms", default=None, help="ClusterBlock GEMMs on bf16-plane operands (csrc/pgemm.hip): off, fwd, wgrad, fwd+wgrad "
                    "(default: off in fp32, fwd+wgrad in bf16 mode) (A/B aid)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of one captured hipGraph per step")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # Started directly (not by torchrun): start the N ranks ourselves, as a CHILD process, before anything in this
        # process has touched the GPU (a process that initialised HIP must never exec / be replaced), and pass its
        # return code on.
        import socket
        import subprocess
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd).returncode)

    # Measurement hygiene: no tuning / diagnostic knob may be active in a benchmark run.  The product library reads no
    # environment at all (the knobs are compiled into the diagnostic build only); the host-side ones are refused here.
    knobs = sorted(k for k in os.environ if k.startswith("VRNET_") and k != "VRNET_BENCH_FORCE_DP")
    if knobs and not args.diagnostic:
        raise SystemExit(f"bench.py: refusing to run with diagnostic environment variables set: {', '.join(knobs)}")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    force_dp = os.environ.get("VRNET_BENCH_FORCE_DP") == "1"    # exercise the RCCL path on a 1-GPU box (torchrun, 1 rank)
    if world > 1 or (force_dp and "RANK" in os.environ):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE {world}: launch with torchrun --nproc-per-node {args.gpus} "
                         "(or run `python bench.py --gpus N` directly, which starts the ranks itself)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import asy_vrnet_amd as A
    from asy_vrnet_amd import hip
    from asy_vrnet_amd.parallel import DataParallelVRNet
    if hip.tuning_build() and not args.diagnostic:
        raise SystemExit("bench.py: the loaded library is the diagnostic build (make tuning); benchmark the product library")
    model = A.EfficientVRNet(4, 9, args.phi, img_size=args.size).to(dev).train()
    A.randomize_state_dict(model.state_dict(), seed=0)
    model.compute_dtype = args.dtype
    if args.serial:
        model.concurrent = False
    model.pair_streams = bool(args.pair)
    model.fused_mlp = not args.no_fused_mlp
    model.weight_planes = not args.no_weight_planes
    model.bn_colstats = not args.no_bn_colstats
    model.overlap_fusion = not args.no_overlap_fusion
    model.mlp_recompute = {"off": False, "all": True}.get(args.mlp_recompute, "auto")
    model.fused_fusion = not args.no_fused_fusion
    model.fused_upsample = not args.no_fused_upsample
    model.early_wgrads = args.early_wgrads
    if args.plane_gemms is not None:
        model.plane_gemms = False if args.plane_gemms == "off" else args.plane_gemms
    net = DataParallelVRNet(model, force_collective=force_dp) if (world > 1 or dist.is_initialized()) else model
    batches = make_batches(args.warmup + args.steps, args.batch, args.size, rank, dev)

    def eager_step(i):
        x, r = batches[i]
        if net is model:
            model.zero_grad(set_to_none=True)
        det, seg = net(x, r)
        loss_of(det, seg).backward()

    launch = "eager"
    step = eager_step
    if not args.no_graph:
        try:
            from asy_vrnet_amd.graph import GraphedStep
            gs = GraphedStep(net, loss_of, args.batch, args.size, dev)
            launch = "hipgraph"

            def step(i):
                gs(*batches[i])
        except Exception as e:                      # capture unsupported: fall back to eager launches
            if dist.is_initialized():
                # data parallel: a per-rank fallback would pair one rank's bucket collectives with another rank's segment
                # collectives (a hang); fail the whole job instead
                raise
            print(f"hipGraph capture failed ({type(e).__name__}: {e}); running eager", file=sys.stderr)

    def fence():
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    fence()
    el = time.perf_counter() - t0
    if dist.is_initialized():
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    ms_per_step = 1e3 * el / args.steps
    value = world * args.batch * args.steps / el

    roof = None
    if not args.no_roofline:
        model.concurrent = False      # serial launches: an event pair then brackets exactly one kernel
        # An event pair around a HOST call times GPU work only while the stream is still busy when the host records the
        # first event; on a drained stream the pair also spans the host's launch latency (round 3: the driver's line read
        # 27 % lower than rocprofv3 on the same build for exactly that reason).  So every instrumented step is issued
        # behind a device-side gate -- a spin kernel longer than the host needs to issue the whole step -- and the host
        # runs ahead of the GPU throughout: every pair then brackets kernel execution (plus the ~1-2 us dispatch gap).
        eager_step(args.warmup)                      # serial-mode warm-up (allocator pools of the serial schedule)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eager_step(args.warmup)
        host_issue_ms = 1e3 * (time.perf_counter() - t0)
        torch.cuda.synchronize()
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        torch.cuda._sleep(20_000_000)
        c1.record()
        torch.cuda.synchronize()
        cycles_per_ms = 20_000_000 / max(c0.elapsed_time(c1), 1e-3)
        gate_ms = 2.2 * host_issue_ms + 20.0         # ConvTimer's event records lengthen the issue time
        ahead = 0
        with ConvTimer(hip) as ct:
            for i in range(args.steps):
                torch.cuda.synchronize()
                torch.cuda._sleep(int(gate_ms * cycles_per_ms))
                gate_end = torch.cuda.Event()
                gate_end.record()
                eager_step(args.warmup + i)
                ahead += 0 if gate_end.query() else 1      # gate still spinning when the last launch was issued
            n, flops, ms = ct.summary("igemm")
            nw, fw, msw = ct.summary("wgrad")
            fam, ideal_ms = ct.by_family("igemm")
            _, ideal_w = ct.by_family("wgrad")
            ncf, bcf, mscf = ct.summary("cluster_fwd")
            ncb, bcb, mscb = ct.summary("cluster_bwd")
            if args.detail:
                ct.detail("igemm", 1e12, "TFLOP/s")
                ct.detail("wgrad", 1e12, "TFLOP/s")
                ct.detail("cluster_fwd", 1e9, "GB/s")
                ct.detail("cluster_bwd", 1e9, "GB/s")
        model.concurrent = True
        ach = flops / (ms * 1e-3) / 1e12
        bf16 = args.dtype == "bf16"
        # Matrix-pipe ceiling of the dominant kernel class = FLOP-weighted over the kernel families that actually ran
        # (hip.last_kernel() per launch): x6 launches are priced against bf16 peak / 6 products = 416.7 TFLOP/s of
        # fp32-equivalent work, fp32-MFMA launches against 157.3, bf16-operand launches against 2500.
        peak = flops / (ideal_ms * 1e-3) / 1e12 if ideal_ms > 0 else FP32_MFMA_PEAK_TFLOPS
        fam_rep = {FAMILY_NAME.get(k, str(k)): {"launches_per_step": v[0] // args.steps, "share_of_flops": round(v[1] / flops, 4),
                                               "achieved": round(v[1] / (v[2] * 1e-3) / 1e12, 2) if v[2] > 0 else None,
                                               "peak": FAMILY_PEAK.get(k)} for k, v in sorted(fam.items())}
        kern = ("implicit-GEMM conv forward + data-gradient (igemm_dma_kernel / igemm_kernel / igemm_bf16_kernel): per launch "
                "the library picks x6 (fp32 products as six exact bf16 x bf16 products on v_mfma_f32_32x32x16_bf16, fp32 "
                "accumulate), the fp32 MFMA (v_mfma_f32_32x32x2_f32) or, with --dtype bf16, bf16-rounded operands; "
                "`families` lists what ran")
        roof = {"bound": "mfma", "kernel": kern,
                "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s (fp32-equivalent)",
                "frac": round(ach / peak, 4),
                "peak_note": "FLOP-weighted matrix-pipe ceiling of the kernel mix that ran (x6 416.7 = bf16 2500 / 6; fp32 MFMA 157.3)",
                "x6_issue_ceiling": {"value": x6_issue_ceiling()[0], "unit": "TFLOP/s (fp32-equivalent)",
                                     "note": "measured, register-resident x6 loop with in-register operand splits (the kernels "
                                             "that take pre-split operands have no such VALU share) -- tools/micro/x6_peak.hip, "
                                             + str(x6_issue_ceiling()[1])},
                "achieved_over_fp32_mfma_peak": round(ach / FP32_MFMA_PEAK_TFLOPS, 4),
                "families": fam_rep,
                "traffic": pmc_traffic_bytes(args.phi) if (args.batch == 8 and args.size == 512 and not bf16) else None,
                "traffic_unit": "HBM bytes per launch (rocprofv3 PMC, profiles/; null unless measured on this exact kernel source)",
                "timing": {"method": "HIP events around every launch of a serialised replay; each replayed step is issued "
                                     "behind a device-side gate so that the host runs ahead and a pair brackets GPU execution only",
                           "gate_ms": round(gate_ms, 1), "host_issue_ms_per_step": round(host_issue_ms, 1),
                           "steps_issued_entirely_ahead_of_the_gpu": f"{ahead}/{args.steps}"},
                # all dense-conv work of the step (forward, data and weight gradients, fused Mlp) over the TIMED step time
                "step_level": {"conv_gflop_per_step": round((flops + fw) / args.steps / 1e9, 2),
                               "achieved": round((flops + fw) / args.steps / (ms_per_step * 1e-3) / 1e12, 2),
                               "unit": "TFLOP/s (fp32-equivalent) = conv FLOPs of one step / ms_per_step",
                               "frac_of_mix_peak": round((flops + fw) / args.steps / (ms_per_step * 1e-3) / 1e12 / peak, 4)},
                "launches_per_step": n // args.steps, "avg_launch_us": round(1e3 * ms / n, 2),
                "avg_launch_gflop": round(flops / n / 1e9, 3),
                "share_of_step": round(ms / args.steps / ms_per_step, 3),
                "wgrad": {"achieved": round(fw / (msw * 1e-3) / 1e12, 2), "launches_per_step": nw // args.steps,
                          "peak": round(fw / (ideal_w * 1e-3) / 1e12, 1) if ideal_w > 0 else None,
                          "share_of_step": round(msw / args.steps / ms_per_step, 3)},
                # the HBM-bound Context-Cluster kernels: algorithmic bytes (3 resp. 5 tensors of B*P*E*D fp32) / time
                "cluster_hbm": {"bound": "hbm", "peak": 8000.0, "unit": "GB/s",
                                "fwd_achieved": round(bcf / (mscf * 1e-3) / 1e9, 1),
                                "bwd_achieved": round(bcb / (mscb * 1e-3) / 1e9, 1),
                                "frac": round((bcf + bcb) / ((mscf + mscb) * 1e-3) / 8e12, 4),
                                "launches_per_step": (ncf + ncb) // args.steps,
                                "avg_launch_mbytes": round((bcf + bcb) / (ncf + ncb) / 1e6, 1)}}
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.phi, args.size, args.cpu_batch or args.batch, 0)

    if rank == 0:
        metric = "images/sec fwd+bwd, 512x512 img+4ch radar, bs=8/GPU"
        if (args.batch, args.size, args.dtype) != (8, 512, "f32"):       # not BASELINE.json's headline configuration
            metric = f"images/sec fwd+bwd, {args.size}x{args.size} img+4ch radar, bs={args.batch}/GPU, {args.dtype} [not the headline config]"
        line = {"metric": metric, "value": round(value, 3),
                "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": args.dtype, "data": "synthetic",
                "config": {"workload": f"EfficientVRNet(phi={args.phi}) forward+backward, {args.size}x{args.size} image + "
                                       f"4x{args.size}x{args.size} radar, bs={args.batch}/GPU, "
                                       + {"f32": "fp32, det+seg heads (BASELINE.json configs[1])",
                                          "f32-mfma": "fp32, every dense conv on the fp32 MFMA (v_mfma_f32_32x32x2_f32) only -- the x6 "
                                                      "kernels and the fused Mlp kernels switched off (comparison run, NOT the "
                                                      "headline configuration)",
                                          "bf16": "bf16-operand dense convs with fp32 accumulation, everything else fp32 "
                                                  "(BASELINE.json configs[2] family; NOT the headline fp32 metric)"}[args.dtype]
                                       + "; random weights",
                           "global_batch": world * args.batch, "image_size": args.size, "parallelism": f"dp{world}", "launch": launch},
                "roofline": roof, "cpu_baseline": cpu,
                "env": {k: v for k, v in os.environ.items() if k.startswith("VRNET_")}, "schema": 3}
        if args.diagnostic:
            line["diagnostic"] = True
            line["metric"] = "DIAGNOSTIC RUN (env knobs / tuning build allowed: not a measurement) -- " + line["metric"]
        print(json.dumps(line))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
