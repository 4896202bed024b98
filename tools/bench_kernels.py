"""Per-shape timing of the conv kernels on the GPU box: records every conv2d / conv2d_wgrad call of one eager
training step, then replays each distinct call standalone under HIP events.

    python tools/bench_kernels.py [--phi l] [--batch 8] [--size 512] > gpurun_out/kernels.txt
"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--phi", default="l")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--cold", action="store_true", help="flush L2/MALL (1 GiB fill) before every timed launch: in-net rates")
    args = ap.parse_args()
    import asy_vrnet_amd as A
    from asy_vrnet_amd import hip
    dev = torch.device("cuda")
    model = A.EfficientVRNet(4, 9, args.phi, img_size=args.size).to(dev).train()
    A.randomize_state_dict(model.state_dict(), seed=0)
    x, r = A.synthetic_inputs(args.batch, args.size, 1, dev)
    calls = []
    o_conv, o_wgrad = hip.conv2d, hip.conv2d_wgrad

    def conv2d(*a, **k):
        calls.append(("conv", a, k))
        o_conv(*a, **k)

    def wgrad(*a, **k):
        calls.append(("wgrad", a, k))
        o_wgrad(*a, **k)
    hip.conv2d, hip.conv2d_wgrad = conv2d, wgrad
    det, seg = model(x, r)
    (sum((d * d).mean() for d in det) + (seg * seg).mean()).backward()
    torch.cuda.synchronize()
    hip.conv2d, hip.conv2d_wgrad = o_conv, o_wgrad
    groups = collections.OrderedDict()
    for kind, a, k in calls:
        if kind == "conv":
            B, H, W, Ci, OH, OW, Co, kh, kw, s, p, d = a[6:18]
            key = (kind, k.get("mode", 0), B, H, W, Ci, OH, OW, Co, kh, s, d, k.get("act", 0), k.get("res") is not None,
                   k.get("ypre") is not None, k.get("aux") is not None, bool(k.get("out_nchw", 0)))
        else:
            B, H, W, Ci, OH, OW, Co, kh, kw, s, p, d = a[7:19]
            key = (kind, 2, B, H, W, Ci, OH, OW, Co, kh, s, d, 0, False, False, False, False)
        groups.setdefault(key, []).append((kind, a, k))
    rows = []
    flush = torch.zeros(1 << 28, device=dev) if args.cold else None
    for key, lst in groups.items():
        kind, a, k = lst[0]
        fn = o_conv if kind == "conv" else o_wgrad
        fn(*a, **k)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if args.cold:
            us = 0.0
            for _ in range(args.reps):
                flush.add_(1.0)
                e0.record()
                fn(*a, **k)
                e1.record()
                torch.cuda.synchronize()
                us += 1e3 * e0.elapsed_time(e1) / args.reps
        else:
            e0.record()
            for _ in range(args.reps):
                fn(*a, **k)
            e1.record()
            torch.cuda.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / args.reps
        _, mode, B, H, W, Ci, OH, OW, Co, kh, s, d = key[:12]
        gf = 2.0 * B * OH * OW * Co * Ci * kh * kh / 1e9
        rows.append((us * len(lst), len(lst), us, gf, key))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    print(f"# phi={args.phi} batch={args.batch} size={args.size}: {len(calls)} conv calls/step, "
          f"{len(rows)} distinct, sum {tot/1e3:.2f} ms/step")
    print("# total_us  n   each_us    GF    TF/s  mode(0 fwd,1 dgrad,2 wgrad) B H W Cin OH OW Cout k s d act res ypre aux nchw")
    for t, n, us, gf, key in rows:
        print(f"{t:9.1f} {n:3d} {us:9.1f} {gf:7.2f} {gf/us*1e-3*1e3:6.1f}  {key[1]} {' '.join(str(v) for v in key[2:12])} "
              f"{key[12]} {int(key[13])} {int(key[14])} {int(key[15])} {int(key[16])}")


if __name__ == "__main__":
    main()
