"""Inference-side timing of the hot path on one GPU (not the BASELINE metric): eval-mode forward of EfficientVRNet +
the box decode (decode_outputs), captured in one hipGraph per batch size.

    python tools/bench_infer.py [--phi l] [--size 512] [--batches 1,8,32] [--dtype f32|bf16]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--phi", default="l")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--batches", default="1,8,32")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"])
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    import asy_vrnet_amd as A
    from asy_vrnet_amd.decode import decode_outputs
    dev = torch.device("cuda")
    model = A.EfficientVRNet(4, 9, args.phi, img_size=args.size).to(dev).eval()
    A.randomize_state_dict(model.state_dict(), seed=0)
    model.compute_dtype = args.dtype
    for bs in [int(b) for b in args.batches.split(",")]:
        x, r = A.synthetic_inputs(bs, args.size, 1, dev)

        def run():
            det, seg = model(x, r)
            return decode_outputs(det, (args.size, args.size)), seg

        with torch.no_grad():
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    run()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = run()
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.steps):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.steps
        print(f"phi={args.phi} {args.size}x{args.size} {args.dtype} eval forward + decode, bs={bs}: {ms:.3f} ms/batch, "
              f"{bs / ms * 1e3:.1f} img/s; boxes {tuple(out[0].shape)}, seg {tuple(out[1].shape)}")


if __name__ == "__main__":
    main()
