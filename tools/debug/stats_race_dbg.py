"""Debug: conv with residual + epilogue statistics, then gn_coef_from_pairs, repeated while another stream keeps the chip busy."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from asy_vrnet_amd import hip
prec = int(sys.argv[1]) if len(sys.argv) > 1 else 2
import ctypes
DUMP = os.environ.get("VRNET_HIP_LIB", "").endswith("dump.so")
if DUMP:
    dump = torch.zeros(4096 * 2 * 64 * 4, device="cuda")
    hip._lib.vrnet_debug_set_dump.argtypes = [ctypes.c_void_p]
    print("dump set rc", hip._lib.vrnet_debug_set_dump(dump.data_ptr()))
torch.manual_seed(0)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
for (B, H, W, ci, co) in [(8, 128, 128, 64, 64)]:
    x = torch.randn(B, H, W, ci, device="cuda"); w = torch.randn(co, ci, 1, 1, device="cuda") / ci ** 0.5
    b = torch.randn(co, device="cuda"); res = torch.randn(B, H, W, co, device="cuda"); ls = torch.rand(co, device="cuda")
    gam, bet = torch.randn(co, device="cuda"), torch.randn(co, device="cuda")
    bx = torch.randn(8, 64, 64, 256, device="cuda"); bw = torch.randn(256, 256, 1, 1, device="cuda") / 16; by = torch.empty(8, 64, 64, 256, device="cuda")
    torch.cuda.synchronize()
    ref = None
    bad_p = bad_a = bad_y = 0
    for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 200):
        with torch.cuda.stream(sB):
            for _ in range(3):
                hip.conv2d(bx, 256, bw, None, by, 256, 8, 64, 64, 256, 64, 64, 256, 1, 1, 1, 0, 1, mode=0, precision=prec)
        with torch.cuda.stream(sA):
            y = torch.empty(B, H, W, co, device="cuda")
            pairs, per = hip.conv_stats_buffer(B, H * W, co, "cuda")
            hip.conv2d(x, ci, w, b, y, co, B, H, W, ci, H, W, co, 1, 1, 1, 0, 1, mode=0, res=res, ldres=co, res_scale=ls,
                       stats=pairs, precision=prec)
            A_, D_, S_, ms = [torch.empty(B, co, device="cuda") for _ in range(3)] + [torch.empty(B, 2, device="cuda")]
            hip.gn_coef_from_pairs(pairs, per, gam, bet, 1e-5, B, H * W, co, A_, D_, S_, ms)
            cur = (y.clone(), pairs.clone(), A_.clone(), dump.clone() if DUMP else None)
        torch.cuda.synchronize()
        if ref is None:
            ref = cur
            continue
        if not torch.equal(cur[1], ref[1]):
            d = (cur[1] != ref[1]).nonzero()
            mb, nb = int(d[0][0]), int(d[0][1])
            if DUMP:
                a_ = ref[3].view(-1, 2, 64, 4)[mb, nb].cpu(); b_ = cur[3].view(-1, 2, 64, 4)[mb, nb].cpu()
                for l in range(64):
                    if not torch.equal(a_[l], b_[l]):
                        print(f"      lane {l}: ref (x0,d1,d2,cnt) {a_[l].tolist()}  cur {b_[l].tolist()}")
            tile = cur[0].view(-1, co)[mb * 32:(mb + 1) * 32, nb * 32:(nb + 1) * 32].double()      # y is correct: recompute per-lane parts
            lanes = torch.arange(64)
            rows = (lanes[:, None] // 8 + 8 * torch.arange(4)[None, :])                              # lane -> its 4 rows
            cols = 4 * (lanes % 8)
            vals = torch.stack([tile[rows[l], :][:, cols[l]:cols[l] + 4].reshape(-1) for l in range(64)]).cpu()   # [64][16]
            x0 = vals[:, 0]
            dd = vals - x0[:, None]
            lane_sq, lane_nx0 = (vals ** 2).sum(1), 16 * x0 ** 2
            deficit = float(ref[1][mb, nb, 1] - cur[1][mb, nb, 1])
            print(f"   tile ({mb},{nb}): deficit {deficit:.6f}; true sumsq {float((tile**2).sum()):.6f} ref {float(ref[1][mb, nb, 1]):.6f}")
            hits = []
            for mask in range(64):
                for val in range(64):
                    if val & ~mask:
                        continue
                    sel = [(l & mask) == val for l in range(64)]
                    for nm, arr in (("lane_sq", lane_sq), ("16x0^2", lane_nx0), ("d2", (dd ** 2).sum(1)), ("2x0d1", 2 * x0 * dd.sum(1)), ("x0d1", x0 * dd.sum(1))):
                        tot = float(sum(arr[l] for l in range(64) if sel[l]))
                        if abs(abs(tot) - deficit) < 2e-3 * max(1.0, deficit):
                            hits.append((nm, hex(mask), hex(val), round(tot, 4)))
            print("      subset hypotheses matching the deficit:", hits[:10])
            print("      16*x0^2 per lane:", [round(float(v), 3) for v in lane_nx0])
            print("      2*x0*d1 per lane:", [round(float(2 * x0[l] * dd[l].sum()), 3) for l in range(64)])
            print("      d2 per lane:", [round(float((dd[l] ** 2).sum()), 3) for l in range(64)])
            print("   iteration", it, "pairs entries that differ:", d[:8].tolist(), "of", cur[1].shape, "ref", ref[1][cur[1] != ref[1]][:6].tolist(), "cur", cur[1][cur[1] != ref[1]][:6].tolist())
        bad_y += not torch.equal(cur[0], ref[0]); bad_p += not torch.equal(cur[1], ref[1]); bad_a += not torch.equal(cur[2], ref[2])
    print(f"B{B} {H}x{W} {ci}->{co} precision {prec} kernel {hip.last_kernel()}: y {bad_y}, pairs {bad_p}, A {bad_a} of 199 differ")
