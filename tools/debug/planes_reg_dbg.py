"""Debug aid for igemm_planes_reg_kernel (diagnostic build + VRNET_PLANES_REG21/22): plain GEMM against fp64, error map by
(row block of 32, column block of 32) so that a wrong fragment / stage mapping shows its pattern."""
import importlib, sys, torch
sys.path.insert(0, ".")
hip = importlib.import_module("asy-vrnet_amd.hip")


def planes(w2d, J, K, sj, sk):
    buf = torch.empty((hip.conv_planes_bytes(J, K),), dtype=torch.uint8, device="cuda")
    nb = (K // 16) * 2 * ((J + 127) // 128)
    tab = torch.tensor([w2d.data_ptr(), J, K, sj, sk, 0, buf.data_ptr(), 0], dtype=torch.int64, device="cuda")
    hip.conv_planes_pack(tab, 1, nb)
    return buf


for (M, Ci, Co) in ((256, 64, 64), (256, 32, 128), (512, 96, 128), (8192, 320, 1280), (32768, 64, 128)):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(M, Ci, generator=g).cuda()
    w = (torch.randn(Co, Ci, generator=g) * 0.05).cuda()
    y = torch.zeros(M, Co, device="cuda")
    pf = planes(w, Co, Ci, Ci, 1)
    B, H, W = 1, M // 64, 64
    hip.conv2d(x, Ci, w, None, y, Co, B, H, W, Ci, H, W, Co, 1, 1, 1, 0, 1, precision=2, w_planes=pf)
    torch.cuda.synchronize()
    ref = (x.double() @ w.double().t())
    err = (y.double() - ref).abs()
    print(f"M{M} K{Ci} N{Co} kernel family {hip.last_kernel()} max err {err.max().item():.3e} (ref scale {ref.abs().max().item():.2f})")
    if err.max().item() > 1e-3:
        e = err[:128, :min(Co, 128)].reshape(4, 32, -1, 32).amax((1, 3))
        print("  error by (row block, col block) of the first tile:\n", e.cpu().numpy().round(3))
        # does y match a K-permuted or half product?
        for name, alt in (("first half of K", x[:, :Ci // 2].double() @ w[:, :Ci // 2].double().t()),
                          ("k16 steps swapped", None)):
            if alt is not None:
                print("  vs", name, (y.double() - alt).abs().max().item())
