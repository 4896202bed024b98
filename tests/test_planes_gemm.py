"""Plane GEMMs (csrc/pgemm.hip): the 1x1 convolutions of the ClusterBlocks on operands that already are bf16 planes
(vr_coc.py:145-147, 187, 205-207 and their autograd), checked against fp64 ATen on the CPU and against the exactness
contract of the plane format (an fp32 tensor IS the sum of its three planes)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import asy_vrnet_amd.hip as h
    return h


def rnd(*shape, seed=0, kind="normal"):
    rng = np.random.default_rng([seed, len(shape)] + list(shape))
    a = rng.standard_normal(shape, dtype=np.float32) if kind == "normal" else rng.random(shape, dtype=np.float32)
    return torch.from_numpy(a)


def close(a, b, tol, what="", floor=1e-6):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(b.abs().max().item(), floor)
    err = (a - b).abs().max().item() / scale
    assert err < tol, f"{what}: rel err {err:.3e} (scale {scale:.3e})"


def split(hip, x2d, np_=3, ld=None):
    """Planes of a row-major fp32 matrix on the GPU (vrnet_planes_from_f32)."""
    R, K = x2d.shape
    ld = K if ld is None else ld
    out = hip.Planes(torch.zeros((np_, R, ld), dtype=torch.bfloat16, device="cuda")[..., :K], ld=ld, plane=R * ld)
    hip.planes_from_f32(x2d, x2d.stride(0), R, K, out)
    return out


def test_plane_split_is_exact(hip):
    """t = p0 + p1 + p2 bit for bit (np = 3), bf16 rounding (np = 1); also values near the ends of the exponent range, and a
    strided source / padded destination."""
    x = rnd(300, 200, seed=1).cuda()
    x[0, :8] = torch.tensor([0.0, -0.0, 1.0, -1.0, 3.0e38, -3.0e38, 1e-30, 1.0 + 2.0 ** -23], device="cuda")
    for src in (x, x[:, :96], x[:, 8:104]):
        p3 = split(hip, src, 3, ld=208)
        assert torch.equal(p3.float(), src), "three planes must reproduce the fp32 tensor exactly"
        assert torch.equal(p3.t[0].float(), src.to(torch.bfloat16).float()), "plane 0 is the bf16 rounding"
        p1 = split(hip, src, 1)
        assert torch.equal(p1.t[0], src.to(torch.bfloat16))
    # K % 8 == 4 with a tight destination (ld == K): a thread stores eight columns at a time, and the one at k0 = K - 4 must
    # not write four zeros into the next row (or, on the last row, past the plane) -- round-4 ADVICE
    for R, K in ((37, 12), (5, 4), (64, 68)):
        src = rnd(R, K, seed=7).cuda()
        for np_ in (3, 1):
            guard = torch.full((np_, R * K + 64), 7.0, dtype=torch.bfloat16, device="cuda")     # 64 sentinels behind every plane
            out = hip.Planes(guard[:, :R * K].view(np_, R, K), ld=K, plane=R * K + 64)
            hip.planes_from_f32(src, K, R, K, out)
            want = src if np_ == 3 else src.to(torch.bfloat16).float()
            assert torch.equal(out.float(), want), (R, K, np_)
            assert bool((guard[:, R * K:] == 7.0).all()), "wrote past the plane"
    inf = torch.tensor([[float("inf"), float("-inf"), float("nan"), 1.0] * 2], device="cuda")
    p = split(hip, inf, 3).float()
    assert torch.isinf(p[0, 0]) or torch.isnan(p[0, 0])      # non-finite values stay non-finite (Inf - Inf = NaN in the residuals)
    assert torch.isnan(p[0, 2]) and p[0, 3] == 1.0


def test_weight_table_split(hip):
    """vrnet_planes_split_f32: several matrices in one launch, the transposed + scaled form of the data-gradient packs."""
    w1, w2 = rnd(320, 96, seed=2).cuda() * 0.05, rnd(64, 320, seed=3).cuda() * 0.05
    ls = rnd(64, seed=4, kind="uniform").cuda()
    o1 = hip.Planes.empty(3, (320, 96), "cuda")
    o2 = hip.Planes.empty(3, (320, 64), "cuda")        # w2^T with ls folded in: rows = Cin (320), contraction = Cout (64)
    b1, b2 = hip.planes_split_blocks(320, 96), hip.planes_split_blocks(320, 64)
    tab = torch.tensor([w1.data_ptr(), 320, 96, 96, 1, 0, o1.t.data_ptr(), o1.ld, o1.plane, 0,
                        w2.data_ptr(), 320, 64, 1, 320, ls.data_ptr(), o2.t.data_ptr(), o2.ld, o2.plane, b1], dtype=torch.int64, device="cuda")
    hip.planes_split(tab, 2, b1 + b2, 3)
    assert torch.equal(o1.float(), w1)
    assert torch.equal(o2.float(), (w2 * ls[:, None]).t().contiguous())


GEMM_CASES = [
    # M, N, K, epilogue
    (8192, 320, 1280, "fc2"),        # stage-2 Mlp fc2: bias + layer-scale residual + statistics
    (8192, 1280, 320, "fc1"),        # stage-2 Mlp fc1: bias, ypre, GELU, plane output
    (8192, 320, 1280, "dgrad"),      # fc1 data gradient
    (8192, 1280, 320, "dgrad_aux"),  # fc2 data gradient: x gelu'(u), plane output only
    (8192, 512, 320, "plain"),       # fc1 | fc_v
    (2048, 512, 2048, "fc2"),        # stage 3
    (1000, 132, 96, "fc2"),          # ragged rows / columns, three stages exactly
    (300, 100, 32, "plain"),         # one stage
    (260, 128, 64, "acc"),           # two stages, accumulate into y
    (32768, 128, 128, "fc2"),        # stage-1 proj
]


@pytest.mark.parametrize("case", GEMM_CASES)
def test_gemm_planes_against_fp64(hip, case):
    M, N, K, ep = case
    a, w = rnd(M, K, seed=5).cuda(), (rnd(N, K, seed=6) / K ** 0.5).cuda()
    bias, ls = rnd(N, seed=7).cuda(), rnd(N, seed=8, kind="uniform").cuda()
    res, aux, y0 = rnd(M, N, seed=9).cuda(), rnd(M, N, seed=10).cuda(), rnd(M, N, seed=11).cuda()
    A, Bp = split(hip, a), split(hip, w)
    D = lambda t: t.double().cpu()
    z = D(a) @ D(w).t()
    y = torch.empty(M, N, device="cuda")
    if ep == "plain":
        hip.gemm_planes(A, Bp, M, N, K, y=y, ldy=N)
        close(y, z, 2e-5, what="y")
    elif ep == "acc":
        y.copy_(y0)
        hip.gemm_planes(A, Bp, M, N, K, bias=bias, y=y, ldy=N, accumulate=1)
        close(y, D(y0) + z + D(bias), 2e-5, what="accumulate")
    elif ep == "fc2":
        hw = 32 * (M // 64) if M % 64 == 0 else 0
        st = torch.zeros((M // 32, (N + 31) // 32, 2), dtype=torch.float64, device="cuda") if hw else None
        hip.gemm_planes(A, Bp, M, N, K, bias=bias, y=y, ldy=N, res=res, ldres=N, res_scale=ls, stats=st, stats_hw=hw)
        ref = D(res) + D(ls) * (z + D(bias))
        close(y, ref, 2e-5, what="y")
        if st is not None:
            yd = y.double()
            close(st[..., 0].sum(), yd.sum(), 1e-9, what="sum of what was stored")
            close(st[..., 1].sum(), (yd * yd).sum(), 1e-9, what="sum of squares of what was stored")
    elif ep == "fc1":
        ypre = torch.empty(M, N, device="cuda")
        hp = hip.Planes.empty(3, (M, N), "cuda")
        hip.gemm_planes(A, Bp, M, N, K, bias=bias, y=y, ldy=N, yp=hp, act=2, ypre=ypre, ldypre=N)
        close(ypre, z + D(bias), 2e-5, what="pre-activation")
        close(y, F.gelu(z + D(bias)), 2e-5, what="GELU")
        assert torch.equal(hp.float(), y), "the plane output is the stored fp32 value, exactly"
        h1 = hip.Planes.empty(1, (M, N), "cuda")
        hip.gemm_planes(A, Bp, M, N, K, bias=bias, yp=h1, act=2)
        assert torch.equal(h1.t[0], y.to(torch.bfloat16)), "np = 1 output: the value rounded to bf16"
        # bf16 pre-activation copy (the Mlp's u in bf16 mode): the fp32 value rounded once, everything else unchanged
        ub, y2 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda"), torch.empty_like(y)
        hip.gemm_planes(A, Bp, M, N, K, bias=bias, y=y2, ldy=N, act=2, ypre=ub, ldypre=N)
        assert torch.equal(ub, ypre.to(torch.bfloat16)) and torch.equal(y2, y)
    elif ep == "dgrad":
        hip.gemm_planes(A, Bp, M, N, K, y=y, ldy=N)
        close(y, z, 2e-5, what="dx")
    elif ep == "dgrad_aux":
        dup = hip.Planes.empty(3, (M, N), "cuda")
        hip.gemm_planes(A, Bp, M, N, K, yp=dup, aux=aux, ldaux=N)
        u = D(aux)
        gp = 0.5 * (1 + torch.erf(u / np.sqrt(2.0))) + u * torch.exp(-0.5 * u * u) / np.sqrt(2 * np.pi)
        close(dup.float(), z * gp, 2e-5, what="du")
        # bf16 GELU' argument: the same as the fp32 copy of those bf16 values, bit for bit
        ab = aux.to(torch.bfloat16)
        d1, d2 = hip.Planes.empty(3, (M, N), "cuda"), hip.Planes.empty(3, (M, N), "cuda")
        hip.gemm_planes(A, Bp, M, N, K, yp=d1, aux=ab, ldaux=N)
        hip.gemm_planes(A, Bp, M, N, K, yp=d2, aux=ab.float(), ldaux=N)
        assert torch.equal(d1.t, d2.t)
    assert hip.last_kernel() in (10, 11)


def test_gemm_planes_matches_the_in_kernel_split_path(hip):
    """Same products, other summation order: the plane GEMM and conv2d at precision 2 agree to fp32 rounding."""
    B, H, W, Ci, Co = 8, 32, 32, 320, 1280
    M = B * H * W
    x, w = rnd(M, Ci, seed=12).cuda(), (rnd(Co, Ci, seed=13) / Ci ** 0.5).cuda()
    y0, y1 = torch.empty(M, Co, device="cuda"), torch.empty(M, Co, device="cuda")
    hip.conv2d(x, Ci, w, None, y0, Co, B, H, W, Ci, H, W, Co, 1, 1, 1, 0, 1, precision=2)
    hip.gemm_planes(split(hip, x), split(hip, w), M, Co, Ci, y=y1, ldy=Co)
    close(y1, y0, 3e-6, what="plane GEMM vs x6 conv")


def test_gemm_planes_bf16_tensors(hip):
    """np = 1: one product of the bf16-rounded operands, fp32 accumulation."""
    M, N, K = 4096, 256, 512
    a, w = rnd(M, K, seed=14).cuda(), (rnd(N, K, seed=15) / K ** 0.5).cuda()
    y = torch.empty(M, N, device="cuda")
    hip.gemm_planes(split(hip, a, 1), split(hip, w, 1), M, N, K, y=y, ldy=N)
    ref = a.to(torch.bfloat16).double().cpu() @ w.to(torch.bfloat16).double().cpu().t()
    close(y, ref, 2e-5, what="bf16 operands")
    assert hip.last_kernel() == 11


def test_gemm_planes_non_finite_and_tiny_operands(hip):
    """As test_x6_non_finite_and_tiny_operands: an Inf stays in its row, tiny operands keep full accuracy."""
    M, N, K = 256, 128, 64
    a, w = rnd(M, K, seed=16).cuda(), rnd(N, K, seed=17).cuda()
    a[5, 7] = float("inf")
    y = torch.empty(M, N, device="cuda")
    hip.gemm_planes(split(hip, a), split(hip, w), M, N, K, y=y, ldy=N)
    bad = ~torch.isfinite(y)
    assert bad[5].all() and not bad[torch.arange(M, device="cuda") != 5].any()
    a2, w2 = (rnd(M, K, seed=18) * 2.0 ** -60).cuda(), (rnd(N, K, seed=19) * 2.0 ** -40).cuda()
    hip.gemm_planes(split(hip, a2), split(hip, w2), M, N, K, y=y, ldy=N)
    close(y, a2.double().cpu() @ w2.double().cpu().t(), 2e-5, what="2^-100 products", floor=0.0)


def test_gemm_planes_repeats_bitwise_beside_a_busy_stream(hip):
    M, N, K = 8192, 320, 1280
    A, Bp = split(hip, rnd(M, K, seed=20).cuda()), split(hip, (rnd(N, K, seed=21) / 36).cuda())
    bx, bw = rnd(8 * 64 * 64, 256, seed=22).cuda(), (rnd(256, 256, 1, 1, seed=23) / 16).cuda()
    by = torch.empty(8 * 64 * 64, 256, device="cuda")
    side, main = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    ref, bad = None, 0
    for it in range(100):
        with torch.cuda.stream(side):
            for _ in range(3):
                hip.conv2d(bx, 256, bw, None, by, 256, 8, 64, 64, 256, 64, 64, 256, 1, 1, 1, 0, 1, mode=0, precision=2)
        with torch.cuda.stream(main):
            y = torch.empty(M, N, device="cuda")
            hip.gemm_planes(A, Bp, M, N, K, y=y, ldy=N)
        torch.cuda.synchronize()
        if ref is None:
            ref = y
        else:
            bad += int(not torch.equal(y, ref))
    assert bad == 0, f"{bad} of 99 launches differ"


def test_gemm_planes_rejects_bad_arguments(hip):
    a, w = split(hip, rnd(128, 48, seed=24).cuda()), split(hip, rnd(128, 48, seed=25).cuda())
    y = torch.empty(128, 128, device="cuda")
    with pytest.raises(RuntimeError, match="K"):
        hip.gemm_planes(a, w, 128, 128, 48, y=y, ldy=128)            # K % 32
    a, w = split(hip, rnd(128, 64, seed=24).cuda()), split(hip, rnd(126, 64, seed=25).cuda())
    with pytest.raises(RuntimeError):
        hip.gemm_planes(a, w, 128, 126, 64, y=y, ldy=128)            # N % 4
    assert not hip.gemm_planes_ok(2048, 64, 256) and hip.gemm_planes_ok(8192, 320, 1280)


WGRAD_CASES = [
    # M, Cin, Cout
    (8192, 1280, 320),       # stage-2 fc2
    (8192, 320, 1280),       # stage-2 fc1
    (8192, 320, 512),        # fc1 | fc_v
    (8192, 256, 320),        # proj
    (2048, 512, 2048),       # stage 3
    (1000, 136, 104),        # ragged rows and channel counts (multiples of 8)
    (288, 96, 128),          # one split of 9 stages
    (32768, 128, 256),
]


@pytest.mark.parametrize("case", WGRAD_CASES)
def test_wgrad_planes_against_fp64(hip, case):
    M, Ci, Co = case
    x, dy = rnd(M, Ci, seed=30).cuda(), rnd(M, Co, seed=31).cuda()
    w, b = (rnd(Co, Ci, seed=32) / Ci ** 0.5).cuda(), rnd(Co, seed=33).cuda()
    ls = rnd(Co, seed=34, kind="uniform").cuda()
    X, DY = split(hip, x), split(hip, dy)
    dw, db, dl = torch.empty(Co, Ci, device="cuda"), torch.empty(Co, device="cuda"), torch.empty(Co, device="cuda")
    hip.wgrad_planes(X, DY, M, Ci, Co, dw, db, ls, w=w, bias=b, dls=dl)
    assert hip.last_kernel() == 12
    D = lambda t: t.double().cpu()
    dw_raw, db_raw = D(dy).t() @ D(x), D(dy).sum(0)
    close(dw, D(ls)[:, None] * dw_raw, 2e-5, what="dw")
    close(db, D(ls) * db_raw, 2e-5, what="db")
    close(dl, (D(w) * dw_raw).sum(1) + D(b) * db_raw, 2e-5, what="dls")
    dw2 = torch.full((Co, Ci), 7.0, device="cuda")
    hip.wgrad_planes(X, DY, M, Ci, Co, dw2, accumulate=1)            # no bias, no scale, accumulate
    close(dw2, 7.0 + dw_raw, 2e-5, what="dw, accumulate")
    # two runs are identical bit for bit (fixed split and reduction order)
    dw3 = torch.full((Co, Ci), 7.0, device="cuda")
    hip.wgrad_planes(X, DY, M, Ci, Co, dw3, accumulate=1)
    assert torch.equal(dw2, dw3)


def test_wgrad_planes_bf16_tensors(hip):
    M, Ci, Co = 8192, 320, 512
    x, dy = rnd(M, Ci, seed=35).cuda(), rnd(M, Co, seed=36).cuda()
    dw, db = torch.empty(Co, Ci, device="cuda"), torch.empty(Co, device="cuda")
    hip.wgrad_planes(split(hip, x, 1), split(hip, dy, 1), M, Ci, Co, dw, db)
    assert hip.last_kernel() == 13
    xb, yb = x.to(torch.bfloat16).double().cpu(), dy.to(torch.bfloat16).double().cpu()
    close(dw, yb.t() @ xb, 2e-5, what="dw (bf16 operands)")
    close(db, yb.sum(0), 2e-5, what="db (bf16 operands)")


def _nhwc(x):
    return x.detach().permute(0, 2, 3, 1).contiguous().cuda()


@pytest.mark.parametrize("np_", [3, 1])
@pytest.mark.parametrize("case", [(2, 4, 32, 16, 16, 2), (2, 8, 32, 32, 32, 2), (1, 4, 24, 64, 64, 2), (2, 8, 32, 16, 16, 1)])
def test_producers_write_the_planes_of_what_they_store(hip, case, np_):
    """The kernels with a plane output (GroupNorm apply forward / backward, Cluster forward / backward) must write, next to
    their fp32 result, exactly its planes (np = 3: the planes add up to the stored value bit for bit; np = 1: its bf16
    rounding) -- also planes-only where the fp32 output is optional."""
    B, E, D, H, W, fold = case
    C = E * D
    def check(planes, stored, what):
        if np_ == 3:
            assert torch.equal(planes.float(), stored), what
        else:
            assert torch.equal(planes.t[0], stored.to(torch.bfloat16)), what
    # ---- GroupNorm apply, forward and backward
    x = (rnd(B, H, W, C, seed=1) * 1.5 + 3.0).cuda()
    gam, bet = (rnd(C, seed=2) * 0.3 + 1).cuda(), (rnd(C, seed=3) * 0.5).cuda()
    nb = (C + 31) // 32
    t = x.double().view(B, H * W // 32, 32, C)
    pairs = torch.zeros(B, H * W // 32, nb, 2, dtype=torch.float64, device="cuda")
    for j in range(nb):
        blk = t[..., 32 * j:32 * (j + 1)]
        pairs[:, :, j, 0], pairs[:, :, j, 1] = blk.sum((2, 3)), (blk * blk).sum((2, 3))
    y, ms = torch.empty(B, H, W, C, device="cuda"), torch.empty(B, 2, device="cuda")
    yp = hip.Planes.empty(np_, (B, H, W, C), "cuda")
    hip.gn_apply_fwd(x, C, pairs, (H * W // 32) * nb, gam, bet, 1e-5, B, H * W, C, y, C, ms, planes=yp)
    y0 = torch.empty_like(y)
    hip.gn_apply_fwd(x, C, pairs, (H * W // 32) * nb, gam, bet, 1e-5, B, H * W, C, y0, C, ms)
    assert torch.equal(y, y0)
    check(yp, y, "gn_apply_fwd planes")
    yp2 = hip.Planes.empty(np_, (B, H, W, C), "cuda")
    hip.gn_apply_fwd(x, C, pairs, (H * W // 32) * nb, gam, bet, 1e-5, B, H * W, C, None, C, ms, planes=yp2)      # planes only
    assert torch.equal(yp2.t, yp.t)
    g, addend = rnd(B, H, W, C, seed=4).cuda(), rnd(B, H, W, C, seed=5).cuda()
    dx, dg, db = torch.empty(B, H, W, C, device="cuda"), torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    dxp = hip.Planes.empty(np_, (B, H, W, C), "cuda")
    hip.gn_apply_bwd(g, C, x, C, ms, gam, B, H * W, C, dx, C, dg, db, 0, add=addend, ldadd=C, planes=dxp)
    dx0 = torch.empty_like(dx)
    hip.gn_apply_bwd(g, C, x, C, ms, gam, B, H * W, C, dx0, C, dg, db, 0, add=addend, ldadd=C)
    assert torch.equal(dx, dx0)
    check(dxp, dx, "gn_apply_bwd planes")
    # ---- Cluster core, forward and backward
    f, v = rnd(B, H, W, C, seed=6).cuda(), rnd(B, H, W, C, seed=7).cuda()
    al, be = torch.tensor([1.3], device="cuda"), torch.tensor([-0.2], device="cuda")
    out, out0 = torch.empty(B, H, W, C, device="cuda"), torch.empty(B, H, W, C, device="cuda")
    idx = torch.empty(B, H, W, E, dtype=torch.uint8, device="cuda")
    wgt = torch.empty(B, H, W, E, device="cuda")
    op = hip.Planes.empty(np_, (B, H, W, C), "cuda")
    hip.cluster_fwd(f, v, C, al, be, out, C, idx, wgt, B, H, W, E, D, fold, planes=op)
    hip.cluster_fwd(f, v, C, al, be, out0, C, idx, wgt, B, H, W, E, D, fold)
    assert torch.equal(out, out0)
    check(op, out, "cluster_fwd planes")
    op2 = hip.Planes.empty(np_, (B, H, W, C), "cuda")
    hip.cluster_fwd(f, v, C, al, be, None, C, idx, wgt, B, H, W, E, D, fold, planes=op2)         # planes only
    assert torch.equal(op2.t, op.t)
    go = rnd(B, H, W, C, seed=8).cuda()
    dfv, dfv0 = torch.empty(B, H, W, 2 * C, device="cuda"), torch.empty(B, H, W, 2 * C, device="cuda")
    dab = torch.zeros(2, device="cuda")
    dfvp = hip.Planes.empty(np_, (B, H, W, 2 * C), "cuda")
    hip.cluster_bwd(f, v, C, al, be, idx, go, C, dfv, dfv[..., C:], 2 * C, dab[0:1], dab[1:2], 0, B, H, W, E, D, fold, planes=dfvp)
    hip.cluster_bwd(f, v, C, al, be, idx, go, C, dfv0, dfv0[..., C:], 2 * C, dab[0:1], dab[1:2], 0, B, H, W, E, D, fold)
    assert torch.equal(dfv, dfv0)
    check(dfvp, dfv, "cluster_bwd planes [df | dv]")
    dfvp2 = hip.Planes.empty(np_, (B, H, W, 2 * C), "cuda")
    hip.cluster_bwd(f, v, C, al, be, idx, go, C, None, None, 2 * C, dab[0:1], dab[1:2], 0, B, H, W, E, D, fold, planes=dfvp2)   # planes only
    assert torch.equal(dfvp2.t, dfvp.t)
    # ---- bf16 f | v and d(out) tensors (round 4, bf16 mode): the kernels widen on load, so the results are those of the
    # fp32 copies of the same values, bit for bit
    if np_ == 1:
        fvh = torch.cat([f, v], -1).to(torch.bfloat16)
        fh, vh, gh = fvh, fvh[..., C:], go.to(torch.bfloat16)
        fw, vw, gw = fvh[..., :C].float().contiguous(), vh.float().contiguous(), gh.float()
        oa, ob = torch.empty(B, H, W, C, device="cuda"), torch.empty(B, H, W, C, device="cuda")
        ia, ib = torch.empty_like(idx), torch.empty_like(idx)
        pa, pb = hip.Planes.empty(1, (B, H, W, C), "cuda"), hip.Planes.empty(1, (B, H, W, C), "cuda")
        hip.cluster_fwd(fh, vh, 2 * C, al, be, oa, C, ia, wgt, B, H, W, E, D, fold, planes=pa)
        hip.cluster_fwd(fw, vw, C, al, be, ob, C, ib, wgt, B, H, W, E, D, fold, planes=pb)
        assert torch.equal(ia, ib) and torch.equal(oa, ob) and torch.equal(pa.t, pb.t), "cluster_fwd bf16 inputs"
        da, db_ = torch.empty(B, H, W, 2 * C, device="cuda"), torch.empty(B, H, W, 2 * C, device="cuda")
        aba, abb = torch.zeros(2, device="cuda"), torch.zeros(2, device="cuda")
        qa, qb = hip.Planes.empty(1, (B, H, W, 2 * C), "cuda"), hip.Planes.empty(1, (B, H, W, 2 * C), "cuda")
        hip.cluster_bwd(fh, vh, 2 * C, al, be, ia, gh, C, da, da[..., C:], 2 * C, aba[0:1], aba[1:2], 0, B, H, W, E, D, fold, planes=qa)
        hip.cluster_bwd(fw, vw, C, al, be, ia, gw, C, db_, db_[..., C:], 2 * C, abb[0:1], abb[1:2], 0, B, H, W, E, D, fold, planes=qb)
        assert torch.equal(da, db_) and torch.equal(qa.t, qb.t) and torch.equal(aba, abb), "cluster_bwd bf16 inputs"
