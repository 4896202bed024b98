"""Kernel-level parity: every C-ABI entry point of libvrnet_hip.so against the CPU oracle
(oracle/vrnet_oracle.py) or the ATen op the reference executes at that point, on identical
seeded inputs.  fp32 tolerance 1e-4 relative to the tensor's max (north star: 1e-3)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 1e-4


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


def nhwc(x):            # NCHW cpu -> NHWC gpu
    return x.detach().permute(0, 2, 3, 1).contiguous().cuda()


def nchw(y):            # NHWC gpu -> NCHW cpu
    return y.permute(0, 3, 1, 2).contiguous().cpu()


def close(a, b, tol=TOL, what="", floor=1e-6):
    """max |a-b| <= tol * max(max|b|, floor).  `floor` = natural magnitude of the quantity when the
    exact value can be identically zero (e.g. d/df of the cluster core when every point is its own centre)."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(b.abs().max().item(), floor)
    err = (a - b).abs().max().item() / scale
    assert err < tol, f"{what}: rel err {err:.3e} (scale {scale:.3e})"


CONV_CASES = [
    # B, H, W, Cin, Cout, k, s, p, d
    (2, 12, 10, 64, 128, 1, 1, 0, 1),
    (1, 7, 9, 32, 9, 1, 1, 0, 1),
    (2, 8, 8, 80, 80, 1, 1, 0, 1),
    (2, 8, 8, 320, 48, 1, 1, 0, 1),
    (2, 20, 20, 4, 3, 3, 1, 1, 1),
    (2, 16, 16, 16, 32, 3, 2, 1, 1),
    (2, 32, 32, 5, 16, 4, 4, 0, 1),
    (2, 16, 16, 32, 32, 3, 1, 6, 6),
    (1, 16, 16, 24, 40, 3, 1, 18, 18),
    (2, 9, 9, 7, 4, 1, 1, 0, 1),
    (3, 1, 1, 32, 32, 1, 1, 0, 1),
    (2, 32, 32, 64, 96, 3, 2, 1, 1),     # stride-2 data gradient: parity-major rows on the LDS-DMA kernel
    (1, 32, 64, 128, 64, 3, 2, 1, 1),
    (2, 32, 32, 48, 96, 3, 2, 1, 1),     # phi=m widths (48, 96, 240, 384): ragged N / K tiles under the stride-2 row order
    (2, 16, 16, 96, 240, 3, 2, 1, 1),
    (2, 16, 16, 240, 384, 3, 2, 1, 1),
    (2, 32, 32, 48, 384, 1, 1, 0, 1),
    (8, 32, 32, 96, 48, 1, 1, 0, 1),
    (2, 16, 16, 960, 240, 1, 1, 0, 1),
    (2, 16, 16, 256, 4, 1, 1, 0, 1),     # narrow outputs over wide inputs (narrowconv.hip): head predictions ...
    (2, 16, 16, 256, 1, 1, 1, 0, 1),
    (1, 32, 32, 128, 9, 1, 1, 0, 1),     # ... and the seg-logit conv
    (3, 7, 5, 64, 5, 1, 1, 0, 1),        # ragged row count
    (2, 8, 8, 192, 12, 1, 1, 0, 1),      # K / 4 = 48 quads on 64 threads
    (2, 8, 8, 512, 4, 1, 1, 0, 1),       # K > 256: only the weight gradient is narrow
    (2, 64, 64, 32, 16, 1, 1, 0, 1),     # 16 outputs: forward / data gradient on the MFMA, weight gradient narrow
    (2, 64, 64, 32, 11, 1, 1, 0, 1),     # 8 threads per row
    (2, 64, 64, 16, 9, 1, 1, 0, 1),
]


def pack(hip, w):
    co, ci, kh, kw = w.shape
    if kh * kw == 1:
        return w.contiguous().cuda()
    out = torch.empty(kh * kw, co, ci, device="cuda")
    hip.pack_weight(w.contiguous().cuda(), out, co, ci, kh, kw)
    return out


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_forward_dgrad_wgrad(hip, case):
    B, H, W, Ci, Co, k, s, p, d = case
    x = rnd(B, Ci, H, W, seed=1).requires_grad_(True)
    w = (rnd(Co, Ci, k, k, seed=2) / np.sqrt(Ci * k * k)).requires_grad_(True)
    b = rnd(Co, seed=3).requires_grad_(True)
    y = F.conv2d(x, w, b, s, p, d)
    OH, OW = y.shape[2:]
    g = rnd(*y.shape, seed=4)
    y.backward(g)
    xg, wp, bg = nhwc(x), pack(hip, w.detach()), b.detach().cuda()
    yg = torch.empty(B, OH, OW, Co, device="cuda")
    hip.conv2d(xg, Ci, wp, bg, yg, Co, B, H, W, Ci, OH, OW, Co, k, k, s, p, d)
    close(nchw(yg), y, what="fwd")
    gg = nhwc(g)
    dx = torch.empty(B, H, W, Ci, device="cuda")
    hip.conv2d(gg, Co, wp, None, dx, Ci, B, H, W, Ci, OH, OW, Co, k, k, s, p, d, mode=1)
    close(nchw(dx), x.grad, what="dgrad")
    dw = torch.empty(Co, Ci, k, k, device="cuda")
    db = torch.empty(Co, device="cuda")
    hip.conv2d_wgrad(xg, Ci, gg, Co, dw, db, None, B, H, W, Ci, OH, OW, Co, k, k, s, p, d)
    close(dw, w.grad, what="wgrad")
    close(db, b.grad, what="bgrad")
    # accumulate + row scale
    rs = rnd(Co, seed=5).cuda()
    hip.conv2d_wgrad(xg, Ci, gg, Co, dw, db, rs, B, H, W, Ci, OH, OW, Co, k, k, s, p, d, accumulate=1)
    close(dw, w.grad * (1 + rs.cpu()[:, None, None, None]), what="wgrad acc")
    close(db, b.grad * (1 + rs.cpu()), what="bgrad acc")


def test_conv_epilogues(hip):
    B, H, W, Ci, Co = 2, 9, 11, 48, 96
    x, w, b = rnd(B, Ci, H, W, seed=1), rnd(Co, Ci, 1, 1, seed=2) / 7, rnd(Co, seed=3)
    res, ls = rnd(B, Co, H, W, seed=4), rnd(Co, seed=5)
    z = F.conv2d(x, w, b)
    xg, wg, bg = nhwc(x), w.cuda(), b.cuda()
    # GELU with pre-activation copy
    y, ypre = torch.empty(B, H, W, Co, device="cuda"), torch.empty(B, H, W, Co, device="cuda")
    hip.conv2d(xg, Ci, wg, bg, y, Co, B, H, W, Ci, H, W, Co, 1, 1, 1, 0, 1, act=2, ypre=ypre, ldypre=Co)
    close(nchw(ypre), z, what="ypre")
    close(nchw(y), F.gelu(z), what="gelu")
    # ReLU
    hip.conv2d(xg, Ci, wg, bg, y, Co, B, H, W, Ci, H, W, Co, 1, 1, 1, 0, 1, act=1)
    close(nchw(y), torch.relu(z), what="relu")
    # layer-scale residual into a channel slice of a wider buffer
    wide = torch.zeros(B, H, W, Co + 32, device="cuda")
    hip.conv2d(xg, Ci, wg, bg, wide[..., 16:], Co + 32, B, H, W, Ci, H, W, Co, 1, 1, 1, 0, 1,
               res=nhwc(res), ldres=Co, res_scale=ls.cuda())
    close(nchw(wide[..., 16:16 + Co]), res + ls[None, :, None, None] * z, what="residual")
    assert wide[..., :16].abs().max().item() == 0 and wide[..., 16 + Co:].abs().max().item() == 0
    # NCHW store into a channel range + accumulate
    out = torch.ones(B, Co + 5, H, W, device="cuda")
    hip.conv2d(xg, Ci, wg, bg, out, 0, B, H, W, Ci, H, W, Co, 1, 1, 1, 0, 1, out_nchw=1, out_ctot=Co + 5, out_coff=5,
               accumulate=1)
    close(out[:, 5:].cpu(), z + 1, what="nchw")
    assert (out[:, :5] == 1).all()
    # dgrad with contraction scale and GELU' epilogue
    g, aux = rnd(B, Co, H, W, seed=6), rnd(B, Ci, H, W, seed=7)
    ks = rnd(Co, seed=8)
    dx = torch.empty(B, H, W, Ci, device="cuda")
    hip.conv2d(nhwc(g), Co, wg, None, dx, Ci, B, H, W, Ci, H, W, Co, 1, 1, 1, 0, 1, mode=1, kscale=ks.cuda(),
               aux=nhwc(aux), ldaux=Ci)
    a = aux.clone().requires_grad_(True)
    F.gelu(a).sum().backward()
    ref = F.conv_transpose2d(g * ks[None, :, None, None], w) * a.grad
    close(nchw(dx), ref, what="dgrad kscale+gelu'")


def test_moments_and_affine(hip):
    B, H, W, C = 3, 13, 7, 20
    x, x2, m = rnd(B, C, H, W, seed=1) + 3.0, rnd(B, C, H, W, seed=2), rnd(B, C, H, W, seed=3)
    xg, x2g, mg = nhwc(x), nhwc(x2), nhwc(m)
    mom = hip.moments(xg, C, B, H * W, C)
    close(mom[..., 0], x.double().sum(dim=(2, 3)), 1e-6, "sum")
    close(mom[..., 1], (x.double() ** 2).sum(dim=(2, 3)), 1e-6, "sumsq")
    mom2 = hip.moments(xg, C, B, H * W, C, x2=x2g, ldx2=C, mask=mg, ldm=C)
    xm = torch.where(m > 0, x, torch.zeros_like(x)).double()
    close(mom2[..., 0], xm.sum(dim=(2, 3)), 1e-6, "masked sum")
    close(mom2[..., 1], (xm * x2.double()).sum(dim=(2, 3)), 1e-6, "masked dot")
    for C2 in (3, 7):       # scalar path
        y = rnd(B, C2, H, W, seed=4)
        mo = hip.moments(nhwc(y), C2, B, H * W, C2)
        close(mo[..., 0], y.double().sum(dim=(2, 3)), 1e-6, "sum scalar")
    A, D1, E, D2 = rnd(B, C, seed=5), rnd(B, C, seed=6), rnd(B, C, seed=7), rnd(B, C, seed=8)
    out = torch.empty(B, H, W, C, device="cuda")
    bc = lambda t: t[:, :, None, None]
    for pre in (0, 1, 2):
        hip.affine(out, C, B, H * W, C, x1=xg, ld1=C, A=A.cuda(), D1=D1.cuda(), pre=pre, masky=mg, ldm=C, x2=x2g, ld2=C,
                   E=E.cuda(), D2=D2.cuda(), bstride=C)
        v = bc(A) * x + bc(D1)
        v = torch.relu(v) if pre == 1 else (torch.where(m > 0, v, torch.zeros_like(v)) if pre == 2 else v)
        close(nchw(out), v + bc(E) * x2 + bc(D2), what=f"affine pre={pre}")
    hip.affine(out, C, B, H * W, C, x1=xg, ld1=C, A=A[0].contiguous().cuda(), bstride=0, accumulate=1)
    close(nchw(out), v + bc(E) * x2 + bc(D2) + A[0][None, :, None, None] * x, what="affine per-channel acc")
    hip.affine(out, C, B, H * W, C, D2=D2.cuda(), bstride=C)
    close(nchw(out), bc(D2).expand(B, C, H, W), what="broadcast")
    hip.affine(out, C, B, H * W, C, x1=xg, ld1=C, A=A.cuda(), S1=D1.cuda(), x2=x2g, ld2=C, E=E.cuda(), S2=D2.cuda(), bstride=C)
    close(nchw(out), bc(A) * (x - bc(D1)) + bc(E) * (x2 - bc(D2)), what="affine shifts")


def test_group_norm_chain(hip):
    B, H, W, C = 2, 16, 16, 48
    x = (rnd(B, C, H, W, seed=1) * 2 + 50).requires_grad_(True)            # |mean| >> std
    gam, bet = (rnd(C, seed=2) * 0.3 + 1).requires_grad_(True), rnd(C, seed=3).requires_grad_(True)
    y = F.group_norm(x, 1, gam, bet, 1e-5)
    g = rnd(B, C, H, W, seed=4)
    y.backward(g)
    xg, gg = nhwc(x), nhwc(g)
    A, D, S = (torch.empty(B, C, device="cuda") for _ in range(3))
    ms = torch.empty(B, 2, device="cuda")
    hip.gn_coef_fwd(hip.moments(xg, C, B, H * W, C), gam.detach().cuda(), bet.detach().cuda(), 1e-5, B, H * W, C, A, D, S, ms)
    out = torch.empty(B, H, W, C, device="cuda")
    hip.affine(out, C, B, H * W, C, x1=xg, ld1=C, A=A, D1=D, S1=S, bstride=C)
    close(nchw(out), y, what="gn fwd")
    A1, D1, S1 = (torch.empty(B, C, device="cuda") for _ in range(3))          # statistics + coefficients in two launches
    ms1 = torch.empty(B, 2, device="cuda")
    hip.gn_stats_fwd(xg, C, gam.detach().cuda(), bet.detach().cuda(), 1e-5, B, H * W, C, A1, D1, S1, ms1)
    for got, want, what in ((A1, A, "A"), (D1, D, "D"), (S1, S, "S"), (ms1, ms, "mean/rstd")):
        close(got, want, tol=1e-6, what="gn_stats_fwd " + what)
    mom2 = hip.moments(gg, C, B, H * W, C, x2=xg, ldx2=C)
    A2, E2, D2, S2 = (torch.empty(B, C, device="cuda") for _ in range(4))
    dgam, dbet = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    hip.gn_coef_bwd(mom2, ms, gam.detach().cuda(), B, H * W, C, A2, E2, D2, S2, dgam, dbet, 0)
    dx = torch.empty(B, H, W, C, device="cuda")
    hip.affine(dx, C, B, H * W, C, x1=gg, ld1=C, A=A2, x2=xg, ld2=C, E=E2, D2=D2, S2=S2, bstride=C)
    close(nchw(dx), x.grad, what="gn dx")
    close(dgam, gam.grad, what="gn dgamma")
    close(dbet, bet.grad, what="gn dbeta")


@pytest.mark.parametrize("shape", [(3, 10, 12, 24), (2, 32, 32, 48), (2, 16, 16, 96), (2, 8, 8, 240),
                                   (2, 256, 256, 4), (8, 128, 128, 64)])      # > 256 chunk partials per channel
@pytest.mark.parametrize("training", [True, False])
def test_batch_norm_relu_chain(hip, training, shape):
    B, H, W, C = shape
    z = (rnd(B, C, H, W, seed=1) * 1.5 + 40.0).requires_grad_(True)      # |mean| >> std: the cancellation-prone regime
    # (the two large maps: all pre-activations positive -- among millions of elements some lie within fp32 rounding of zero,
    #  and their ReLU mask bit then differs between any two evaluations, torch's own fp32 and fp64 included)
    shift = 8.0 if B * H * W > 10000 else 0.0
    gam, bet = (rnd(C, seed=2) * 0.3 + 1).requires_grad_(True), (rnd(C, seed=3) * 0.5 + shift).requires_grad_(True)
    rm, rv = rnd(C, seed=4) * 0.1, rnd(C, seed=5, kind="uniform") + 0.5
    rm0, rv0 = rm.clone(), rv.clone()
    y = torch.relu(F.batch_norm(z, rm, rv, gam, bet, training, 0.03, 1e-3))
    g = rnd(B, C, H, W, seed=6)
    y.backward(g)
    zg, gg = nhwc(z), nhwc(g)
    rmg, rvg, nbt = rm0.cuda(), rv0.cuda(), torch.zeros((), dtype=torch.int64, device="cuda")
    A, D, S = (torch.empty(C, device="cuda") for _ in range(3))
    ms = torch.empty(C, 2, device="cuda")
    mom = hip.moments(zg, C, B, H * W, C) if training else None
    hip.bn_coef_fwd(mom, gam.detach().cuda(), bet.detach().cuda(), 1e-3, 0.03, rmg, rvg, nbt, training, B, H * W, C, A, D, S, ms)
    out = torch.empty(B, H, W, C, device="cuda")
    hip.affine(out, C, B, H * W, C, x1=zg, ld1=C, A=A, D1=D, S1=S, pre=1)
    close(nchw(out), y, what="bn+relu fwd")
    close(rmg, rm, what="running_mean")
    close(rvg, rv, what="running_var")
    assert int(nbt.item()) == (1 if training else 0)
    mom2 = hip.moments(gg, C, B, H * W, C, x2=zg, ldx2=C, mask=out, ldm=C)
    A2, E2, D2, S2 = (torch.empty(C, device="cuda") for _ in range(4))
    dgam, dbet = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    hip.bn_coef_bwd(mom2, ms, gam.detach().cuda(), training, B, H * W, C, A2, E2, D2, S2, dgam, dbet, 0)
    dz = torch.empty(B, H, W, C, device="cuda")
    hip.affine(dz, C, B, H * W, C, x1=gg, ld1=C, A=A2, pre=2, masky=out, ldm=C, x2=zg, ld2=C, E=E2, D2=D2, S2=S2)
    close(nchw(dz), z.grad, what="bn dz")
    close(dgam, gam.grad, what="bn dgamma")
    close(dbet, bet.grad, what="bn dbeta")
    # the two-launch forms (moments kernel + one reduce-and-coefficients kernel) the network uses
    if training:
        rm2, rv2, nbt2 = rm0.cuda(), rv0.cuda(), torch.zeros((), dtype=torch.int64, device="cuda")
        A3, D3, S3 = (torch.empty(C, device="cuda") for _ in range(3))
        ms3 = torch.empty(C, 2, device="cuda")
        hip.bn_stats_fwd(zg, C, gam.detach().cuda(), bet.detach().cuda(), 1e-3, 0.03, rm2, rv2, nbt2, B, H * W, C, A3, D3, S3, ms3)
        for a_, b_, nm in ((A3, A, "A"), (D3, D, "D"), (S3, S, "S"), (ms3, ms, "mean_rstd"), (rm2, rmg, "rm"), (rv2, rvg, "rv")):
            close(a_, b_, 1e-6, what="bn_stats_fwd " + nm)
        assert int(nbt2.item()) == 1
    A4, E4, D4, S4 = (torch.empty(C, device="cuda") for _ in range(4))
    dg4, db4 = torch.full((C,), 3.0, device="cuda"), torch.full((C,), 3.0, device="cuda")
    hip.bn_stats_bwd(gg, C, zg, C, out, C, ms, gam.detach().cuda(), training, B, H * W, C, A4, E4, D4, S4, dg4, db4, 1)
    for a_, b_, nm in ((A4, A2, "A"), (E4, E2, "E"), (D4, D2, "D"), (S4, S2, "S"), (dg4, dgam + 3, "dgamma"), (db4, dbet + 3, "dbeta")):
        close(a_, b_, 1e-5, what="bn_stats_bwd " + nm, floor=1e-3)
    # round 4: the same backward WITHOUT reading the ReLU output -- the mask is recomputed from z with the forward
    # coefficients, and must be the forward's mask bit for bit: identical results, not merely close ones
    A5, E5, D5, S5 = (torch.empty(C, device="cuda") for _ in range(4))
    dg5, db5 = torch.full((C,), 3.0, device="cuda"), torch.full((C,), 3.0, device="cuda")
    hip.bn_stats_bwd_zmask(gg, C, zg, C, (A, D, S), ms, gam.detach().cuda(), training, B, H * W, C, A5, E5, D5, S5, dg5, db5, 1)
    for a_, b_, nm in ((A5, A4, "A"), (E5, E4, "E"), (D5, D4, "D"), (S5, S4, "S"), (dg5, dg4, "dgamma"), (db5, db4, "dbeta")):
        assert torch.equal(a_, b_), "bn_stats_bwd_zmask " + nm
    dz5 = torch.empty(B, H, W, C, device="cuda")
    hip.bn_apply_bwd_zmask(gg, C, zg, C, (A, D, S), A2, E2, D2, S2, dz5, C, B, H * W, C)
    assert torch.equal(dz5, dz), "bn_apply_bwd_zmask"


CLUSTER_CASES = [
    # B, E, D, H, W, fold
    (2, 4, 32, 32, 32, 2),     # N = 256 (every backbone stage at 512 px)
    (2, 4, 24, 16, 16, 2),     # N = 64, neck head_dim
    (1, 4, 24, 64, 64, 2),     # N = 1024 (neck p3 at 512 px)
    (2, 8, 32, 16, 16, 1),     # fold 1 (stage 3)
    (1, 2, 32, 10, 14, 2),     # odd 5x7 regions: overlapping pooling windows
    (2, 4, 32, 4, 4, 2),       # N = 4
    (2, 4, 24, 2, 2, 2),       # N = 1: all four centres coincide -> ties -> index 0
    (1, 4, 32, 128, 128, 8),   # fold 8 (stage 0)
    (1, 4, 24, 128, 128, 2),   # N = 4096 (neck p3 at 1024 px): streaming kernel
    (1, 2, 32, 34, 30, 2),     # N = 255 odd (register kernel, ragged last pass) / 17x15 regions
    (1, 2, 32, 36, 30, 2),     # N = 270: smallest streaming case, odd windows
]


@pytest.mark.parametrize("case", CLUSTER_CASES)
def test_cluster_core(hip, case):
    from oracle import vrnet_oracle as O
    B, E, D, H, W, fold = case
    f = rnd(B, E * D, H, W, seed=1).requires_grad_(True)
    v = rnd(B, E * D, H, W, seed=2).requires_grad_(True)
    alpha, beta = torch.tensor([1.3], requires_grad=True), torch.tensor([-0.2], requires_grad=True)
    fg, vg = nhwc(f), nhwc(v)
    out = torch.empty(B, H, W, E * D, device="cuda")
    idx = torch.empty(B, H, W, E, dtype=torch.uint8, device="cuda")
    wgt = torch.empty(B, H, W, E, device="cuda")
    hip.cluster_fwd(fg, vg, E * D, alpha.detach().cuda(), beta.detach().cuda(), out, E * D, idx, wgt, B, H, W, E, D, fold)
    if (H // fold) * (W // fold) > 256:
        with pytest.raises(RuntimeError, match="similarity map"):
            hip.cluster_fwd(fg, vg, E * D, alpha.detach().cuda(), beta.detach().cuda(), out, E * D, idx, None, B, H, W, E, D, fold)
    idx_cpu = idx.permute(0, 3, 1, 2).contiguous().cpu().long()
    rep = {}
    ref, ref_idx = O.cluster_core(f, v, alpha, beta, E, fold, forced_idx=idx_cpu, report=rep)
    # every disagreement with the oracle's own argmax must be a numerical near-tie
    assert rep["mismatch"] <= max(2, rep["points"] // 5000), rep
    assert rep["max_gap"] < 1e-5, rep
    close(nchw(out), ref, what="cluster fwd")
    g = rnd(B, E * D, H, W, seed=3)
    ref.backward(g)
    df, dv = torch.empty_like(fg), torch.empty_like(vg)
    dab = torch.zeros(2, device="cuda")
    hip.cluster_bwd(fg, vg, E * D, alpha.detach().cuda(), beta.detach().cuda(), idx, nhwc(g), E * D, df, dv, E * D,
                    dab[0:1], dab[1:2], 0, B, H, W, E, D, fold)
    close(nchw(dv), v.grad, what="cluster dv")
    close(nchw(df), f.grad, 5e-4, what="cluster df", floor=1e-2)
    close(dab[0:1], alpha.grad, 5e-4, what="dalpha", floor=1e-2)
    close(dab[1:2], beta.grad, 5e-4, what="dbeta", floor=1e-2)
    # bitwise reproducible
    out2 = torch.empty_like(out)
    hip.cluster_fwd(fg, vg, E * D, alpha.detach().cuda(), beta.detach().cuda(), out2, E * D, idx, wgt, B, H, W, E, D, fold)
    assert torch.equal(out, out2)
    # round 4: regions of more than 256 points -- the forward leaves its per-region state, the backward starts from it (two
    # passes fewer): every output bit for bit what the four-pass backward gives
    nst = hip.cluster_state_floats(B, H, W, E, fold)
    assert (nst > 0) == ((H // fold) * (W // fold) > 256)
    if nst:
        state = torch.full((nst,), float("nan"), device="cuda")
        idx_s, wgt_s, out_s = torch.empty_like(idx), torch.empty_like(wgt), torch.empty_like(out)
        hip.cluster_fwd(fg, vg, E * D, alpha.detach().cuda(), beta.detach().cuda(), out_s, E * D, idx_s, wgt_s, B, H, W, E, D, fold,
                        state=state)
        assert torch.equal(out_s, out) and torch.equal(idx_s, idx) and not bool(torch.isnan(state.view(-1, 264)[:, :260]).any())
        df_s, dv_s, dab_s = torch.empty_like(df), torch.empty_like(dv), torch.zeros(2, device="cuda")
        hip.cluster_bwd(fg, vg, E * D, alpha.detach().cuda(), beta.detach().cuda(), idx_s, nhwc(g), E * D, df_s, dv_s, E * D,
                        dab_s[0:1], dab_s[1:2], 0, B, H, W, E, D, fold, saved=(wgt_s, state))
        assert torch.equal(df_s, df) and torch.equal(dv_s, dv) and torch.equal(dab_s, dab)
        with pytest.raises(RuntimeError, match="come together"):
            hip.cluster_bwd(fg, vg, E * D, alpha.detach().cuda(), beta.detach().cuda(), idx_s, nhwc(g), E * D, df_s, dv_s, E * D,
                            dab_s[0:1], dab_s[1:2], 0, B, H, W, E, D, fold, saved=(wgt_s, None))
    # teacher-forced forward (vrnet_cluster_fwd_forced_f32): its own assignment gives the same output; a different one (every
    # point to the next centre) gives what the oracle computes for that assignment
    out3 = torch.empty_like(out)
    hip.cluster_fwd(fg, vg, E * D, alpha.detach().cuda(), beta.detach().cuda(), out3, E * D, idx, wgt, B, H, W, E, D, fold, forced=True)
    assert torch.equal(out, out3)
    idx_rot = ((idx.long() + 1) % 4).to(torch.uint8)
    keep = idx_rot.clone()
    hip.cluster_fwd(fg, vg, E * D, alpha.detach().cuda(), beta.detach().cuda(), out3, E * D, idx_rot, wgt, B, H, W, E, D, fold, forced=True)
    assert torch.equal(idx_rot, keep)                       # read, not written
    with torch.no_grad():
        ref_rot, _ = O.cluster_core(f.detach(), v.detach(), alpha.detach(), beta.detach(), E, fold,
                                    forced_idx=idx_rot.permute(0, 3, 1, 2).contiguous().cpu().long())
    close(nchw(out3), ref_rot, what="cluster fwd, forced assignment")


def test_cluster_rejects_bad_shapes(hip):
    f = torch.zeros(1, 6, 6, 32, device="cuda")
    o = torch.empty_like(f)
    idx = torch.empty(1, 6, 6, 1, dtype=torch.uint8, device="cuda")
    ab = torch.ones(1, device="cuda")
    with pytest.raises(RuntimeError, match="divided by fold"):
        hip.cluster_fwd(f, f, 32, ab, ab, o, 32, idx, None, 1, 6, 6, 1, 32, 4)


@pytest.mark.parametrize("C", [64, 256, 6, 40])      # vector kernel (64, 256), scalar kernel (6; 40: 256 % 10 != 0)
def test_depthwise(hip, C):
    B, H, W = 2, 9, 8
    x = rnd(B, C, H, W, seed=1).requires_grad_(True)
    w = rnd(C, 1, 3, 3, seed=2).requires_grad_(True)
    y = F.conv2d(x, w, None, 1, 1, 1, C)
    g = rnd(B, C, H, W, seed=3)
    y.backward(g)
    xg, gg, wg = nhwc(x), nhwc(g), w.detach().cuda()
    out = torch.empty(B, H, W, C, device="cuda")
    hip.dwconv3x3(xg, C, wg, out, C, B, H, W, C)
    close(nchw(out), y, what="dw fwd")
    hip.dwconv3x3(gg, C, wg, out, C, B, H, W, C, flip=1)
    close(nchw(out), x.grad, what="dw dgrad")
    dw = torch.empty(C, 1, 3, 3, device="cuda")
    hip.dwconv3x3_wgrad(xg, C, gg, C, dw, B, H, W, C)
    close(dw, w.grad, what="dw wgrad")


@pytest.mark.parametrize("shape", [(8, 64, 64, 256), (2, 32, 32, 256), (3, 16, 16, 256), (2, 16, 32, 64), (2, 48, 48, 32), (1, 64, 64, 512),
                                   (2, 128, 128, 256), (1, 32, 128, 64), (1, 16, 256, 512)])      # rows wider than the workgroup's runs (1 024 px)
def test_depthwise_wgrad_sliding_window(hip, shape):
    """The head's depthwise convs (decouplehead.py:23-34: 256 channels at 64 / 32 / 16 px) take the sliding-window weight-gradient
    kernel (four channels and a 16-pixel run per thread); against fp64 ATen, accumulate on and off, twice for bitwise equality."""
    B, H, W, C = shape
    x, g = rnd(B, C, H, W, seed=1), rnd(B, C, H, W, seed=3)
    w = rnd(C, 1, 3, 3, seed=2).double().requires_grad_(True)
    F.conv2d(x.double(), w, None, 1, 1, 1, C).backward(g.double())
    xg, gg = nhwc(x), nhwc(g)
    dw, dw2 = torch.empty(C, 1, 3, 3, device="cuda"), torch.full((C, 1, 3, 3), 2.0, device="cuda")
    hip.dwconv3x3_wgrad(xg, C, gg, C, dw, B, H, W, C)
    close(dw, w.grad, 2e-5, what="dw wgrad")
    hip.dwconv3x3_wgrad(xg, C, gg, C, dw2, B, H, W, C, accumulate=1)
    close(dw2, w.grad + 2.0, 2e-5, what="dw wgrad, accumulate")
    dw3 = torch.empty_like(dw)
    hip.dwconv3x3_wgrad(xg, C, gg, C, dw3, B, H, W, C)
    assert torch.equal(dw, dw3)


@pytest.mark.parametrize("scale,nchw_out", [(2, 0), (4, 0), (4, 1), (2, 1)])
@pytest.mark.parametrize("C", [9, 8, 128])      # 9: scalar kernels; 8 / 128: four channels per thread (NHWC)
def test_upsample(hip, scale, nchw_out, C):
    B, H, W = 2, 6, 5
    x = rnd(B, C, H, W, seed=1).requires_grad_(True)
    y = F.interpolate(x, scale_factor=scale, mode="bilinear", align_corners=True)
    g = rnd(*y.shape, seed=2)
    y.backward(g)
    xg = nhwc(x)
    OH, OW = H * scale, W * scale
    if nchw_out:
        out = torch.empty(B, C, OH, OW, device="cuda")
        hip.upsample(xg, C, out, 0, B, H, W, C, scale, out_nchw=1)
        close(out.cpu(), y, what="up fwd nchw")
        gg = g.cuda()
    else:
        out = torch.empty(B, OH, OW, C, device="cuda")
        hip.upsample(xg, C, out, C, B, H, W, C, scale)
        close(nchw(out), y, what="up fwd")
        gg = nhwc(g)
    dx = torch.empty(B, H, W, C, device="cuda")
    hip.upsample_bwd(gg, C, nchw_out, dx, C, B, H, W, C, scale)
    close(nchw(dx), x.grad, what="up bwd")
    dx2 = torch.full((B, H, W, C), 2.0, device="cuda")
    hip.upsample_bwd(gg, C, nchw_out, dx2, C, B, H, W, C, scale, accumulate=1)
    assert torch.equal(dx2, dx + 2.0)


def test_image_gain(hip):
    B, H, W, C = 2, 8, 8, 16
    p = torch.relu(rnd(B, C, H, W, seed=1)).requires_grad_(True)
    x = rnd(B, C, H, W, seed=2).requires_grad_(True)
    out = (1 + (p - p.min()) / (p.max() - p.min())) * x
    g = rnd(B, C, H, W, seed=3)
    out.backward(g)
    pg, xg, gg = nhwc(p), nhwc(x), nhwc(g)
    mm = torch.empty(2, device="cuda")
    hip.minmax(pg, pg.numel(), mm)
    assert mm[0].item() == p.min().item() and mm[1].item() == p.max().item()
    o = torch.empty_like(xg)
    hip.enhance_mul(pg, xg, mm, o, o.numel())
    close(nchw(o), out, what="gain fwd")
    dx, dp = torch.empty_like(xg), torch.empty_like(xg)
    hip.enhance_bwd(gg, xg, pg, mm, dx, dp, dx.numel())
    close(nchw(dx), x.grad, what="gain dx")
    close(nchw(dp), p.grad, what="gain dp")


@pytest.mark.parametrize("C,G", [(64, 8), (32, 4), (16, 8), (48, 4), (240, 4), (96, 8)])      # last three: phi=m widths
def test_shuffle_attention(hip, C, G):
    from oracle import vrnet_oracle as O
    B, H, W = 2, 7, 9
    cp = C // (2 * G)
    x = (rnd(B, C, H, W, seed=1) + 8.0).requires_grad_(True)
    names = ["cweight", "cbias", "sweight", "sbias"]
    P = {"m." + n: rnd(1, cp, 1, 1, seed=10 + i).requires_grad_(True) for i, n in enumerate(names)}
    P["m.gn.weight"] = (rnd(cp, seed=20) * 0.3 + 1).requires_grad_(True)
    P["m.gn.bias"] = rnd(cp, seed=21).requires_grad_(True)
    y = O.shuffle_attention(P, "m", x, G)
    g = rnd(B, C, H, W, seed=3)
    y.backward(g)
    xg, gg = nhwc(x), nhwc(g)
    order = ["m.cweight", "m.cbias", "m.sweight", "m.sbias", "m.gn.weight", "m.gn.bias"]
    params = [P[k].detach().reshape(-1).contiguous().cuda() for k in order]
    mom = hip.moments(xg, C, B, H * W, C)
    Pq, Qq, Mn = (torch.empty(B, C, device="cuda") for _ in range(3))
    hip.sa_coef_fwd(mom, *params, B, H * W, C, G, Pq, Qq, Mn)
    out = torch.empty(B, H, W, C, device="cuda")
    hip.sa_apply(xg, C, Pq, Qq, Mn, out, C, B, H * W, C)
    close(nchw(out), y, what="sa fwd")
    grads = [torch.zeros(cp, device="cuda") for _ in range(6)]
    dx = torch.empty_like(xg)
    EF = torch.empty(2, B, C, device="cuda")
    hip.sa_bwd(gg, C, xg, C, Pq, Qq, Mn, mom, params, dx, C, grads, EF, B, H * W, C, G, 0, 0)
    close(nchw(dx), x.grad, what="sa dx")
    for gk, k in zip(grads, order):
        close(gk, P[k].grad.reshape(-1), 2e-4, what="sa d" + k)


@pytest.mark.parametrize("C", [7, 64, 256, 96, 480])
def test_eca(hip, C):
    from oracle import vrnet_oracle as O
    B, H, W = 2, 6, 5
    k = O.eca_kernel_size(C)
    x = rnd(B, C, H, W, seed=1).requires_grad_(True)
    P = {"m.conv.weight": rnd(1, 1, k, seed=2).requires_grad_(True)}
    y = O.eca(P, "m", x)
    g = rnd(B, C, H, W, seed=3)
    y.backward(g)
    xg, gg = nhwc(x), nhwc(g)
    wk = P["m.conv.weight"].detach().reshape(-1).cuda()
    mom = hip.moments(xg, C, B, H * W, C)
    gate = torch.empty(B, C, device="cuda")
    hip.eca_coef_fwd(mom, wk, k, B, H * W, C, gate)
    out = torch.empty_like(xg)
    hip.affine(out, C, B, H * W, C, x1=xg, ld1=C, A=gate, bstride=C)
    close(nchw(out), y, what="eca fwd")
    mom2 = hip.moments(gg, C, B, H * W, C, x2=xg, ldx2=C)
    Fc, dwk = torch.empty(B, C, device="cuda"), torch.empty(k, device="cuda")
    hip.eca_coef_bwd(mom2, mom, gate, wk, k, B, H * W, C, Fc, dwk, 0)
    dx = torch.empty_like(xg)
    hip.affine(dx, C, B, H * W, C, x1=gg, ld1=C, A=gate, D2=Fc, bstride=C)
    close(nchw(dx), x.grad, what="eca dx")
    close(dwk, P["m.conv.weight"].grad.reshape(-1), what="eca dwk")


def test_layer_scale_coef_and_layout(hip):
    B, H, W, C = 2, 5, 6, 12
    dx, o, ls = rnd(B, C, H, W, seed=1), rnd(B, C, H, W, seed=2), rnd(C, seed=3)
    mom2 = hip.moments(nhwc(dx), C, B, H * W, C, x2=nhwc(o), ldx2=C)
    dls, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    hip.ls_coef_bwd(mom2, ls.cuda(), B, C, dls, db, 0)
    close(dls, (dx * o).sum(dim=(0, 2, 3)), what="dls")
    close(db, ls * dx.sum(dim=(0, 2, 3)), what="dbias")
    # cat + 2-group shuffle through strided copies, and the adjoint
    a, b = rnd(B, C, H, W, seed=4), rnd(B, C, H, W, seed=5)
    buf = torch.empty(B, H, W, 2 * C, device="cuda")
    hip.copy_channels(nhwc(a), C, 1, buf, 2 * C, 2, B * H * W, C)
    hip.copy_channels(nhwc(b), C, 1, buf[..., 1:], 2 * C, 2, B * H * W, C)
    from oracle import vrnet_oracle as O
    close(nchw(buf), O.shuffle2(torch.cat([a, b], 1)), 1e-7, "cat+shuffle")
    back = torch.ones(B, H, W, C, device="cuda")
    hip.copy_channels(buf[..., 1:], 2 * C, 2, back, C, 1, B * H * W, C, accumulate=1)
    close(nchw(back), b + 1, 1e-7, "adjoint")
    # transposes
    t = rnd(B, C, H, W, seed=6)
    dst = torch.empty(B, H, W, C + 4, device="cuda")
    hip.nchw_to_nhwc(t.cuda(), dst, C + 4, B, C, H * W)
    close(nchw(dst[..., :C]), t, 1e-7, "nchw->nhwc")
    out = torch.zeros(B, C, H, W, device="cuda")
    hip.nhwc_to_nchw(dst, C + 4, out, B, C, H * W)
    close(out.cpu(), t, 1e-7, "nhwc->nchw")
    hip.add_(out, out.clone())
    close(out.cpu(), 2 * t, 1e-7, "add")
    hip.fill_(out, 3.0)
    assert (out == 3).all()
    mf = torch.empty(B, C, device="cuda")
    hip.moments_to_float(hip.moments(nhwc(t), C, B, H * W, C), mf, B * C, 1.0 / (H * W))
    close(mf, t.mean(dim=(2, 3)), what="gap")


@pytest.mark.parametrize("C,k", [(3, 4), (4, 4), (5, 2)])
def test_patch_embed_as_gather_gemm(hip, C, k):
    """cat([x, fea_pos]) -> conv k x k / stride k (vr_coc.py:583-586, 99-102) through patch_gather + 1x1 GEMM with
    OHWI weights, and its three gradients, against torch's conv2d on the materialised concat."""
    B, H, W, CP, Co = 2, 8 * k, 6 * k, 2, 24
    x = rnd(B, C, H, W, seed=1).requires_grad_(True)
    pos = rnd(H, W, CP, seed=2)
    w = (rnd(Co, C + CP, k, k, seed=3) * 0.2).requires_grad_(True)
    bias = rnd(Co, seed=4).requires_grad_(True)
    cat = torch.cat([x, pos.permute(2, 0, 1).unsqueeze(0).expand(B, -1, -1, -1)], dim=1)
    y = F.conv2d(cat, w, bias, stride=k)
    g = rnd(*y.shape, seed=5)
    y.backward(g)
    OH, OW, KT = H // k, W // k, k * k * (C + CP)
    xg, posg, wg = nhwc(x), pos.cuda(), w.detach().cuda()
    patches = torch.empty(B, OH, OW, KT, device="cuda")
    hip.patch_gather(xg, C, posg, patches, B, H, W, C, CP, k)
    w2 = torch.empty(Co, KT, device="cuda")
    hip.weight_ohwi(wg, w2, Co, C + CP, k, k, 0)
    out = torch.empty(B, OH, OW, Co, device="cuda")
    hip.conv2d(patches, KT, w2, bias.detach().cuda(), out, Co, B, OH, OW, KT, OH, OW, Co, 1, 1, 1, 0, 1, mode=0)
    close(nchw(out), y, what="patch embed fwd")
    gg = nhwc(g)
    gw2, gb = torch.empty(Co, KT, device="cuda"), torch.empty(Co, device="cuda")
    hip.conv2d_wgrad(patches, KT, gg, Co, gw2, gb, None, B, OH, OW, KT, OH, OW, Co, 1, 1, 1, 0, 1)
    gw = torch.full((Co, C + CP, k, k), 0.5, device="cuda")
    hip.weight_ohwi(gw2, gw, Co, C + CP, k, k, 1, accumulate=1)
    close(gw.cpu() - 0.5, w.grad, what="patch embed dw")
    close(gb, bias.grad, what="patch embed db")
    dp = torch.empty(B, OH, OW, KT, device="cuda")
    hip.conv2d(gg, Co, w2, None, dp, KT, B, OH, OW, KT, OH, OW, Co, 1, 1, 1, 0, 1, mode=1)
    dx = torch.full((B, H, W, C), 0.25, device="cuda")
    hip.patch_scatter(dp, dx, C, B, H, W, C, CP, k, accumulate=1)
    close(nchw(dx) - 0.25, x.grad, what="patch embed dx")


def test_conv_output_statistics(hip):
    """`stats` output of conv2d: fp64 (sum, sum of squares) of the stored outputs per 32x32 tile, feeding GroupNorm."""
    B, H, W, Ci, Co = 2, 16, 8, 64, 96
    x, w, bias = rnd(B, Ci, H, W, seed=1), rnd(Co, Ci, 1, 1, seed=2) * 0.2, rnd(Co, seed=3)
    res, ls = rnd(B, Co, H, W, seed=4) + 30.0, rnd(Co, seed=5) * 0.3 + 1            # |mean| >> std
    y = res + ls.view(1, -1, 1, 1) * F.conv2d(x, w, bias)
    pairs, per = hip.conv_stats_buffer(B, H * W, Co, "cuda")
    assert per == (H * W // 32) * 3
    out = torch.empty(B, H, W, Co, device="cuda")
    hip.conv2d(nhwc(x), Ci, w.cuda(), bias.cuda(), out, Co, B, H, W, Ci, H, W, Co, 1, 1, 1, 0, 1, mode=0, res=nhwc(res), ldres=Co,
               res_scale=ls.cuda(), stats=pairs)
    close(nchw(out), y, what="conv + residual")
    got = pairs.view(B, -1, 2).sum(1).cpu()
    yd = y.double().reshape(B, -1)
    want = torch.stack([yd.sum(1), (yd * yd).sum(1)], 1)
    assert torch.allclose(got, want, rtol=1e-6), (got, want)
    gam, bet = rnd(Co, seed=6) * 0.3 + 1, rnd(Co, seed=7)
    A, D, S = (torch.empty(B, Co, device="cuda") for _ in range(3))
    ms = torch.empty(B, 2, device="cuda")
    hip.gn_coef_from_pairs(pairs, per, gam.cuda(), bet.cuda(), 1e-5, B, H * W, Co, A, D, S, ms)
    o2 = torch.empty(B, H, W, Co, device="cuda")
    hip.affine(o2, Co, B, H * W, Co, x1=out, ld1=Co, A=A, D1=D, S1=S, bstride=Co)
    close(nchw(o2), F.group_norm(y, 1, gam, bet, 1e-5), what="GroupNorm from conv statistics")
    assert hip.conv_stats_buffer(B, 40, Co, "cuda") == (None, 0)
    with pytest.raises(RuntimeError, match="statistics"):
        hip.conv2d(out, Co, w.cuda(), None, torch.empty(B, H, W, Ci, device="cuda"), Ci, B, H, W, Ci, H, W, Co, 1, 1, 1, 0, 1,
                   mode=1, stats=pairs)                               # data gradients have no GroupNorm consumer


@pytest.mark.parametrize("case", [(2, 16, 16, 64, 96, 1, 1, 0, 1), (2, 16, 16, 80, 128, 3, 1, 1, 1), (2, 32, 32, 64, 64, 3, 2, 1, 1),
                                  (1, 8, 8, 320, 48, 1, 1, 0, 1), (2, 16, 16, 128, 64, 3, 1, 6, 6)])
def test_conv_bf16_operands(hip, case):
    """precision = 1: operands rounded to bf16 (round-to-nearest-even) when staged, fp32 accumulate.  Reference =
    the fp32 convolution of the bf16-rounded tensors, so the comparison is tight (summation order only)."""
    B, H, W, Ci, Co, k, s, p, d = case
    x, w = rnd(B, Ci, H, W, seed=1), rnd(Co, Ci, k, k, seed=2) * (1.0 / (Ci * k * k) ** 0.5)
    bias = rnd(Co, seed=3)
    rb = lambda t: t.bfloat16().float()
    y = F.conv2d(rb(x), rb(w), bias, s, p, d)
    OH, OW = y.shape[2:]
    assert hip.bf16_conv_ok(Ci, Ci, Co, 0)
    out = torch.empty(B, OH, OW, Co, device="cuda")
    hip.conv2d(nhwc(x), Ci, pack(hip, w), bias.cuda(), out, Co, B, H, W, Ci, OH, OW, Co, k, k, s, p, d, mode=0, precision=1)
    close(nchw(out), y, tol=2e-5, what="bf16 conv fwd")
    # data gradient with a layer scale folded into the transposed pack
    g, ks = rnd(B, Co, OH, OW, seed=4), rnd(Co, seed=5) * 0.3 + 1
    wt = torch.empty(k * k, Ci, Co, device="cuda")
    hip.pack_weight_t(w.contiguous().cuda(), ks.cuda(), wt, Co, Ci, k, k)
    xr = rb(x).requires_grad_(True)
    F.conv2d(xr, rb(w * ks.view(-1, 1, 1, 1)), None, s, p, d).backward(rb(g))
    if hip.bf16_conv_ok(Co, Ci, Co, 1):
        dx = torch.empty(B, H, W, Ci, device="cuda")
        hip.conv2d(nhwc(g), Co, wt, None, dx, Ci, B, H, W, Ci, OH, OW, Co, k, k, s, p, d, mode=1, precision=1)
        close(nchw(dx), xr.grad, tol=2e-5, what="bf16 conv dgrad")
    with pytest.raises(RuntimeError, match="bf16"):
        hip.conv2d(nhwc(x), Ci, pack(hip, w), None, out, Co, B, H, W, Ci, OH, OW, Co, k, k, s, p, d, mode=0, precision=1,
                   kscale=ks.cuda())


@pytest.mark.parametrize("case", [(2, 16, 16, 64, 96, 1, 1, 0, 1), (4, 32, 32, 128, 64, 1, 1, 0, 1), (2, 16, 16, 80, 128, 3, 1, 1, 1),
                                  (2, 32, 32, 64, 64, 3, 2, 1, 1)])
def test_wgrad_bf16_operands(hip, case):
    """Weight gradient with bf16-rounded dy and x (transposing LDS reads): against autograd on the rounded tensors."""
    B, H, W, Ci, Co, k, s, p, d = case
    x, w = rnd(B, Ci, H, W, seed=1), (rnd(Co, Ci, k, k, seed=2) * 0.1).requires_grad_(True)
    bias = rnd(Co, seed=3).requires_grad_(True)
    rb = lambda t: t.bfloat16().float()
    y = F.conv2d(rb(x), w, bias, s, p, d)
    g = rnd(*y.shape, seed=4)
    OH, OW = y.shape[2:]
    y.backward(rb(g))
    dw_ref = w.grad.clone()
    w.grad = None; bias.grad = None
    F.conv2d(rb(x), w, bias, s, p, d).backward(g)            # bias gradient: fp32 sums of the unrounded dy
    assert hip.bf16_wgrad_ok(Ci, Co, Ci, Co)
    dw, db = torch.empty(Co, Ci, k, k, device="cuda"), torch.empty(Co, device="cuda")
    hip.conv2d_wgrad(nhwc(x), Ci, nhwc(g), Co, dw, db, None, B, H, W, Ci, OH, OW, Co, k, k, s, p, d, precision=1)
    close(dw, dw_ref, tol=3e-5, what="bf16 wgrad dw")
    close(db, bias.grad, tol=1e-5, what="bf16 wgrad db")
    hip.conv2d_wgrad(nhwc(x), Ci, nhwc(g), Co, dw, db, None, B, H, W, Ci, OH, OW, Co, k, k, s, p, d, accumulate=1, precision=1)
    close(dw, 2 * dw_ref, tol=3e-5, what="bf16 wgrad accumulate")


# ---- two-stream launches (image + radar chain of a backbone stage as one batch, vr_coc.py:589-600): every entry point
# ---- with a second parameter set must equal two single launches on the two halves of the batch
PAIR_CONV_CASES = [
    # B (both streams), H, W, Cin, Cout, k, s, p
    (4, 8, 8, 64, 128, 1, 1, 0),      # 128 rows per stream: LDS-DMA kernel
    (4, 16, 16, 48, 96, 1, 1, 0),
    (8, 32, 32, 64, 64, 1, 1, 0),     # 4096 rows per stream
    (16, 32, 32, 32, 48, 1, 1, 0),    # M = 16384, short K: register-staged kernel
    (4, 16, 16, 32, 64, 3, 2, 1),     # the stage reducers (3x3 / s2): parity-major data gradient
    (8, 32, 32, 64, 96, 3, 2, 1),
    (4, 16, 16, 40, 24, 1, 1, 0),     # <= 32 output channels
]


@pytest.mark.parametrize("case", PAIR_CONV_CASES)
@pytest.mark.parametrize("precision", [0, 1])
def test_two_stream_conv(hip, case, precision):
    B, H, W, Ci, Co, k, s, p = case
    if precision == 1 and not (hip.bf16_conv_ok(Ci, Ci, Co, 0) and hip.bf16_conv_ok(Co, Ci, Co, 1) and hip.bf16_wgrad_ok(Ci, Co, Ci, Co)):
        pytest.skip("shape not on the bf16 path")
    Bh = B // 2
    OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    x = nhwc(rnd(B, Ci, H, W, seed=1))
    ws = [rnd(Co, Ci, k, k, seed=2 + i) / np.sqrt(Ci * k * k) for i in range(2)]
    wp = [pack(hip, w) for w in ws]
    bs = [rnd(Co, seed=4 + i).cuda() for i in range(2)]
    ls = [rnd(Co, seed=6 + i).cuda() for i in range(2)]
    res = nhwc(rnd(B, Co, OH, OW, seed=8))
    y, ypre = torch.empty(B, OH, OW, Co, device="cuda"), torch.empty(B, OH, OW, Co, device="cuda")
    stats_ok = precision == 0 and (OH * OW) % 32 == 0 and Co > 32 and Co % 4 == 0
    st, per = hip.conv_stats_buffer(B, OH * OW, Co, x.device) if stats_ok else (None, 0)
    hip.conv2d(x, Ci, wp[0], bs[0], y, Co, B, H, W, Ci, OH, OW, Co, k, k, s, p, 1, act=2, ypre=ypre, ldypre=Co, res=res,
               ldres=Co, res_scale=ls[0], stats=st, precision=precision, pair_rows=Bh * OH * OW, w2=wp[1], bias2=bs[1],
               res_scale2=ls[1])
    for i in range(2):
        sl = slice(i * Bh, (i + 1) * Bh)
        yr, pr = torch.empty(Bh, OH, OW, Co, device="cuda"), torch.empty(Bh, OH, OW, Co, device="cuda")
        hip.conv2d(x[sl], Ci, wp[i], bs[i], yr, Co, Bh, H, W, Ci, OH, OW, Co, k, k, s, p, 1, act=2, ypre=pr, ldypre=Co,
                   res=res[sl], ldres=Co, res_scale=ls[i], precision=precision)
        close(y[sl], yr, 1e-5, what=f"pair fwd stream {i}")
        close(ypre[sl], pr, 1e-5, what=f"pair ypre stream {i}")
    if st is not None:
        tot = st.view(B, -1, 2).sum(1).cpu()
        ref = torch.stack([y.double().sum((1, 2, 3)), (y.double() ** 2).sum((1, 2, 3))], 1).cpu()
        close(tot, ref, 1e-6, what="pair stats")
    # data gradient with contraction scale
    g = nhwc(rnd(B, Co, OH, OW, seed=9))
    wt = wp
    ks = ls
    if precision == 1:
        wt = []
        for i in range(2):
            t = torch.empty(k * k, Ci, Co, device="cuda")
            hip.pack_weight_t(ws[i].cuda(), ls[i], t, Co, Ci, k, k)
            wt.append(t)
        ks = [None, None]
    dx = torch.empty(B, H, W, Ci, device="cuda")
    hip.conv2d(g, Co, wt[0], None, dx, Ci, B, H, W, Ci, OH, OW, Co, k, k, s, p, 1, mode=1, kscale=ks[0], precision=precision,
               pair_rows=Bh * H * W, w2=wt[1], kscale2=ks[1])
    for i in range(2):
        sl = slice(i * Bh, (i + 1) * Bh)
        dr = torch.empty(Bh, H, W, Ci, device="cuda")
        hip.conv2d(g[sl], Co, wt[i], None, dr, Ci, Bh, H, W, Ci, OH, OW, Co, k, k, s, p, 1, mode=1, kscale=ks[i],
                   precision=precision)
        close(dx[sl], dr, 1e-5, what=f"pair dgrad stream {i}")
    # weight / bias gradient with row scale, then accumulate
    dws = [torch.empty(Co, Ci, k, k, device="cuda") for _ in range(2)]
    dbs = [torch.empty(Co, device="cuda") for _ in range(2)]
    hip.conv2d_wgrad(x, Ci, g, Co, dws[0], dbs[0], ls[0], B, H, W, Ci, OH, OW, Co, k, k, s, p, 1, precision=precision,
                     dw2=dws[1], dbias2=dbs[1], row_scale2=ls[1])
    for i in range(2):
        sl = slice(i * Bh, (i + 1) * Bh)
        dwr, dbr = torch.empty(Co, Ci, k, k, device="cuda"), torch.empty(Co, device="cuda")
        hip.conv2d_wgrad(x[sl], Ci, g[sl], Co, dwr, dbr, ls[i], Bh, H, W, Ci, OH, OW, Co, k, k, s, p, 1, precision=precision)
        close(dws[i], dwr, 1e-5, what=f"pair wgrad stream {i}")
        close(dbs[i], dbr, 1e-5, what=f"pair bgrad stream {i}")
    keep = [d.clone() for d in dws]
    hip.conv2d_wgrad(x, Ci, g, Co, dws[0], dbs[0], ls[0], B, H, W, Ci, OH, OW, Co, k, k, s, p, 1, accumulate=1,
                     precision=precision, dw2=dws[1], dbias2=dbs[1], row_scale2=ls[1])
    for i in range(2):
        close(dws[i], 2 * keep[i], 1e-5, what=f"pair wgrad accumulate stream {i}")


def test_two_stream_conv_rejects_bad_arguments(hip):
    x = torch.zeros(4, 8, 8, 64, device="cuda")
    w = torch.zeros(64, 64, device="cuda")
    y = torch.empty(4, 8, 8, 64, device="cuda")
    with pytest.raises(RuntimeError, match="two-stream"):
        hip.conv2d(x, 64, w, None, y, 64, 4, 8, 8, 64, 8, 8, 64, 1, 1, 1, 0, 1, pair_rows=128)          # no second set
    with pytest.raises(RuntimeError, match="two-stream"):
        hip.conv2d(x, 64, w, None, y, 64, 4, 8, 8, 64, 8, 8, 64, 1, 1, 1, 0, 1, pair_rows=100, w2=w)    # not whole tiles
    with pytest.raises(RuntimeError, match="equal row counts"):
        hip.conv2d(x, 64, w, None, y, 64, 4, 8, 8, 64, 8, 8, 64, 1, 1, 1, 0, 1, pair_rows=256, w2=w)    # M = 256: nothing left


@pytest.mark.parametrize("case", [(4, 4, 32, 32, 32, 2), (2, 8, 32, 16, 16, 1), (2, 4, 24, 64, 64, 2), (4, 4, 24, 16, 16, 2)])
def test_two_stream_cluster(hip, case):
    B, E, D, H, W, fold = case
    Bh = B // 2
    f, v, g = nhwc(rnd(B, E * D, H, W, seed=1)), nhwc(rnd(B, E * D, H, W, seed=2)), nhwc(rnd(B, E * D, H, W, seed=3))
    al = [torch.tensor([1.3], device="cuda"), torch.tensor([0.7], device="cuda")]
    be = [torch.tensor([-0.2], device="cuda"), torch.tensor([0.3], device="cuda")]
    out = torch.empty(B, H, W, E * D, device="cuda")
    idx = torch.empty(B, H, W, E, dtype=torch.uint8, device="cuda")
    wgt = torch.empty(B, H, W, E, device="cuda")
    hip.cluster_fwd(f, v, E * D, al[0], be[0], out, E * D, idx, wgt, B, H, W, E, D, fold, alpha2=al[1], beta2=be[1])
    df, dv = torch.empty_like(f), torch.empty_like(v)
    dab = torch.zeros(4, device="cuda")
    hip.cluster_bwd(f, v, E * D, al[0], be[0], idx, g, E * D, df, dv, E * D, dab[0:1], dab[1:2], 0, B, H, W, E, D, fold,
                    alpha2=al[1], beta2=be[1], dalpha2=dab[2:3], dbeta2=dab[3:4])
    for i in range(2):
        sl = slice(i * Bh, (i + 1) * Bh)
        o1 = torch.empty(Bh, H, W, E * D, device="cuda")
        i1 = torch.empty(Bh, H, W, E, dtype=torch.uint8, device="cuda")
        w1 = torch.empty(Bh, H, W, E, device="cuda")
        hip.cluster_fwd(f[sl], v[sl], E * D, al[i], be[i], o1, E * D, i1, w1, Bh, H, W, E, D, fold)
        assert torch.equal(out[sl], o1) and torch.equal(idx[sl], i1)
        df1, dv1 = torch.empty_like(o1), torch.empty_like(o1)
        d1 = torch.zeros(2, device="cuda")
        hip.cluster_bwd(f[sl], v[sl], E * D, al[i], be[i], i1, g[sl], E * D, df1, dv1, E * D, d1[0:1], d1[1:2], 0, Bh, H, W,
                        E, D, fold)
        assert torch.equal(df[sl], df1) and torch.equal(dv[sl], dv1)
        close(dab[2 * i:2 * i + 2], d1, 1e-6, what=f"pair dalpha/dbeta stream {i}")


def test_two_stream_group_norm_and_layer_scale(hip):
    B, H, W, C = 4, 16, 16, 64
    Bh, HW = B // 2, H * W
    x, dy = nhwc(rnd(B, C, H, W, seed=1) * 2 + 0.5), nhwc(rnd(B, C, H, W, seed=2))
    gam = [rnd(C, seed=3 + i).cuda() for i in range(2)]
    bet = [rnd(C, seed=5 + i).cuda() for i in range(2)]

    def bufs(b):
        return [torch.empty(b, C, device="cuda") for _ in range(3)] + [torch.empty(b, 2, device="cuda")]
    A, D, S, ms = bufs(B)
    hip.gn_stats_fwd(x, C, gam[0], bet[0], 1e-5, B, HW, C, A, D, S, ms, gamma2=gam[1], beta2=bet[1])
    mom2 = hip.moments(dy, C, B, HW, C, x2=x, ldx2=C)
    A2, E2, D2, S2 = [torch.empty(B, C, device="cuda") for _ in range(4)]
    dg = [torch.empty(C, device="cuda") for _ in range(4)]
    hip.gn_coef_bwd(mom2, ms, gam[0], B, HW, C, A2, E2, D2, S2, dg[0], dg[1], 0, gamma2=gam[1], dgamma2=dg[2], dbeta2=dg[3])
    dl = [torch.empty(C, device="cuda") for _ in range(4)]
    hip.ls_coef_bwd(mom2, gam[0], B, C, dl[0], dl[1], 0, pair=1, ls2=gam[1], dls2=dl[2], dbias2=dl[3])
    for i in range(2):
        sl = slice(i * Bh, (i + 1) * Bh)
        A1, D1, S1, ms1 = bufs(Bh)
        hip.gn_stats_fwd(x[sl], C, gam[i], bet[i], 1e-5, Bh, HW, C, A1, D1, S1, ms1)
        assert torch.equal(A[sl], A1) and torch.equal(D[sl], D1) and torch.equal(S[sl], S1) and torch.equal(ms[sl], ms1)
        m1 = hip.moments(dy[sl], C, Bh, HW, C, x2=x[sl], ldx2=C)
        a, e, d, s_ = [torch.empty(Bh, C, device="cuda") for _ in range(4)]
        g1, b1 = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
        hip.gn_coef_bwd(m1, ms1, gam[i], Bh, HW, C, a, e, d, s_, g1, b1, 0)
        assert torch.equal(A2[sl], a) and torch.equal(E2[sl], e) and torch.equal(D2[sl], d)
        assert torch.equal(dg[2 * i], g1) and torch.equal(dg[2 * i + 1], b1)
        l1, lb1 = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
        hip.ls_coef_bwd(m1, gam[i], Bh, C, l1, lb1, 0)
        assert torch.equal(dl[2 * i], l1) and torch.equal(dl[2 * i + 1], lb1)


def test_wrappers_reject_host_and_strided_tensors(hip):
    """The per-op wrappers validate every tensor argument: no CPU tensor, wrong dtype or
    element-strided view reaches a kernel as a raw pointer."""
    g = torch.zeros(2, 4, 4, 8, device="cuda")
    with pytest.raises(RuntimeError, match="HIP device"):
        hip.add_(g, torch.zeros(2, 4, 4, 8))
    with pytest.raises(RuntimeError, match="unsupported dtype"):
        hip.add_(g, g.half())
    with pytest.raises(RuntimeError, match="innermost dimension"):
        hip.add_(g[..., ::2], g[..., ::2])


@pytest.mark.parametrize("pair", [False, True])
@pytest.mark.parametrize("precision", [0, 1])
def test_layer_scale_gradient_from_weight_gradient_slabs(hip, pair, precision):
    """d ls of x + ls * (W h + b) (vr_coc.py:266-271) out of the fc2 weight-gradient launch: equals autograd's sum over
    pixels of dy * branch without the branch output ever being stored."""
    B, H, W, Ci, Co = (4, 16, 16, 128, 64)
    S = 2 if pair else 1
    Bh = B // S
    h = rnd(B, Ci, H, W, seed=1)
    g = rnd(B, Co, H, W, seed=2)
    ws = [(rnd(Co, Ci, 1, 1, seed=3 + i) / np.sqrt(Ci)).requires_grad_(True) for i in range(S)]
    bs = [rnd(Co, seed=5 + i).requires_grad_(True) for i in range(S)]
    ls = [rnd(Co, seed=7 + i).requires_grad_(True) for i in range(S)]
    rb = (lambda t: t.bfloat16().float()) if precision else (lambda t: t)
    for i in range(S):
        sl = slice(i * Bh, (i + 1) * Bh)
        t = F.conv2d(rb(h[sl]), ws[i] + (rb(ws[i].detach()) - ws[i].detach()), bs[i])
        (ls[i][None, :, None, None] * t * (g[sl] + ((rb(g[sl]) - g[sl]) if precision else 0))).sum().backward()
    hg, gg = nhwc(h), nhwc(g)
    dw = [torch.empty(Co, Ci, 1, 1, device="cuda") for _ in range(S)]
    db = [torch.empty(Co, device="cuda") for _ in range(S)]
    dl = [torch.empty(Co, device="cuda") for _ in range(S)]
    wc, bc, lc = [w.detach().cuda() for w in ws], [b.detach().cuda() for b in bs], [l.detach().cuda() for l in ls]
    kw = dict(dw2=dw[1], dbias2=db[1], row_scale2=lc[1], w2=wc[1], bias2=bc[1], dls2=dl[1]) if pair else {}
    hip.conv2d_wgrad(hg, Ci, gg, Co, dw[0], db[0], lc[0], B, H, W, Ci, H, W, Co, 1, 1, 1, 0, 1, precision=precision,
                     w=wc[0], bias=bc[0], dls=dl[0], **kw)
    tol = 2e-2 if precision else TOL
    for i in range(S):
        close(dl[i], ls[i].grad, tol, what=f"dls stream {i}")
        close(dw[i], ws[i].grad, tol, what=f"dw stream {i}")
        close(db[i], bs[i].grad, tol, what=f"db stream {i}")
    with pytest.raises(RuntimeError, match="layer-scale gradient"):
        hip.conv2d_wgrad(hg, Ci, gg, Co, dw[0], None, lc[0], B, H, W, Ci, H, W, Co, 1, 1, 1, 0, 1, w=wc[0], bias=bc[0], dls=dl[0])
    # out-of-place addend of the affine kernel (the block backward never updates a gradient map in place)
    x = nhwc(rnd(B, Co, H, W, seed=9))
    out = torch.empty_like(x)
    A = rnd(B, Co, seed=10).cuda()
    hip.affine(out, Co, B, H * W, Co, x1=gg, ld1=Co, A=A, bstride=Co, add=x, ldadd=Co)
    close(out, gg * A[:, None, None, :] + x, 1e-6, what="affine add")


def test_torch_library_custom_ops(hip):
    """torch.ops.vrnet.* (asy_vrnet_amd/ops.py): schema + fake kernels (opcheck) and autograd against the ATen conv / the
    oracle's Cluster core."""
    import asy_vrnet_amd.ops  # noqa: F401  (registers the ops)
    from oracle import vrnet_oracle as O
    x = rnd(2, 12, 12, 48, seed=1).cuda().requires_grad_(True)
    w = (rnd(64, 48, 3, 3, seed=2) / 20).cuda().requires_grad_(True)
    b = rnd(64, seed=3).cuda().requires_grad_(True)
    y = torch.ops.vrnet.conv2d_nhwc(x, w, b, 1, 1, 1)
    ref = F.conv2d(x.permute(0, 3, 1, 2), w, b, 1, 1, 1).permute(0, 2, 3, 1)
    close(y, ref, what="custom op conv")
    g = rnd(*y.shape, seed=4).cuda()
    gx, gw, gb = torch.autograd.grad(y, (x, w, b), g)
    rx, rw, rb = torch.autograd.grad(ref, (x, w, b), g)
    close(gx, rx, what="custom op conv dx"); close(gw, rw, what="custom op conv dw"); close(gb, rb, what="custom op conv db")
    torch.library.opcheck(torch.ops.vrnet.conv2d_nhwc.default, (x.detach(), w.detach(), b.detach(), 1, 1, 1),
                          test_utils=("test_schema", "test_faketensor"))
    f = rnd(2, 16, 16, 128, seed=5).cuda().requires_grad_(True)
    v = rnd(2, 16, 16, 128, seed=6).cuda().requires_grad_(True)
    al, be = torch.tensor([1.3], device="cuda", requires_grad=True), torch.tensor([-0.2], device="cuda", requires_grad=True)
    out, idx = torch.ops.vrnet.cluster(f, v, al, be, 4, 2)
    fo, vo = f.detach().cpu().permute(0, 3, 1, 2).requires_grad_(True), v.detach().cpu().permute(0, 3, 1, 2).requires_grad_(True)
    ao, bo = al.detach().cpu().requires_grad_(True), be.detach().cpu().requires_grad_(True)
    ref, _ = O.cluster_core(fo, vo, ao, bo, 4, 2, forced_idx=idx.permute(0, 3, 1, 2).contiguous().cpu().long(), report={})
    close(nchw(out), ref, what="custom op cluster")
    gg = rnd(2, 128, 16, 16, seed=7)
    ref.backward(gg)
    df, dv, da, db = torch.autograd.grad(out, (f, v, al, be), nhwc(gg))
    close(nchw(dv), vo.grad, what="custom op cluster dv")
    close(nchw(df), fo.grad, 5e-4, what="custom op cluster df", floor=1e-2)
    close(da, ao.grad, 5e-4, what="custom op cluster dalpha", floor=1e-2)
    torch.library.opcheck(torch.ops.vrnet.cluster.default, (f.detach(), v.detach(), al.detach(), be.detach(), 4, 2),
                          test_utils=("test_schema", "test_faketensor"))
    with pytest.raises(Exception):
        torch.ops.vrnet.cluster(f.detach().cpu(), v.detach().cpu(), al.detach().cpu(), be.detach().cpu(), 4, 2)


# ---- precision 2 ("x6": fp32 products as six exact bf16 x bf16 products on the LDS-DMA tile kernels) against the fp32
# ---- MFMA path of the same entry point, on shapes large enough for the tile kernels to be chosen (>= 256 tiles)
X6_CASES = [
    # B, H, W, Cin, Cout, k, s, p, d
    (2, 128, 128, 64, 128, 1, 1, 0, 1),     # tile 21 (256 x 2 column tiles)
    (2, 128, 128, 64, 512, 1, 1, 0, 1),     # tile 22 forward, tile 21 data gradient (Cin = 64 columns)
    (2, 128, 128, 512, 64, 1, 1, 0, 1),     # Mlp.fc2 at stage 0
    (8, 32, 32, 320, 1280, 1, 1, 0, 1),     # stage 2 MLP
    (8, 32, 32, 1280, 320, 1, 1, 0, 1),
    (4, 64, 64, 128, 96, 1, 1, 0, 1),       # 96 columns: ragged 128-wide / 64-wide tiles
    (4, 64, 64, 72, 200, 1, 1, 0, 1),       # contraction not a multiple of 16, columns not a multiple of 64
    (2, 128, 128, 64, 64, 3, 1, 1, 1),      # radar_projection 3x3
    (2, 128, 128, 64, 128, 3, 2, 1, 1),     # reducer 3x3 / s2: parity-major data gradient
    (8, 64, 64, 128, 320, 3, 2, 1, 1),
    (8, 64, 64, 64, 64, 3, 1, 6, 6),        # dilated
    (2, 64, 64, 128, 128, 3, 1, 1, 1),      # radar_projection at stage 1, bs 2: only the weight gradient has an x6 kernel
    (2, 32, 32, 320, 320, 3, 1, 1, 1),
    (2, 16, 16, 512, 512, 3, 1, 1, 1),
    (2, 64, 64, 128, 1024, 1, 1, 0, 1),
    # round 4, split contraction: 16 x 16 maps at batch 8 (2048 rows) -- S workgroups per 128 x 64 tile share the K loop
    (8, 16, 16, 2048, 512, 1, 1, 0, 1),     # stage-3 Mlp.fc2: forward split; the data gradient has 2048 columns (unsplit)
    (8, 16, 16, 512, 2048, 1, 1, 0, 1),     # stage-3 Mlp.fc1: data gradient split
    (8, 16, 16, 512, 512, 3, 1, 1, 1),      # 3x3: the split may start inside a tap
    (8, 16, 16, 512, 512, 3, 1, 12, 12),    # dilated: most taps dead for most rows, per-tile live-tap lists
    (8, 16, 16, 320, 512, 3, 1, 1, 1),
    (8, 16, 16, 520, 264, 1, 1, 0, 1),      # ragged columns and a contraction that is no multiple of 16
]


def _conv_suite_inputs(case):
    B, H, W, Ci, Co, k, s, p, d = case
    OH, OW = (H + 2 * p - d * (k - 1) - 1) // s + 1, (W + 2 * p - d * (k - 1) - 1) // s + 1
    return dict(x=rnd(B, Ci, H, W, seed=1), w=rnd(Co, Ci, k, k, seed=2) / np.sqrt(Ci * k * k), b=rnd(Co, seed=3), ls=rnd(Co, seed=4),
                res=rnd(B, Co, OH, OW, seed=5), g=rnd(B, Co, OH, OW, seed=6), aux=rnd(B, Ci, H, W, seed=7), dx0=rnd(B, Ci, H, W, seed=8),
                OH=OH, OW=OW)


def _conv_suite(hip, case, prec):
    """Every way the network launches a dense conv, at one precision: forward with bias + pre-activation copy + GELU +
    layer-scale residual + output statistics; data gradient with contraction scale, GELU' epilogue and accumulation; weight
    / bias gradient with row scale and (1x1) the layer-scale gradient; weight gradient without bias, accumulating.
    Returns (y, ypre, per-sample (sum, sumsq), dx, dw, db, dls, kernel families, dw_nb)."""
    B, H, W, Ci, Co, k, s, p, d = case
    t = _conv_suite_inputs(case)
    OH, OW = t["OH"], t["OW"]
    x, wp, b, ls, res = nhwc(t["x"]), pack(hip, t["w"]), t["b"].cuda(), t["ls"].cuda(), nhwc(t["res"])
    y, ypre = torch.empty(B, OH, OW, Co, device="cuda"), torch.empty(B, OH, OW, Co, device="cuda")
    st, per = hip.conv_stats_buffer(B, OH * OW, Co, x.device)
    hip.conv2d(x, Ci, wp, b, y, Co, B, H, W, Ci, OH, OW, Co, k, k, s, p, d, act=2, ypre=ypre, ldypre=Co, res=res, ldres=Co,
               res_scale=ls, stats=st, precision=prec)
    fam_f = hip.last_kernel()
    g, aux, dx = nhwc(t["g"]), nhwc(t["aux"]), nhwc(t["dx0"])     # dx: accumulate into existing contents
    hip.conv2d(g, Co, wp, None, dx, Ci, B, H, W, Ci, OH, OW, Co, k, k, s, p, d, mode=1, kscale=ls, aux=aux, ldaux=Ci,
               accumulate=1, precision=prec)
    fam_d = hip.last_kernel()
    dw, db, dl = torch.empty(Co, Ci, k, k, device="cuda"), torch.empty(Co, device="cuda"), torch.empty(Co, device="cuda")
    kw = dict(w=t["w"].cuda().contiguous(), bias=b, dls=dl) if k == 1 else {}
    hip.conv2d_wgrad(x, Ci, g, Co, dw, db, ls, B, H, W, Ci, OH, OW, Co, k, k, s, p, d, precision=prec, **kw)
    fam_w = hip.last_kernel()
    dw_nb = torch.full((Co, Ci, k, k), 7.0, device="cuda")     # without a bias gradient (BaseConv), accumulating
    hip.conv2d_wgrad(x, Ci, g, Co, dw_nb, None, None, B, H, W, Ci, OH, OW, Co, k, k, s, p, d, accumulate=1, precision=prec)
    return (y, ypre, st.view(B, -1, 2).sum(1) if st is not None else None, dx, dw, db, dl if k == 1 else None,
            (fam_f, fam_d, fam_w), dw_nb)


def test_split_contraction_plan(hip):
    """Which shapes the library splits: small maps with a long contraction; never a shape that fills the chip unsplit."""
    assert hip.conv2d_dma_plan(2048, 512, 2048) == (21, 4)
    assert hip.conv2d_dma_plan(2048, 512, 512 * 9)[1] >= 2
    assert hip.conv2d_dma_plan(2048, 2048, 512) == (21, 1)          # 512 tiles already
    assert hip.conv2d_dma_plan(8192, 320, 1280)[1] == 1
    assert hip.conv2d_dma_plan(2048, 512, 512) == (0, 1)            # 32 K16 steps: a split of fewer does not pay for its finishing launch
    assert hip.conv2d_dma_plan(2048, 256, 64) == (0, 1)             # too short to split
    assert hip.conv2d_dma_plan(2048, 32, 4096) == (0, 1)


@pytest.mark.parametrize("case", X6_CASES)
def test_x6_conv_matches_fp32_mfma_path(hip, case):
    outs = {prec: _conv_suite(hip, case, prec) for prec in (0, 2)}
    assert 6 in outs[2][7], outs[2][7]                            # at least one launch ran on an x6 kernel
    assert 6 not in outs[0][7]
    for name, a, b_ in zip(("y", "ypre", "stats", "dx", "dw", "db", "dls"), outs[2], outs[0]):
        if a is not None:
            close(a, b_, 2e-5, what=f"x6 {name} {outs[2][7]}")
    close(outs[2][8], outs[0][8], 2e-5, what="x6 dw without bias, accumulate")


@pytest.mark.parametrize("case", X6_CASES)
def test_x6_conv_against_fp64_aten(hip, case):
    """The dominant kernels (precision 2: fp32 products as six bf16 x bf16 products) DIRECTLY against fp64 ATen on the CPU --
    forward, data gradient, weight / bias / layer-scale gradient with every epilogue the network uses, on shapes large
    enough for the x6 tile kernels to be dispatched (incl. ragged tiles, a contraction that is no multiple of 16, 3x3, stride
    2 with parity-major rows, dilation).  2e-5 of each tensor's largest magnitude."""
    B, H, W, Ci, Co, k, s, p, d = case
    t = _conv_suite_inputs(case)
    out = _conv_suite(hip, case, 2)
    assert 6 in out[7], out[7]
    D = lambda a: a.double()
    x, w, b = D(t["x"]).requires_grad_(True), D(t["w"]).requires_grad_(True), D(t["b"]).requires_grad_(True)
    ls, g = D(t["ls"])[None, :, None, None], D(t["g"])
    z = F.conv2d(x, w, b, s, p, d)
    y = D(t["res"]) + ls * F.gelu(z)
    (z * (g * ls)).sum().backward()          # x.grad = conv^T(g * ls), w.grad = ls * sum g x, b.grad = ls * sum g
    a = D(t["aux"])
    gp = 0.5 * (1 + torch.erf(a / np.sqrt(2.0))) + a * torch.exp(-0.5 * a * a) / np.sqrt(2 * np.pi)
    dx = D(t["dx0"]) + x.grad * gp
    close(nchw(out[1]), z, 2e-5, what=f"ypre {out[7]}")
    close(nchw(out[0]), y, 2e-5, what="y")
    if out[2] is not None:
        yd = y.detach()
        close(out[2], torch.stack([yd.sum((1, 2, 3)), (yd * yd).sum((1, 2, 3))], 1), 2e-5, what="statistics")
    close(nchw(out[3]), dx, 2e-5, what="dx")
    close(out[4], w.grad, 2e-5, what="dw")
    close(out[5], b.grad, 2e-5, what="db")
    if out[6] is not None:
        dw_raw, db_raw = w.grad / ls.view(-1, 1, 1, 1), b.grad / ls.flatten()
        close(out[6], (w.detach() * dw_raw).sum((1, 2, 3)) + b.detach() * db_raw, 2e-5, what="dls")
    close(out[8], 7.0 + w.grad / ls.view(-1, 1, 1, 1), 2e-5, what="dw without bias, accumulate")


@pytest.mark.parametrize("precision", [2, 0])
def test_conv_epilogue_statistics_repeat_bitwise_beside_a_busy_stream(hip, precision):
    """The conv epilogue's (sum, sum of squares) pairs -- the GroupNorm statistics of the consumer -- must not depend on
    what else shares the CU.  (A build whose fp32 statistics code had been SLP-vectorised into packed-fp32 sequences
    produced a short sum of squares in ~3 % of launches next to a second stream: tools/debug/stats_race_dbg.py.)"""
    B, H, W, ci, co = 8, 128, 128, 64, 64
    x, w = rnd(B, H, W, ci, seed=1).cuda(), (rnd(co, ci, 1, 1, seed=2) / ci ** 0.5).cuda()
    b, res, ls = rnd(co, seed=3).cuda(), rnd(B, H, W, co, seed=4).cuda(), rnd(co, seed=5, kind="uniform").cuda()
    bx, bw = rnd(8, 64, 64, 256, seed=6).cuda(), (rnd(256, 256, 1, 1, seed=7) / 16).cuda()
    by = torch.empty(8, 64, 64, 256, device="cuda")
    side, main = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    ref, bad = None, 0
    for it in range(400):
        with torch.cuda.stream(side):
            for _ in range(3):
                hip.conv2d(bx, 256, bw, None, by, 256, 8, 64, 64, 256, 64, 64, 256, 1, 1, 1, 0, 1, mode=0, precision=precision)
        with torch.cuda.stream(main):
            y = torch.empty(B, H, W, co, device="cuda")
            pairs, per = hip.conv_stats_buffer(B, H * W, co, "cuda")
            hip.conv2d(x, ci, w, b, y, co, B, H, W, ci, H, W, co, 1, 1, 1, 0, 1, mode=0, res=res, ldres=co, res_scale=ls,
                       stats=pairs, precision=precision)
            cur = (y.clone(), pairs.clone())
        torch.cuda.synchronize()
        if ref is None:
            ref = cur
            tot = y.double().view(-1, co)          # the pairs are the tile sums of what was stored
            close(pairs[..., 0].sum(), tot.sum(), 1e-9, what="sum")
            close(pairs[..., 1].sum(), (tot * tot).sum(), 1e-9, what="sum of squares")
        else:
            bad += int(not (torch.equal(cur[0], ref[0]) and torch.equal(cur[1], ref[1])))
    assert bad == 0, f"{bad} of 399 launches differ"


# ---- fused Mlp (fc1 -> GELU -> fc2 in one kernel per direction, vr_coc.py:195-223) against fp64 ATen ------------------
def _bf16r(t):
    return t.to(torch.bfloat16).to(t.dtype)


MLP_CASES = [
    # B, H, W, C, hid
    (2, 32, 32, 64, 512),          # stage 0 of phi = l
    (2, 16, 16, 128, 1024),        # stage 1
    (1, 8, 12, 64, 128),           # 96 rows: the last wave of the only workgroup is idle
    (3, 8, 8, 128, 96),            # 3 chunks: odd chunk count
    (1, 4, 8, 64, 32),             # a single chunk, a single live wave
]


@pytest.mark.parametrize("precision", [2, 1])
@pytest.mark.parametrize("case", MLP_CASES)
def test_fused_mlp_against_fp64(hip, case, precision):
    B, H, W, C, hid = case
    M = B * H * W
    assert hip.mlp_fused_ok(C, hid, M)
    x, res = rnd(M, C, seed=1), rnd(M, C, seed=2)
    w1, b1 = rnd(hid, C, seed=3) / np.sqrt(C), rnd(hid, seed=4)
    w2, b2 = rnd(C, hid, seed=5) / np.sqrt(hid), rnd(C, seed=6)
    ls, dy = rnd(C, seed=7), rnd(M, C, seed=8)
    d = lambda t: t.double()
    rd = _bf16r if precision == 1 else (lambda t: t)          # precision 1: both operands of both GEMMs rounded to bf16
    # ---- reference (fp64 ATen on the CPU)
    u_ref = d(rd(x)) @ d(rd(w1)).T + d(b1)
    h_ref = F.gelu(u_ref)
    y_ref = d(res) + d(ls) * (d(rd(h_ref.float())) @ d(rd(w2)).T + d(b2))
    # ---- HIP
    xg, resg = x.cuda(), res.cuda()
    fwd, bwd = hip.mlp_pack(w1.cuda(), w2.cuda(), C, hid, precision)
    y, u = torch.empty(M, C, device="cuda"), torch.empty(M, hid, device="cuda")
    pairs, per = hip.conv_stats_buffer(B, H * W, C, "cuda")
    assert pairs is not None
    hip.mlp_fwd(xg, C, fwd, b1.cuda(), b2.cuda(), resg, C, ls.cuda(), y, C, u, hid, pairs, M, C, hid, precision)
    assert hip.last_kernel() == (7 if precision == 2 else 8)
    tol = 2e-5 if precision == 2 else 2e-3       # (precision 1: one bf16 rounding of h / du that falls the other way = 2^-8 of that element)
    close(u, u_ref, 2e-5, what="u")
    close(y, y_ref, tol, what="y")
    yd = y.double().view(B, H * W // 32, 32, C // 32, 32)
    close(pairs[..., 0].view(B, H * W // 32, C // 32), yd.sum((2, 4)), 1e-9, what="tile sums")
    close(pairs[..., 1].view(B, H * W // 32, C // 32), (yd * yd).sum((2, 4)), 1e-9, what="tile sums of squares")
    # without the pre-activation output and without statistics (eval / no_grad)
    y2 = torch.empty(M, C, device="cuda")
    hip.mlp_fwd(xg, C, fwd, b1.cuda(), b2.cuda(), resg, C, ls.cuda(), y2, C, None, 0, None, M, C, hid, precision)
    assert torch.equal(y2, y)
    # ---- backward: given u (as stored by the forward kernel) and dy
    uu = u.double().cpu()
    cdf = 0.5 * (1 + torch.erf(uu / np.sqrt(2.0)))
    gp = cdf + uu * torch.exp(-0.5 * uu * uu) / np.sqrt(2 * np.pi)
    dh = d(rd((dy * ls))) @ d(rd(w2))
    du_ref = dh * gp
    dx_ref = d(rd(du_ref.float())) @ d(rd(w1))
    hb, du, dx = torch.empty(M, hid, device="cuda"), torch.empty(M, hid, device="cuda"), torch.empty(M, C, device="cuda")
    hip.mlp_bwd(dy.cuda(), C, ls.cuda(), bwd, u, hid, hb, hid, du, hid, dx, C, M, C, hid, precision)
    close(hb, uu * cdf, 2e-6, what="recomputed h")
    close(du, du_ref, 2e-5, what="du")
    close(dx, dx_ref, tol, what="dx")
    # ---- the backward that RECOMPUTES u from x (round 5: vrnet_mlp_bwd_rc_f32; no stored hidden tensor): at precision 2 its u is
    # the forward's accumulator again -- same fragments, same MFMA order -- so h, du, dx are the stored-u kernel's bits
    if precision == 2 and hip.mlp_rc_ok(C, hid):
        rc = hip.mlp_pack_rc(w1.cuda(), w2.cuda(), C, hid, 2)
        hb2, du2, dx2 = torch.empty_like(hb), torch.empty_like(du), torch.empty_like(dx)
        hip.mlp_bwd_rc(dy.cuda(), C, ls.cuda(), rc, xg, C, b1.cuda(), hb2, hid, du2, hid, dx2, C, M, C, hid, 2)
        assert hip.last_kernel() == 7
        assert torch.equal(hb2, hb) and torch.equal(du2, du) and torch.equal(dx2, dx)
    if precision == 1 and hip.mlp_rc_ok(C, hid):
        # bf16-rounded operands, bf16 h / du (precision 4), u recomputed in fp32: against the fp64 reference from the exact u
        rc = hip.mlp_pack_rc(w1.cuda(), w2.cuda(), C, hid, 4)
        hb4 = torch.empty(M, hid, dtype=torch.bfloat16, device="cuda")
        du4, dx4 = torch.empty_like(hb4), torch.empty_like(dx)
        hip.mlp_bwd_rc(dy.cuda(), C, ls.cuda(), rc, xg, C, b1.cuda(), hb4, hid, du4, hid, dx4, C, M, C, hid, 4)
        ue = u_ref
        cdf_e = 0.5 * (1 + torch.erf(ue / np.sqrt(2.0)))
        gp_e = cdf_e + ue * torch.exp(-0.5 * ue * ue) / np.sqrt(2 * np.pi)
        close(hb4.float(), ue * cdf_e, 6e-3, what="rc h (bf16)")
        close(du4.float(), dh * gp_e, 6e-3, what="rc du (bf16)")
        close(dx4, d(rd((dh * gp_e).float())) @ d(rd(w1)), 4e-3, what="rc dx")


@pytest.mark.parametrize("case", MLP_CASES[:3])
def test_fused_mlp_bf16_hidden_tensors(hip, case):
    """precision 4 (compute_dtype "bf16" with bf16 tensors): the kernels of precision 1 with the hidden-sized tensors -- u forward,
    h and du backward -- stored as bf16.  Forward output identical to precision 1 (u is rounded only on its way to memory);
    the stored tensors are the bf16 roundings of what precision 1 stores / computes from the rounded u."""
    B, H, W, C, hid = case
    M = B * H * W
    x, res = rnd(M, C, seed=1).cuda(), rnd(M, C, seed=2).cuda()
    w1, b1 = (rnd(hid, C, seed=3) / np.sqrt(C)).cuda(), rnd(hid, seed=4).cuda()
    w2, b2 = (rnd(C, hid, seed=5) / np.sqrt(hid)).cuda(), rnd(C, seed=6).cuda()
    ls, dy = rnd(C, seed=7).cuda(), rnd(M, C, seed=8).cuda()
    fwd, bwd = hip.mlp_pack(w1, w2, C, hid, 1)
    y1, u1 = torch.empty(M, C, device="cuda"), torch.empty(M, hid, device="cuda")
    hip.mlp_fwd(x, C, fwd, b1, b2, res, C, ls, y1, C, u1, hid, None, M, C, hid, 1)
    y4, u4 = torch.empty(M, C, device="cuda"), torch.empty(M, hid, dtype=torch.bfloat16, device="cuda")
    hip.mlp_fwd(x, C, fwd, b1, b2, res, C, ls, y4, C, u4, hid, None, M, C, hid, 4)
    assert torch.equal(y4, y1) and torch.equal(u4, u1.to(torch.bfloat16))
    # backward from the bf16 pre-activation: precision 1 fed the SAME (rounded) u must give the same h, du (before rounding), dx
    ub = u4.float()
    hb1, du1, dx1 = (torch.empty(M, hid, device="cuda") for _ in range(2)) , None, torch.empty(M, C, device="cuda")
    hb1, du1 = hb1
    hip.mlp_bwd(dy, C, ls, bwd, ub, hid, hb1, hid, du1, hid, dx1, C, M, C, hid, 1)
    hb4, du4 = (torch.empty(M, hid, dtype=torch.bfloat16, device="cuda") for _ in range(2))
    dx4 = torch.empty(M, C, device="cuda")
    hip.mlp_bwd(dy, C, ls, bwd, u4, hid, hb4, hid, du4, hid, dx4, C, M, C, hid, 4)
    assert torch.equal(dx4, dx1)
    assert torch.equal(hb4, hb1.to(torch.bfloat16)) and torch.equal(du4, du1.to(torch.bfloat16))


def test_fused_mlp_rejects_bad_arguments(hip):
    assert not hip.mlp_fused_ok(96, 512, 1024) and not hip.mlp_fused_ok(64, 500, 1024) and not hip.mlp_fused_ok(64, 512, 1000)
    w1, w2 = torch.zeros(512, 64, device="cuda"), torch.zeros(64, 512, device="cuda")
    fwd, _ = hip.mlp_pack(w1, w2, 64, 512, 2)
    x, y = torch.zeros(128, 64, device="cuda"), torch.zeros(128, 64, device="cuda")
    with pytest.raises(RuntimeError, match="row stride"):
        hip.mlp_fwd(x, 32, fwd, None, None, None, 0, None, y, 64, None, 0, None, 128, 64, 512, 2)
    with pytest.raises(RuntimeError, match="precision"):
        hip.mlp_fwd(x, 64, fwd, None, None, None, 0, None, y, 64, None, 0, None, 128, 64, 512, 0)
    with pytest.raises(RuntimeError, match="no fused Mlp kernel"):
        hip.mlp_fwd(x, 64, fwd, None, None, None, 0, None, y, 64, None, 0, None, 128, 96, 512, 2)


def test_gelu_and_derivative_accuracy(hip):
    """The kernels' exact-erf GELU (common.h: branch-free erf(z) = 1 - Q(s) exp(-z^2)) against fp64, through an identity 1x1
    conv on the fp32 MFMA (exact for a 0/1 weight matrix): absolute error at the level of torch's own fp32 formula."""
    C, M = 64, 4096
    vals = torch.cat([torch.linspace(-9, 9, C * M - 8), torch.tensor([0.0, -0.0, 1e-20, -1e-20, 30.0, -30.0, 1e-4, -1e-4])])
    x = vals[torch.randperm(C * M, generator=torch.Generator().manual_seed(0))].view(1, 64, 64, C).cuda().contiguous()
    eye = torch.eye(C).view(C, C, 1, 1).cuda().contiguous()
    y = torch.empty_like(x)
    hip.conv2d(x, C, eye, None, y, C, 1, 64, 64, C, 64, 64, C, 1, 1, 1, 0, 1, act=2, precision=0)
    xd = x.double().cpu()
    ref = 0.5 * xd * (1 + torch.erf(xd / np.sqrt(2.0)))
    err = (y.double().cpu() - ref).abs()
    assert (err / xd.abs().clamp_min(1.0)).max().item() < 2.5e-7          # (torch's own fp32 formula: 1.1e-7 |x|)
    g = torch.ones_like(x)
    dx = torch.empty_like(x)
    hip.conv2d(g, C, eye, None, dx, C, 1, 64, 64, C, 64, 64, C, 1, 1, 1, 0, 1, mode=1, aux=x, ldaux=C, precision=0)
    gref = 0.5 * (1 + torch.erf(xd / np.sqrt(2.0))) + xd * torch.exp(-0.5 * xd * xd) / np.sqrt(2 * np.pi)
    assert (dx.double().cpu() - gref).abs().max().item() < 3e-7
    # non-finite pre-activations: NaN stays NaN (and, as in any fp32 GEMM, poisons only its own pixel's outputs)
    sp = torch.zeros(1, 8, 8, C, device="cuda")
    sp[0, 0, 0, 2] = float("nan")
    ys = torch.empty_like(sp)
    hip.conv2d(sp, C, eye, None, ys, C, 1, 8, 8, C, 8, 8, C, 1, 1, 1, 0, 1, act=2, precision=0)
    assert torch.isnan(ys[0, 0, 0, 2]) and torch.isfinite(ys[0, 1:]).all()


def test_narrow_conv_head_layout(hip):
    """The head's prediction convs as the program issues them (decouplehead.py:74-86): NCHW stores into channel ranges of one
    (B, 5 + nc, h, w) tensor, data / weight gradients from channel SLICES of the (B, h, w, 5 + nc) gradient (row stride 9,
    odd offsets), the data gradients of reg and obj accumulated into one buffer."""
    B, H, W, K, ctot = 2, 16, 16, 256, 9
    g = rnd(B, K, H, W, seed=1)
    ws = [rnd(n, K, 1, 1, seed=2 + i) / 16 for i, n in enumerate((4, 1, 4))]
    bs = [rnd(n, seed=5 + i) for i, n in enumerate((4, 1, 4))]
    out = torch.zeros(B, ctot, H, W, device="cuda")
    gg = nhwc(g)
    for w, b, off in zip(ws, bs, (0, 4, 5)):
        hip.conv2d(gg, K, w.cuda(), b.cuda(), out, 0, B, H, W, K, H, W, w.shape[0], 1, 1, 1, 0, 1, out_nchw=1, out_ctot=ctot, out_coff=off)
        assert hip.last_kernel() == 5
    ref = torch.cat([F.conv2d(g, w, b) for w, b in zip(ws, bs)], 1)
    close(out.cpu(), ref, what="nchw slices")
    d = nhwc(rnd(B, ctot, H, W, seed=9))                      # (B, H, W, 9) gradient
    dx = torch.empty(B, H, W, K, device="cuda")
    acc = 0
    for w, off in zip(ws, (0, 4, 5)):
        hip.conv2d(d[..., off:], ctot, w.cuda(), None, dx, K, B, H, W, K, H, W, w.shape[0], 1, 1, 1, 0, 1, mode=1, accumulate=acc)
        assert hip.last_kernel() == 5
        acc = 1
    dref = sum(F.conv_transpose2d(nchw(d)[:, off:off + w.shape[0]], w) for w, off in zip(ws, (0, 4, 5)))
    close(nchw(dx), dref, what="dgrad from slices, accumulated")
    for w, off in zip(ws, (0, 4, 5)):
        n = w.shape[0]
        dw, db = torch.empty(n, K, 1, 1, device="cuda"), torch.empty(n, device="cuda")
        hip.conv2d_wgrad(gg, K, d[..., off:], ctot, dw, db, None, B, H, W, K, H, W, n, 1, 1, 1, 0, 1)
        assert hip.last_kernel() == 5
        dn = nchw(d)[:, off:off + n]
        close(dw, torch.einsum("bnhw,bkhw->nk", dn, g).view(n, K, 1, 1), what="wgrad from a slice")
        close(db, dn.sum((0, 2, 3)), what="bgrad from a slice")


def test_x6_non_finite_and_tiny_operands(hip):
    """Documented edge semantics of the six-product scheme (include/vrnet_hip.h): an Inf operand splits into (Inf, NaN, NaN)
    and so yields NaN where fp32 arithmetic would give +-Inf; NaN stays NaN; both stay confined to their own output rows.
    Operands down to 2^-100 keep full fp32 accuracy; below ~2^-110 the low planes fall into the bf16 denormal range."""
    B, H, W, Ci, Co = 2, 128, 128, 64, 128
    x = rnd(B, H, W, Ci, seed=1).cuda()
    w = (rnd(Co, Ci, 1, 1, seed=2) / 8).cuda()
    y0 = torch.empty(B, H, W, Co, device="cuda")
    hip.conv2d(x, Ci, w, None, y0, Co, B, H, W, Ci, H, W, Co, 1, 1, 1, 0, 1, precision=2)
    assert hip.last_kernel() == 6
    xi = x.clone()
    xi[0, 0, 0, 3] = float("inf")
    xi[1, 5, 7, 9] = float("nan")
    y = torch.empty_like(y0)
    hip.conv2d(xi, Ci, w, None, y, Co, B, H, W, Ci, H, W, Co, 1, 1, 1, 0, 1, precision=2)
    assert torch.isnan(y[0, 0, 0]).all() and torch.isnan(y[1, 5, 7]).all()          # (fp32 arithmetic: +-Inf resp. NaN)
    y[0, 0, 0], y[1, 5, 7] = y0[0, 0, 0], y0[1, 5, 7]
    assert torch.equal(y, y0)                                                        # every other row untouched
    ref = (x.double().view(-1, Ci) @ w.double().view(Co, Ci).T).view(B, H, W, Co)
    for e, tol in ((-100, 2e-5), (-120, 1e-2)):
        sc = 2.0 ** e
        hip.conv2d(x * sc, Ci, w, None, y, Co, B, H, W, Ci, H, W, Co, 1, 1, 1, 0, 1, precision=2)
        err = ((y.double() / sc - ref).abs().max() / ref.abs().max()).item()
        print(f"operands scaled by 2^{e}: rel err {err:.2e}")
        assert err < tol, (e, err)


def test_torch_library_custom_ops_round3(hip):
    """The remaining fused ops as torch.ops.vrnet.* (ops.py): values and autograd against ATen, schema / fake kernels by
    opcheck, and the autocast registration (fp32 inside an autocast region)."""
    import asy_vrnet_amd.ops  # noqa: F401
    opcheck = lambda op, args: torch.library.opcheck(op, args, test_utils=("test_schema", "test_faketensor"))
    # ---- mlp: res + ls * fc2(gelu(fc1(x)))
    B, H, W, C, hid = 2, 8, 8, 64, 256
    mk = lambda *s_, seed, sc=1.0: (rnd(*s_, seed=seed) * sc).cuda().requires_grad_(True)
    x, res = mk(B, H, W, C, seed=1), mk(B, H, W, C, seed=2)
    w1, b1, w2, b2, ls = mk(hid, C, 1, 1, seed=3, sc=1 / 8), mk(hid, seed=4), mk(C, hid, 1, 1, seed=5, sc=1 / 16), mk(C, seed=6), mk(C, seed=7)
    y, _ = torch.ops.vrnet.mlp(x, w1, b1, w2, b2, res, ls)
    ref = res + ls * (F.gelu(x @ w1.view(hid, C).T + b1) @ w2.view(C, hid).T + b2)
    close(y, ref, what="mlp op")
    g = rnd(B, H, W, C, seed=8).cuda()
    mine = torch.autograd.grad(y, (x, w1, b1, w2, b2, res, ls), g)
    want = torch.autograd.grad(ref, (x, w1, b1, w2, b2, res, ls), g)
    for a, b_, nm in zip(mine, want, ("dx", "dw1", "db1", "dw2", "db2", "dres", "dls")):
        close(a, b_, 2e-4, what="mlp op " + nm)
    opcheck(torch.ops.vrnet.mlp.default, tuple(t.detach() for t in (x, w1, b1, w2, b2, res, ls)))
    # ---- GroupNorm(1, C)
    gam, bet = mk(C, seed=9), mk(C, seed=10)
    y, _ = torch.ops.vrnet.group_norm1(x, gam, bet, 1e-5)
    ref = F.group_norm(x.permute(0, 3, 1, 2), 1, gam, bet, 1e-5).permute(0, 2, 3, 1)
    close(y, ref, what="group_norm1")
    for a, b_, nm in zip(torch.autograd.grad(y, (x, gam, bet), g), torch.autograd.grad(ref, (x, gam, bet), g), ("dx", "dgamma", "dbeta")):
        close(a, b_, 2e-4, what="group_norm1 " + nm)
    opcheck(torch.ops.vrnet.group_norm1.default, (x.detach(), gam.detach(), bet.detach(), 1e-5))
    # ---- BatchNorm + ReLU (train mode), functional running statistics
    rm, rv = rnd(C, seed=11).cuda() * 0.1, rnd(C, seed=12, kind="uniform").cuda() + 0.5
    y, _, rm2, rv2 = torch.ops.vrnet.batch_norm_act(x, gam, bet, rm, rv, True, 0.03, 1e-3, True)
    rmr, rvr = rm.clone(), rv.clone()
    ref = torch.relu(F.batch_norm(x.permute(0, 3, 1, 2), rmr, rvr, gam, bet, True, 0.03, 1e-3)).permute(0, 2, 3, 1)
    close(y, ref, what="batch_norm_act")
    close(rm2, rmr, what="running_mean"); close(rv2, rvr, what="running_var")
    for a, b_, nm in zip(torch.autograd.grad(y, (x, gam, bet), g), torch.autograd.grad(ref, (x, gam, bet), g), ("dx", "dgamma", "dbeta")):
        close(a, b_, 5e-4, what="batch_norm_act " + nm)
    opcheck(torch.ops.vrnet.batch_norm_act.default, (x.detach(), gam.detach(), bet.detach(), rm, rv, True, 0.03, 1e-3, True))
    # ---- depthwise 3x3 and bilinear upsampling
    wd = mk(C, 1, 3, 3, seed=13)
    y = torch.ops.vrnet.dwconv3x3(x, wd)
    ref = F.conv2d(x.permute(0, 3, 1, 2), wd, None, 1, 1, 1, C).permute(0, 2, 3, 1)
    close(y, ref, what="dwconv3x3 op")
    for a, b_, nm in zip(torch.autograd.grad(y, (x, wd), g), torch.autograd.grad(ref, (x, wd), g), ("dx", "dw")):
        close(a, b_, 2e-4, what="dwconv3x3 op " + nm)
    opcheck(torch.ops.vrnet.dwconv3x3.default, (x.detach(), wd.detach()))
    y = torch.ops.vrnet.upsample_bilinear(x, 2)
    ref = F.interpolate(x.permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=True).permute(0, 2, 3, 1)
    close(y, ref, what="upsample op")
    g2 = rnd(B, 2 * H, 2 * W, C, seed=14).cuda()
    close(torch.autograd.grad(y, x, g2)[0], torch.autograd.grad(ref, x, g2)[0], 2e-4, what="upsample op dx")
    opcheck(torch.ops.vrnet.upsample_bilinear.default, (x.detach(), 2))
    # ---- autocast: half-precision arguments are cast to fp32 on the way in, the result is fp32
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ya = torch.ops.vrnet.dwconv3x3(x.detach().bfloat16(), wd.detach())
    assert ya.dtype == torch.float32
    close(ya, torch.ops.vrnet.dwconv3x3(x.detach().bfloat16().float(), wd.detach()), 1e-6, what="autocast")


def test_torch_library_custom_ops_round4(hip):
    """torch.ops.vrnet.{shuffle_attention, eca, image_enhance, radar_enhance} (ops.py): values and autograd against
    the CPU oracle's restatement of the reference modules, schema / fake kernels by opcheck, autocast registration."""
    import asy_vrnet_amd.ops  # noqa: F401
    from oracle import vrnet_oracle as O
    opcheck = lambda op, args: torch.library.opcheck(op, args, test_utils=("test_schema", "test_faketensor"))
    B, H, W, C, G = 2, 8, 6, 64, 8
    cp = C // (2 * G)
    x = rnd(B, C, H, W, seed=1).requires_grad_(True)
    names = ["cweight", "cbias", "sweight", "sbias"]
    P = {"m." + n: rnd(1, cp, 1, 1, seed=10 + i).requires_grad_(True) for i, n in enumerate(names)}
    P["m.gn.weight"] = (rnd(cp, seed=20) * 0.3 + 1).requires_grad_(True)
    P["m.gn.bias"] = rnd(cp, seed=21).requires_grad_(True)
    order = ["m.cweight", "m.cbias", "m.sweight", "m.sbias", "m.gn.weight", "m.gn.bias"]
    g = rnd(B, C, H, W, seed=3)
    ref = O.shuffle_attention(P, "m", x, G)
    want = torch.autograd.grad(ref, [x] + [P[k] for k in order], g)
    xg = nhwc(x).requires_grad_(True)
    prm = [P[k].detach().cuda().requires_grad_(True) for k in order]
    y = torch.ops.vrnet.shuffle_attention(xg, *prm, G)[0]
    close(nchw(y), ref, what="shuffle_attention op")
    mine = torch.autograd.grad(y, [xg] + prm, nhwc(g))
    close(nchw(mine[0]), want[0], 2e-4, what="shuffle_attention op dx")
    for a, b_, k in zip(mine[1:], want[1:], order):
        assert a.shape == b_.shape
        close(a, b_, 2e-4, what="shuffle_attention op d" + k)
    opcheck(torch.ops.vrnet.shuffle_attention.default, (xg.detach(), *[t.detach() for t in prm], G))
    # ---- ECA
    k = O.eca_kernel_size(C)
    Pe = {"m.conv.weight": rnd(1, 1, k, seed=2).requires_grad_(True)}
    ref = O.eca(Pe, "m", x)
    want = torch.autograd.grad(ref, [x, Pe["m.conv.weight"]], g)
    we = Pe["m.conv.weight"].detach().cuda().requires_grad_(True)
    y = torch.ops.vrnet.eca(xg, we)[0]
    close(nchw(y), ref, what="eca op")
    mine = torch.autograd.grad(y, [xg, we], nhwc(g))
    close(nchw(mine[0]), want[0], 2e-4, what="eca op dx")
    close(mine[1], want[1], 2e-4, what="eca op dw")
    opcheck(torch.ops.vrnet.eca.default, (xg.detach(), we.detach()))
    # ---- the gain of ImageEnhanceByRadar: (1 + data_normal(p)) * x, data_normal over the whole batch tensor (vr_coc.py:59-67)
    p = torch.relu(rnd(B, C, H, W, seed=4)).requires_grad_(True)
    d = p.max() - p.min()
    ref = (1 + (p - p.min()) / d) * x
    want = torch.autograd.grad(ref, [p, x], g)
    pg = nhwc(p).requires_grad_(True)
    t = torch.ops.vrnet.image_enhance(pg, xg)[0]
    close(nchw(t), ref, what="image_enhance op")
    mine = torch.autograd.grad(t, [pg, xg], nhwc(g))
    close(nchw(mine[0]), want[0], 2e-4, what="image_enhance op dp")
    close(nchw(mine[1]), want[1], 2e-4, what="image_enhance op dx")
    opcheck(torch.ops.vrnet.image_enhance.default, (pg.detach(), xg.detach()))
    # ---- the gate of RadarEnhanceByImage: eca(shuffle_channels(cat([a, r]), 2))
    r = rnd(B, C, H, W, seed=5).requires_grad_(True)
    k2 = O.eca_kernel_size(2 * C)
    Pr = {"m.conv.weight": rnd(1, 1, k2, seed=6).requires_grad_(True)}
    ref = O.eca(Pr, "m", O.shuffle2(torch.cat([x, r], 1)))
    g2 = rnd(B, 2 * C, H, W, seed=7)
    want = torch.autograd.grad(ref, [x, r, Pr["m.conv.weight"]], g2)
    rg = nhwc(r).requires_grad_(True)
    wr = Pr["m.conv.weight"].detach().cuda().requires_grad_(True)
    u = torch.ops.vrnet.radar_enhance(xg, rg, wr)[0]
    close(nchw(u), ref, what="radar_enhance op")
    mine = torch.autograd.grad(u, [xg, rg, wr], nhwc(g2))
    close(nchw(mine[0]), want[0], 2e-4, what="radar_enhance op da")
    close(nchw(mine[1]), want[1], 2e-4, what="radar_enhance op dr")
    close(mine[2], want[2], 2e-4, what="radar_enhance op dw")
    opcheck(torch.ops.vrnet.radar_enhance.default, (xg.detach(), rg.detach(), wr.detach()))
    # ---- autocast: half-precision arguments are cast to fp32 on the way in
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ya = torch.ops.vrnet.eca(xg.detach().bfloat16(), we.detach())[0]
    assert ya.dtype == torch.float32
    close(ya, torch.ops.vrnet.eca(xg.detach().bfloat16().float(), we.detach())[0], 1e-6, what="autocast")


@pytest.mark.parametrize("shape", [(2, 16, 16, 64), (3, 8, 8, 320), (8, 128, 128, 64), (2, 32, 32, 128)])
def test_group_norm_one_and_two_launch_forms(hip, shape):
    """GroupNorm(1, C) as the network runs it on the block chain: forward in ONE launch from the producer's tile pairs
    (vrnet_gn_apply_fwd), backward in TWO (moments + apply with in-kernel coefficients and parameter gradients,
    vrnet_gn_apply_bwd), against ATen in fp64; |mean| >> std on purpose."""
    B, H, W, C = shape
    x = (rnd(B, H, W, C, seed=1) * 1.5 + 10.0)
    gam, bet = rnd(C, seed=2) * 0.3 + 1, rnd(C, seed=3) * 0.5
    g, addend = rnd(B, H, W, C, seed=4), rnd(B, H, W, C, seed=5)
    xd = x.double().requires_grad_(True)
    gd, bd = gam.double().requires_grad_(True), bet.double().requires_grad_(True)
    ref = F.group_norm(xd.permute(0, 3, 1, 2), 1, gd, bd, 1e-5).permute(0, 2, 3, 1)
    ref.backward(g.double())
    # tile pairs as a conv epilogue leaves them: (sum, sumsq) per 32 rows x 32 channels, fp64
    nb = (C + 31) // 32
    t = x.double().view(B, H * W // 32, 32, C)
    pairs = torch.zeros(B, H * W // 32, nb, 2, dtype=torch.float64)
    for j in range(nb):
        blk = t[..., 32 * j:32 * (j + 1)]
        pairs[:, :, j, 0], pairs[:, :, j, 1] = blk.sum((2, 3)), (blk * blk).sum((2, 3))
    xg, y, ms = x.cuda(), torch.empty(B, H, W, C, device="cuda"), torch.empty(B, 2, device="cuda")
    hip.gn_apply_fwd(xg, C, pairs.cuda(), (H * W // 32) * nb, gam.cuda(), bet.cuda(), 1e-5, B, H * W, C, y, C, ms)
    close(y, ref, 2e-5, what="gn forward")
    mean = x.double().mean((1, 2, 3))
    close(ms[:, 0], mean, 1e-6, what="mean")
    close(ms[:, 1], 1 / torch.sqrt(x.double().var((1, 2, 3), unbiased=False) + 1e-5), 1e-5, what="rstd")
    dx, dg, db = torch.empty(B, H, W, C, device="cuda"), torch.full((C,), 2.0, device="cuda"), torch.full((C,), 2.0, device="cuda")
    hip.gn_apply_bwd(g.cuda(), C, xg, C, ms, gam.cuda(), B, H * W, C, dx, C, dg, db, 1, add=addend.cuda(), ldadd=C)
    close(dx, xd.grad + addend.double(), 5e-5, what="gn dx + add")
    close(dg, gd.grad + 2, 5e-5, what="dgamma (accumulated)")
    close(db, bd.grad + 2, 5e-5, what="dbeta (accumulated)")
    dx2 = addend.cuda().clone()                                   # in-place accumulate: add == out
    hip.gn_apply_bwd(g.cuda(), C, xg, C, ms, gam.cuda(), B, H * W, C, dx2, C, dg, db, 0, add=dx2, ldadd=C)
    assert torch.equal(dx2, dx)
    close(dg, gd.grad, 5e-5, what="dgamma")


@pytest.mark.parametrize("case", [(2, 16, 16, 64, 96, 1, 1, 0, 1, 0), (8, 32, 32, 320, 128, 1, 1, 0, 1, 2), (2, 128, 128, 64, 128, 1, 1, 0, 1, 2),
                                  (2, 32, 32, 64, 64, 3, 1, 1, 1, 2), (2, 16, 16, 96, 48, 3, 2, 1, 1, 0), (2, 32, 32, 128, 64, 1, 1, 0, 1, 1)])
def test_conv_column_statistics(hip, case):
    """conv2d `colstats`: per-channel (sum, sum of squares) partials of the stored outputs per 32-row tile -- train-mode
    BatchNorm statistics without a pass over z -- and, on a data gradient, the GroupNorm-backward moments (sum dy, sum dy * x)
    with gamma-weighted tile totals; both against fp64 ATen, and through the coefficient kernels that consume them."""
    B, H, W, Ci, Co, k, s, p, d, prec = case
    OH, OW = (H + 2 * p - d * (k - 1) - 1) // s + 1, (W + 2 * p - d * (k - 1) - 1) // s + 1
    x = rnd(B, Ci, H, W, seed=1)
    w, b = rnd(Co, Ci, k, k, seed=2) / np.sqrt(Ci * k * k), rnd(Co, seed=3)
    if prec == 1:
        x, w = _bf16r(x), _bf16r(w)
    z = F.conv2d(x.double(), w.double(), b.double(), s, p, d)
    xg, wp = nhwc(x), pack(hip, w)
    y = torch.empty(B, OH, OW, Co, device="cuda")
    part, _ = hip.colstats_buffers(B, OH * OW, Co, "cuda")
    hip.conv2d(xg, Ci, wp, b.cuda(), y, Co, B, H, W, Ci, OH, OW, Co, k, k, s, p, d, precision=prec, colstats=(part, None, 0, None, None))
    yd = y.double().cpu().view(B * OH * OW // 32, 32, Co)
    close(part[..., 0], yd.sum(1), 1e-9, what="tile column sums")
    close(part[..., 1], (yd * yd).sum(1), 1e-9, what="tile column sums of squares")
    close(part[..., 0].sum(0), z.sum((0, 2, 3)), 2e-4 if prec == 1 else 2e-5, what="channel sums vs ATen")
    # ... and the BatchNorm coefficients from them
    gam, bet = rnd(Co, seed=4).cuda() * 0.3 + 1, rnd(Co, seed=5).cuda()
    rm, rv, nbt = torch.zeros(Co, device="cuda"), torch.ones(Co, device="cuda"), torch.zeros((), dtype=torch.int64, device="cuda")
    A, D, S, ms = (torch.empty(Co, device="cuda") for _ in range(3)) , None, None, torch.empty(Co, 2, device="cuda")
    A, D, S = A
    hip.bn_coef_fwd_from_partials(part, gam, bet, 1e-3, 0.03, rm, rv, nbt, B, OH * OW, Co, A, D, S, ms)
    yy = y.double().cpu().view(-1, Co)
    close(S, yy.mean(0), 1e-6, what="bn mean")
    close(A, gam.double().cpu() / torch.sqrt(yy.var(0, unbiased=False) + 1e-3), 1e-5, what="bn scale")
    close(rv, 0.97 + 0.03 * yy.var(0, unbiased=True), 1e-5, what="running_var")
    assert int(nbt) == 1
    if s != 1:
        return
    # ---- data gradient with GroupNorm-backward moments: dxn = conv^T(g); sums of dxn and dxn * xin per channel, gamma totals
    g = rnd(B, Co, OH, OW, seed=6)
    xin, gamma = rnd(B, H, W, Ci, seed=7).cuda(), (rnd(Ci, seed=8) * 0.3 + 1).cuda()
    dx = torch.empty(B, H, W, Ci, device="cuda")
    part, tot = hip.colstats_buffers(B, H * W, Ci, "cuda", totals=True)
    hip.conv2d(nhwc(g), Co, wp, None, dx, Ci, B, H, W, Ci, OH, OW, Co, k, k, s, p, d, mode=1, precision=0 if prec == 1 else prec,
               colstats=(part, xin, Ci, gamma, tot))
    dd, xx = dx.double().cpu().view(B * H * W // 32, 32, Ci), xin.double().cpu().view(B * H * W // 32, 32, Ci)
    close(part[..., 0], dd.sum(1), 1e-9, what="dgrad tile column sums")
    close(part[..., 1], (dd * xx).sum(1), 1e-9, what="dgrad tile column sums of dy * x")
    gd = gamma.double().cpu()
    nb = (Ci + 31) // 32
    pad = torch.zeros(B * H * W // 32, nb * 32, 2, dtype=torch.float64)
    pad[:, :Ci, 0], pad[:, :Ci, 1] = dd.sum(1) * gd, (dd * xx).sum(1) * gd
    close(tot, pad.view(-1, nb, 32, 2).sum(2), 1e-9, what="gamma-weighted tile totals")
    # the one-launch GroupNorm backward that consumes them, against the two-launch form
    ms = torch.stack([xin.double().mean((1, 2, 3)), 1 / torch.sqrt(xin.double().var((1, 2, 3), unbiased=False) + 1e-5)], 1).float().cuda()
    outs = []
    for form in (0, 1):
        o, dg, db = torch.empty(B, H, W, Ci, device="cuda"), torch.empty(Ci, device="cuda"), torch.empty(Ci, device="cuda")
        if form == 0:
            hip.gn_apply_bwd(dx, Ci, xin, Ci, ms, gamma, B, H * W, Ci, o, Ci, dg, db, 0)
        else:
            hip.gn_apply_bwd_from_partials(dx, Ci, xin, Ci, part, tot, ms, gamma, B, H * W, Ci, o, Ci, dg, db, 0)
        outs.append((o, dg, db))
    for a, b_, nm in zip(outs[1], outs[0], ("dx", "dgamma", "dbeta")):
        close(a, b_, 1e-5, what="gn backward from partials: " + nm)


def _planes(hip, w2d, J, K, sj, sk, kscale=None):
    """bf16 planes of one weight through the multi-tensor pack entry point (a one-entry table)."""
    buf = torch.empty((hip.conv_planes_bytes(J, K),), dtype=torch.uint8, device="cuda")
    nb = (K // 16) * 2 * ((J + 127) // 128)
    tab = torch.tensor([w2d.data_ptr(), J, K, sj, sk, 0 if kscale is None else kscale.data_ptr(), buf.data_ptr(), 0], dtype=torch.int64,
                       device="cuda")
    hip.conv_planes_pack(tab, 1, nb)
    return buf


@pytest.mark.parametrize("case", [(2, 128, 128, 64, 128), (2, 128, 128, 64, 512), (2, 128, 128, 512, 64), (8, 32, 32, 320, 1280),
                                  (8, 32, 32, 1280, 320), (4, 64, 64, 128, 96), (8, 64, 64, 256, 192), (2, 128, 128, 80, 64),
                                  (8, 16, 16, 2048, 512), (8, 16, 16, 512, 2048), (8, 16, 16, 1024, 512),      # split contraction (4, 4, 2 splits)
                                  (8, 64, 64, 128, 256), (8, 128, 128, 64, 128), (3, 128, 128, 64, 192), (2, 128, 128, 128, 128)])      # K = 64 / 128 over >= 32 768 rows: resident-B streaming kernel
def test_x6_conv_with_presplit_weights(hip, case):
    """precision 2 with `w_planes` (weights split once into bf16 planes by vrnet_conv_planes_pack_f32, kernel family 9):
    forward with the full epilogue and data gradient with the layer scale folded into the pack, against fp64 ATen."""
    B, H, W, Ci, Co = case
    t = _conv_suite_inputs((B, H, W, Ci, Co, 1, 1, 0, 1))
    x, w, b, ls, res = nhwc(t["x"]), t["w"].cuda().contiguous(), t["b"].cuda(), t["ls"].cuda(), nhwc(t["res"])
    pf = _planes(hip, w, Co, Ci, Ci, 1)
    y, ypre = torch.empty(B, H, W, Co, device="cuda"), torch.empty(B, H, W, Co, device="cuda")
    st, per = hip.conv_stats_buffer(B, H * W, Co, x.device)
    hip.conv2d(x, Ci, w, b, y, Co, B, H, W, Ci, H, W, Co, 1, 1, 1, 0, 1, act=2, ypre=ypre, ldypre=Co, res=res, ldres=Co, res_scale=ls,
               stats=st, precision=2, w_planes=pf)
    fam_f = hip.last_kernel()
    D = lambda a: a.double()
    z = F.conv2d(D(t["x"]), D(t["w"]), D(t["b"]))
    close(nchw(ypre), z, 2e-5, what="ypre")
    close(nchw(y), D(t["res"]) + D(t["ls"])[None, :, None, None] * F.gelu(z), 2e-5, what="y")
    yd = y.double().cpu()
    close(st.view(B, -1, 2).sum(1), torch.stack([yd.sum((1, 2, 3)), (yd * yd).sum((1, 2, 3))], 1), 1e-9, what="statistics of the stored outputs")
    # data gradient: the pack holds w^T with the contraction scale folded in
    pb = _planes(hip, w, Ci, Co, 1, Ci, kscale=ls)
    g, aux, dx = nhwc(t["g"]), nhwc(t["aux"]), nhwc(t["dx0"])
    hip.conv2d(g, Co, w, None, dx, Ci, B, H, W, Ci, H, W, Co, 1, 1, 1, 0, 1, mode=1, kscale=ls, aux=aux, ldaux=Ci, accumulate=1, precision=2,
               w_planes=pb)
    fam_d = hip.last_kernel()
    a = D(t["aux"])
    gp = 0.5 * (1 + torch.erf(a / np.sqrt(2.0))) + a * torch.exp(-0.5 * a * a) / np.sqrt(2 * np.pi)
    ref = D(t["dx0"]) + F.conv_transpose2d(D(t["g"]) * D(t["ls"])[None, :, None, None], D(t["w"])) * gp
    close(nchw(dx), ref, 2e-5, what="dx")
    assert 9 in (fam_f, fam_d), (fam_f, fam_d)        # at least one of the two launches had a tile kernel and used the planes


@pytest.mark.parametrize("interleave", [False, True])
@pytest.mark.parametrize("case", [(37, 8, 8), (1024, 64, 64), (200, 12, 20), (50, 6, 10), (33, 6, 6), (4096, 128, 128)])      # (6, .): scalar kernel
def test_cat2_and_adjoint(hip, case, interleave):
    """torch.cat([a, b], channels) (+ shuffle_channels(groups=2)) in one launch, and its adjoint with per-source accumulate
    flags and an absent source (vr_coc.py:70-80, coc_fpn_dual.py:120-130)."""
    rows, Ca, Cb = case
    if interleave and Ca != Cb:
        with pytest.raises(RuntimeError, match="equal width"):
            hip.cat2(torch.zeros(rows, Ca, device="cuda"), Ca, Ca, torch.zeros(rows, Cb, device="cuda"), Cb, Cb,
                     torch.zeros(rows, Ca + Cb, device="cuda"), Ca + Cb, rows, True)
        return
    g = torch.Generator().manual_seed(1)
    a, b = torch.randn(rows, Ca + 4, generator=g).cuda(), torch.randn(rows, Cb, generator=g).cuda()      # a: row stride > width
    out = torch.empty(rows, Ca + Cb, device="cuda")
    hip.cat2(a, Ca + 4, Ca, b, Cb, Cb, out, Ca + Cb, rows, interleave)
    ref = torch.cat([a[:, :Ca], b], 1)
    if interleave:
        ref = ref.view(rows, 2, Ca).transpose(1, 2).reshape(rows, 2 * Ca)          # shuffle_channels(groups=2)
    assert torch.equal(out, ref)
    gr = torch.randn(rows, Ca + Cb, generator=g).cuda()
    ga0, gb0 = torch.randn(rows, Ca, generator=g).cuda(), torch.randn(rows, Cb, generator=g).cuda()
    ga, gb = ga0.clone(), gb0.clone()
    hip.cat2(ga, Ca, Ca, gb, Cb, Cb, gr, Ca + Cb, rows, interleave, dir=1, accumulate_a=1, accumulate_b=0)
    ra, rb = (gr[:, 0::2], gr[:, 1::2]) if interleave else (gr[:, :Ca], gr[:, Ca:])
    assert torch.equal(ga, ga0 + ra) and torch.equal(gb, rb)
    gb2 = gb0.clone()
    hip.cat2(None, Ca, Ca, gb2, Cb, Cb, gr, Ca + Cb, rows, interleave, dir=1, accumulate_b=1)
    assert torch.equal(gb2, gb0 + rb)


def test_torch_library_cat_shuffle_and_batch_formats(hip):
    """torch.ops.vrnet.cat_shuffle (cat + 2-group channel shuffle, autograd) and torch.ops.vrnet.batch_formats (input formats
    from bytes) against ATen / the host functions; schemas and fake kernels by opcheck."""
    import asy_vrnet_amd.ops  # noqa: F401
    from asy_vrnet_amd import data
    opcheck = lambda op, args: torch.library.opcheck(op, args, test_utils=("test_schema", "test_faketensor"))
    a = rnd(2, 6, 5, 16, seed=1).cuda().requires_grad_(True)
    b = rnd(2, 6, 5, 16, seed=2).cuda().requires_grad_(True)
    for il in (False, True):
        y = torch.ops.vrnet.cat_shuffle(a, b, il)
        ref = torch.cat([a, b], -1)
        if il:
            ref = ref.view(2, 6, 5, 2, 16).transpose(3, 4).reshape(2, 6, 5, 32)
        assert torch.equal(y, ref)
        g = rnd(2, 6, 5, 32, seed=3).cuda()
        for m_, w_ in zip(torch.autograd.grad(y, (a, b), g), torch.autograd.grad(ref, (a, b), g)):
            assert torch.equal(m_, w_)
        opcheck(torch.ops.vrnet.cat_shuffle.default, (a.detach(), b.detach(), il))
    rng = np.random.default_rng(0)
    img = torch.from_numpy(rng.integers(0, 256, (2, 8, 12, 3), dtype=np.uint8)).cuda()
    png = torch.from_numpy(rng.integers(0, 256, (2, 8, 12), dtype=np.uint8)).cuda()
    images, lab, onehot = torch.ops.vrnet.batch_formats(img, png, 9)
    want = np.stack([np.transpose(data.preprocess_input(im), [2, 0, 1]) for im in img.cpu().numpy()]).astype(np.float32)
    assert np.array_equal(images.cpu().numpy(), want)
    assert np.array_equal(lab.cpu().numpy(), np.minimum(png.cpu().numpy(), 9).astype(np.int64))
    assert np.array_equal(onehot.cpu().numpy().argmax(-1), lab.cpu().numpy())
    opcheck(torch.ops.vrnet.batch_formats.default, (img, png, 9))
