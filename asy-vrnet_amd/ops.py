"""The fused kernels as PyTorch custom ops in the ``vrnet::`` namespace (what a native replacement exports).

``torch.ops.vrnet.cluster`` and ``torch.ops.vrnet.conv2d_nhwc`` wrap the same C-ABI entry points the whole-network
program uses (hip.py), registered with ``torch.library`` so that they carry schemas, fake (meta) kernels for shape
inference and autograd formulas: they compose with ordinary torch code, ``torch.autograd`` and ``torch.library.opcheck``.
The CUDA dispatch key is the HIP device on ROCm.  There is no CPU kernel: calling them with CPU tensors raises.

    out, idx = torch.ops.vrnet.cluster(f, v, alpha, beta, heads, fold)     # Cluster core, vr_coc.py:158-190
    y = torch.ops.vrnet.conv2d_nhwc(x, w, bias, stride, pad, dil)           # NHWC implicit-GEMM conv (fp32-accurate x6 / MFMA)
"""
import torch

from . import hip


@torch.library.custom_op("vrnet::cluster", mutates_args=(), device_types="cuda")
def cluster(f: torch.Tensor, v: torch.Tensor, alpha: torch.Tensor, beta: torch.Tensor, heads: int, fold: int) -> tuple[torch.Tensor, torch.Tensor]:
    """f, v: (B,H,W,E*D) NHWC fp32; alpha, beta: 1-element tensors.  Returns (out (B,H,W,E*D), idx (B,H,W,E) uint8)."""
    f, v = f.contiguous(), v.contiguous()
    B, H, W, ED = f.shape
    if ED % heads:
        raise RuntimeError("vrnet::cluster: channels must be heads * head_dim")
    out = torch.empty_like(f)
    idx = torch.empty((B, H, W, heads), dtype=torch.uint8, device=f.device)
    wgt = torch.empty((B, H, W, heads), dtype=torch.float32, device=f.device)
    with torch.cuda.device(f.device):
        hip.cluster_fwd(f, v, ED, alpha.reshape(1), beta.reshape(1), out, ED, idx, wgt, B, H, W, heads, ED // heads, fold)
    return out, idx


@cluster.register_fake
def _(f, v, alpha, beta, heads, fold):
    B, H, W, ED = f.shape
    return torch.empty_like(f), f.new_empty((B, H, W, heads), dtype=torch.uint8)


@torch.library.custom_op("vrnet::cluster_backward", mutates_args=(), device_types="cuda")
def cluster_backward(g: torch.Tensor, f: torch.Tensor, v: torch.Tensor, alpha: torch.Tensor, beta: torch.Tensor,
                     idx: torch.Tensor, heads: int, fold: int) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    g, f, v = g.contiguous(), f.contiguous(), v.contiguous()
    B, H, W, ED = f.shape
    df, dv = torch.empty_like(f), torch.empty_like(v)
    dab = torch.empty(2, dtype=torch.float32, device=f.device)
    with torch.cuda.device(f.device):
        hip.cluster_bwd(f, v, ED, alpha.reshape(1), beta.reshape(1), idx, g, ED, df, dv, ED, dab[0:1], dab[1:2], 0, B, H, W,
                        heads, ED // heads, fold)
    return df, dv, dab[0:1].clone().reshape(alpha.shape), dab[1:2].clone().reshape(beta.shape)


@cluster_backward.register_fake
def _(g, f, v, alpha, beta, idx, heads, fold):
    return torch.empty_like(f), torch.empty_like(v), torch.empty_like(alpha), torch.empty_like(beta)


def _cluster_setup(ctx, inputs, output):
    f, v, alpha, beta, heads, fold = inputs
    ctx.save_for_backward(f, v, alpha, beta, output[1])
    ctx.heads, ctx.fold = heads, fold


def _cluster_bwd(ctx, g_out, g_idx):
    f, v, alpha, beta, idx = ctx.saved_tensors
    df, dv, da, db = torch.ops.vrnet.cluster_backward(g_out, f, v, alpha, beta, idx, ctx.heads, ctx.fold)
    return df, dv, da, db, None, None


cluster.register_autograd(_cluster_bwd, setup_context=_cluster_setup)


def _geom(x, w, stride, pad, dil):
    B, H, W, ci = x.shape
    co, ci2, kh, kw = w.shape
    if ci2 != ci:
        raise RuntimeError(f"vrnet::conv2d_nhwc: input has {ci} channels, weight expects {ci2}")
    OH = (H + 2 * pad - dil * (kh - 1) - 1) // stride + 1
    OW = (W + 2 * pad - dil * (kw - 1) - 1) // stride + 1
    return B, H, W, ci, co, kh, kw, OH, OW


def _packed(w):
    co, ci, kh, kw = w.shape
    if kh * kw == 1:
        return w.contiguous()
    p = torch.empty((kh * kw, co, ci), dtype=torch.float32, device=w.device)
    hip.pack_weight(w.contiguous(), p, co, ci, kh, kw)
    return p


@torch.library.custom_op("vrnet::conv2d_nhwc", mutates_args=(), device_types="cuda")
def conv2d_nhwc(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor | None, stride: int, pad: int, dil: int) -> torch.Tensor:
    """x: (B,H,W,Cin) NHWC fp32; w: (Cout,Cin,kh,kw) OIHW as in nn.Conv2d.state_dict().  Returns (B,OH,OW,Cout)."""
    x = x.contiguous()
    B, H, W, ci, co, kh, kw, OH, OW = _geom(x, w, stride, pad, dil)
    y = torch.empty((B, OH, OW, co), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        hip.conv2d(x, ci, _packed(w), bias, y, co, B, H, W, ci, OH, OW, co, kh, kw, stride, pad, dil, precision=2)
    return y


@conv2d_nhwc.register_fake
def _(x, w, bias, stride, pad, dil):
    B, H, W, ci, co, kh, kw, OH, OW = _geom(x, w, stride, pad, dil)
    return x.new_empty((B, OH, OW, co))


@torch.library.custom_op("vrnet::conv2d_nhwc_backward", mutates_args=(), device_types="cuda")
def conv2d_nhwc_backward(g: torch.Tensor, x: torch.Tensor, w: torch.Tensor, stride: int, pad: int, dil: int) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    g, x = g.contiguous(), x.contiguous()
    B, H, W, ci, co, kh, kw, OH, OW = _geom(x, w, stride, pad, dil)
    dx = torch.empty_like(x)
    dw, db = torch.empty_like(w, memory_format=torch.contiguous_format), torch.empty(co, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        hip.conv2d(g, co, _packed(w), None, dx, ci, B, H, W, ci, OH, OW, co, kh, kw, stride, pad, dil, mode=1, precision=2)
        hip.conv2d_wgrad(x, ci, g, co, dw, db, None, B, H, W, ci, OH, OW, co, kh, kw, stride, pad, dil, precision=2)
    return dx, dw, db


@conv2d_nhwc_backward.register_fake
def _(g, x, w, stride, pad, dil):
    return torch.empty_like(x), torch.empty_like(w), w.new_empty((w.shape[0],))


def _conv_setup(ctx, inputs, output):
    x, w, bias, stride, pad, dil = inputs
    ctx.save_for_backward(x, w)
    ctx.geom = (stride, pad, dil)
    ctx.has_bias = bias is not None


def _conv_bwd(ctx, g):
    x, w = ctx.saved_tensors
    dx, dw, db = torch.ops.vrnet.conv2d_nhwc_backward(g, x, w, *ctx.geom)
    return dx, dw, (db if ctx.has_bias else None), None, None, None


conv2d_nhwc.register_autograd(_conv_bwd, setup_context=_conv_setup)


# ------------------------------------------------------------------------------------------------------------------
# Further fused ops of the path, same pattern: custom op + fake kernel + backward op + autograd + autocast.
#   y = torch.ops.vrnet.mlp(x, w1, b1, w2, b2, res, ls)            # res + ls * fc2(gelu(fc1(x))): ONE kernel per direction
#   y = torch.ops.vrnet.group_norm1(x, gamma, beta, eps)             # GroupNorm(1, C) over NHWC, vr_coc.py:105-111
#   y = torch.ops.vrnet.batch_norm_act(x, gamma, beta, running_mean, running_var, training, momentum, eps, relu)
#   y = torch.ops.vrnet.dwconv3x3(x, w)                              # depthwise 3x3, stride 1, pad 1, NHWC
#   y = torch.ops.vrnet.upsample_bilinear(x, scale)                  # align_corners=True, NHWC
def _rows(x):
    B, H, W, C = x.shape
    return B, H, W, C, B * H * W


@torch.library.custom_op("vrnet::mlp", mutates_args=(), device_types="cuda")
def mlp(x: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor, res: torch.Tensor,
        ls: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """x, res: (B,H,W,C) NHWC fp32; w1: (HID,C,1,1), w2: (C,HID,1,1) as in Mlp.state_dict(); ls: (C,) layer scale.
    Returns (res + ls * (gelu(x w1^T + b1) w2^T + b2), pre-activation (B,H,W,HID) for the backward pass)."""
    x, res = x.contiguous(), res.contiguous()
    B, H, W, C, M = _rows(x)
    hid = w1.shape[0]
    if not hip.mlp_fused_ok(C, hid, M):
        raise RuntimeError(f"vrnet::mlp: no fused kernel for C={C}, hidden={hid}, rows={M} (C in (64, 128), hidden % 32 == 0, rows % 32 == 0)")
    y, u = torch.empty_like(x), torch.empty((B, H, W, hid), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        fwd, _ = hip.mlp_pack(w1.contiguous(), w2.contiguous(), C, hid, 2, want_bwd=False)
        hip.mlp_fwd(x, C, fwd, b1, b2, res, C, ls, y, C, u, hid, None, M, C, hid, 2)
    return y, u


@mlp.register_fake
def _(x, w1, b1, w2, b2, res, ls):
    return torch.empty_like(x), x.new_empty((*x.shape[:3], w1.shape[0]))


@torch.library.custom_op("vrnet::mlp_backward", mutates_args=(), device_types="cuda")
def mlp_backward(g: torch.Tensor, x: torch.Tensor, u: torch.Tensor, w1: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor,
                 ls: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    g, x = g.contiguous(), x.contiguous()
    B, H, W, C, M = _rows(x)
    hid = w1.shape[0]
    h, du, dx = torch.empty_like(u), torch.empty_like(u), torch.empty_like(x)
    dw1, db1 = torch.empty_like(w1, memory_format=torch.contiguous_format), torch.empty(hid, device=x.device)
    dw2, db2, dls = torch.empty_like(w2, memory_format=torch.contiguous_format), torch.empty(C, device=x.device), torch.empty(C, device=x.device)
    with torch.cuda.device(x.device):
        _, bwd = hip.mlp_pack(w1.contiguous(), w2.contiguous(), C, hid, 2)
        hip.mlp_bwd(g, C, ls, bwd, u, hid, h, hid, du, hid, dx, C, M, C, hid, 2)
        hip.conv2d_wgrad(h, hid, g, C, dw2, db2, ls, B, H, W, hid, H, W, C, 1, 1, 1, 0, 1, precision=2, w=w2.contiguous(), bias=b2, dls=dls)
        hip.conv2d_wgrad(x, C, du, hid, dw1, db1, None, B, H, W, C, H, W, hid, 1, 1, 1, 0, 1, precision=2)
    return dx, dw1, db1, dw2, db2, dls


@mlp_backward.register_fake
def _(g, x, u, w1, w2, b2, ls):
    return torch.empty_like(x), torch.empty_like(w1), w1.new_empty((w1.shape[0],)), torch.empty_like(w2), torch.empty_like(ls), torch.empty_like(ls)


def _mlp_setup(ctx, inputs, output):
    x, w1, b1, w2, b2, res, ls = inputs
    ctx.save_for_backward(x, output[1], w1, w2, b2, ls)


def _mlp_bwd(ctx, g, g_u):
    x, u, w1, w2, b2, ls = ctx.saved_tensors
    dx, dw1, db1, dw2, db2, dls = torch.ops.vrnet.mlp_backward(g, x, u, w1, w2, b2, ls)
    return dx, dw1, db1, dw2, db2, g, dls


mlp.register_autograd(_mlp_bwd, setup_context=_mlp_setup)


@torch.library.custom_op("vrnet::group_norm1", mutates_args=(), device_types="cuda")
def group_norm1(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float) -> tuple[torch.Tensor, torch.Tensor]:
    """GroupNorm(1, C) of an NHWC tensor: statistics in fp64, torch's (x - mean) * rstd * gamma + beta order.
    Returns (y, (B,2) mean / rstd)."""
    x = x.contiguous()
    B, H, W, C, _ = _rows(x)
    A, D, S = (torch.empty((B, C), device=x.device) for _ in range(3))
    ms = torch.empty((B, 2), device=x.device)
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        hip.gn_stats_fwd(x, C, gamma, beta, eps, B, H * W, C, A, D, S, ms)
        hip.affine(y, C, B, H * W, C, x1=x, ld1=C, A=A, D1=D, S1=S, bstride=C)
    return y, ms


@group_norm1.register_fake
def _(x, gamma, beta, eps):
    return torch.empty_like(x), x.new_empty((x.shape[0], 2))


@torch.library.custom_op("vrnet::group_norm1_backward", mutates_args=(), device_types="cuda")
def group_norm1_backward(g: torch.Tensor, x: torch.Tensor, ms: torch.Tensor, gamma: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    g, x = g.contiguous(), x.contiguous()
    B, H, W, C, _ = _rows(x)
    dx, dg, db = torch.empty_like(x), torch.empty_like(gamma), torch.empty_like(gamma)
    A, E, D, S = (torch.empty((B, C), device=x.device) for _ in range(4))
    with torch.cuda.device(x.device):
        mom2 = hip.moments(g, C, B, H * W, C, x2=x, ldx2=C)
        hip.gn_coef_bwd(mom2, ms, gamma, B, H * W, C, A, E, D, S, dg, db, 0)
        hip.affine(dx, C, B, H * W, C, x1=g, ld1=C, A=A, x2=x, ld2=C, E=E, D2=D, S2=S, bstride=C)
    return dx, dg, db


@group_norm1_backward.register_fake
def _(g, x, ms, gamma):
    return torch.empty_like(x), torch.empty_like(gamma), torch.empty_like(gamma)


def _gn_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], output[1], inputs[1])


def _gn_bwd(ctx, g, g_ms):
    x, ms, gamma = ctx.saved_tensors
    dx, dg, db = torch.ops.vrnet.group_norm1_backward(g, x, ms, gamma)
    return dx, dg, db, None


group_norm1.register_autograd(_gn_bwd, setup_context=_gn_setup)


@torch.library.custom_op("vrnet::batch_norm_act", mutates_args=(), device_types="cuda")
def batch_norm_act(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, running_mean: torch.Tensor, running_var: torch.Tensor,
                   training: bool, momentum: float, eps: float, relu: bool) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """BatchNorm2d over NHWC (+ ReLU), BaseConv's tail (normal_conv.py:45-49): batch statistics in fp64; functional -- the
    updated running statistics (unbiased variance, as nn.BatchNorm2d) are RETURNED, the caller copies them into the
    module's buffers.  Returns (y, (C,2) mean / rstd, new running_mean, new running_var)."""
    x = x.contiguous()
    B, H, W, C, _ = _rows(x)
    A, D, S = (torch.empty(C, device=x.device) for _ in range(3))
    ms = torch.empty((C, 2), device=x.device)
    rm, rv = running_mean.clone(), running_var.clone()
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        if training:
            hip.bn_stats_fwd(x, C, gamma, beta, eps, momentum, rm, rv, None, B, H * W, C, A, D, S, ms)
        else:
            hip.bn_coef_fwd(None, gamma, beta, eps, momentum, rm, rv, None, False, B, H * W, C, A, D, S, ms)
        hip.affine(y, C, B, H * W, C, x1=x, ld1=C, A=A, D1=D, S1=S, pre=1 if relu else 0)
    return y, ms, rm, rv


@batch_norm_act.register_fake
def _(x, gamma, beta, running_mean, running_var, training, momentum, eps, relu):
    return torch.empty_like(x), x.new_empty((x.shape[-1], 2)), torch.empty_like(running_mean), torch.empty_like(running_var)


@torch.library.custom_op("vrnet::batch_norm_act_backward", mutates_args=(), device_types="cuda")
def batch_norm_act_backward(g: torch.Tensor, x: torch.Tensor, y: torch.Tensor, ms: torch.Tensor, gamma: torch.Tensor, training: bool,
                            relu: bool) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    g, x = g.contiguous(), x.contiguous()
    B, H, W, C, _ = _rows(x)
    dx, dg, db = torch.empty_like(x), torch.empty_like(gamma), torch.empty_like(gamma)
    A, E, D, S = (torch.empty(C, device=x.device) for _ in range(4))
    mask = y if relu else None
    with torch.cuda.device(x.device):
        hip.bn_stats_bwd(g, C, x, C, mask, C if relu else 0, ms, gamma, training, B, H * W, C, A, E, D, S, dg, db, 0)
        hip.affine(dx, C, B, H * W, C, x1=g, ld1=C, A=A, pre=2 if relu else 0, masky=mask, ldm=C if relu else 0, x2=x, ld2=C, E=E,
                   D2=D, S2=S)
    return dx, dg, db


@batch_norm_act_backward.register_fake
def _(g, x, y, ms, gamma, training, relu):
    return torch.empty_like(x), torch.empty_like(gamma), torch.empty_like(gamma)


def _bn_setup(ctx, inputs, output):
    x, gamma, beta, rm, rv, training, momentum, eps, relu = inputs
    ctx.save_for_backward(x, output[0], output[1], gamma)
    ctx.flags = (training, relu)


def _bn_bwd(ctx, g, g_ms, g_rm, g_rv):
    x, y, ms, gamma = ctx.saved_tensors
    dx, dg, db = torch.ops.vrnet.batch_norm_act_backward(g, x, y, ms, gamma, *ctx.flags)
    return dx, dg, db, None, None, None, None, None, None


batch_norm_act.register_autograd(_bn_bwd, setup_context=_bn_setup)


@torch.library.custom_op("vrnet::dwconv3x3", mutates_args=(), device_types="cuda")
def dwconv3x3(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """Depthwise 3x3, stride 1, pad 1 (DWConv.dconv, normal_conv.py:23-33); x (B,H,W,C) NHWC, w (C,1,3,3)."""
    x = x.contiguous()
    B, H, W, C, _ = _rows(x)
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        hip.dwconv3x3(x, C, w.contiguous(), y, C, B, H, W, C)
    return y


@dwconv3x3.register_fake
def _(x, w):
    return torch.empty_like(x)


@torch.library.custom_op("vrnet::dwconv3x3_backward", mutates_args=(), device_types="cuda")
def dwconv3x3_backward(g: torch.Tensor, x: torch.Tensor, w: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    g, x = g.contiguous(), x.contiguous()
    B, H, W, C, _ = _rows(x)
    dx, dw = torch.empty_like(x), torch.empty_like(w, memory_format=torch.contiguous_format)
    with torch.cuda.device(x.device):
        hip.dwconv3x3(g, C, w.contiguous(), dx, C, B, H, W, C, flip=1)
        hip.dwconv3x3_wgrad(x, C, g, C, dw, B, H, W, C)
    return dx, dw


@dwconv3x3_backward.register_fake
def _(g, x, w):
    return torch.empty_like(x), torch.empty_like(w)


dwconv3x3.register_autograd(lambda ctx, g: tuple(torch.ops.vrnet.dwconv3x3_backward(g, *ctx.saved_tensors)),
                            setup_context=lambda ctx, inputs, output: ctx.save_for_backward(*inputs))


@torch.library.custom_op("vrnet::upsample_bilinear", mutates_args=(), device_types="cuda")
def upsample_bilinear(x: torch.Tensor, scale: int) -> torch.Tensor:
    """nn.Upsample(scale_factor, mode='bilinear', align_corners=True) on NHWC (coc_fpn_dual.py:19-22)."""
    x = x.contiguous()
    B, H, W, C, _ = _rows(x)
    y = torch.empty((B, H * scale, W * scale, C), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        hip.upsample(x, C, y, C, B, H, W, C, scale)
    return y


@upsample_bilinear.register_fake
def _(x, scale):
    B, H, W, C = x.shape
    return x.new_empty((B, H * scale, W * scale, C))


@torch.library.custom_op("vrnet::upsample_bilinear_backward", mutates_args=(), device_types="cuda")
def upsample_bilinear_backward(g: torch.Tensor, scale: int) -> torch.Tensor:
    g = g.contiguous()
    B, HS, WS, C = g.shape
    dx = torch.empty((B, HS // scale, WS // scale, C), dtype=torch.float32, device=g.device)
    with torch.cuda.device(g.device):
        hip.upsample_bwd(g, C, 0, dx, C, B, HS // scale, WS // scale, C, scale)
    return dx


@upsample_bilinear_backward.register_fake
def _(g, scale):
    B, HS, WS, C = g.shape
    return g.new_empty((B, HS // scale, WS // scale, C))


def _up_setup(ctx, inputs, output):
    ctx.scale = inputs[1]


upsample_bilinear.register_autograd(lambda ctx, g: (torch.ops.vrnet.upsample_bilinear_backward(g, ctx.scale), None),
                                    setup_context=_up_setup)



@torch.library.custom_op("vrnet::cat_shuffle", mutates_args=(), device_types="cuda")
def cat_shuffle(a: torch.Tensor, b: torch.Tensor, interleave: bool) -> torch.Tensor:
    """torch.cat([a, b], channels) of two NHWC maps, followed by shuffle_channels(groups=2) when `interleave` (equal widths:
    channel 2 j = a_j, 2 j + 1 = b_j) -- vr_coc.py:70-80, coc_fpn_dual.py:120-130 -- in one launch."""
    a, b = a.contiguous(), b.contiguous()
    rows = a.numel() // a.shape[-1]
    out = a.new_empty(a.shape[:-1] + (a.shape[-1] + b.shape[-1],))
    with torch.cuda.device(a.device):
        hip.cat2(a, a.shape[-1], a.shape[-1], b, b.shape[-1], b.shape[-1], out, out.shape[-1], rows, interleave)
    return out


@cat_shuffle.register_fake
def _(a, b, interleave):
    return a.new_empty(a.shape[:-1] + (a.shape[-1] + b.shape[-1],))


@torch.library.custom_op("vrnet::cat_shuffle_backward", mutates_args=(), device_types="cuda")
def cat_shuffle_backward(g: torch.Tensor, ca: int, interleave: bool) -> tuple[torch.Tensor, torch.Tensor]:
    g = g.contiguous()
    ct = g.shape[-1]
    rows = g.numel() // ct
    ga, gb = g.new_empty(g.shape[:-1] + (ca,)), g.new_empty(g.shape[:-1] + (ct - ca,))
    with torch.cuda.device(g.device):
        hip.cat2(ga, ca, ca, gb, ct - ca, ct - ca, g, ct, rows, interleave, dir=1)
    return ga, gb


@cat_shuffle_backward.register_fake
def _(g, ca, interleave):
    return g.new_empty(g.shape[:-1] + (ca,)), g.new_empty(g.shape[:-1] + (g.shape[-1] - ca,))


def _cat_setup(ctx, inputs, output):
    ctx.ca, ctx.interleave = inputs[0].shape[-1], inputs[2]


cat_shuffle.register_autograd(lambda ctx, g: (*torch.ops.vrnet.cat_shuffle_backward(g, ctx.ca, ctx.interleave), None),
                              setup_context=_cat_setup)


@torch.library.custom_op("vrnet::batch_formats", mutates_args=(), device_types="cuda")
def batch_formats(images_u8: torch.Tensor, pngs_u8: torch.Tensor, num_classes_seg: int) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """A letterboxed batch from bytes (utils/dataloader.py:88-107, 440-457): images (B,H,W,3) uint8 -> (B,3,H,W) float32
    normalised as preprocess_input does; labels (B,H,W) uint8 -> (B,H,W) int64 with the ignore class and the
    (B,H,W,nc+1) float32 one-hot."""
    with torch.cuda.device(images_u8.device):
        return hip.batch_formats(images_u8.contiguous(), pngs_u8.contiguous(), num_classes_seg)


@batch_formats.register_fake
def _(images_u8, pngs_u8, num_classes_seg):
    B, H, W, _ = images_u8.shape
    return (images_u8.new_empty((B, 3, H, W), dtype=torch.float32), pngs_u8.new_empty((B, H, W), dtype=torch.int64),
            pngs_u8.new_empty((B, H, W, num_classes_seg + 1), dtype=torch.float32))


# ---- ShuffleAttention (backbone/attention_modules/shuffle_attention.py:48-72) --------------------------------------------
@torch.library.custom_op("vrnet::shuffle_attention", mutates_args=(), device_types="cuda")
def shuffle_attention(x: torch.Tensor, cweight: torch.Tensor, cbias: torch.Tensor, sweight: torch.Tensor, sbias: torch.Tensor,
                      gn_weight: torch.Tensor, gn_bias: torch.Tensor, groups: int) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """x: (B,H,W,C) NHWC fp32; the six parameters as in ShuffleAttention.state_dict() (any shape, C / (2 G) elements each).
    Returns (y, per-(sample, channel) moments, P, Q, Mn): the last four are what the backward pass re-uses."""
    x = x.contiguous()
    B, H, W, C = x.shape
    params = [t.contiguous().reshape(-1) for t in (cweight, cbias, sweight, sbias, gn_weight, gn_bias)]
    with torch.cuda.device(x.device):
        mom = hip.moments(x, C, B, H * W, C)
        P, Q, Mn = (torch.empty((B, C), dtype=torch.float32, device=x.device) for _ in range(3))
        hip.sa_coef_fwd(mom, *params, B, H * W, C, groups, P, Q, Mn)
        y = torch.empty_like(x)
        hip.sa_apply(x, C, P, Q, Mn, y, C, B, H * W, C)
    return y, mom, P, Q, Mn


@shuffle_attention.register_fake
def _(x, cweight, cbias, sweight, sbias, gn_weight, gn_bias, groups):
    B, _, _, C = x.shape
    return (torch.empty_like(x), x.new_empty((B, C, 2), dtype=torch.float64), x.new_empty((B, C)), x.new_empty((B, C)),
            x.new_empty((B, C)))


@torch.library.custom_op("vrnet::shuffle_attention_backward", mutates_args=(), device_types="cuda")
def shuffle_attention_backward(g: torch.Tensor, x: torch.Tensor, mom: torch.Tensor, P: torch.Tensor, Q: torch.Tensor, Mn: torch.Tensor,
                               cweight: torch.Tensor, cbias: torch.Tensor, sweight: torch.Tensor, sbias: torch.Tensor,
                               gn_weight: torch.Tensor, gn_bias: torch.Tensor, groups: int) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    g, x = g.contiguous(), x.contiguous()
    B, H, W, C = x.shape
    plist = (cweight, cbias, sweight, sbias, gn_weight, gn_bias)
    params = [t.contiguous().reshape(-1) for t in plist]
    grads = [torch.empty(t.numel(), dtype=torch.float32, device=x.device) for t in plist]
    dx = torch.empty_like(x)
    with torch.cuda.device(x.device):
        hip.sa_bwd(g, C, x, C, P, Q, Mn, mom, params, dx, C, grads, torch.empty((2, B, C), dtype=torch.float32, device=x.device),
                   B, H * W, C, groups, 0, 0)
    return (dx, *[gr.reshape(t.shape) for gr, t in zip(grads, plist)])


@shuffle_attention_backward.register_fake
def _(g, x, mom, P, Q, Mn, cweight, cbias, sweight, sbias, gn_weight, gn_bias, groups):
    return (torch.empty_like(x), *[torch.empty_like(t) for t in (cweight, cbias, sweight, sbias, gn_weight, gn_bias)])


def _sa_setup(ctx, inputs, output):
    x, *params, groups = inputs
    ctx.save_for_backward(x, *output[1:], *params)
    ctx.groups = groups


def _sa_bwd(ctx, g, *unused):
    x, mom, P, Q, Mn, *params = ctx.saved_tensors
    return (*torch.ops.vrnet.shuffle_attention_backward(g, x, mom, P, Q, Mn, *params, ctx.groups), None)


shuffle_attention.register_autograd(_sa_bwd, setup_context=_sa_setup)


# ---- ECA gate (backbone/attention_modules/eca.py:16-22) ------------------------------------------------------------------
@torch.library.custom_op("vrnet::eca", mutates_args=(), device_types="cuda")
def eca(x: torch.Tensor, weight: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """x: (B,H,W,C) NHWC fp32; weight: the Conv1d(1, 1, k) weight.  y = x * sigmoid(conv1d(mean over pixels)).
    Returns (y, gate (B,C), moments)."""
    x = x.contiguous()
    B, H, W, C = x.shape
    wk = weight.contiguous().reshape(-1)
    with torch.cuda.device(x.device):
        mom = hip.moments(x, C, B, H * W, C)
        gate = torch.empty((B, C), dtype=torch.float32, device=x.device)
        hip.eca_coef_fwd(mom, wk, wk.numel(), B, H * W, C, gate)
        y = torch.empty_like(x)
        hip.affine(y, C, B, H * W, C, x1=x, ld1=C, A=gate, bstride=C)
    return y, gate, mom


@eca.register_fake
def _(x, weight):
    B, _, _, C = x.shape
    return torch.empty_like(x), x.new_empty((B, C)), x.new_empty((B, C, 2), dtype=torch.float64)


@torch.library.custom_op("vrnet::eca_backward", mutates_args=(), device_types="cuda")
def eca_backward(g: torch.Tensor, x: torch.Tensor, gate: torch.Tensor, mom: torch.Tensor, weight: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    g, x = g.contiguous(), x.contiguous()
    B, H, W, C = x.shape
    wk = weight.contiguous().reshape(-1)
    with torch.cuda.device(x.device):
        mom2 = hip.moments(g, C, B, H * W, C, x2=x, ldx2=C)
        Fc = torch.empty((B, C), dtype=torch.float32, device=x.device)
        dwk = torch.empty(wk.numel(), dtype=torch.float32, device=x.device)
        hip.eca_coef_bwd(mom2, mom, gate, wk, wk.numel(), B, H * W, C, Fc, dwk, 0)
        dx = torch.empty_like(x)
        hip.affine(dx, C, B, H * W, C, x1=g, ld1=C, A=gate, D2=Fc, bstride=C)
    return dx, dwk.reshape(weight.shape)


@eca_backward.register_fake
def _(g, x, gate, mom, weight):
    return torch.empty_like(x), torch.empty_like(weight)


def _eca_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], output[1], output[2], inputs[1])


def _eca_bwd(ctx, g, *unused):
    x, gate, mom, weight = ctx.saved_tensors
    return torch.ops.vrnet.eca_backward(g, x, gate, mom, weight)


eca.register_autograd(_eca_bwd, setup_context=_eca_setup)


# ---- the gain of ImageEnhanceByRadar (vr_coc.py:59-67, 314): (1 + data_normal(p)) * x -----------------------------------
@torch.library.custom_op("vrnet::image_enhance", mutates_args=(), device_types="cuda")
def image_enhance(p: torch.Tensor, x: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """p: the projected radar map ReLU(BN(conv3x3(radar))), x: the image map, both (B,H,W,C) NHWC fp32.  data_normal maps p to
    [0, 1] with the minimum / maximum over the WHOLE batch tensor.  As in the reference (vr_coc.py:59-67), a constant map
    (max == min, e.g. every ReLU output zero) divides 0 by 0: the kernels compute (p - min) / (max - min) unguarded and the
    result is NaN, exactly the reference's behaviour for such a degenerate batch (no guard is added on either side).
    Returns (t = (1 + data_normal(p)) * x, (min, max))."""
    p, x = p.contiguous(), x.contiguous()
    with torch.cuda.device(x.device):
        mm = torch.empty(2, dtype=torch.float32, device=x.device)
        hip.minmax(p, p.numel(), mm)
        t = torch.empty_like(x)
        hip.enhance_mul(p, x, mm, t, p.numel())
    return t, mm


@image_enhance.register_fake
def _(p, x):
    return torch.empty_like(x), x.new_empty((2,))


@torch.library.custom_op("vrnet::image_enhance_backward", mutates_args=(), device_types="cuda")
def image_enhance_backward(g: torch.Tensor, p: torch.Tensor, x: torch.Tensor, mm: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    g, p, x = g.contiguous(), p.contiguous(), x.contiguous()
    dx, dp = torch.empty_like(x), torch.empty_like(p)
    with torch.cuda.device(x.device):
        hip.enhance_bwd(g, x, p, mm, dx, dp, p.numel())
    return dp, dx


@image_enhance_backward.register_fake
def _(g, p, x, mm):
    return torch.empty_like(p), torch.empty_like(x)


def _ie_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], inputs[1], output[1])


image_enhance.register_autograd(lambda ctx, g, g_mm: torch.ops.vrnet.image_enhance_backward(g, *ctx.saved_tensors),
                                setup_context=_ie_setup)


# ---- the gate of RadarEnhanceByImage (vr_coc.py:331-359): eca(shuffle(cat(image features, radar features))) -------------
@torch.library.custom_op("vrnet::radar_enhance", mutates_args=(), device_types="cuda")
def radar_enhance(a: torch.Tensor, r: torch.Tensor, weight: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """a: image features (after ShuffleAttention, or the raw image at the input stage), r: radar features, NHWC fp32; weight: the
    ECA Conv1d weight.  u = eca(shuffle_channels(cat([a, r], C), 2)) -- the channel-wise gate in front of the inverse
    projection -- with the concat + shuffle as ONE strided launch and the gate as moments + coefficient + scale.
    Returns (u, gate, moments)."""
    cat = torch.ops.vrnet.cat_shuffle(a, r, _re_interleave(a, r))
    return torch.ops.vrnet.eca(cat, weight)


def _re_interleave(a, r):
    """Whether shuffle_channels(cat, 2) (vr_coc.py:70-80) interleaves: it does whenever Ca + Cb is even, and is the identity
    for an odd sum (the 3 + 4 input level).  The kernel behind cat_shuffle interleaves two halves of EQUAL width -- every
    site of the model; unequal widths with an even sum would need the general half-interleave of the concatenation (the
    halves then straddle the two sources): refused rather than returned in another channel order."""
    ca, cb = a.shape[-1], r.shape[-1]
    if ca != cb and (ca + cb) % 2 == 0:
        raise RuntimeError(f"vrnet::radar_enhance: widths {ca} + {cb} (unequal, even sum) need the general 2-group shuffle of "
                           "the concatenation, which this op does not implement (the model only has equal widths and 3 + 4)")
    return ca == cb


@radar_enhance.register_fake
def _(a, r, weight):
    B, Ct = a.shape[0], a.shape[-1] + r.shape[-1]
    return a.new_empty(a.shape[:-1] + (Ct,)), a.new_empty((B, Ct)), a.new_empty((B, Ct, 2), dtype=torch.float64)


@torch.library.custom_op("vrnet::radar_enhance_backward", mutates_args=(), device_types="cuda")
def radar_enhance_backward(g: torch.Tensor, a: torch.Tensor, r: torch.Tensor, gate: torch.Tensor, mom: torch.Tensor,
                           weight: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    il = _re_interleave(a, r)
    cat = torch.ops.vrnet.cat_shuffle(a, r, il)                      # recomputed: one strided copy instead of a stored tensor
    dcat, dw = torch.ops.vrnet.eca_backward(g, cat, gate, mom, weight)
    da, dr = torch.ops.vrnet.cat_shuffle_backward(dcat, a.shape[-1], il)
    return da, dr, dw


@radar_enhance_backward.register_fake
def _(g, a, r, gate, mom, weight):
    return torch.empty_like(a), torch.empty_like(r), torch.empty_like(weight)


def _re_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], inputs[1], output[1], output[2], inputs[2])


def _re_bwd(ctx, g, *unused):
    a, r, gate, mom, weight = ctx.saved_tensors
    return torch.ops.vrnet.radar_enhance_backward(g, a, r, gate, mom, weight)


radar_enhance.register_autograd(_re_bwd, setup_context=_re_setup)


# Autocast policy (the reference trains under torch.cuda.amp.autocast, utils/utils_fit.py:86-88): these ops compute in fp32
# whatever the autocast dtype, i.e. floating-point arguments are cast to fp32 on the way in.
for _op in ("cluster", "conv2d_nhwc", "mlp", "group_norm1", "batch_norm_act", "dwconv3x3", "upsample_bilinear", "cat_shuffle",
            "shuffle_attention", "eca", "image_enhance", "radar_enhance"):
    torch.library.register_autocast(f"vrnet::{_op}", "cuda", torch.float32)
