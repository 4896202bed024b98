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
