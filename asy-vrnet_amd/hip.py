"""ctypes binding of the C-ABI in include/vrnet_hip.h (libvrnet_hip.so, gfx950).

PyTorch is used for device memory and streams only: every function below passes raw device
pointers, sizes and the current HIP stream to the library.  There is no fallback: a missing
library raises at import of this module, a failing call raises RuntimeError with the
library's message (mirroring the reference's assert at backbone/fusion/vr_coc.py:163-164).
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VRNET_HIP_LIB") or os.path.join(_HERE, "csrc", "libvrnet_hip.so")   # override: diagnostic builds only
ABI_VERSION = 10

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build the HIP library first (python -c 'import __graft_entry__ as g; g.build()' "
        "or make -C asy-vrnet_amd/csrc). The hot path has no CPU/eager fallback.")
_lib = ctypes.CDLL(LIB_PATH)

P, L, I, F, D = ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_float, ctypes.c_double
_SIGS = {
    "vrnet_abi_version": ([], I),
    "vrnet_last_error": ([], ctypes.c_char_p),
    "vrnet_last_kernel": ([], I),
    "vrnet_tuning_build": ([], I),
    "vrnet_kernel_launches": ([I], L),
    "vrnet_device_arch": ([ctypes.c_char_p, I], I),
    "vrnet_conv2d_f32": ([P, L, P, P, P, L] + [I] * 14 + [P, L, P, L, P, P, P, L, I, I, I, I, P, I, I, P, P, P, P, P, P, P, L, P], I),
    "vrnet_conv2d_dma_plan": ([L, I, L, P], I),
    "vrnet_conv2d_splitk_workspace": ([L, I, L], L),
    "vrnet_conv_planes_bytes": ([I, I], L),
    "vrnet_conv_planes_pack_f32": ([P, I, L, P], I),
    "vrnet_pack_weight_t_f32": ([P, P, P, I, I, I, I, P], I),
    "vrnet_gemm_planes_ok": ([L, I, I], I),
    "vrnet_gemm_planes_f32": ([P, L, L, P, L, L, I, L, I, I, P, P, L, P, L, L, I, I, P, L, P, L, P, P, L, I, P, L, P, I, P], I),
    "vrnet_planes_split_blocks": ([L, L], L),
    "vrnet_planes_split_f32": ([P, I, L, I, P], I),
    "vrnet_planes_from_f32": ([P, L, L, L, P, L, L, I, P], I),
    "vrnet_wgrad_planes_ok": ([L, I, I], I),
    "vrnet_wgrad_planes_workspace": ([L, I, I], L),
    "vrnet_wgrad_planes_f32": ([P, L, L, P, L, L, I, L, I, I, P, P, P, I, P, P, P, P, L, P], I),
    "vrnet_conv2d_dma_tile": ([L, I], I),
    "vrnet_conv2d_wgrad_workspace": ([I] * 8, L),
    "vrnet_conv2d_wgrad_f32": ([P, L, P, L, P, P, P] + [I] * 14 + [P, P, P, P, P, P, P, P, P, P, L, P], I),
    "vrnet_pack_weight_f32": ([P, P, I, I, I, I, P], I),
    "vrnet_mlp_fused_ok": ([I, I, L], I),
    "vrnet_mlp_rc_ok": ([I, I, L], I),
    "vrnet_mlp_pack_bytes": ([I, I, I], L),
    "vrnet_mlp_pack_f32": ([P, P, I, I, I, P, P, P], I),
    "vrnet_mlp_fwd_f32": ([P, L, P, P, P, P, L, P, P, L, P, L, P, L, I, I, I, P], I),
    "vrnet_mlp_pack_rc_bytes": ([I, I, I], L),
    "vrnet_mlp_pack_rc_f32": ([P, P, I, I, I, P, P], I),
    "vrnet_mlp_bwd_rc_f32": ([P, L, P, P, P, L, P, P, L, P, L, P, L, L, I, I, I, P], I),
    "vrnet_mlp_bwd_f32": ([P, L, P, P, P, L, P, L, P, L, P, L, L, I, I, I, P], I),
    "vrnet_moments_workspace": ([I, L, I], L),
    "vrnet_moments_f32": ([P, L, P, L, P, L, I, L, I, P, P, L, P], I),
    "vrnet_affine_f32": ([P, L, P, P, P, I, P, L, P, L, P, P, P, L, P, L, I, L, I, I, P, L, P], I),
    "vrnet_gn_coef_fwd": ([P, P, P, F, I, L, I, P, P, P, P, P], I),
    "vrnet_gn_coef_from_pairs": ([P, L, P, P, F, I, L, I, P, P, P, P, P, P, P], I),
    "vrnet_gn_stats_fwd": ([P, L, P, P, F, I, L, I, P, P, P, P, P, P, P, L, P], I),
    "vrnet_gn_coef_bwd": ([P, P, P, I, L, I, P, P, P, P, P, P, I, P, P, P, P], I),
    "vrnet_gn_apply_fwd": ([P, L, P, L, P, P, F, I, L, I, P, L, P, P], I),
    "vrnet_gn_apply_fwd_planes": ([P, L, P, L, P, P, F, I, L, I, P, L, P, P, P], I),
    "vrnet_gn_apply_bwd_planes": ([P, L, P, L, P, P, I, L, I, P, L, P, L, P, P, I, P, P, L, P], I),
    "vrnet_cluster_fwd_planes_f32": ([P, P, L, I, P, P, P, L, P, P, I, I, I, I, I, I, I, P, P, P], I),
    "vrnet_cluster_state_floats": ([I, I, I, I, I], L),
    "vrnet_cluster_bwd_planes_f32": ([P, P, L, I, P, P, P, P, L, P, P, L, P, P, I, I, I, I, I, I, I, P, P, P, P, L, P], I),
    "vrnet_gn_bwd_workspace": ([I, L, I], L),
    "vrnet_gn_apply_bwd_from_partials": ([P, L, P, L, P, P, P, P, I, L, I, P, L, P, L, P, P, I, P], I),
    "vrnet_bn_coef_fwd_from_partials": ([P, P, P, F, F, P, P, P, I, L, I, P, P, P, P, P], I),
    "vrnet_gn_apply_bwd": ([P, L, P, L, P, P, I, L, I, P, L, P, L, P, P, I, P, L, P], I),
    "vrnet_bn_coef_fwd": ([P, P, P, F, F, P, P, P, I, I, L, I, P, P, P, P, P], I),
    "vrnet_bn_coef_bwd": ([P, P, P, I, I, L, I, P, P, P, P, P, P, I, P], I),
    "vrnet_bn_stats_fwd": ([P, L, P, P, F, F, P, P, P, I, L, I, P, P, P, P, P, L, P], I),
    "vrnet_bn_stats_bwd_zmask": ([P, L, P, L, P, P, P, P, P, I, I, L, I, P, P, P, P, P, P, I, P, L, P], I),
    "vrnet_bn_apply_bwd_zmask": ([P, L, P, L, P, P, P, P, P, P, P, P, L, I, L, I, P], I),
    "vrnet_bn_stats_bwd": ([P, L, P, L, P, L, P, P, I, I, L, I, P, P, P, P, P, P, I, P, L, P], I),
    "vrnet_eca_coef_fwd": ([P, P, I, I, L, I, P, P], I),
    "vrnet_eca_coef_bwd": ([P, P, P, P, I, I, L, I, P, P, I, P], I),
    "vrnet_ls_coef_bwd": ([P, P, I, I, P, P, I, I, P, P, P, P], I),
    "vrnet_moments_to_float": ([P, P, L, D, I, P], I),
    "vrnet_copy_channels_f32": ([P, L, I, P, L, I, L, I, I, P], I),
    "vrnet_cat2_f32": ([P, L, I, P, L, I, P, L, L, I, I, I, I, P], I),
    "vrnet_patch_gather_f32": ([P, L, P, P, I, I, I, I, I, I, P], I),
    "vrnet_patch_scatter_f32": ([P, P, L, I, I, I, I, I, I, I, P], I),
    "vrnet_weight_ohwi_f32": ([P, P, I, I, I, I, I, I, P], I),
    "vrnet_nchw_to_nhwc_f32": ([P, P, L, I, I, L, P], I),
    "vrnet_nhwc_to_nchw_f32": ([P, L, P, I, I, L, I, P], I),
    "vrnet_add_f32": ([P, P, L, P], I),
    "vrnet_fill_f32": ([P, F, L, P], I),
    "vrnet_cluster_fwd_f32": ([P, P, L, P, P, P, L, P, P, I, I, I, I, I, I, P, P, P], I),
    "vrnet_cluster_fwd_forced_f32": ([P, P, L, P, P, P, L, P, P, I, I, I, I, I, I, P, P, P], I),
    "vrnet_cluster_bwd_workspace": ([I, I, I], L),
    "vrnet_cluster_bwd_workspace2": ([I, I, I, I, I], L),
    "vrnet_cluster_bwd_f32": ([P, P, L, P, P, P, P, L, P, P, L, P, P, I, I, I, I, I, I, I, P, P, P, P, P, L, P], I),
    "vrnet_cluster_ab_reduce_multi": ([I, P, P, P, P, P, P], I),
    "vrnet_dwconv3x3_f32": ([P, L, P, P, L, I, I, I, I, I, I, P], I),
    "vrnet_dwconv3x3_wgrad_workspace": ([I, I, I, I], L),
    "vrnet_dwconv3x3_wgrad_f32": ([P, L, P, L, P, I, I, I, I, I, P, L, P], I),
    "vrnet_upsample_bilinear_f32": ([P, L, P, L, I, I, I, I, I, I, P], I),
    "vrnet_bn_relu_upsample_bilinear_f32": ([P, L, P, P, P, P, L, I, I, I, I, I, I, P], I),
    "vrnet_upsample_bilinear_bwd_f32": ([P, L, I, P, L, I, I, I, I, I, I, P], I),
    "vrnet_reduce_workspace": ([], L),
    "vrnet_minmax_f32": ([P, L, P, P, L, P], I),
    "vrnet_enhance_mul_f32": ([P, P, P, P, L, P], I),
    "vrnet_enhance_fwd_f32": ([P, P, P, P, L, P, L, P], I),
    "vrnet_fusion_chunks": ([L, I], I),
    "vrnet_fusion_fold_chunks": ([L, I], I),
    "vrnet_bn_relu_minmax_f32": ([P, P, P, P, P, L, I, P, P], I),
    "vrnet_bn_relu_res_stats_f32": ([P, P, P, P, P, P, L, I, P, P], I),
    "vrnet_enhance_stats_f32": ([P, P, P, I, P, P, L, I, P, P], I),
    "vrnet_bn_bwd_enhance_f32": ([P, P, P, P, P, P, P, P, P, P, L, I, P, P], I),
    "vrnet_enhance_bwd_stats_f32": ([P, P, P, P, P, I, P, P, P, P, P, P, L, I, I, P, P], I),
    "vrnet_bn_bwd_next_stats_f32": ([P, P, P, P, P, P, P, P, P, P, P, L, I, P, P], I),
    "vrnet_bn_coef_fwd_from_chunks": ([P, I, L, P, P, F, F, P, P, P, I, P, P, P, P, P], I),
    "vrnet_bn_coef_bwd_from_chunks": ([P, I, L, P, P, I, I, P, P, P, P, P, P, I, P], I),
    "vrnet_enhance_bwd_f32": ([P, P, P, P, P, P, L, I, P, L, P], I),
    "vrnet_sa_coef_fwd": ([P] * 7 + [I, L, I, I, P, P, P, P], I),
    "vrnet_sa_cat_sums_f32": ([P, L, P, P, P, P, L, P, L, I, L, I, P, P, L, P], I),
    "vrnet_sa_apply_f32": ([P, L, P, P, P, P, L, I, L, I, P], I),
    "vrnet_sa_bwd_workspace": ([I, L, I], L),
    "vrnet_decode_outputs_f32": ([P, P, P, I, I, I, F, F, P, P], I),
    "vrnet_batch_formats_u8": ([P, P, I, I, I, I, P, P, P, P], I),
    "vrnet_yolo_loss_workspace": ([I, L, I, I], L),
    "vrnet_yolo_loss_f32": ([P, P, P, P, P, I, I, I, P, P, I, F, P, P, P, P, P, L, P], I),
    "vrnet_seg_loss_workspace": ([I, I, L], L),
    "vrnet_seg_loss_f32": ([P, P, P, P, I, I, L, I, I, F, F, F, F, F, P, P, P, L, P], I),
    "vrnet_mean_square_workspace": ([I, P], L),
    "vrnet_mean_square_f32": ([I, P, P, P, P, L, P], I),
    "vrnet_mean_square_bwd_f32": ([I, P, P, P, P, P], I),
    "vrnet_mt_sgd_f32": ([P, P, P, P, P, I, I, I, F, F, I, I, P], I),
    "vrnet_mt_adam_f32": ([P, P, P, P, P, I, I, I, F, F, F, F, I, P], I),
    "vrnet_mt_ema_f32": ([P, P, P, P, I, I, I, F, P], I),
    "vrnet_clock_stamp": ([P, P], I),
    "vrnet_mt_copy_f32": ([P, P, P, P, I, I, I, P], I),
    "vrnet_sa_bwd_f32": ([P, L, P, L, P, P, P, P] + [P] * 6 + [P, L] + [P] * 6 + [P, I, L, I, I, I, I, P, L, P], I),
}
for _name, (_args, _res) in _SIGS.items():
    _fn = getattr(_lib, _name)          # AttributeError here = header/library mismatch
    _fn.argtypes = _args
    _fn.restype = _res

if _lib.vrnet_abi_version() != ABI_VERSION:
    raise ImportError(f"libvrnet_hip.so ABI {_lib.vrnet_abi_version()} != expected {ABI_VERSION}")

EXPORTED = tuple(_SIGS)


def _check(rc, name):
    if rc != 0:
        raise RuntimeError(f"{name}: {_lib.vrnet_last_error().decode()}")


class ConvColStats(ctypes.Structure):
    """vrnet_conv_colstats (include/vrnet_hip.h): optional column statistics of a conv's stored outputs."""
    _fields_ = [("partial", P), ("x2", P), ("ldx2", L), ("gamma", P), ("tile_totals", P)]


class PlanesOut(ctypes.Structure):
    """vrnet_planes_out (include/vrnet_hip.h): optional bf16-plane output of a producing kernel."""
    _fields_ = [("p", P), ("ld", L), ("plane", L), ("np", I)]


def _planes_out(pl):
    return None if pl is None else ctypes.byref(PlanesOut(ptr(pl.t), pl.ld, pl.plane, pl.np))


_DTYPES = frozenset((torch.float32, torch.float64, torch.uint8, torch.int64, torch.int32, torch.bfloat16))


def tuning_build():
    """True when the loaded library is the diagnostic build (environment knobs / launch-skipping ablations compiled in)."""
    return bool(_lib.vrnet_tuning_build())


def kernel_launches(family):
    """Launches of a kernel family (codes of last_kernel) by this thread so far."""
    return _lib.vrnet_kernel_launches(family)


def last_kernel():
    """Kernel family of this thread's last conv2d / conv2d_wgrad call (include/vrnet_hip.h)."""
    return _lib.vrnet_last_kernel()


def ptr(t):
    """Device address of a tensor argument.  Every wrapper below passes its tensors through here, so a CPU tensor, an
    unsupported dtype or a view whose innermost stride is not 1 (the kernels take a ROW stride, never an element
    stride) raises instead of handing a meaningless pointer to a kernel."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("vrnet HIP path needs tensors on a HIP device (there is no CPU fallback)")
    if t.dtype not in _DTYPES:
        raise RuntimeError(f"vrnet HIP path: unsupported dtype {t.dtype}")
    if t.dim() and t.stride(-1) != 1 and t.shape[-1] != 1:
        raise RuntimeError(f"vrnet HIP path: innermost dimension must be contiguous (shape {tuple(t.shape)}, "
                           f"strides {t.stride()})")
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def empty(*shape, dtype=torch.float32, like=None, device=None):
    return torch.empty(shape, dtype=dtype, device=device if device is not None else like.device)


class Workspace:
    """Grow-only scratch arena (bytes) per (device, stream): kernels on forked streams run concurrently and
    must not share scratch.  Outgrown arenas are retired, never freed: a captured HIP graph may still
    reference them."""

    def __init__(self):
        self.buf = {}
        self.retired = []

    def get(self, nbytes, device):
        key = (device, torch.cuda.current_stream(device).cuda_stream)
        b = self.buf.get(key)
        if b is None or b.numel() < nbytes:
            if b is not None:
                self.retired.append(b)
            b = torch.empty(max(int(nbytes) * 2, 8 << 20), dtype=torch.uint8, device=device)
            self.buf[key] = b
        return b


_ws = Workspace()


# --------------------------------------------------------------------------------------- wrappers
def conv2d(a, lda, w, bias, y, ldy, B, H, W, Cin, OH, OW, Cout, kh, kw, stride, pad, dil, mode=0, act=0,
           ypre=None, ldypre=0, res=None, ldres=0, res_scale=None, kscale=None, aux=None, ldaux=0,
           out_nchw=0, out_ctot=0, out_coff=0, accumulate=0, stats=None, precision=0, pair_rows=0, w2=None, bias2=None,
           res_scale2=None, kscale2=None, colstats=None, w_planes=None):
    """pair_rows > 0: two-stream launch, GEMM rows >= pair_rows use (w2, bias2, res_scale2, kscale2).
    colstats = (partial, x2, ldx2, gamma, tile_totals) (None entries allowed): column statistics of the stored outputs."""
    cs = None
    if colstats is not None:
        part, x2, ldx2, gam, tot = colstats
        cs = ctypes.byref(ConvColStats(ptr(part), ptr(x2), ldx2, ptr(gam), ptr(tot)))
    # split contraction of the small maps (precision 2): slabs of raw accumulators in this stream's scratch arena
    ws, wsb = None, 0
    if precision == 2:
        rows, cols, ktot = (B * OH * OW, Cout, Cin * kh * kw) if mode == 0 else (B * H * W, Cin, Cout * kh * kw)
        wsb = _lib.vrnet_conv2d_splitk_workspace(rows, cols, ktot)
        if wsb:
            ws = _ws.get(wsb, a.device)
    _check(_lib.vrnet_conv2d_f32(ptr(a), lda, ptr(w), ptr(bias), ptr(y), ldy, B, H, W, Cin, OH, OW, Cout, kh, kw,
                                 stride, pad, dil, mode, act, ptr(ypre), ldypre, ptr(res), ldres, ptr(res_scale),
                                 ptr(kscale), ptr(aux), ldaux, out_nchw, out_ctot, out_coff, accumulate, ptr(stats),
                                 precision, pair_rows, ptr(w2), ptr(bias2), ptr(res_scale2), ptr(kscale2), ptr(w_planes), cs, ptr(ws), wsb, stream()),
           "conv2d")


def conv_planes_bytes(J, K):
    return _lib.vrnet_conv_planes_bytes(J, K)


def conv_planes_pack(table, nentries, total_blocks):
    """table: int64 device tensor, 8 values per entry (vrnet_conv_planes_pack_f32)."""
    _check(_lib.vrnet_conv_planes_pack_f32(ptr(table), nentries, total_blocks, stream()), "conv_planes_pack")


# ---- plane tensors (csrc/pgemm.hip): an fp32 tensor as three bf16 planes, t = p0 + p1 + p2 exactly, or (np = 1) a bf16 tensor
class Planes:
    """bf16 planes of a (rows, C) matrix: `t` is a (np, ..., C) bfloat16 tensor, plane q = t[q]; `ld` row stride, `plane`
    plane stride (elements).  np = 3: the fp32 values, exactly (float(t).sum(0) in the order p2 + p1 + p0 reproduces them);
    np = 1: the values rounded to bf16."""
    __slots__ = ("t", "np", "ld", "plane", "C")

    def __init__(self, t, ld=None, plane=None):
        assert t.dtype == torch.bfloat16 and t.stride(-1) == 1
        self.t, self.np, self.C = t, t.shape[0], t.shape[-1]
        self.ld = t.stride(-2) if ld is None else ld
        self.plane = t.stride(0) if plane is None else plane

    @staticmethod
    def empty(np_, shape, device):
        return Planes(torch.empty((np_,) + tuple(shape), dtype=torch.bfloat16, device=device))

    def float(self):
        """The fp32 tensor the planes stand for (exact for np = 3)."""
        f = self.t.float()
        return f[0] if self.np == 1 else (f[2] + f[1]) + f[0]


def gemm_planes_ok(rows, cols, K):
    """Whether gemm_planes has a kernel for a rows x cols product over a contraction of K."""
    return bool(_lib.vrnet_gemm_planes_ok(rows, cols, K))


def gemm_planes(a, b, M, N, K, bias=None, y=None, ldy=0, yp=None, act=0, ypre=None, ldypre=0, res=None, ldres=0,
                res_scale=None, aux=None, ldaux=0, accumulate=0, stats=None, stats_hw=0, colstats=None):
    """y / yp = epilogue(A . B^T) on plane operands a, b (Planes, same np); y: fp32 tensor or None, yp: Planes or None.
    ypre / aux may be bfloat16 tensors (the Mlp's pre-activation in bf16 mode; strides in elements)."""
    assert a.np == b.np
    half_side = (1 if (ypre is not None and ypre.dtype == torch.bfloat16) else 0) | \
        (2 if (aux is not None and aux.dtype == torch.bfloat16) else 0)
    cs = None
    if colstats is not None:
        part, x2, ldx2, gam, tot = colstats
        cs = ctypes.byref(ConvColStats(ptr(part), ptr(x2), ldx2, ptr(gam), ptr(tot)))
    _check(_lib.vrnet_gemm_planes_f32(ptr(a.t), a.ld, a.plane, ptr(b.t), b.ld, b.plane, a.np, M, N, K, ptr(bias), ptr(y), ldy,
                                      None if yp is None else ptr(yp.t), 0 if yp is None else yp.ld,
                                      0 if yp is None else yp.plane, 0 if yp is None else yp.np, act, ptr(ypre), ldypre,
                                      ptr(res), ldres, ptr(res_scale), ptr(aux), ldaux, accumulate, ptr(stats), stats_hw, cs,
                                      half_side, stream()), "gemm_planes")


def planes_split_blocks(R, K):
    return _lib.vrnet_planes_split_blocks(R, K)


def planes_split(table, nentries, total_blocks, np_):
    """table: int64 device tensor, 10 values per entry (vrnet_planes_split_f32)."""
    _check(_lib.vrnet_planes_split_f32(ptr(table), nentries, total_blocks, np_, stream()), "planes_split")


def planes_from_f32(src, lds, R, K, out):
    """out (Planes) = the row-major fp32 matrix `src` (R rows of K values, row stride lds)."""
    _check(_lib.vrnet_planes_from_f32(ptr(src), lds, R, K, ptr(out.t), out.ld, out.plane, out.np, stream()), "planes_from_f32")


def wgrad_planes_ok(M, Cin, Cout):
    return bool(_lib.vrnet_wgrad_planes_ok(M, Cin, Cout))


def wgrad_planes(x, dy, M, Cin, Cout, dw, dbias=None, row_scale=None, accumulate=0, w=None, bias=None, dls=None):
    """dw (+ dbias, + dls) of a 1x1 conv from plane operands x (M x Cin) and dy (M x Cout) (vrnet_wgrad_planes_f32)."""
    assert x.np == dy.np
    ws = _ws.get(_lib.vrnet_wgrad_planes_workspace(M, Cin, Cout), dw.device)
    _check(_lib.vrnet_wgrad_planes_f32(ptr(x.t), x.ld, x.plane, ptr(dy.t), dy.ld, dy.plane, x.np, M, Cin, Cout, ptr(dw), ptr(dbias),
                                       ptr(row_scale), accumulate, ptr(w), ptr(bias), ptr(dls), ptr(ws), ws.numel(), stream()),
           "wgrad_planes")


def bf16_conv_ok(lda, Cin, Cout, mode):
    """Shapes the bf16-operand path of conv2d accepts (mirrors the check in vrnet_conv2d_f32)."""
    ck, cn = (Cin, Cout) if mode == 0 else (Cout, Cin)
    return ck % 4 == 0 and lda % 4 == 0 and cn > 32 and cn % 4 == 0


def conv2d_dma_tile(rows, cols):
    """22 / 21 / 0: tile of the LDS-DMA x6 / bf16 kernels for a GEMM of rows x cols (0 = no such kernel)."""
    return _lib.vrnet_conv2d_dma_tile(rows, cols)


def conv2d_dma_plan(rows, cols, ktot):
    """(tile, splits) of conv2d at precision 2 for a GEMM of rows x cols with a contraction of ktot: the unsplit tile with
    splits = 1, or tile 21 with the K loop split over `splits` workgroups per tile (small maps), or (0, 1)."""
    s = ctypes.c_int(1)
    t = _lib.vrnet_conv2d_dma_plan(rows, cols, ktot, ctypes.byref(s))
    return t, s.value


def pack_weight_t(w_oihw, kscale, out, Cout, Cin, kh, kw):
    _check(_lib.vrnet_pack_weight_t_f32(ptr(w_oihw), ptr(kscale), ptr(out), Cout, Cin, kh, kw, stream()), "pack_weight_t")


def conv_stats_buffer(B, HW, Cout, device):
    """(buffer, pairs per sample) for the `stats` output of conv2d, or (None, 0) when the shape does not qualify."""
    if HW % 32 or Cout <= 32 or Cout % 4:
        return None, 0
    nb = (Cout + 31) // 32
    return torch.empty((B * HW // 32, nb, 2), dtype=torch.float64, device=device), (HW // 32) * nb


def conv2d_wgrad(x, ldx, dy, lddy, dw, dbias, row_scale, B, H, W, Cin, OH, OW, Cout, kh, kw, stride, pad, dil,
                 accumulate=0, precision=0, dw2=None, dbias2=None, row_scale2=None, w=None, bias=None, dls=None,
                 w2=None, bias2=None, dls2=None):
    """dw2 given: two-stream launch, samples [B/2, B) contribute to (dw2, dbias2, row_scale2).
    dls given (1x1 convs): layer-scale gradient rowdot(w, dw_raw) + bias * db_raw."""
    nbytes = _lib.vrnet_conv2d_wgrad_workspace(B, OH, OW, Cin, Cout, kh, kw, 1 if dw2 is not None else 0)
    ws = _ws.get(nbytes, x.device)
    _check(_lib.vrnet_conv2d_wgrad_f32(ptr(x), ldx, ptr(dy), lddy, ptr(dw), ptr(dbias), ptr(row_scale), B, H, W, Cin,
                                       OH, OW, Cout, kh, kw, stride, pad, dil, accumulate, precision, ptr(dw2),
                                       ptr(dbias2), ptr(row_scale2), ptr(w), ptr(bias), ptr(dls), ptr(w2), ptr(bias2),
                                       ptr(dls2), ptr(ws), ws.numel(), stream()), "conv2d_wgrad")


def bf16_wgrad_ok(ldx, lddy, Cin, Cout):
    return Cin % 4 == 0 and Cout % 4 == 0 and ldx % 4 == 0 and lddy % 4 == 0 and Cin > 32 and Cout > 32


def pack_weight(w_oihw, out, Cout, Cin, kh, kw):
    _check(_lib.vrnet_pack_weight_f32(ptr(w_oihw), ptr(out), Cout, Cin, kh, kw, stream()), "pack_weight")


def mlp_fused_ok(C, HID, M):
    """Whether the fused fc1 -> GELU -> fc2 kernels exist for block width C, hidden width HID and M rows."""
    return bool(_lib.vrnet_mlp_fused_ok(C, HID, M))


def mlp_pack(w1, w2, C, HID, precision, want_bwd=True):
    """(forward planes, backward planes) of an Mlp's weights: bf16 planes in MFMA fragment order (vrnet_mlp_pack_f32)."""
    n = _lib.vrnet_mlp_pack_bytes(C, HID, precision)
    fwd = torch.empty((n,), dtype=torch.uint8, device=w1.device)
    bwd = torch.empty((n,), dtype=torch.uint8, device=w1.device) if want_bwd else None
    _check(_lib.vrnet_mlp_pack_f32(ptr(w1), ptr(w2), C, HID, precision, ptr(fwd), ptr(bwd), stream()), "mlp_pack")
    return fwd, bwd


def mlp_rc_ok(C, HID, M=32):
    """Whether the recompute form of the fused Mlp backward (mlp_bwd_rc) exists for this block (the library's own predicate:
    vrnet_mlp_rc_ok, which vrnet_mlp_pack_rc_f32 and vrnet_mlp_bwd_rc_f32 check too)."""
    return bool(_lib.vrnet_mlp_rc_ok(C, HID, M))


def mlp_pack_rc(w1, w2, C, HID, precision):
    """Planes of the backward kernel that recomputes the pre-activation: per chunk [fc2^T | fc1^T | fc1]."""
    pack = torch.empty((_lib.vrnet_mlp_pack_rc_bytes(C, HID, precision),), dtype=torch.uint8, device=w1.device)
    _check(_lib.vrnet_mlp_pack_rc_f32(ptr(w1), ptr(w2), C, HID, precision, ptr(pack), stream()), "mlp_pack_rc")
    return pack


def mlp_bwd_rc(dy, lddy, dy_scale, pack, x, ldx, b1, h, ldh, du, lddu, dx, lddx, M, C, HID, precision):
    _check(_lib.vrnet_mlp_bwd_rc_f32(ptr(dy), lddy, ptr(dy_scale), ptr(pack), ptr(x), ldx, ptr(b1), ptr(h), ldh, ptr(du), lddu,
                                     ptr(dx), lddx, M, C, HID, precision, stream()), "mlp_bwd_rc")


def mlp_fwd(x, ldx, pack, b1, b2, res, ldres, res_scale, y, ldy, upre, ldu, stats, M, C, HID, precision):
    _check(_lib.vrnet_mlp_fwd_f32(ptr(x), ldx, ptr(pack), ptr(b1), ptr(b2), ptr(res), ldres, ptr(res_scale), ptr(y), ldy,
                                  ptr(upre), ldu, ptr(stats), M, C, HID, precision, stream()), "mlp_fwd")


def mlp_bwd(dy, lddy, dy_scale, pack, upre, ldu, h, ldh, du, lddu, dx, lddx, M, C, HID, precision):
    _check(_lib.vrnet_mlp_bwd_f32(ptr(dy), lddy, ptr(dy_scale), ptr(pack), ptr(upre), ldu, ptr(h), ldh, ptr(du), lddu,
                                  ptr(dx), lddx, M, C, HID, precision, stream()), "mlp_bwd")


def moments(x, ldx, B, HW, C, x2=None, ldx2=0, mask=None, ldm=0, out=None):
    if out is None:
        out = torch.empty((B, C, 2), dtype=torch.float64, device=x.device)
    ws = _ws.get(_lib.vrnet_moments_workspace(B, HW, C), x.device)
    _check(_lib.vrnet_moments_f32(ptr(x), ldx, ptr(x2), ldx2, ptr(mask), ldm, B, HW, C, ptr(out), ptr(ws), ws.numel(),
                                  stream()), "moments")
    return out


def affine(out, ldo, B, HW, C, x1=None, ld1=0, A=None, D1=None, pre=0, masky=None, ldm=0, x2=None, ld2=0, E=None,
           D2=None, bstride=0, accumulate=0, S1=None, S2=None, add=None, ldadd=0):
    """out = pre(A*(x1-S1)+D1) + E*(x2-S2) + D2 [+ out (accumulate=1) | + add (out of place)]."""
    _check(_lib.vrnet_affine_f32(ptr(x1), ld1, ptr(A), ptr(D1), ptr(S1), pre, ptr(masky), ldm, ptr(x2), ld2, ptr(E),
                                 ptr(D2), ptr(S2), bstride, ptr(out), ldo, B, HW, C, accumulate, ptr(add), ldadd, stream()),
           "affine")


def gn_coef_fwd(mom, gamma, beta, eps, B, HW, C, A, Dc, S, mean_rstd):
    _check(_lib.vrnet_gn_coef_fwd(ptr(mom), ptr(gamma), ptr(beta), eps, B, HW, C, ptr(A), ptr(Dc), ptr(S),
                                  ptr(mean_rstd), stream()), "gn_coef_fwd")


def gn_coef_from_pairs(pairs, per_sample, gamma, beta, eps, B, HW, C, A, Dc, S, mean_rstd, gamma2=None, beta2=None):
    _check(_lib.vrnet_gn_coef_from_pairs(ptr(pairs), per_sample, ptr(gamma), ptr(beta), eps, B, HW, C, ptr(A), ptr(Dc),
                                         ptr(S), ptr(mean_rstd), ptr(gamma2), ptr(beta2), stream()), "gn_coef_from_pairs")


def gn_stats_fwd(x, ldx, gamma, beta, eps, B, HW, C, A, Dc, S, mean_rstd, gamma2=None, beta2=None):
    ws = _ws.get(_lib.vrnet_moments_workspace(B, HW, C), x.device)
    _check(_lib.vrnet_gn_stats_fwd(ptr(x), ldx, ptr(gamma), ptr(beta), eps, B, HW, C, ptr(A), ptr(Dc), ptr(S),
                                   ptr(mean_rstd), ptr(gamma2), ptr(beta2), ptr(ws), ws.numel(), stream()), "gn_stats_fwd")


def gn_apply_ok(C, *lds):
    """Shapes the one-launch GroupNorm kernels accept: C % 4 == 0 and row strides that are multiples of 4."""
    return C % 4 == 0 and all(ld % 4 == 0 for ld in lds)


def gn_apply_fwd(x, ldx, pairs, per_sample, gamma, beta, eps, B, HW, C, y, ldy, mean_rstd, planes=None):
    """planes (Planes): the result also (y given) or only (y None) as bf16 planes."""
    if planes is None:
        _check(_lib.vrnet_gn_apply_fwd(ptr(x), ldx, ptr(pairs), per_sample, ptr(gamma), ptr(beta), eps, B, HW, C, ptr(y), ldy,
                                       ptr(mean_rstd), stream()), "gn_apply_fwd")
    else:
        _check(_lib.vrnet_gn_apply_fwd_planes(ptr(x), ldx, ptr(pairs), per_sample, ptr(gamma), ptr(beta), eps, B, HW, C, ptr(y), ldy,
                                              ptr(mean_rstd), _planes_out(planes), stream()), "gn_apply_fwd_planes")


def gn_apply_bwd(dy, lddy, x, ldx, mean_rstd, gamma, B, HW, C, out, ldo, dgamma, dbeta, accumulate_params, add=None, ldadd=0,
                 planes=None):
    ws = _ws.get(_lib.vrnet_gn_bwd_workspace(B, HW, C), x.device)
    if planes is None:
        _check(_lib.vrnet_gn_apply_bwd(ptr(dy), lddy, ptr(x), ldx, ptr(mean_rstd), ptr(gamma), B, HW, C, ptr(add), ldadd, ptr(out),
                                       ldo, ptr(dgamma), ptr(dbeta), accumulate_params, ptr(ws), ws.numel(), stream()), "gn_apply_bwd")
    else:
        _check(_lib.vrnet_gn_apply_bwd_planes(ptr(dy), lddy, ptr(x), ldx, ptr(mean_rstd), ptr(gamma), B, HW, C, ptr(add), ldadd,
                                              ptr(out), ldo, ptr(dgamma), ptr(dbeta), accumulate_params, _planes_out(planes),
                                              ptr(ws), ws.numel(), stream()), "gn_apply_bwd_planes")


def colstats_ok(HW, Cout, *lds):
    """Shapes for which a conv can leave the column statistics of its output (see conv2d `colstats`)."""
    return HW % 32 == 0 and Cout > 32 and Cout % 4 == 0 and all(ld % 4 == 0 for ld in lds)


def colstats_buffers(B, HW, C, device, totals=False):
    part = torch.empty((B * HW // 32, C, 2), dtype=torch.float64, device=device)
    tot = torch.empty((B * HW // 32, (C + 31) // 32, 2), dtype=torch.float64, device=device) if totals else None
    return part, tot


def gn_apply_bwd_from_partials(dy, lddy, x, ldx, partial, tile_totals, mean_rstd, gamma, B, HW, C, out, ldo, dgamma, dbeta,
                               accumulate_params, add=None, ldadd=0):
    _check(_lib.vrnet_gn_apply_bwd_from_partials(ptr(dy), lddy, ptr(x), ldx, ptr(partial), ptr(tile_totals), ptr(mean_rstd),
                                                 ptr(gamma), B, HW, C, ptr(add), ldadd, ptr(out), ldo, ptr(dgamma), ptr(dbeta),
                                                 accumulate_params, stream()), "gn_apply_bwd_from_partials")


def bn_coef_fwd_from_partials(partial, gamma, beta, eps, momentum, rm, rv, nbt, B, HW, C, A, Dc, S, mean_rstd):
    _check(_lib.vrnet_bn_coef_fwd_from_partials(ptr(partial), ptr(gamma), ptr(beta), eps, momentum, ptr(rm), ptr(rv), ptr(nbt), B,
                                                HW, C, ptr(A), ptr(Dc), ptr(S), ptr(mean_rstd), stream()), "bn_coef_fwd_from_partials")


def gn_coef_bwd(mom2, mean_rstd, gamma, B, HW, C, A, E, Dc, S, dgamma, dbeta, accumulate, gamma2=None, dgamma2=None,
                dbeta2=None):
    _check(_lib.vrnet_gn_coef_bwd(ptr(mom2), ptr(mean_rstd), ptr(gamma), B, HW, C, ptr(A), ptr(E), ptr(Dc), ptr(S),
                                  ptr(dgamma), ptr(dbeta), accumulate, ptr(gamma2), ptr(dgamma2), ptr(dbeta2), stream()),
           "gn_coef_bwd")


def bn_coef_fwd(mom, gamma, beta, eps, momentum, rm, rv, nbt, training, B, HW, C, A, Dc, S, mean_rstd):
    _check(_lib.vrnet_bn_coef_fwd(ptr(mom), ptr(gamma), ptr(beta), eps, momentum, ptr(rm), ptr(rv), ptr(nbt),
                                  int(training), B, HW, C, ptr(A), ptr(Dc), ptr(S), ptr(mean_rstd), stream()),
           "bn_coef_fwd")


def bn_coef_bwd(mom2, mean_rstd, gamma, training, B, HW, C, A, E, Dc, S, dgamma, dbeta, accumulate):
    _check(_lib.vrnet_bn_coef_bwd(ptr(mom2), ptr(mean_rstd), ptr(gamma), int(training), B, HW, C, ptr(A), ptr(E),
                                  ptr(Dc), ptr(S), ptr(dgamma), ptr(dbeta), accumulate, stream()), "bn_coef_bwd")


def bn_stats_fwd(x, ldx, gamma, beta, eps, momentum, rm, rv, nbt, B, HW, C, A, Dc, S, mean_rstd):
    """Train-mode BatchNorm: batch moments of x + coefficients + running statistics in two launches."""
    ws = _ws.get(_lib.vrnet_moments_workspace(B, HW, C), x.device)
    _check(_lib.vrnet_bn_stats_fwd(ptr(x), ldx, ptr(gamma), ptr(beta), eps, momentum, ptr(rm), ptr(rv), ptr(nbt), B, HW, C,
                                   ptr(A), ptr(Dc), ptr(S), ptr(mean_rstd), ptr(ws), ws.numel(), stream()), "bn_stats_fwd")


def bn_stats_bwd(dy, lddy, z, ldz, mask, ldm, mean_rstd, gamma, training, B, HW, C, A, E, Dc, S, dgamma, dbeta, accumulate):
    ws = _ws.get(_lib.vrnet_moments_workspace(B, HW, C), dy.device)
    _check(_lib.vrnet_bn_stats_bwd(ptr(dy), lddy, ptr(z), ldz, ptr(mask), ldm, ptr(mean_rstd), ptr(gamma), int(training), B,
                                   HW, C, ptr(A), ptr(E), ptr(Dc), ptr(S), ptr(dgamma), ptr(dbeta), accumulate, ptr(ws),
                                   ws.numel(), stream()), "bn_stats_bwd")


def bn_stats_bwd_zmask(dy, lddy, z, ldz, fwd, mean_rstd, gamma, training, B, HW, C, A, E, Dc, S, dgamma, dbeta, accumulate):
    """bn_stats_bwd with the ReLU mask recomputed from z: fwd = the forward apply's (A, D, S)."""
    ws = _ws.get(_lib.vrnet_moments_workspace(B, HW, C), dy.device)
    _check(_lib.vrnet_bn_stats_bwd_zmask(ptr(dy), lddy, ptr(z), ldz, ptr(fwd[0]), ptr(fwd[1]), ptr(fwd[2]), ptr(mean_rstd),
                                         ptr(gamma), int(training), B, HW, C, ptr(A), ptr(E), ptr(Dc), ptr(S), ptr(dgamma),
                                         ptr(dbeta), accumulate, ptr(ws), ws.numel(), stream()), "bn_stats_bwd_zmask")


def bn_apply_bwd_zmask(dy, lddy, z, ldz, fwd, A, E, Dc, S, dz, lddz, B, HW, C):
    _check(_lib.vrnet_bn_apply_bwd_zmask(ptr(dy), lddy, ptr(z), ldz, ptr(fwd[0]), ptr(fwd[1]), ptr(fwd[2]), ptr(A), ptr(E),
                                         ptr(Dc), ptr(S), ptr(dz), lddz, B, HW, C, stream()), "bn_apply_bwd_zmask")


def eca_coef_fwd(mom, wk, k, B, HW, C, gate):
    _check(_lib.vrnet_eca_coef_fwd(ptr(mom), ptr(wk), k, B, HW, C, ptr(gate), stream()), "eca_coef_fwd")


def eca_coef_bwd(mom2, mom, gate, wk, k, B, HW, C, Fc, dwk, accumulate):
    _check(_lib.vrnet_eca_coef_bwd(ptr(mom2), ptr(mom), ptr(gate), ptr(wk), k, B, HW, C, ptr(Fc), ptr(dwk), accumulate,
                                   stream()), "eca_coef_bwd")


def ls_coef_bwd(mom2, ls, B, C, dls, dbias, accumulate, pair=0, ls2=None, dls2=None, dbias2=None):
    _check(_lib.vrnet_ls_coef_bwd(ptr(mom2), ptr(ls), B, C, ptr(dls), ptr(dbias), accumulate, pair, ptr(ls2), ptr(dls2),
                                  ptr(dbias2), stream()), "ls_coef_bwd")


def moments_to_float(mom, out, n, scale, which=0):
    _check(_lib.vrnet_moments_to_float(ptr(mom), ptr(out), n, scale, which, stream()), "moments_to_float")


def patch_gather(x, ldx, pos, out, B, H, W, C, CP, k):
    _check(_lib.vrnet_patch_gather_f32(ptr(x), ldx, ptr(pos), ptr(out), B, H, W, C, CP, k, stream()), "patch_gather")


def patch_scatter(dp, dx, lddx, B, H, W, C, CP, k, accumulate=0):
    _check(_lib.vrnet_patch_scatter_f32(ptr(dp), ptr(dx), lddx, B, H, W, C, CP, k, accumulate, stream()), "patch_scatter")


def weight_ohwi(src, dst, Cout, Cin, kh, kw, direction, accumulate=0):
    _check(_lib.vrnet_weight_ohwi_f32(ptr(src), ptr(dst), Cout, Cin, kh, kw, direction, accumulate, stream()), "weight_ohwi")


def copy_channels(src, lds, scs, dst, ldd, dcs, rows, C, accumulate=0):
    _check(_lib.vrnet_copy_channels_f32(ptr(src), lds, scs, ptr(dst), ldd, dcs, rows, C, accumulate, stream()),
           "copy_channels")


def cat2(a, lda, Ca, b, ldb, Cb, cat, ldc, rows, interleave, dir=0, accumulate_a=0, accumulate_b=0):
    """dir 0: cat = torch.cat([a, b], channels) (+ 2-group shuffle when interleave); dir 1: the adjoint (a / b may be None)."""
    _check(_lib.vrnet_cat2_f32(ptr(a), lda, Ca, ptr(b), ldb, Cb, ptr(cat), ldc, rows, int(interleave), dir, accumulate_a,
                               accumulate_b, stream()), "cat2")


def nchw_to_nhwc(src, dst, ldd, B, C, HW):
    _check(_lib.vrnet_nchw_to_nhwc_f32(ptr(src), ptr(dst), ldd, B, C, HW, stream()), "nchw_to_nhwc")


def nhwc_to_nchw(src, lds, dst, B, C, HW, accumulate=0):
    _check(_lib.vrnet_nhwc_to_nchw_f32(ptr(src), lds, ptr(dst), B, C, HW, accumulate, stream()), "nhwc_to_nchw")


def add_(dst, src):
    _check(_lib.vrnet_add_f32(ptr(dst), ptr(src), dst.numel(), stream()), "add")


def clock_stamp(buf, slot):
    """buf[slot] (int64) = the device clock (100 MHz) when the stream reaches this point (diagnostic)."""
    _check(_lib.vrnet_clock_stamp(buf.data_ptr() + 8 * slot, stream()), "clock_stamp")


def fill_(dst, value):
    _check(_lib.vrnet_fill_f32(ptr(dst), float(value), dst.numel(), stream()), "fill")


def cluster_state_floats(B, H, W, E, fold):
    """Floats of per-region forward state for regions of more than 256 points (0: none needed)."""
    return _lib.vrnet_cluster_state_floats(B, H, W, E, fold)


def cluster_fwd(f, v, ld, alpha, beta, out, ldo, idx, wgt, B, H, W, E, Dh, fold, alpha2=None, beta2=None, forced=False,
                planes=None, state=None):
    """forced: idx is given (read), not computed (teacher-forced assignment for parity comparisons).
    planes (Planes): `out` as bf16 planes, beside the fp32 `out` or instead of it (out None); single-stream launches.
    f / v may then be bfloat16 tensors (ld in elements).  state (cluster_state_floats floats): the forward's per-region state
    for cluster_bwd(saved=(wgt, state))."""
    if planes is not None or f.dtype == torch.bfloat16 or state is not None:
        assert alpha2 is None and f.dtype == v.dtype
        _check(_lib.vrnet_cluster_fwd_planes_f32(ptr(f), ptr(v), ld, 1 if f.dtype == torch.bfloat16 else 0, ptr(alpha), ptr(beta),
                                                 ptr(out), ldo, ptr(idx), ptr(wgt), B, H, W, E, Dh, fold, 1 if forced else 0,
                                                 _planes_out(planes), ptr(state), stream()), "cluster_fwd_planes")
        return
    fn = _lib.vrnet_cluster_fwd_forced_f32 if forced else _lib.vrnet_cluster_fwd_f32
    _check(fn(ptr(f), ptr(v), ld, ptr(alpha), ptr(beta), ptr(out), ldo, ptr(idx), ptr(wgt), B, H,
                                      W, E, Dh, fold, ptr(alpha2), ptr(beta2), stream()), "cluster_fwd")


def cluster_bwd(f, v, ld, alpha, beta, idx, dout, lddo, df, dv, lddf, dalpha, dbeta, accumulate_ab, B, H, W, E, Dh,
                fold, alpha2=None, beta2=None, dalpha2=None, dbeta2=None, planes=None, saved=None):
    """planes (Planes of 2 E Dh columns): a second copy of [df | dv] as bf16 planes (single-stream launches).
    saved = (wgt, state) of the forward (regions of more than 256 points): the backward skips its first two passes."""
    nb = _lib.vrnet_cluster_bwd_workspace2(B, H, W, E, fold)
    # dalpha is None: the (d alpha, d beta) partials stay in the workspace -- then a buffer of its own, returned to the caller,
    # who reduces a whole section's modules with one cluster_ab_reduce_multi
    ws = torch.empty(nb, dtype=torch.uint8, device=f.device) if dalpha is None else _ws.get(nb, f.device)
    if planes is not None or f.dtype == torch.bfloat16 or saved is not None or dalpha is None:
        assert alpha2 is None and f.dtype == v.dtype == dout.dtype
        wf, st = saved if saved is not None else (None, None)
        _check(_lib.vrnet_cluster_bwd_planes_f32(ptr(f), ptr(v), ld, 1 if f.dtype == torch.bfloat16 else 0, ptr(alpha), ptr(beta),
                                                 ptr(idx), ptr(dout), lddo, ptr(df), ptr(dv), lddf, ptr(dalpha), ptr(dbeta),
                                                 accumulate_ab, B, H, W, E, Dh, fold, _planes_out(planes), ptr(wf), ptr(st),
                                                 ptr(ws), ws.numel(), stream()), "cluster_bwd_planes")
        return ws if dalpha is None else None
    _check(_lib.vrnet_cluster_bwd_f32(ptr(f), ptr(v), ld, ptr(alpha), ptr(beta), ptr(idx), ptr(dout), lddo, ptr(df),
                                      ptr(dv), lddf, ptr(dalpha), ptr(dbeta), accumulate_ab, B, H, W, E, Dh, fold,
                                      ptr(alpha2), ptr(beta2), ptr(dalpha2), ptr(dbeta2), ptr(ws), ws.numel(), stream()),
           "cluster_bwd")


def cluster_ab_reduce_multi(entries):
    """entries: [(workspace a deferred cluster_bwd returned, B * E * fold^2, dalpha, dbeta, accumulate)]: ONE launch."""
    n = len(entries)
    PA, LA, IA = ctypes.c_void_p * n, ctypes.c_long * n, ctypes.c_int * n
    _check(_lib.vrnet_cluster_ab_reduce_multi(n, PA(*[ptr(e[0]) for e in entries]), LA(*[int(e[1]) for e in entries]),
                                              PA(*[ptr(e[2]) for e in entries]), PA(*[ptr(e[3]) for e in entries]),
                                              IA(*[int(e[4]) for e in entries]), stream()), "cluster_ab_reduce_multi")


def dwconv3x3(x, ldx, w, y, ldy, B, H, W, C, flip=0, accumulate=0):
    _check(_lib.vrnet_dwconv3x3_f32(ptr(x), ldx, ptr(w), ptr(y), ldy, B, H, W, C, flip, accumulate, stream()),
           "dwconv3x3")


def dwconv3x3_wgrad(x, ldx, dy, lddy, dw, B, H, W, C, accumulate=0):
    ws = _ws.get(_lib.vrnet_dwconv3x3_wgrad_workspace(B, H, W, C), x.device)
    _check(_lib.vrnet_dwconv3x3_wgrad_f32(ptr(x), ldx, ptr(dy), lddy, ptr(dw), B, H, W, C, accumulate, ptr(ws),
                                          ws.numel(), stream()), "dwconv3x3_wgrad")


def upsample(x, ldx, y, ldy, B, H, W, C, scale, out_nchw=0):
    _check(_lib.vrnet_upsample_bilinear_f32(ptr(x), ldx, ptr(y), ldy, B, H, W, C, scale, out_nchw, stream()),
           "upsample")


def bn_relu_upsample(z, ldz, A, Dc, S, y, ldy, B, H, W, C, scale, out_nchw=0):
    """y = upsample(ReLU(A (z - S) + Dc)): BatchNorm apply + ReLU on the taps (CoCUpsample without its low-resolution output)."""
    _check(_lib.vrnet_bn_relu_upsample_bilinear_f32(ptr(z), ldz, ptr(A), ptr(Dc), ptr(S), ptr(y), ldy, B, H, W, C, scale, out_nchw,
                                                    stream()), "bn_relu_upsample")


def upsample_bwd(dy, lddy, dy_nchw, dx, lddx, B, H, W, C, scale, accumulate=0):
    _check(_lib.vrnet_upsample_bilinear_bwd_f32(ptr(dy), lddy, dy_nchw, ptr(dx), lddx, B, H, W, C, scale, accumulate,
                                                stream()), "upsample_bwd")


def minmax(p, n, mm):
    ws = _ws.get(_lib.vrnet_reduce_workspace(), p.device)
    _check(_lib.vrnet_minmax_f32(ptr(p), n, ptr(mm), ptr(ws), ws.numel(), stream()), "minmax")


def enhance_mul(p, x, mm, out, n):
    _check(_lib.vrnet_enhance_mul_f32(ptr(p), ptr(x), ptr(mm), ptr(out), n, stream()), "enhance_mul")


def enhance_fwd(p, x, mm, out, n):
    """minmax + enhance_mul in two launches; mm receives (min, max)."""
    ws = _ws.get(_lib.vrnet_reduce_workspace(), p.device)
    _check(_lib.vrnet_enhance_fwd_f32(ptr(p), ptr(x), ptr(mm), ptr(out), n, ptr(ws), ws.numel(), stream()), "enhance_fwd")


def enhance_bwd(dt, x, p, mm, dx, dp, n, accumulate_dx=0):
    ws = _ws.get(_lib.vrnet_reduce_workspace(), p.device)
    _check(_lib.vrnet_enhance_bwd_f32(ptr(dt), ptr(x), ptr(p), ptr(mm), ptr(dx), ptr(dp), n, accumulate_dx, ptr(ws),
                                      ws.numel(), stream()), "enhance_bwd")


def sa_coef_fwd(mom, cw, cb, sw, sb, gnw, gnb, B, HW, C, G, Pq, Qq, Mn):
    _check(_lib.vrnet_sa_coef_fwd(ptr(mom), ptr(cw), ptr(cb), ptr(sw), ptr(sb), ptr(gnw), ptr(gnb), B, HW, C, G, ptr(Pq),
                                  ptr(Qq), ptr(Mn), stream()), "sa_coef_fwd")


def sa_apply(x, ldx, Pq, Qq, Mn, y, ldy, B, HW, C):
    _check(_lib.vrnet_sa_apply_f32(ptr(x), ldx, ptr(Pq), ptr(Qq), ptr(Mn), ptr(y), ldy, B, HW, C, stream()), "sa_apply")


def sa_cat_sums(x, ldx, Pq, Qq, Mn, r, ldr, cat, ldc, B, HW, C):
    """ShuffleAttention apply + cat with r + 2-group shuffle -> cat, and the (B, 2C, 2) channel sums of cat (for the ECA gate)."""
    mom = torch.empty((B, 2 * C, 2), dtype=torch.float64, device=x.device)
    ws = _ws.get(_lib.vrnet_moments_workspace(B, HW, 2 * C), x.device)
    _check(_lib.vrnet_sa_cat_sums_f32(ptr(x), ldx, ptr(Pq), ptr(Qq), ptr(Mn), ptr(r), ldr, ptr(cat), ldc, B, HW, C, ptr(mom), ptr(ws),
                                      ws.numel(), stream()), "sa_cat_sums")
    return mom


def sa_bwd(dy, lddy, x, ldx, Pq, Qq, Mn, mom, params, dx, lddx, grads, EF, B, HW, C, G, accumulate_dx,
           accumulate_params):
    ws = _ws.get(_lib.vrnet_sa_bwd_workspace(B, HW, C), x.device)
    _check(_lib.vrnet_sa_bwd_f32(ptr(dy), lddy, ptr(x), ldx, ptr(Pq), ptr(Qq), ptr(Mn), ptr(mom),
                                 *[ptr(t) for t in params],
                                 ptr(dx), lddx, *[ptr(t) for t in grads], ptr(EF), B, HW, C, G, accumulate_dx,
                                 accumulate_params, ptr(ws), ws.numel(), stream()), "sa_bwd")


# ---- multi-tensor updates (optimizer step, EMA): tables are int64/int32/float32 device tensors built by optim.py
def mt_sgd(addrs, sizes, chunk_tensor, chunk_index, weight_decay, n_tensors, n_chunks, chunk_elems, lr, momentum, nesterov,
           first_step):
    _check(_lib.vrnet_mt_sgd_f32(ptr(addrs), ptr(sizes), ptr(chunk_tensor), ptr(chunk_index), ptr(weight_decay), n_tensors,
                                 n_chunks, chunk_elems, lr, momentum, int(nesterov), int(first_step), stream()), "mt_sgd")


def mt_adam(addrs, sizes, chunk_tensor, chunk_index, weight_decay, n_tensors, n_chunks, chunk_elems, lr, beta1, beta2, eps,
            step):
    _check(_lib.vrnet_mt_adam_f32(ptr(addrs), ptr(sizes), ptr(chunk_tensor), ptr(chunk_index), ptr(weight_decay), n_tensors,
                                  n_chunks, chunk_elems, lr, beta1, beta2, eps, step, stream()), "mt_adam")


def mt_ema(addrs, sizes, chunk_tensor, chunk_index, n_tensors, n_chunks, chunk_elems, decay):
    _check(_lib.vrnet_mt_ema_f32(ptr(addrs), ptr(sizes), ptr(chunk_tensor), ptr(chunk_index), n_tensors, n_chunks,
                                 chunk_elems, decay, stream()), "mt_ema")


def mt_copy(addrs, sizes, chunk_tensor, chunk_index, n_tensors, n_chunks, chunk_elems):
    _check(_lib.vrnet_mt_copy_f32(ptr(addrs), ptr(sizes), ptr(chunk_tensor), ptr(chunk_index), n_tensors, n_chunks,
                                  chunk_elems, stream()), "mt_copy")


def decode_outputs(levels, input_h, input_w, out):
    """levels: list of contiguous (B, C, h, w) fp32 GPU tensors; out: (B, sum h*w, C)."""
    n = len(levels)
    ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in levels])
    hs = (ctypes.c_int * n)(*[t.shape[2] for t in levels])
    ws = (ctypes.c_int * n)(*[t.shape[3] for t in levels])
    _check(_lib.vrnet_decode_outputs_f32(ptrs, hs, ws, n, levels[0].shape[0], levels[0].shape[1], float(input_h),
                                         float(input_w), ptr(out), stream()), "decode_outputs")


def batch_formats(img_u8, png_u8, num_classes_seg, images=None, png_out=None, onehot=None):
    """Letterboxed batch as bytes -> the tensors the reference's collate function ships (vrnet_batch_formats_u8):
    img_u8 (B,H,W,3) uint8 -> images (B,3,H,W) f32; png_u8 (B,H,W) uint8 -> png (B,H,W) int64, onehot (B,H,W,ns+1) f32."""
    src = img_u8 if img_u8 is not None else png_u8
    B, H, W = src.shape[:3]
    for t, sh in ((img_u8, (B, H, W, 3)), (png_u8, (B, H, W))):
        if t is not None and (t.dtype != torch.uint8 or tuple(t.shape) != sh or not t.is_contiguous() or not t.is_cuda):
            raise RuntimeError(f"batch_formats: expected a contiguous uint8 GPU tensor of shape {sh}, got {t.dtype} {tuple(t.shape)}")
    if img_u8 is not None and images is None:
        images = torch.empty((B, 3, H, W), dtype=torch.float32, device=src.device)
    if png_u8 is not None:
        if png_out is None:
            png_out = torch.empty((B, H, W), dtype=torch.int64, device=src.device)
        if onehot is None:
            onehot = torch.empty((B, H, W, num_classes_seg + 1), dtype=torch.float32, device=src.device)
    _check(_lib.vrnet_batch_formats_u8(ptr(img_u8), ptr(png_u8), B, H, W, int(num_classes_seg), ptr(images), ptr(png_out),
                                       ptr(onehot), stream()), "batch_formats")
    return images, png_out, onehot


def yolo_loss(levels, grads, strides, labels, counts, max_gt, grad_scale, out, fg=None, matched=None, piou=None):
    """levels / grads: lists of contiguous (B, C, h, w) fp32 GPU tensors (grads None = value only)."""
    n = len(levels)
    B, C = levels[0].shape[:2]
    A = sum(t.shape[2] * t.shape[3] for t in levels)
    lv = (ctypes.c_void_p * n)(*[t.data_ptr() for t in levels])
    gr = (ctypes.c_void_p * n)(*[t.data_ptr() for t in grads]) if grads is not None else None
    hs = (ctypes.c_int * n)(*[t.shape[2] for t in levels])
    ws_ = (ctypes.c_int * n)(*[t.shape[3] for t in levels])
    st = (ctypes.c_float * n)(*[float(s) for s in strides])
    ws = _ws.get(_lib.vrnet_yolo_loss_workspace(B, A, max_gt, C - 5), levels[0].device)
    _check(_lib.vrnet_yolo_loss_f32(lv, gr, hs, ws_, st, n, B, C, ptr(labels), ptr(counts), max_gt, float(grad_scale),
                                    ptr(out), ptr(fg), ptr(matched), ptr(piou), ptr(ws), ws.numel(), stream()), "yolo_loss")


def seg_loss(x, png, onehot, weights, focal, dice, alpha, gamma, beta, smooth, grad_scale, out, dx):
    B, C, H, W = x.shape
    ws = _ws.get(_lib.vrnet_seg_loss_workspace(B, C, H * W), x.device)
    _check(_lib.vrnet_seg_loss_f32(ptr(x), ptr(png), ptr(onehot), ptr(weights), B, C, H * W, int(focal), int(bool(dice)),
                                   float(alpha), float(gamma), float(beta), float(smooth), float(grad_scale), ptr(out),
                                   ptr(dx), ptr(ws), ws.numel(), stream()), "seg_loss")


def mean_square(tensors):
    """loss = sum_k mean(t_k^2) (a 1-element fp32 tensor) over contiguous fp32 tensors: two launches."""
    k = len(tensors)
    PA, LA = ctypes.c_void_p * k, ctypes.c_long * k
    n = LA(*[t.numel() for t in tensors])
    ws = _ws.get(_lib.vrnet_mean_square_workspace(k, n), tensors[0].device)
    loss = torch.empty((1,), dtype=torch.float32, device=tensors[0].device)
    _check(_lib.vrnet_mean_square_f32(k, PA(*[ptr(t) for t in tensors]), n, ptr(loss), ptr(ws), ws.numel(), stream()), "mean_square")
    return loss


def mean_square_bwd(tensors, g, grads):
    """grads[k] = (2 g / n_k) tensors[k]; g: 1-element device tensor."""
    k = len(tensors)
    PA, LA = ctypes.c_void_p * k, ctypes.c_long * k
    _check(_lib.vrnet_mean_square_bwd_f32(k, PA(*[ptr(t) for t in tensors]), LA(*[t.numel() for t in tensors]), ptr(g),
                                          PA(*[ptr(t) for t in grads]), stream()), "mean_square_bwd")


# ---- fused passes of the fusion blocks (csrc/fusion.hip)
def fusion_chunks(n, C):
    """Workgroups / partial entries of the fused fusion-block kernels for a contiguous tensor (0: shape not supported)."""
    return _lib.vrnet_fusion_chunks(n, C)


def fusion_fold_chunks(n, C):
    """Entries of the (min, max) / four-sum partials (bn_relu_minmax, bn_bwd_enhance)."""
    return _lib.vrnet_fusion_fold_chunks(n, C)


def bn_relu_minmax(z, A, D, S, p, n, C, mmpart):
    _check(_lib.vrnet_bn_relu_minmax_f32(ptr(z), ptr(A), ptr(D), ptr(S), ptr(p), n, C, ptr(mmpart), stream()), "bn_relu_minmax")


def bn_relu_res_stats(z, A, D, S, res, s, n, C, colpart):
    _check(_lib.vrnet_bn_relu_res_stats_f32(ptr(z), ptr(A), ptr(D), ptr(S), ptr(res), ptr(s), n, C, ptr(colpart), stream()),
           "bn_relu_res_stats")


def enhance_stats(p, x, mmpart, nmm, mm, t, n, C, colpart):
    _check(_lib.vrnet_enhance_stats_f32(ptr(p), ptr(x), ptr(mmpart), nmm, ptr(mm), ptr(t), n, C, ptr(colpart), stream()),
           "enhance_stats")


def bn_bwd_enhance(g, t, A, E, D, S, x, p, mm, dt, n, C, sums4):
    _check(_lib.vrnet_bn_bwd_enhance_f32(ptr(g), ptr(t), ptr(A), ptr(E), ptr(D), ptr(S), ptr(x), ptr(p), ptr(mm), ptr(dt), n, C,
                                         ptr(sums4), stream()), "bn_bwd_enhance")


def enhance_bwd_stats(dt, x, p, mm, sums4, nsums, z, fA, fD, fS, dx, dp, n, C, accumulate_dx, colpart):
    _check(_lib.vrnet_enhance_bwd_stats_f32(ptr(dt), ptr(x), ptr(p), ptr(mm), ptr(sums4), nsums, ptr(z), ptr(fA), ptr(fD), ptr(fS),
                                            ptr(dx), ptr(dp), n, C, accumulate_dx, ptr(colpart), stream()), "enhance_bwd_stats")


def bn_bwd_next_stats(g, s, A, E, D, S, z, fwd, ds, n, C, colpart):
    _check(_lib.vrnet_bn_bwd_next_stats_f32(ptr(g), ptr(s), ptr(A), ptr(E), ptr(D), ptr(S), ptr(z), ptr(fwd[0]), ptr(fwd[1]),
                                            ptr(fwd[2]), ptr(ds), n, C, ptr(colpart), stream()), "bn_bwd_next_stats")


def bn_coef_fwd_from_chunks(partial, nchunks, count, gamma, beta, eps, momentum, rmean, rvar, nbt, C, A, Dc, S, mean_rstd):
    _check(_lib.vrnet_bn_coef_fwd_from_chunks(ptr(partial), nchunks, count, ptr(gamma), ptr(beta), eps, momentum, ptr(rmean), ptr(rvar),
                                              ptr(nbt), C, ptr(A), ptr(Dc), ptr(S), ptr(mean_rstd), stream()), "bn_coef_fwd_from_chunks")


def bn_coef_bwd_from_chunks(partial, nchunks, count, mean_rstd, gamma, training, C, A, E, Dc, S, dgamma, dbeta, accumulate):
    _check(_lib.vrnet_bn_coef_bwd_from_chunks(ptr(partial), nchunks, count, ptr(mean_rstd), ptr(gamma), 1 if training else 0, C, ptr(A),
                                              ptr(E), ptr(Dc), ptr(S), ptr(dgamma), ptr(dbeta), accumulate, stream()),
           "bn_coef_bwd_from_chunks")
