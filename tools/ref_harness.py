"""Throw-away harness that imports the upstream reference (read-only, at /root/reference)
in THIS container so golden vectors can be generated (tools/make_golden.py).

Never imported by the shipped package, tests marked gpu, smoke() or bench.py:
/root/reference does not exist on the GPU box.

Shims: in-memory stub modules for timm / thop / torchinfo and an
nn.GroupNorm patch accepting num_groups == 0 (torch 1.9 behaviour needed by
backbone/attention_modules/shuffle_attention.py:15 when channel // (2*G) == 0).
"""
import functools
import os
import sys
import types

import torch
import torch.nn as nn

REF_ROOT = os.environ.get("VRNET_REFERENCE_ROOT", "/root/reference")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _install_stubs():
    class DropPath(nn.Module):
        def __init__(self, p=0.0):
            super().__init__()
            self.p = p

        def forward(self, x):
            return x

    def to_2tuple(v):
        return tuple(v) if isinstance(v, (tuple, list)) else (v, v)

    _stub("timm")
    _stub("timm.data", IMAGENET_DEFAULT_MEAN=(0.485, 0.456, 0.406), IMAGENET_DEFAULT_STD=(0.229, 0.224, 0.225))
    _stub("timm.models")
    _stub("timm.models.layers", DropPath=DropPath, trunc_normal_=nn.init.trunc_normal_)
    _stub("timm.models.layers.helpers", to_2tuple=to_2tuple)
    _stub("timm.models.registry", register_model=lambda f: f)
    _stub("thop", profile=lambda *a, **k: (0, 0), clever_format=lambda *a, **k: ("0", "0"))
    _stub("torchinfo", summary=lambda *a, **k: "")


def _patch_groupnorm():
    if getattr(nn.GroupNorm, "_vrnet_patched", False):
        return
    orig_init = nn.GroupNorm.__init__

    def init(self, num_groups, num_channels, eps=1e-5, affine=True, device=None, dtype=None):
        if num_groups == 0:
            nn.Module.__init__(self)
            self.num_groups, self.num_channels, self.eps, self.affine = 0, num_channels, eps, affine
            self.weight = nn.Parameter(torch.empty(num_channels))
            self.bias = nn.Parameter(torch.empty(num_channels))
            return
        orig_init(self, num_groups, num_channels, eps=eps, affine=affine, device=device, dtype=dtype)

    nn.GroupNorm.__init__ = init
    nn.GroupNorm._vrnet_patched = True


_loaded = {}


def load_reference():
    """Returns a namespace with the reference's live classes."""
    if _loaded:
        return types.SimpleNamespace(**_loaded)
    sys.dont_write_bytecode = True
    _install_stubs()
    _patch_groupnorm()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import nets.efficient_vrnet as evr
    import neck.coc_fpn_dual as neck
    import backbone.fusion.vr_coc as vr
    import backbone.vision.context_cluster as vcc
    import backbone.attention_modules.shuffle_attention as sa
    import backbone.attention_modules.eca as eca
    import backbone.conv_utils.normal_conv as nc
    import head.decouplehead as head
    _loaded.update(evr=evr, neck=neck, vr=vr, vcc=vcc, sa=sa, eca=eca, nc=nc, head=head,
                   orig_coc_small=neck.coc_small)
    return types.SimpleNamespace(**_loaded)


def build_reference_model(num_classes=4, num_seg_classes=9, phi="nano", img_size=512):
    """EfficientVRNet with fea_pos sized for img_size."""
    ref = load_reference()
    iw, ih = (img_size, img_size) if isinstance(img_size, int) else img_size      # (H, W): the reference's img_w is the FIRST
    ref.neck.coc_small = functools.partial(ref.orig_coc_small, img_w=iw, img_h=ih)   # spatial axis (vr_coc.py:402-407, 583-585)
    try:
        model = ref.evr.EfficientVRNet(num_classes=num_classes, num_seg_classes=num_seg_classes, phi=phi)
    finally:
        ref.neck.coc_small = ref.orig_coc_small
    return model
