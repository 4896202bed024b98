"""Host-side mirror of the reference module tree for the fusion hot path.

The classes below hold PARAMETERS ONLY, registered under the reference's attribute names so
that ``state_dict()`` has the reference's 887 keys (names, shapes, dtypes, order) and the
callers' introspection keeps working:

* ``train.py:460-473`` groups parameters by ``isinstance(v, nn.BatchNorm2d) or "bn" in k``,
  ``hasattr(v, "weight")`` / ``hasattr(v, "bias")``  -> leaves are real ``nn.Conv2d`` /
  ``nn.BatchNorm2d`` / ``nn.GroupNorm`` / ``nn.Conv1d`` modules;
* ``nets/yolo_training.py:482-501`` (``weights_init``) keys on class names containing
  ``'Conv'`` / ``'BatchNorm2d'`` and ``hasattr(m, 'weight')`` -> containers expose no ``.weight``;
* ``train.py:440`` freezes ``model.backbone.backbone.parameters()``.

No leaf module's ``forward`` is ever called by the hot path: compute happens in
``program.py`` which launches the HIP kernels of ``libvrnet_hip.so`` on these parameters.
Citations are file:line in the reference tree.
"""
import math

import torch
import torch.nn as nn

WIDTH = {"nano": 0.25, "tiny": 0.375, "s": 0.50, "m": 0.75, "l": 1.00}   # nets/efficient_vrnet.py:16-17
DEPTH = {"nano": 0.33, "tiny": 0.33, "s": 0.33, "m": 0.67, "l": 1.00}


def _no_forward(self, *a, **k):
    raise RuntimeError(
        f"{type(self).__name__} is a parameter holder of the HIP hot path; call EfficientVRNet.forward")


class _Holder(nn.Module):
    forward = _no_forward


class _EmptyGroupNorm(_Holder):
    """nn.GroupNorm(0, 0) of torch 1.9 (shuffle_attention.py:15 with channel // (2*G) == 0):
    two zero-sized parameters that are part of the state_dict surface."""

    def __init__(self):
        super().__init__()
        self.num_groups, self.num_channels, self.eps = 0, 0, 1e-5
        self.weight = nn.Parameter(torch.empty(0))
        self.bias = nn.Parameter(torch.empty(0))


class BaseConv(_Holder):
    """backbone/conv_utils/normal_conv.py:36-49  conv(bias=False) -> BN(eps 1e-3, momentum .03) -> ReLU."""

    def __init__(self, cin, cout, ksize, stride=1, ds_conv=False):
        super().__init__()
        self.ksize, self.stride, self.ds_conv = ksize, stride, ds_conv
        pad = (ksize - 1) // 2
        if ds_conv:
            self.conv = DWConv(cin, cout, ksize, stride, pad, bias=False)
        else:
            self.conv = nn.Conv2d(cin, cout, ksize, stride, pad, bias=False)
        self.bn = nn.BatchNorm2d(cout, eps=0.001, momentum=0.03)
        self.act = nn.ReLU(inplace=True)


class DWConv(_Holder):
    """normal_conv.py:23-33  depthwise kxk + pointwise 1x1."""

    def __init__(self, cin, cout, ksize, stride=1, padding=0, bias=True):
        super().__init__()
        self.dconv = nn.Conv2d(cin, cin, ksize, stride, padding, groups=cin, bias=bias)
        self.pconv = nn.Conv2d(cin, cout, 1, 1, bias=bias)


class PointRecuder(_Holder):
    """backbone/fusion/vr_coc.py:83-102  a single conv (norm is Identity in the live model)."""

    def __init__(self, patch_size, stride, padding, in_chans, embed_dim):
        super().__init__()
        self.proj = nn.Conv2d(in_chans, embed_dim, patch_size, stride, padding)
        self.norm = nn.Identity()


class GroupNorm(nn.GroupNorm):
    """vr_coc.py:105-111  GroupNorm with one group."""

    def __init__(self, num_channels):
        super().__init__(1, num_channels)

    forward = _no_forward


class Cluster(_Holder):
    """vr_coc.py:128-153"""

    def __init__(self, dim, out_dim, fold, heads, head_dim):
        super().__init__()
        self.heads, self.head_dim, self.fold = heads, head_dim, fold
        self.fc1 = nn.Conv2d(dim, heads * head_dim, 1)
        self.fc2 = nn.Conv2d(heads * head_dim, out_dim, 1)
        self.fc_v = nn.Conv2d(dim, heads * head_dim, 1)
        self.sim_alpha = nn.Parameter(torch.ones(1))
        self.sim_beta = nn.Parameter(torch.zeros(1))


class Mlp(_Holder):
    """vr_coc.py:195-215  (trunc_normal_(.02) weights, zero bias)."""

    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Conv2d(dim, hidden, 1)
        self.act = nn.GELU()
        self.fc2 = nn.Conv2d(hidden, dim, 1)
        for m in (self.fc1, self.fc2):
            nn.init.trunc_normal_(m.weight, std=0.02)
            nn.init.constant_(m.bias, 0)


class ClusterBlock(_Holder):
    """vr_coc.py:226-262 (backbone) / backbone/vision/context_cluster.py:198-235 (neck defaults
    heads=4, head_dim=24, fold=2, mlp_ratio=4)."""

    def __init__(self, dim, mlp_ratio=4.0, fold=2, heads=4, head_dim=24, layer_scale_init_value=1e-5):
        super().__init__()
        self.dim = dim
        self.norm1 = GroupNorm(dim)
        self.token_mixer = Cluster(dim, dim, fold, heads, head_dim)
        self.norm2 = GroupNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        self.layer_scale_1 = nn.Parameter(layer_scale_init_value * torch.ones(dim))
        self.layer_scale_2 = nn.Parameter(layer_scale_init_value * torch.ones(dim))


class ShuffleAttention(_Holder):
    """backbone/attention_modules/shuffle_attention.py:8-20"""

    def __init__(self, channel, G=8):
        super().__init__()
        self.G, self.channel = G, channel
        cp = channel // (2 * G)
        self.gn = nn.GroupNorm(cp, cp) if cp > 0 else _EmptyGroupNorm()
        self.cweight = nn.Parameter(torch.zeros(1, cp, 1, 1))
        self.cbias = nn.Parameter(torch.ones(1, cp, 1, 1))
        self.sweight = nn.Parameter(torch.zeros(1, cp, 1, 1))
        self.sbias = nn.Parameter(torch.ones(1, cp, 1, 1))


class eca_block(_Holder):
    """backbone/attention_modules/eca.py:6-14"""

    def __init__(self, channel, b=1, gamma=2):
        super().__init__()
        k = int(abs((math.log(channel, 2) + b) / gamma))
        k = k if k % 2 else k + 1
        self.kernel_size = k
        self.conv = nn.Conv1d(1, 1, kernel_size=k, padding=(k - 1) // 2, bias=False)


class ImageEnhanceByRadar(_Holder):
    """vr_coc.py:303-310"""

    def __init__(self, radar_in_channels, image_in_channels):
        super().__init__()
        self.radar_projection = BaseConv(radar_in_channels, image_in_channels, 3, 1)
        self.norm = nn.BatchNorm2d(image_in_channels)


class RadarEnhanceByImage(_Holder):
    """vr_coc.py:319-329"""

    def __init__(self, radar_in_channels, image_in_channels, initial=False):
        super().__init__()
        self.initial = initial
        self.image_attn = ShuffleAttention(image_in_channels, G=4)
        self.channel_attn = eca_block(radar_in_channels + image_in_channels)
        self.inverse_projection = BaseConv(radar_in_channels + image_in_channels, radar_in_channels, 1, 1)
        self.norm = nn.BatchNorm2d(radar_in_channels)


class VRCoC(_Holder):
    """vr_coc.py:362-489 with the coc_small configuration (vr_coc.py:759-785): layers [2,2,6,2],
    heads [4,4,8,8] x head_dim 32, folds [8,4,2,1], mlp ratios [8,8,4,4], 3x3/s2/p1 reducers."""

    LAYERS = (2, 2, 6, 2)
    HEADS = (4, 4, 8, 8)
    HEAD_DIM = (32, 32, 32, 32)
    FOLD = (8, 4, 2, 1)
    MLP = (8, 8, 4, 4)

    def __init__(self, width=1.0, img_w=512, img_h=512):
        super().__init__()
        dims = [int(64 * width), int(128 * width), int(320 * width), int(512 * width)]
        self.embed_dims = dims
        for name in ("fea_pos", "fea_pos_r"):       # vr_coc.py:402-413
            rw = torch.arange(0, img_w, dtype=torch.float32) / (img_w - 1.0)
            rh = torch.arange(0, img_h, dtype=torch.float32) / (img_h - 1.0)
            pos = torch.stack(torch.meshgrid(rw, rh, indexing="ij"), dim=-1).float() - 0.5
            self.register_buffer(name, pos)
        self.image_initial = PointRecuder(1, 1, 0, 3, 3)
        self.radar_initial = PointRecuder(1, 1, 0, 4, 4)
        self.radar_enhance_by_image1 = RadarEnhanceByImage(image_in_channels=3, radar_in_channels=4, initial=True)
        self.image_enhance_by_radar1 = ImageEnhanceByRadar(image_in_channels=3, radar_in_channels=4)
        self.patch_embed = PointRecuder(4, 4, 0, 5, dims[0])
        self.patch_embed_radar = PointRecuder(4, 4, 0, 6, dims[0])
        network, network_radar = [], []
        for i in range(4):
            for net in (network, network_radar):
                net.append(nn.Sequential(*[
                    ClusterBlock(dims[i], self.MLP[i], self.FOLD[i], self.HEADS[i], self.HEAD_DIM[i])
                    for _ in range(self.LAYERS[i])]))
            network.append(ImageEnhanceByRadar(image_in_channels=dims[i], radar_in_channels=dims[i]))
            network_radar.append(RadarEnhanceByImage(image_in_channels=dims[i], radar_in_channels=dims[i]))
            if i < 3:
                for net in (network, network_radar):
                    net.append(PointRecuder(3, 2, 1, dims[i], dims[i + 1]))
        self.network = nn.ModuleList(network)
        self.network_radar = nn.ModuleList(network_radar)


class CoCUpsample(_Holder):
    """neck/coc_fpn_dual.py:15-22"""

    def __init__(self, cin, cout, scale=2):
        super().__init__()
        self.scale = scale
        self.upsample = nn.Sequential(BaseConv(cin, cout, 1, 1),
                                      nn.Upsample(scale_factor=scale, mode="bilinear", align_corners=True))


class CoC_Conv(_Holder):
    """coc_fpn_dual.py:29-34"""

    def __init__(self, cin, cout):
        super().__init__()
        self.coc = ClusterBlock(dim=cin)
        self.conv_att = BaseConv(cin, cout, 1, 1)


class ASPP(_Holder):
    """coc_fpn_dual.py:46-77"""

    def __init__(self, dim_in, dim_out, bn_mom=0.1):
        super().__init__()

        def branch(k, d):
            return nn.Sequential(nn.Conv2d(dim_in, dim_out, k, 1, padding=0 if k == 1 else d, dilation=d, bias=True),
                                 nn.BatchNorm2d(dim_out, momentum=bn_mom), nn.ReLU(inplace=True))
        self.branch1 = branch(1, 1)
        self.branch2 = branch(3, 6)
        self.branch3 = branch(3, 12)
        self.branch4 = branch(3, 18)
        self.branch5_conv = nn.Conv2d(dim_in, dim_out, 1, 1, 0, bias=True)
        self.branch5_bn = nn.BatchNorm2d(dim_out, momentum=bn_mom)
        self.branch5_relu = nn.ReLU(inplace=True)
        self.conv_cat = nn.Sequential(nn.Conv2d(dim_out * 5, dim_out, 1, 1, padding=0, bias=True),
                                      nn.BatchNorm2d(dim_out, momentum=bn_mom), nn.ReLU(inplace=True))


class CoCFpnDual(_Holder):
    """coc_fpn_dual.py:133-182"""

    def __init__(self, num_seg_class=9, width=1.0, img_size=(512, 512)):
        super().__init__()
        self.backbone = VRCoC(width=width, img_w=img_size[0], img_h=img_size[1])
        self.num_seg_class = num_seg_class
        c = [int(v * width) for v in (64, 128, 320, 512)]
        self.in_channels = c
        self.aspp = ASPP(c[3], c[3])
        self.upsample5_4 = CoCUpsample(c[3], c[2])
        self.sc_attn_seg4 = ShuffleAttention(c[2] * 2)
        self.upsample4_3 = CoCUpsample(c[2] * 2, c[1])
        self.sc_attn_seg3 = ShuffleAttention(c[1] * 2)
        self.upsample3_2 = CoCUpsample(c[1] * 2, c[0])
        self.sc_attn_seg2 = ShuffleAttention(c[0] * 2)
        self.upsample2_0 = CoCUpsample(c[0] * 2, num_seg_class, scale=4)
        self.p5_out_det = CoC_Conv(c[3], c[3])
        self.p5_4_det = CoCUpsample(c[3], c[2])
        self.p4_out_det = CoC_Conv(c[2] * 2, c[2])
        self.p4_3_det = CoCUpsample(c[2], c[1])
        self.p3_out_det = CoC_Conv(c[1] * 2, c[1])


class DecoupleHead(_Holder):
    """head/decouplehead.py:7-40  (`depthwise` is accepted and ignored, as in the reference)."""

    def __init__(self, num_classes, width=1.0, in_channels=(128, 320, 512), depthwise=False):
        super().__init__()
        self.num_classes = num_classes
        hid = int(256 * width)
        self.cls_convs, self.reg_convs = nn.ModuleList(), nn.ModuleList()
        self.cls_preds, self.reg_preds, self.obj_preds = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        self.stems = nn.ModuleList()
        for cin in in_channels:
            self.stems.append(BaseConv(int(cin * width), hid, 1, 1))
            self.cls_convs.append(nn.Sequential(BaseConv(hid, hid, 3, 1, ds_conv=True), BaseConv(hid, hid, 3, 1, ds_conv=True)))
            self.cls_preds.append(nn.Conv2d(hid, num_classes, 1, 1, 0))
            self.reg_convs.append(nn.Sequential(BaseConv(hid, hid, 3, 1, ds_conv=True), BaseConv(hid, hid, 3, 1, ds_conv=True)))
            self.reg_preds.append(nn.Conv2d(hid, 4, 1, 1, 0))
            self.obj_preds.append(nn.Conv2d(hid, 1, 1, 1, 0))
