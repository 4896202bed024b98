"""CPU oracle for the ASY-VRNet fusion hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
The shipped package (asy-vrnet_amd/) never does: its forward/backward run on the HIP
library (libvrnet_hip.so) and fail loudly when that library is missing.

What it is: a functional, pure-torch fp32 (dtype-generic: fp64 works too) restatement of
the reference algorithm, written from the op semantics over a flat
``{state_dict key: tensor}`` dict.  Layout NCHW, autograd provides the backward.
Pinned against the reference itself: tools/make_golden.py imports /root/reference in the
build container, runs it on seeded inputs/parameters and commits outputs + gradients as
tests/golden/*.npz; tests/test_oracle_golden.py checks this file against those vectors.
(The reference repo holds no tests or golden vectors of its own - SURVEY.md section 4.)

Reference lines restated by each function are cited in the docstrings as file:line
relative to the reference root.
"""
import math

import torch
import torch.nn.functional as F

WIDTH = {"nano": 0.25, "tiny": 0.375, "s": 0.50, "m": 0.75, "l": 1.00}  # nets/efficient_vrnet.py:17
STAGE_BLOCKS = (2, 2, 6, 2)          # backbone/fusion/vr_coc.py:761
STAGE_HEADS = (4, 4, 8, 8)           # vr_coc.py:770
STAGE_HEAD_DIM = (32, 32, 32, 32)    # vr_coc.py:771
STAGE_FOLD = (8, 4, 2, 1)            # vr_coc.py:768-769
STAGE_MLP = (8, 8, 4, 4)             # vr_coc.py:764
NECK_HEADS, NECK_HEAD_DIM, NECK_FOLD, NECK_MLP = 4, 24, 2, 4   # backbone/vision/context_cluster.py:211-216


def stage_dims(width):
    """vr_coc.py:763"""
    return [int(64 * width), int(128 * width), int(320 * width), int(512 * width)]


# ----------------------------------------------------------------------------- primitives
class Ctx:
    """Carries mode + side outputs (BN running-stat updates, Cluster assignment maps)."""

    def __init__(self, training, forced_idx=None, forced_relu=None):
        self.training = training
        # Teacher forcing of the ReLU masks behind BatchNorm (the other discontinuity of the path): when given,
        # relu site `key` (= its BatchNorm's state_dict prefix) multiplies by forced_relu[key] (B,C,H,W bool) instead
        # of its own (z > 0) and records in relu_report[key] how many elements disagree and the largest |z| among
        # them: a legitimate flip is a pre-activation within rounding of zero.
        self.forced_relu = forced_relu
        self.relu_report = {}
        self.new_stats = {}
        self.idx = {}
        self.taps = {}
        # Teacher forcing of the hard assignment (flip-aware parity): when given,
        # cluster `pre` uses forced_idx[pre] instead of its own argmax and records, in
        # idx_report[pre], how many points disagree with its own argmax and the largest similarity
        # gap (own best - forced choice) among them: a legitimate flip is a near-tie.
        self.forced_idx = forced_idx
        self.idx_report = {}


# Operand rounding of the dense convolutions, mirroring the build's `compute_dtype = "bf16"` mode (BASELINE configs
# "bf16 with MFMA conv path"; the reference's counterpart is autocast): None, or "bf16" = activations and weights are
# rounded to bfloat16 (round-to-nearest-even, straight-through gradient) before an fp32/fp64 convolution, for exactly
# the layers the build runs on its bf16 kernels (hip.bf16_conv_ok: contraction % 4 == 0, more than 32 output channels).
OPERAND_ROUND = None


def _round_ste(t):
    return t + (t.detach().float().bfloat16().to(t.dtype) - t.detach())


def conv(P, pre, x, stride=1, pad=0, dil=1, groups=1):
    w = P[pre + ".weight"]
    if OPERAND_ROUND == "bf16" and groups == 1:
        co, ci, kh, kw = w.shape
        patch = kh == stride and kh > 1 and pad == 0            # patch embedding: the build contracts over kh*kw*ci at once
        ck = ci * kh * kw if patch else ci
        if ck % 4 == 0 and co > 32 and co % 4 == 0:
            x, w = _round_ste(x), _round_ste(w)
    return F.conv2d(x, w, P.get(pre + ".bias"), stride, pad, dil, groups)


def batch_norm(P, pre, x, ctx, eps=1e-5, momentum=0.1):
    """nn.BatchNorm2d semantics: batch statistics (biased var) in training, running stats in
    eval; running_var updated with the unbiased estimate."""
    w, b = P[pre + ".weight"], P[pre + ".bias"]
    if ctx.training:
        n = x.numel() // x.shape[1]
        if n <= 1:
            raise ValueError("Expected more than 1 value per channel when training")
        mean = x.mean(dim=(0, 2, 3))
        var = ((x - mean[None, :, None, None]) ** 2).mean(dim=(0, 2, 3))
        with torch.no_grad():
            rm = (1 - momentum) * P[pre + ".running_mean"] + momentum * mean
            rv = (1 - momentum) * P[pre + ".running_var"] + momentum * var * (n / (n - 1))
            ctx.new_stats[pre + ".running_mean"] = rm.detach()
            ctx.new_stats[pre + ".running_var"] = rv.detach()
    else:
        mean, var = P[pre + ".running_mean"], P[pre + ".running_var"]
    inv = torch.rsqrt(var + eps)
    return (x - mean[None, :, None, None]) * (inv * w)[None, :, None, None] + b[None, :, None, None]


def group_norm1(P, pre, x, eps=1e-5):
    """GroupNorm with ONE group (vr_coc.py:105-111): per-sample stats over C*H*W."""
    mean = x.mean(dim=(1, 2, 3), keepdim=True)
    var = ((x - mean) ** 2).mean(dim=(1, 2, 3), keepdim=True)
    xh = (x - mean) * torch.rsqrt(var + eps)
    return xh * P[pre + ".weight"][None, :, None, None] + P[pre + ".bias"][None, :, None, None]


def relu_site(ctx, key, z):
    """ReLU behind the BatchNorm `key`; mask-aware when ctx.forced_relu holds a mask for it."""
    forced = None if ctx.forced_relu is None else ctx.forced_relu.get(key)
    if forced is None:
        return torch.relu(z)
    forced = forced.to(z.device)
    own = z.detach() > 0
    diff = own != forced
    n = int(diff.sum())
    ctx.relu_report[key] = {"elements": z.numel(), "mismatch": n,
                            "max_abs": float(z.detach().abs()[diff].max()) if n else 0.0,
                            "scale": float(z.detach().abs().max())}
    # value = relu(z) exactly (never negative: a forced-active element whose own z is -1e-8 must not become THE
    # minimum of data_normal's batch-wide min, which would route the whole min-gradient through that one element);
    # gradient = the forced mask
    lin = z * forced.to(z.dtype)
    return lin + (torch.relu(z) - lin).detach()


def base_conv(P, pre, x, k, ctx, stride=1):
    """BaseConv (backbone/conv_utils/normal_conv.py:37-49): conv(no bias) -> BN(1e-3, .03) -> ReLU."""
    z = conv(P, pre + ".conv", x, stride, (k - 1) // 2)
    return relu_site(ctx, pre + ".bn", batch_norm(P, pre + ".bn", z, ctx, eps=1e-3, momentum=0.03))


def ds_base_conv(P, pre, x, ctx):
    """BaseConv(ds_conv=True) (normal_conv.py:23-33,43): depthwise 3x3 -> pointwise 1x1 -> BN -> ReLU."""
    c = x.shape[1]
    z = conv(P, pre + ".conv.dconv", x, 1, 1, 1, groups=c)
    z = conv(P, pre + ".conv.pconv", z)
    return relu_site(ctx, pre + ".bn", batch_norm(P, pre + ".bn", z, ctx, eps=1e-3, momentum=0.03))


def shuffle2(x):
    """2-group channel interleave, no-op for odd channel count (vr_coc.py:70-80, coc_fpn_dual.py:120-130)."""
    b, c, h, w = x.shape
    if c % 2:
        return x
    return x.view(b, 2, c // 2, h, w).transpose(1, 2).reshape(b, c, h, w)


def pool_windows(n):
    """AdaptiveAvgPool windows for output size 2 over n inputs."""
    return [(math.floor(i * n / 2), math.ceil((i + 1) * n / 2)) for i in range(2)]


def center_matrix(h, w, dtype):
    """(4, h*w) averaging matrix of the 2x2 adaptive-avg-pool centre proposals (vr_coc.py:150,168-169)."""
    q = torch.zeros(4, h * w, dtype=dtype)
    for i, (r0, r1) in enumerate(pool_windows(h)):
        for j, (c0, c1) in enumerate(pool_windows(w)):
            m = torch.zeros(h, w, dtype=dtype)
            m[r0:r1, c0:c1] = 1.0 / ((r1 - r0) * (c1 - c0))
            q[2 * i + j] = m.reshape(-1)
    return q


def cluster_core(f, v, alpha, beta, heads, fold, forced_idx=None, report=None):
    """The Context-Cluster token mixer between fc1/fc_v and fc2 (vr_coc.py:158-190, cos-sim :114-125).

    f, v: (B, heads*D, H, W).  Returns (out (B, heads*D, H, W), idx (B, heads, H, W) int64)."""
    B, ED, H, W = f.shape
    E, D = heads, ED // heads
    fh = fw = fold if fold > 1 else 1
    h, w = H // fh, W // fw
    assert h * fh == H and w * fw == W, "feature map not divisible by fold"

    def to_regions(t):   # (B,E*D,H,W) -> (R, N, D), R=(b e f1 f2), n = i*w + j
        return t.view(B, E, D, fh, h, fw, w).permute(0, 1, 3, 5, 4, 6, 2).reshape(B * E * fh * fw, h * w, D)

    fr, vr = to_regions(f), to_regions(v)
    Q = center_matrix(h, w, f.dtype)
    c = torch.einsum("mn,rnd->rmd", Q, fr)
    vc = torch.einsum("mn,rnd->rmd", Q, vr)
    fn = fr / fr.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    cn = c / c.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    sim = torch.sigmoid(beta + alpha * torch.einsum("rmd,rnd->rmn", cn, fn))     # (R,4,N)
    idx = sim.argmax(dim=1)                                                      # first max on ties
    if forced_idx is not None:
        fi = forced_idx.to(idx.device).long().view(B, E, fh, h, fw, w).permute(0, 1, 2, 4, 3, 5).reshape(-1, h * w)
        if report is not None:
            with torch.no_grad():
                best = sim.gather(1, idx[:, None, :]).squeeze(1)
                got = sim.gather(1, fi[:, None, :]).squeeze(1)
                diff = fi != idx
                report["mismatch"] = int(diff.sum())
                report["points"] = idx.numel()
                report["max_gap"] = float((best - got)[diff].max()) if diff.any() else 0.0
        idx = fi
    wgt = sim.gather(1, idx[:, None, :]).squeeze(1)                              # (R,N)
    onehot = F.one_hot(idx, 4).to(f.dtype)                                       # (R,N,4)
    cnt = onehot.sum(dim=1)                                                      # (R,4)
    agg = torch.einsum("rnm,rnd->rmd", onehot * wgt[:, :, None], vr)
    agg = (agg + vc) / (cnt[:, :, None] + 1.0)
    out = wgt[:, :, None] * agg.gather(1, idx[:, :, None].expand(-1, -1, D))      # (R,N,D)
    out = out.view(B, E, fh, fw, h, w, D).permute(0, 1, 6, 2, 4, 3, 5).reshape(B, ED, H, W)
    idx_map = idx.view(B, E, fh, fw, h, w).permute(0, 1, 2, 4, 3, 5).reshape(B, E, H, W)
    return out, idx_map


def cluster(P, pre, x, heads, fold, ctx):
    """Cluster.forward (vr_coc.py:155-192)."""
    v = conv(P, pre + ".fc_v", x)
    f = conv(P, pre + ".fc1", x)
    forced = None if ctx.forced_idx is None else ctx.forced_idx[pre]
    rep = ctx.idx_report.setdefault(pre, {})
    out, idx = cluster_core(f, v, P[pre + ".sim_alpha"], P[pre + ".sim_beta"], heads, fold, forced, rep)
    ctx.idx[pre] = idx
    return conv(P, pre + ".fc2", out)


def mlp(P, pre, x):
    """Mlp.forward (vr_coc.py:217-223): 1x1 -> exact-erf GELU -> 1x1."""
    return conv(P, pre + ".fc2", F.gelu(conv(P, pre + ".fc1", x)))


def cluster_block(P, pre, x, heads, fold, ctx):
    """ClusterBlock.forward (vr_coc.py:264-271 / vision/context_cluster.py:237-244)."""
    ls1 = P[pre + ".layer_scale_1"][None, :, None, None]
    ls2 = P[pre + ".layer_scale_2"][None, :, None, None]
    x = x + ls1 * cluster(P, pre + ".token_mixer", group_norm1(P, pre + ".norm1", x), heads, fold, ctx)
    x = x + ls2 * mlp(P, pre + ".mlp", group_norm1(P, pre + ".norm2", x))
    return x


def shuffle_attention(P, pre, x, G):
    """ShuffleAttention.forward (backbone/attention_modules/shuffle_attention.py:48-72)."""
    b, c, h, w = x.shape
    cp = c // (2 * G)
    xg = x.reshape(b * G, c // G, h, w)
    x0, x1 = xg[:, :cp], xg[:, cp:]
    gate0 = torch.sigmoid(P[pre + ".cweight"] * x0.mean(dim=(2, 3), keepdim=True) + P[pre + ".cbias"])
    y0 = x0 * gate0
    mu = x1.mean(dim=(2, 3), keepdim=True)
    var = ((x1 - mu) ** 2).mean(dim=(2, 3), keepdim=True)
    xn = (x1 - mu) * torch.rsqrt(var + 1e-5)
    xn = xn * P[pre + ".gn.weight"][None, :, None, None] + P[pre + ".gn.bias"][None, :, None, None]
    y1 = x1 * torch.sigmoid(P[pre + ".sweight"] * xn + P[pre + ".sbias"])
    out = torch.cat([y0, y1], dim=1).reshape(b, c, h, w)
    return shuffle2(out)


def eca_kernel_size(channels):
    """eca.py:9-10"""
    k = int(abs((math.log(channels, 2) + 1) / 2))
    return k if k % 2 else k + 1


def eca(P, pre, x):
    """eca_block.forward (backbone/attention_modules/eca.py:16-22)."""
    wk = P[pre + ".conv.weight"]                       # (1,1,k)
    k = wk.shape[-1]
    g = x.mean(dim=(2, 3))                             # (B,C)
    g = F.conv1d(g[:, None, :], wk, padding=(k - 1) // 2)[:, 0, :]
    return x * torch.sigmoid(g)[:, :, None, None]


def image_enhance_by_radar(P, pre, image, radar, ctx):
    """ImageEnhanceByRadar.forward + data_normal (vr_coc.py:312-316, 59-67).  The `d_min < 0`
    branch is dead (ReLU output), the min/max span the whole batch tensor."""
    p = base_conv(P, pre + ".radar_projection", radar, 3, ctx)
    n = (p - p.min()) / (p.max() - p.min())
    return batch_norm(P, pre + ".norm", (1 + n) * image, ctx)


def radar_enhance_by_image(P, pre, image, radar, ctx, initial=False):
    """RadarEnhanceByImage.forward (vr_coc.py:331-359)."""
    a = image if initial else shuffle_attention(P, pre + ".image_attn", image, 4)
    u = eca(P, pre + ".channel_attn", shuffle2(torch.cat([a, radar], dim=1)))
    z = base_conv(P, pre + ".inverse_projection", u, 1, ctx)
    return batch_norm(P, pre + ".norm", z + radar, ctx)


def bilinear_up(x, scale):
    """nn.Upsample(scale_factor, 'bilinear', align_corners=True) (coc_fpn_dual.py:21)."""
    return F.interpolate(x, scale_factor=scale, mode="bilinear", align_corners=True)


def coc_upsample(P, pre, x, scale, ctx):
    """CoCUpsample.forward (coc_fpn_dual.py:24-26)."""
    return bilinear_up(base_conv(P, pre + ".upsample.0", x, 1, ctx), scale)


def coc_conv(P, pre, x, ctx):
    """CoC_Conv.forward (coc_fpn_dual.py:36-39): neck ClusterBlock -> BaseConv 1x1."""
    x = cluster_block(P, pre + ".coc", x, NECK_HEADS, NECK_FOLD, ctx)
    return base_conv(P, pre + ".conv_att", x, 1, ctx)


def aspp(P, pre, x, ctx):
    """ASPP.forward (coc_fpn_dual.py:79-104)."""
    def branch(name, d):
        z = conv(P, f"{pre}.{name}.0", x, 1, 0 if d == 0 else d, max(d, 1))
        return relu_site(ctx, f"{pre}.{name}.1", batch_norm(P, f"{pre}.{name}.1", z, ctx))
    outs = [branch("branch1", 0), branch("branch2", 6), branch("branch3", 12), branch("branch4", 18)]
    g = x.mean(dim=(2, 3), keepdim=True)
    g = relu_site(ctx, pre + ".branch5_bn", batch_norm(P, pre + ".branch5_bn", conv(P, pre + ".branch5_conv", g), ctx))
    outs.append(g.expand(-1, -1, x.shape[2], x.shape[3]))      # bilinear from 1x1, align_corners: constant
    z = conv(P, pre + ".conv_cat.0", torch.cat(outs, dim=1))
    return relu_site(ctx, pre + ".conv_cat.1", batch_norm(P, pre + ".conv_cat.1", z, ctx))


# ----------------------------------------------------------------------------- assembly
def backbone(P, pre, x, r, width, ctx):
    """VRCoC.forward = forward_embeddings + forward_tokens (vr_coc.py:575-675)."""
    x = conv(P, pre + ".image_initial.proj", x)
    r = conv(P, pre + ".radar_initial.proj", r)
    x = image_enhance_by_radar(P, pre + ".image_enhance_by_radar1", x, r, ctx)
    r = radar_enhance_by_image(P, pre + ".radar_enhance_by_image1", x, r, ctx, initial=True)
    pos = P[pre + ".fea_pos"].permute(2, 0, 1)[None].expand(x.shape[0], -1, -1, -1).to(x.dtype)
    x = conv(P, pre + ".patch_embed.proj", torch.cat([x, pos], dim=1), stride=4)
    r = conv(P, pre + ".patch_embed_radar.proj", torch.cat([r, pos], dim=1), stride=4)   # fea_pos, not fea_pos_r (:585)
    outs, outs_r = [], []
    for i in range(4):
        for j in range(STAGE_BLOCKS[i]):
            x = cluster_block(P, f"{pre}.network.{3 * i}.{j}", x, STAGE_HEADS[i], STAGE_FOLD[i], ctx)
            r = cluster_block(P, f"{pre}.network_radar.{3 * i}.{j}", r, STAGE_HEADS[i], STAGE_FOLD[i], ctx)
        x = image_enhance_by_radar(P, f"{pre}.network.{3 * i + 1}", x, r, ctx)
        r = radar_enhance_by_image(P, f"{pre}.network_radar.{3 * i + 1}", x, r, ctx)
        if i in (0, 3):
            outs.append(x)
            outs_r.append(r)
        if i < 3:
            x = conv(P, f"{pre}.network.{3 * i + 2}.proj", x, stride=2, pad=1)
            r = conv(P, f"{pre}.network_radar.{3 * i + 2}.proj", r, stride=2, pad=1)
            if i < 2:        # taps after reducers 0 and 1 only (vr_coc.py:611-614, 632-635, 651-657)
                outs.append(x)
                outs_r.append(r)
    return outs, outs_r


def neck(P, pre, x, r, width, ctx):
    """CoCFpnDual.forward (coc_fpn_dual.py:184-224)."""
    (x2, x3, x4, x5), (r2, r3, r4, r5) = backbone(P, pre + ".backbone", x, r, width, ctx)
    ctx.taps.update(x2=x2, x3=x3, x4=x4, x5=x5, r2=r2, r3=r3, r4=r4, r5=r5)
    x5 = aspp(P, pre + ".aspp", x5, ctx)
    t = torch.cat([x4, coc_upsample(P, pre + ".upsample5_4", x5, 2, ctx)], dim=1)
    t = shuffle_attention(P, pre + ".sc_attn_seg4", shuffle2(t), 8)
    t = torch.cat([coc_upsample(P, pre + ".upsample4_3", t, 2, ctx), x3], dim=1)
    t = shuffle_attention(P, pre + ".sc_attn_seg3", shuffle2(t), 8)
    t = torch.cat([coc_upsample(P, pre + ".upsample3_2", t, 2, ctx), x2], dim=1)
    t = shuffle_attention(P, pre + ".sc_attn_seg2", shuffle2(t), 8)
    seg = coc_upsample(P, pre + ".upsample2_0", t, 4, ctx)
    p5 = coc_conv(P, pre + ".p5_out_det", r5, ctx)
    p4 = coc_conv(P, pre + ".p4_out_det", torch.cat([r4, coc_upsample(P, pre + ".p5_4_det", p5, 2, ctx)], dim=1), ctx)
    p3 = coc_conv(P, pre + ".p3_out_det", torch.cat([r3, coc_upsample(P, pre + ".p4_3_det", p4, 2, ctx)], dim=1), ctx)
    return (p3, p4, p5), seg


def head(P, pre, feats, ctx):
    """DecoupleHead.forward (head/decouplehead.py:42-88)."""
    outs = []
    for k, x in enumerate(feats):
        x = base_conv(P, f"{pre}.stems.{k}", x, 1, ctx)
        c = ds_base_conv(P, f"{pre}.cls_convs.{k}.1", ds_base_conv(P, f"{pre}.cls_convs.{k}.0", x, ctx), ctx)
        g = ds_base_conv(P, f"{pre}.reg_convs.{k}.1", ds_base_conv(P, f"{pre}.reg_convs.{k}.0", x, ctx), ctx)
        outs.append(torch.cat([conv(P, f"{pre}.reg_preds.{k}", g), conv(P, f"{pre}.obj_preds.{k}", g),
                               conv(P, f"{pre}.cls_preds.{k}", c)], dim=1))
    return outs


def forward(P, x, x_radar, phi="nano", training=True, forced_idx=None, forced_relu=None):
    """EfficientVRNet.forward (nets/efficient_vrnet.py:24-27).  Returns (det list[3], seg, ctx)."""
    ctx = Ctx(training, forced_idx, forced_relu)
    width = WIDTH[phi]
    feats, seg = neck(P, "backbone", x, x_radar, width, ctx)
    det = head(P, "head", feats, ctx)
    return det, seg, ctx


def synthetic_loss(det, seg):
    """Fixed scalar used to drive backward in parity tests and the benchmark."""
    return sum((d * d).mean() for d in det) + (seg * seg).mean()
