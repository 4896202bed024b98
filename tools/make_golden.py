"""Generates tests/golden/*.npz by running the upstream reference (PyTorch, CPU) in the build
container.  Only arrays leave this script: inputs and parameters are reproduced from seeds
by asy_vrnet_amd.init_utils (name-keyed numpy RNG), so fixtures hold just the reference's
outputs, gradients and updated BN statistics.

    python tools/make_golden.py            # writes tests/golden/

Requires /root/reference (not present on the GPU box; never run there).
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

from ref_harness import build_reference_model, load_reference  # noqa: E402
import asy_vrnet_amd as A  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
torch.manual_seed(0)


def loss_of(det, seg):
    return sum((d * d).mean() for d in det) + (seg * seg).mean()


FULL_GRAD_KEYS = [
    "backbone.backbone.network.0.0.token_mixer.sim_alpha",
    "backbone.backbone.network.0.0.token_mixer.sim_beta",
    "backbone.backbone.network_radar.6.3.token_mixer.sim_alpha",
    "backbone.backbone.network_radar.6.3.token_mixer.sim_beta",
    "backbone.backbone.network.0.1.layer_scale_1",
    "backbone.backbone.network.9.1.layer_scale_2",
    "backbone.backbone.network.3.0.norm1.weight",
    "backbone.backbone.network.3.0.token_mixer.fc1.weight",
    "backbone.backbone.network.3.0.mlp.fc2.bias",
    "backbone.backbone.network_radar.4.image_attn.cweight",
    "backbone.backbone.network_radar.4.image_attn.sbias",
    "backbone.backbone.network_radar.4.image_attn.gn.weight",
    "backbone.backbone.network_radar.4.channel_attn.conv.weight",
    "backbone.backbone.network_radar.1.inverse_projection.conv.weight",
    "backbone.backbone.network.1.radar_projection.conv.weight",
    "backbone.backbone.network.1.norm.weight",
    "backbone.backbone.network.2.proj.weight",
    "backbone.backbone.patch_embed.proj.weight",
    "backbone.backbone.patch_embed_radar.proj.bias",
    "backbone.backbone.image_initial.proj.weight",
    "backbone.backbone.radar_initial.proj.weight",
    "backbone.backbone.image_enhance_by_radar1.radar_projection.conv.weight",
    "backbone.backbone.radar_enhance_by_image1.inverse_projection.conv.weight",
    "backbone.backbone.radar_enhance_by_image1.channel_attn.conv.weight",
    "backbone.aspp.branch3.0.weight",
    "backbone.aspp.branch5_conv.weight",
    "backbone.aspp.conv_cat.1.weight",
    "backbone.sc_attn_seg3.sweight",
    "backbone.sc_attn_seg4.cbias",
    "backbone.upsample2_0.upsample.0.conv.weight",
    "backbone.p4_out_det.coc.token_mixer.sim_alpha",
    "backbone.p3_out_det.coc.layer_scale_1",
    "backbone.p3_out_det.conv_att.bn.bias",
    "head.stems.1.conv.weight",
    "head.cls_convs.0.1.conv.dconv.weight",
    "head.reg_convs.2.0.conv.pconv.weight",
    "head.cls_preds.1.weight",
    "head.obj_preds.0.bias",
]
STAT_KEYS = [
    "backbone.backbone.network.1.radar_projection.bn",
    "backbone.backbone.network_radar.10.norm",
    "backbone.aspp.branch5_bn",
    "backbone.backbone.image_enhance_by_radar1.norm",
    "head.stems.0.bn",
]


class Discontinuities:
    """Records the reference's two kinds of hard decisions during one forward pass, without restating its code:
    the Cluster arg-max (vr_coc.py:173-176: `mask.scatter_(1, sim_max_idx, 1.)` -- the index tensor of that scatter_ call
    is captured while a Cluster module is running) and the sign of every BatchNorm2d output (the ReLU masks of the
    conv -> BN -> ReLU sites are a subset of them, selected by name in the tests)."""

    def __init__(self, model):
        self.idx, self.bn_pos, self.cur = {}, {}, None
        self.handles = []
        for name, mod in model.named_modules():
            if hasattr(mod, "sim_alpha") and hasattr(mod, "fc_v"):
                self.handles.append(mod.register_forward_pre_hook(lambda m, a, n=name: setattr(self, "cur", (n, m, a[0].shape))))
                self.handles.append(mod.register_forward_hook(lambda m, a, o: setattr(self, "cur", None)))
            elif isinstance(mod, torch.nn.BatchNorm2d):
                self.handles.append(mod.register_forward_hook(lambda m, a, o, n=name: self.bn_pos.__setitem__(n, (o > 0).numpy())))
        self.orig = torch.Tensor.scatter_
        rec = self

        def scatter_(t, dim, index, *a, **k):
            if rec.cur is not None and dim == 1 and index.dtype == torch.int64:
                name, mod, shp = rec.cur
                B, _, H, W = shp
                E, f1, f2 = mod.heads, max(mod.fold_w, 1), max(mod.fold_h, 1)
                if not (mod.fold_w > 1 and mod.fold_h > 1):
                    f1 = f2 = 1
                w, h = H // f1, W // f2
                m = index.reshape(B, E, f1, f2, w, h).permute(0, 1, 2, 4, 3, 5).reshape(B, E, H, W)     # (b e f1 f2)(w h) -> b e (f1 w)(f2 h)
                rec.idx[name] = m.to(torch.uint8).numpy()
            return rec.orig(t, dim, index, *a, **k)
        torch.Tensor.scatter_ = scatter_

    def close(self):
        torch.Tensor.scatter_ = self.orig
        for h in self.handles:
            h.remove()

    @staticmethod
    def relu_sites(phi, size):
        """Names of the BatchNorms that have a ReLU behind them: the keys the CPU oracle's relu_site() is called with on one
        forward pass (= the keys of the HIP path's recorded masks, tests/parity.py)."""
        from oracle import vrnet_oracle as O
        keys, orig = [], O.relu_site
        O.relu_site = lambda ctx, key, z: (keys.append(key), orig(ctx, key, z))[1]
        try:
            m = A.EfficientVRNet(4, 9, phi, img_size=size)
            A.randomize_state_dict(m.state_dict(), seed=1)
            x, r = A.synthetic_inputs(2, size, 1)
            with torch.no_grad():
                O.forward({k: v.clone() for k, v in m.state_dict().items()}, x, r, phi, True)
        finally:
            O.relu_site = orig
        return set(keys)

    def arrays(self, relu_keys):
        """2 bits per assignment, 1 bit per BatchNorm output sign (only the BatchNorms with a ReLU behind them); shapes in
        the meta record."""
        arrs, shapes = {}, {}
        self.bn_pos = {k: v for k, v in self.bn_pos.items() if k in relu_keys}
        assert len(self.bn_pos) == len(relu_keys), (sorted(relu_keys - set(self.bn_pos)))
        for k, v in self.idx.items():
            assert v.max() < 4
            f = v.reshape(-1).astype(np.uint8)
            f = np.concatenate([f, np.zeros((-len(f)) % 4, np.uint8)]).reshape(-1, 4)
            arrs["i:" + k] = (f[:, 0] | (f[:, 1] << 2) | (f[:, 2] << 4) | (f[:, 3] << 6)).astype(np.uint8)
            shapes["i:" + k] = list(v.shape)
        for k, v in self.bn_pos.items():
            arrs["b:" + k] = np.packbits(v.reshape(-1))
            shapes["b:" + k] = list(v.shape)
        return arrs, shapes


def whole_net(name, phi, size, batch, training, pseed, iseed, seg_stride=1, with_grads=True, dtype=torch.float32, decisions=False):
    """dtype=torch.float64: the reference itself evaluated in double (`m.double()`), i.e. the exact outputs and
    gradients of the reference ALGORITHM -- the pin for the oracle's backward at the benchmark resolution, where
    the reference's fp32 backward is 1-25 % away from the exact gradient."""
    m = build_reference_model(phi=phi, img_size=size)
    A.randomize_state_dict(m.state_dict(), seed=pseed)
    m.train(training)
    x, r = A.synthetic_inputs(batch, size, iseed)
    if dtype != torch.float32:
        m, x, r = m.to(dtype), x.to(dtype), r.to(dtype)
    x.requires_grad_(with_grads)
    r.requires_grad_(with_grads)
    disc = Discontinuities(m) if decisions else None
    try:
        det, seg = m(x, r)
    finally:
        if disc is not None:
            disc.close()
    if disc is not None:      # the reference's hard decisions, in a file of their own: <name>_decisions.npz
        arrs, shapes = disc.arrays(Discontinuities.relu_sites(phi, min(size, 128)))
        # two files (each stays below 1.5 MB): the Cluster assignments, the ReLU masks
        np.savez_compressed(os.path.join(OUT, name + "_decisions.npz"), **{k: v for k, v in arrs.items() if k[0] == "i"})
        np.savez_compressed(os.path.join(OUT, name + "_relu_masks.npz"), **{k: v for k, v in arrs.items() if k[0] == "b"})
        with open(os.path.join(OUT, name + "_decisions.json"), "w") as f:
            json.dump(shapes, f)
        print(name + "_decisions", sum(a.nbytes for a in arrs.values()) // 1024, "KiB raw,", len(disc.idx), "Cluster maps,",
              len(disc.bn_pos), "BatchNorm sign maps")
    rec = {"det0": det[0], "det1": det[1], "det2": det[2], "seg": seg[:, :, ::seg_stride, ::seg_stride],
           "seg_sum": seg.double().sum(), "seg_abs_sum": seg.double().abs().sum()}
    meta = dict(phi=phi, size=size, batch=batch, training=training, pseed=pseed, iseed=iseed,
                seg_stride=seg_stride, dtype=str(dtype).replace("torch.", ""))
    if with_grads:
        loss_of(det, seg).backward()
        rec["loss"] = loss_of(det, seg)
        rec["dx"] = x.grad[:, :, ::seg_stride, ::seg_stride]
        rec["dr"] = r.grad[:, :, ::seg_stride, ::seg_stride]
        names, norms = [], []
        for k, p in m.named_parameters():
            if p.numel() == 0:
                continue
            names.append(k)
            norms.append(0.0 if p.grad is None else p.grad.double().norm().item())
        rec["grad_norms"] = np.asarray(norms)
        meta["grad_names"] = names
        pd = dict(m.named_parameters())
        for k in FULL_GRAD_KEYS:
            rec["g:" + k] = pd[k].grad
    if training:
        sd = m.state_dict()
        for k in STAT_KEYS:
            rec["s:" + k + ".running_mean"] = sd[k + ".running_mean"]
            rec["s:" + k + ".running_var"] = sd[k + ".running_var"]
    save(name, rec, meta)


def save(name, rec, meta):
    arrs = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in rec.items()}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrs)
    with open(os.path.join(OUT, name + ".json"), "w") as f:
        json.dump(meta, f, indent=0)
    print(name, sum(a.nbytes for a in arrs.values()) // 1024, "KiB raw")


def module_case(name, module, inputs, meta, seed=3):
    """Reference sub-module with name-keyed random parameters; loss = mean(out * G) with a fixed
    seeded projection G ~ N(0,1) (mean(out^2) is degenerate behind a trailing BatchNorm)."""
    A.randomize_state_dict(module.state_dict(), seed=seed)
    module.train(True)
    ins = [t.clone().requires_grad_(True) for t in inputs]
    out = module(*ins)
    (out * rnd(tuple(out.shape), 999)).mean().backward()
    rec = {"out": out}
    for i, t in enumerate(ins):
        rec[f"din{i}"] = t.grad
    for k, p in module.named_parameters():
        if p.numel() and p.grad is not None:
            rec["g:" + k] = p.grad
    meta = dict(meta, seed=seed)
    save(name, rec, meta)


def rnd(shape, seed, kind="normal"):
    rng = np.random.default_rng(seed)
    a = rng.standard_normal(shape, dtype=np.float32) if kind == "normal" else rng.random(shape, dtype=np.float32)
    return torch.from_numpy(a)


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = load_reference()
    if len(sys.argv) > 1 and sys.argv[1] == "fp64":        # only the double-precision 512 px case
        whole_net("net_nano_512_train_fp64", "nano", 512, 2, True, 3, 9, seg_stride=16, dtype=torch.float64)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "rect":        # only the rectangular case (round 5): H != W
        whole_net("net_nano_128x192_train", "nano", [128, 192], 2, True, 5, 11, seg_stride=2)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "decisions":   # only the two cases that also pin the reference's hard decisions
        whole_net("net_nano_128_train", "nano", 128, 2, True, 2, 8, decisions=True)
        whole_net("net_nano_512_train", "nano", 512, 2, True, 3, 9, seg_stride=8, decisions=True)
        return
    # ---- state_dict surface (names, shapes, dtypes) for nano and l
    for phi in ("nano", "l"):
        m = build_reference_model(phi=phi, img_size=512)
        surf = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m.state_dict().items()]
        with open(os.path.join(OUT, f"state_dict_surface_{phi}.json"), "w") as f:
            json.dump(surf, f)
        print(phi, len(surf), "keys")
    # ---- whole net
    whole_net("net_nano_64_train", "nano", 64, 2, True, 1, 7)
    whole_net("net_nano_64_eval", "nano", 64, 2, False, 1, 7, with_grads=False)
    whole_net("net_nano_128_train", "nano", 128, 2, True, 2, 8, decisions=True)
    whole_net("net_tiny_128_eval", "tiny", 128, 1, False, 4, 10, with_grads=False)
    whole_net("net_nano_512_train", "nano", 512, 2, True, 3, 9, seg_stride=8, decisions=True)
    whole_net("net_nano_512_train_fp64", "nano", 512, 2, True, 3, 9, seg_stride=16, dtype=torch.float64)
    whole_net("net_nano_128x192_train", "nano", [128, 192], 2, True, 5, 11, seg_stride=2)
    # ---- sub-modules (seeded inputs: shape, seed, kind recorded in meta)
    def case(name, mod, shapes, kinds=None, **meta):
        kinds = kinds or ["normal"] * len(shapes)
        ins = [rnd(s, 100 + i, k) for i, (s, k) in enumerate(zip(shapes, kinds))]
        module_case(name, mod, ins, dict(meta, shapes=[list(s) for s in shapes], kinds=kinds))

    CB = ref.vr.ClusterBlock
    case("mod_clusterblock_n256_d32", CB(dim=32, mlp_ratio=8, heads=4, head_dim=32, fold_w=2, fold_h=2),
         [(2, 32, 32, 32)], kind="clusterblock", dim=32, heads=4, head_dim=32, fold=2, mlp_ratio=8)
    case("mod_clusterblock_n64_d24", ref.vcc.ClusterBlock(dim=48), [(2, 48, 16, 16)],
         kind="clusterblock", dim=48, heads=4, head_dim=24, fold=2, mlp_ratio=4)
    case("mod_clusterblock_n1024_d24", ref.vcc.ClusterBlock(dim=16), [(1, 16, 64, 64)],
         kind="clusterblock", dim=16, heads=4, head_dim=24, fold=2, mlp_ratio=4)
    case("mod_clusterblock_fold1_d32", CB(dim=32, mlp_ratio=4, heads=8, head_dim=32, fold_w=1, fold_h=1),
         [(2, 32, 16, 16)], kind="clusterblock", dim=32, heads=8, head_dim=32, fold=1, mlp_ratio=4)
    case("mod_clusterblock_odd_d32", CB(dim=16, mlp_ratio=4, heads=2, head_dim=32, fold_w=2, fold_h=2),
         [(1, 16, 10, 14)], kind="clusterblock", dim=16, heads=2, head_dim=32, fold=2, mlp_ratio=4)
    case("mod_image_enhance", ref.vr.ImageEnhanceByRadar(radar_in_channels=32, image_in_channels=32),
         [(2, 32, 16, 16), (2, 32, 16, 16)], kind="image_enhance", c_img=32, c_rad=32)
    case("mod_image_enhance_in", ref.vr.ImageEnhanceByRadar(radar_in_channels=4, image_in_channels=3),
         [(2, 3, 32, 32), (2, 4, 32, 32)], kinds=["normal", "uniform"], kind="image_enhance", c_img=3, c_rad=4)
    case("mod_radar_enhance", ref.vr.RadarEnhanceByImage(radar_in_channels=32, image_in_channels=32),
         [(2, 32, 16, 16), (2, 32, 16, 16)], kind="radar_enhance", c_img=32, c_rad=32, initial=False)
    case("mod_radar_enhance_in", ref.vr.RadarEnhanceByImage(radar_in_channels=4, image_in_channels=3, initial=True),
         [(2, 3, 32, 32), (2, 4, 32, 32)], kind="radar_enhance", c_img=3, c_rad=4, initial=True)
    case("mod_shuffle_attention_g8", ref.sa.ShuffleAttention(channel=64, G=8), [(2, 64, 8, 8)],
         kind="shuffle_attention", channel=64, G=8)
    case("mod_shuffle_attention_g4", ref.sa.ShuffleAttention(channel=32, G=4), [(2, 32, 16, 16)],
         kind="shuffle_attention", channel=32, G=4)
    case("mod_eca_c64", ref.eca.eca_block(channel=64), [(2, 64, 8, 8)], kind="eca", channel=64)
    case("mod_aspp", ref.neck.ASPP(dim_in=32, dim_out=32), [(2, 32, 16, 16)], kind="aspp", dim=32)
    case("mod_coc_upsample_x2", ref.neck.CoCUpsample(32, 16), [(2, 32, 8, 8)], kind="coc_upsample", cin=32, cout=16, scale=2)
    case("mod_coc_upsample_x4", ref.neck.CoCUpsample(32, 9, scale=4), [(2, 32, 8, 8)], kind="coc_upsample", cin=32, cout=9, scale=4)
    case("mod_baseconv_ds", ref.nc.BaseConv(32, 32, 3, 1, ds_conv=True), [(2, 32, 8, 8)], kind="baseconv_ds", c=32)
    case("mod_reducer", ref.vr.PointRecuder(patch_size=3, stride=2, padding=1, in_chans=16, embed_dim=32),
         [(2, 16, 16, 16)], kind="reducer", cin=16, cout=32)


if __name__ == "__main__":
    main()
