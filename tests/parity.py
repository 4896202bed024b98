"""Flip-aware parity of the HIP hot path against the CPU oracle (test infrastructure).

The Cluster op assigns every point to its arg-max centre (vr_coc.py:173-176).  Two correct fp32
implementations with different summation orders disagree on a handful of numerically tied points,
and one flipped point changes downstream features by O(1) (measured: the oracle in fp64 vs fp32
flips 0.07 % of points at 512 px and moves det outputs by 7 %).  Parity is therefore stated as:

 1. teacher forcing: the oracle re-runs the forward with the HIP path's assignment maps; every
    point where the oracle's own arg-max differs must be a near-tie (similarity gap < 1e-4 under
    the oracle's arithmetic), and such points must be rare (< 0.1 %);
 2. conditional on the assignments, det maps and seg logits agree within 1e-3 relative
    (the north star's tolerance);
 2b. the ReLU masks behind BatchNorm are the path's other discontinuity (at tiny fixture sizes a BatchNorm+ReLU layer
    sees 32-64 values per channel, and one pre-activation within rounding of zero moves a weight-gradient row by
    ~1/32 when its mask bit differs between two correct implementations).  They are handled the same way as the
    arg-max: the oracle is teacher-forced with the HIP path's masks, every element where the oracle's own (z > 0)
    differs must be within rounding of zero (|z| < 1e-4 x the layer's largest |z|) and such elements must be rare
    (< 0.01 %).  No seed is hand-picked to avoid them;
 3. gradients are compared with the oracle evaluated in fp64 (the exact gradient for those
    assignments).  Several of them are ill-conditioned in fp32 (BatchNorm over a few dozen values,
    BN-invariant directions): the oracle's OWN fp32 evaluation deviates from fp64 by up to a few
    per cent on small maps, and the reference's fp32 backward by up to 25 % at 512 px
    (tests/test_oracle_golden.py).  The bar is therefore: error vs fp64 <= max(5e-3, 3 x the error of
    the fp32 oracle vs fp64) -- i.e. as accurate as the reference arithmetic itself.
"""
import torch

import asy_vrnet_amd as A
from oracle import vrnet_oracle as O


def rel_err(a, b, floor=1e-6):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(floor)).item()


def hip_relu_masks(model):
    """{BatchNorm prefix: (B,C,H,W) bool cpu} from the last HIP forward (model.record_relu_masks = True)."""
    return {k: v.permute(0, 3, 1, 2).contiguous().cpu() for k, v in model._last_relu_masks.items()}


def hip_idx_maps(model):
    """{oracle cluster prefix: (B,E,H,W) int64 cpu} from the last HIP forward."""
    return {k: v.permute(0, 3, 1, 2).contiguous().cpu().long() for k, v in model._last_idx_maps.items()}


def compare_with_oracle(model, batch, size, iseed, check_grads=True, oracle_dtype=torch.float32, tol=1e-3,
                        gtol=5e-3, conv_round=None, per_param=False):
    """per_param=True keeps rep["_per_param"]: (parameter, gradient error) in forward order (tools/parity_dbg.py)."""
    O.OPERAND_ROUND = conv_round
    try:
        rep = _compare_with_oracle(model, batch, size, iseed, check_grads, oracle_dtype, tol, gtol)
        if not per_param:
            rep.pop("_per_param", None)
        return rep
    finally:
        O.OPERAND_ROUND = None


def _compare_with_oracle(model, batch, size, iseed, check_grads, oracle_dtype, tol, gtol):
    dev = next(model.parameters()).device
    x, r = A.synthetic_inputs(batch, size, iseed)
    sd0 = {k: v.detach().clone().cpu() for k, v in model.state_dict().items()}
    xg, rg = x.to(dev).requires_grad_(check_grads), r.to(dev).requires_grad_(check_grads)
    model.zero_grad(set_to_none=True)
    model.record_relu_masks = True
    try:
        det, seg = model(xg, rg)
    finally:
        model.record_relu_masks = False
    if check_grads:
        O.synthetic_loss(det, seg).backward()
    torch.cuda.synchronize()
    forced = hip_idx_maps(model)
    masks = hip_relu_masks(model)
    pnames = {k for k, _ in model.named_parameters()}
    P = {}
    for k, v in sd0.items():
        t = v.detach().clone().to(oracle_dtype) if v.dtype.is_floating_point else v
        if check_grads and k in pnames and t.numel():
            t.requires_grad_(True)
        P[k] = t
    xo = x.detach().clone().to(oracle_dtype).requires_grad_(check_grads)
    ro = r.detach().clone().to(oracle_dtype).requires_grad_(check_grads)
    det_o, seg_o, ctx = O.forward(P, xo, ro, model.phi, model.training, forced_idx=forced, forced_relu=masks)
    rep = {"flips": sum(v.get("mismatch", 0) for v in ctx.idx_report.values()),
           "points": sum(v.get("points", 0) for v in ctx.idx_report.values()),
           "max_gap": max([v.get("max_gap", 0.0) for v in ctx.idx_report.values()] + [0.0])}
    assert len(ctx.relu_report) == len(masks) > 0, "every BatchNorm+ReLU site of the oracle must have been teacher-forced"
    rep["relu_flips"] = sum(v["mismatch"] for v in ctx.relu_report.values())
    rep["relu_elements"] = sum(v["elements"] for v in ctx.relu_report.values())
    rep["relu_max_rel"] = max(v["max_abs"] / max(v["scale"], 1e-30) for v in ctx.relu_report.values())
    rep["det_err"] = max(rel_err(a, b) for a, b in zip(det, det_o))
    rep["seg_err"] = rel_err(seg, seg_o)
    ok = rep["det_err"] < tol and rep["seg_err"] < tol and rep["max_gap"] < 1e-4 and \
        rep["flips"] <= max(3, rep["points"] // 1000) and rep["relu_max_rel"] < 1e-4 and \
        rep["relu_flips"] <= max(3, rep["relu_elements"] // 10000)
    if model.training:
        sd1 = model.state_dict()
        rep["stat_err"] = max(rel_err(sd1[k], v) for k, v in ctx.new_stats.items())
        ok = ok and rep["stat_err"] < tol
    if check_grads:
        O.synthetic_loss(det_o, seg_o).backward()
        # the same computation in the reference's precision (fp32), same assignments: the noise yardstick
        P32 = {}
        for k, v in sd0.items():
            t = v.detach().clone()
            if k in pnames and t.numel():
                t.requires_grad_(True)
            P32[k] = t
        x32, r32 = x.detach().clone().requires_grad_(True), r.detach().clone().requires_grad_(True)
        det32, seg32, _ = O.forward(P32, x32, r32, model.phi, model.training, forced_idx=forced, forced_relu=masks)
        O.synthetic_loss(det32, seg32).backward()
        rep["dx_err"] = rel_err(xg.grad, xo.grad)
        rep["dr_err"] = rel_err(rg.grad, ro.grad)
        ref_in = max(rel_err(x32.grad, xo.grad), rel_err(r32.grad, ro.grad))
        worst, worst_k, ref_worst = 0.0, None, 0.0
        per_param = rep.setdefault("_per_param", [])       # (name, error) in forward order; diagnostic
        gmax = max(float(P[k].grad.abs().max()) for k in pnames if P[k].numel() and P[k].grad is not None)
        for k, p in model.named_parameters():
            if p.numel() == 0:
                continue
            go = P[k].grad
            assert p.grad is not None, f"no gradient for {k}"
            # floor: gradients that are analytically ~0 hold rounding noise on both sides
            e = rel_err(p.grad, go, floor=1e-4 * gmax)
            ref_worst = max(ref_worst, rel_err(P32[k].grad, go, floor=1e-4 * gmax))
            per_param.append((k, e))
            if e > worst:
                worst, worst_k = e, k
        rep["grad_err"], rep["grad_worst"], rep["ref32_grad_err"], rep["ref32_in_err"] = worst, worst_k, ref_worst, ref_in
        ok = ok and max(rep["dx_err"], rep["dr_err"]) < max(gtol, 3 * ref_in) and worst < max(gtol, 3 * ref_worst)
    rep["ok"] = bool(ok)
    return rep


def load_reference_decisions(golden_dir, name):
    """The reference's own hard decisions for a golden case (tools/make_golden.py, class Discontinuities):
    ({cluster prefix: (B,E,H,W) uint8 assignment}, {BatchNorm prefix: (B,C,H,W) bool: output > 0})."""
    import json
    import os
    import numpy as np
    shapes = json.load(open(os.path.join(golden_dir, name + "_decisions.json")))
    zi = np.load(os.path.join(golden_dir, name + "_decisions.npz"))
    zb = np.load(os.path.join(golden_dir, name + "_relu_masks.npz"))
    idx, masks = {}, {}
    for k in zi.files:
        shp = shapes[k]
        n = int(np.prod(shp))
        p = zi[k]
        v = np.stack([p & 3, (p >> 2) & 3, (p >> 4) & 3, (p >> 6) & 3], 1).reshape(-1)[:n]
        idx[k[2:]] = torch.from_numpy(v.reshape(shp).astype(np.uint8))
    for k in zb.files:
        shp = shapes[k]
        masks[k[2:]] = torch.from_numpy(np.unpackbits(zb[k])[:int(np.prod(shp))].reshape(shp).astype(bool))
    return idx, masks
