"""bf16-operand mode against the oracle with the same operand rounding (test infrastructure)."""
import torch

from oracle import vrnet_oracle as O
from tests.parity import hip_idx_maps, rel_err


def bf16_report(A, m, phi, batch, size, iseed):
    """Runs `m` (training mode) with compute_dtype = "bf16" and the fp64 oracle with OPERAND_ROUND = "bf16",
    teacher-forced with the kernels' Cluster assignments; gradients are compared in aggregate (relative L2, cosine)."""
    m.compute_dtype = "bf16"
    x, r = A.synthetic_inputs(batch, size, iseed)
    sd0 = {k: v.detach().clone().cpu() for k, v in m.state_dict().items()}
    m.zero_grad(set_to_none=True)
    try:
        det, seg = m(x.cuda(), r.cuda())
        O.synthetic_loss(det, seg).backward()
    finally:
        m.compute_dtype = "f32"
    forced = hip_idx_maps(m)
    pn = {k for k, _ in m.named_parameters()}
    P = {k: (v.double().requires_grad_(k in pn and v.numel() > 0) if v.dtype.is_floating_point else v) for k, v in sd0.items()}
    O.OPERAND_ROUND = "bf16"
    try:
        det_o, seg_o, ctx = O.forward(P, x.double(), r.double(), phi, True, forced_idx=forced)
        O.synthetic_loss(det_o, seg_o).backward()
    finally:
        O.OPERAND_ROUND = None
    flips = sum(v.get("mismatch", 0) for v in ctx.idx_report.values())
    points = sum(v.get("points", 0) for v in ctx.idx_report.values())
    num = den = dot = n1 = 0.0
    for k, p in m.named_parameters():
        if p.numel() == 0 or P[k].grad is None:
            continue
        a, b = p.grad.double().cpu(), P[k].grad
        num += float(((a - b) ** 2).sum()); den += float((b ** 2).sum()); dot += float((a * b).sum()); n1 += float((a ** 2).sum())
    return {"flips": flips, "points": points, "det_err": max(rel_err(a, b) for a, b in zip(det, det_o)),
            "seg_err": rel_err(seg, seg_o), "grad_rel_l2": (num / den) ** 0.5, "grad_cos": dot / (n1 * den) ** 0.5}
