"""BASELINE.json configs[1] (phi=l, 512x512 image + 4-channel radar, fp32) on the GPU.

 * against the oracle at the full model/resolution with a batch the CPU finishes in seconds (bs=2, fp64 oracle);
 * at the full batch (bs=8) through size-independent properties of forward+backward: bit-exact repeatability
   (every reduction on the path is order-fixed: slab split-K, fixed chunk trees), linearity of the backward pass
   in the upstream gradient, equivariance under a permutation of the batch (to rounding in training mode, where BatchNorm sums
   over the batch; exactly in eval mode).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def A():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import asy_vrnet_amd
    return asy_vrnet_amd


@pytest.fixture(scope="module")
def net(A):
    m = A.EfficientVRNet(4, 9, "l", img_size=512).cuda()
    A.randomize_state_dict(m.state_dict(), seed=2)
    return m


def rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-12)).item()


def run(net, x, r, gdet, gseg):
    """forward + backward with upstream gradients (gdet, gseg); returns outputs and parameter gradients."""
    net.zero_grad(set_to_none=True)
    det, seg = net(x, r)
    torch.autograd.backward([*det, seg], [*gdet, gseg])
    grads = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
    return [d.detach().clone() for d in det], seg.detach().clone(), grads


def upstream(det, seg, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return ([torch.randn(d.shape, device="cuda", generator=g) / d.numel() for d in det],
            torch.randn(seg.shape, device="cuda", generator=g) / seg.numel())


def test_l_512_against_oracle(A):
    from tests.parity import compare_with_oracle
    m = A.EfficientVRNet(4, 9, "l", img_size=512).cuda().train()
    A.randomize_state_dict(m.state_dict(), seed=21)
    rep = compare_with_oracle(m, 2, 512, iseed=5, check_grads=True, oracle_dtype=torch.float64)
    print(rep)
    assert rep["ok"], rep


def test_bs8_train_step_is_repeatable_linear_and_permutation_equivariant(A, net):
    net.train()
    x, r = A.synthetic_inputs(8, 512, 11, "cuda")
    bn_state = {k: v.clone() for k, v in net.state_dict().items()}

    def fresh():
        net.load_state_dict(bn_state)           # running statistics back to the same start

    with torch.no_grad():
        fresh()
        d0, s0 = net(x, r)
    g1d, g1s = upstream(d0, s0, 1)
    g2d, g2s = upstream(d0, s0, 2)
    fresh(); det_a, seg_a, ga = run(net, x, r, g1d, g1s)
    sd_a = {k: v.clone() for k, v in net.state_dict().items()}
    fresh(); det_b, seg_b, gb = run(net, x, r, g1d, g1s)
    # 1. repeatable, bit for bit (outputs, gradients, BatchNorm running statistics)
    assert torch.equal(seg_a, seg_b) and all(torch.equal(p, q) for p, q in zip(det_a, det_b))
    assert ga.keys() == gb.keys() and all(torch.equal(ga[k], gb[k]) for k in ga)
    assert all(torch.equal(v, sd_a[k]) for k, v in net.state_dict().items())
    # 2. the backward pass is linear in the upstream gradient
    fresh(); _, _, g2 = run(net, x, r, g2d, g2s)
    fresh(); _, _, g12 = run(net, x, r, [2.0 * a - 0.5 * b for a, b in zip(g1d, g2d)], 2.0 * g1s - 0.5 * g2s)
    errs = {k: rel(g12[k], 2.0 * ga[k] - 0.5 * g2[k]) for k in ga if ga[k].abs().max() > 0}
    bad = sorted(((e, k, float(ga[k].abs().max()), float(g2[k].abs().max())) for k, e in errs.items() if e > 1e-3), reverse=True)
    print("linearity outliers:", bad[:12], len(bad), len(errs))
    num = sum(float(((g12[k].double() - (2.0 * ga[k].double() - 0.5 * g2[k].double())) ** 2).sum()) for k in ga)
    den = sum(float((g12[k].double() ** 2).sum()) for k in ga)
    print("linearity aggregate", (num / den) ** 0.5)
    assert (num / den) ** 0.5 < 1e-4
    # 3. permuting the batch permutes the outputs and leaves the parameter gradients alone (BatchNorm statistics
    #    and weight gradients are sums over the batch: only the summation order moves)
    perm = torch.tensor([5, 2, 7, 0, 3, 6, 1, 4], device="cuda")
    fresh(); det_p, seg_p, gp = run(net, x[perm].contiguous(), r[perm].contiguous(), [g[perm].contiguous() for g in g1d], g1s[perm].contiguous())
    assert rel(seg_p, seg_a[perm]) < 1e-3 and all(rel(p, q[perm]) < 1e-3 for p, q in zip(det_p, det_a)), \
        (rel(seg_p, seg_a[perm]), [rel(p, q[perm]) for p, q in zip(det_p, det_a)])
    num = sum(float(((gp[k].double() - ga[k].double()) ** 2).sum()) for k in ga)
    den = sum(float((ga[k].double() ** 2).sum()) for k in ga)
    assert (num / den) ** 0.5 < 1e-3, (num / den) ** 0.5


def test_bs8_eval_is_permutation_equivariant_bit_for_bit(A, net):
    """Eval mode: the only cross-image term left is the batch-wide min/max of ImageEnhanceByRadar / RadarEnhanceByImage
    (vr_coc.py:60-64 normalises over the whole tensor, so the reference itself couples the images of a batch and a
    bs=8 pass is NOT two bs=4 passes).  min and max are exact and order-free, every other reduction stays inside
    one image, so permuting the batch must permute the outputs exactly."""
    net.eval()
    x, r = A.synthetic_inputs(8, 512, 12, "cuda")
    perm = torch.tensor([3, 6, 0, 5, 1, 7, 4, 2], device="cuda")
    with torch.no_grad():
        det8, seg8 = net(x, r)
        detp, segp = net(x[perm].contiguous(), r[perm].contiguous())
    assert torch.equal(segp, seg8[perm])
    assert all(torch.equal(a, b[perm]) for a, b in zip(detp, det8))
