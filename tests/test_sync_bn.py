"""Synchronised BatchNorm (the reference's `sync_bn` option, train.py:356-357) through the real kernels on one GPU: the two
ranks of a world-size-2 job are played one after the other, a stub collective hands each the other's totals.  A rank's
half of the outputs, of dz and of the running statistics must equal plain BatchNorm over the concatenated batch; its
parameter gradients are the LOCAL sums (they add up to the full batch's)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


class TwoRankStub:
    """parallel.SyncBatchNormStats for one of two ranks: `others` = the other rank's totals in call order."""

    world = 2

    def __init__(self, others=()):
        self.others, self.seen = list(others), []

    def begin_forward(self, batch_local, device):
        pass

    def count(self, batch_local, per_sample, batch_total=None):
        return (batch_total if batch_total is not None else 2 * batch_local) * per_sample

    def total(self, mom):
        tot = mom.sum(0, keepdim=True)
        self.seen.append(tot.clone())
        return tot + self.others.pop(0) if self.others else tot


@pytest.mark.parametrize("relu", [True, False])
@pytest.mark.parametrize("shape", [(2, 8, 8, 32), (3, 5, 7, 48)])
def test_sync_bn_equals_full_batch(shape, relu):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import asy_vrnet_amd.program as program
    Bh, H, W, C = shape                                  # samples per rank
    dev = torch.device("cuda", torch.cuda.current_device())
    gen = torch.Generator().manual_seed(5)
    z_full = (torch.randn(2 * Bh, H, W, C, generator=gen) * 1.7 + 0.3).cuda()
    dy_full = torch.randn(2 * Bh, H, W, C, generator=gen).cuda()
    w0, b0 = torch.rand(C, generator=gen) + 0.5, torch.randn(C, generator=gen)

    def fresh_bn():
        bn = torch.nn.BatchNorm2d(C, eps=1e-3, momentum=0.03).cuda().train()
        with torch.no_grad():
            bn.weight.copy_(w0)
            bn.bias.copy_(b0)
        return bn

    def run(z, dy, stub):
        bn = fresh_bn()
        rt = program.RT(dev, True, True)
        rt.sync_bn = stub
        za = program.Act(z.contiguous())
        y, ms = program.bn_forward(rt, za, bn, relu=relu)
        dz = program.bn_backward(rt, bn, za, ms, dy.contiguous(), C, mask=y if relu else None)
        torch.cuda.synchronize()
        return y.t.clone(), dz.clone(), rt.pgrads[bn.weight].clone(), rt.pgrads[bn.bias].clone(), bn

    y_ref, dz_ref, gw_ref, gb_ref, bn_ref = run(z_full, dy_full, None)            # rank-local BatchNorm over the whole batch
    halves = [(z_full[:Bh], dy_full[:Bh]), (z_full[Bh:], dy_full[Bh:])]
    # pass 1: every rank's forward totals (they do not depend on the normalisation)
    fwd = []
    for z, dy in halves:
        st = TwoRankStub()
        run(z, dy, st)
        fwd.append(st.seen[0])
    # pass 2: with the global forward statistics in place, every rank's backward totals
    bwd = []
    for r, (z, dy) in enumerate(halves):
        st = TwoRankStub([fwd[1 - r]])
        run(z, dy, st)
        bwd.append(st.seen[1])
    # pass 3: the synchronised step of each rank
    gw_sum, gb_sum = 0, 0
    for r, (z, dy) in enumerate(halves):
        y, dz, gw, gb, bn = run(z, dy, TwoRankStub([fwd[1 - r], bwd[1 - r]]))
        sl = slice(r * Bh, (r + 1) * Bh)
        assert torch.allclose(y, y_ref[sl], rtol=1e-5, atol=1e-5)
        assert torch.allclose(dz, dz_ref[sl], rtol=1e-4, atol=1e-5)
        assert torch.allclose(bn.running_mean, bn_ref.running_mean, rtol=1e-6, atol=1e-7)
        assert torch.allclose(bn.running_var, bn_ref.running_var, rtol=1e-6, atol=1e-7)
        assert int(bn.num_batches_tracked) == 1
        gw_sum, gb_sum = gw_sum + gw, gb_sum + gb
    assert torch.allclose(gw_sum, gw_ref, rtol=1e-4, atol=1e-4)
    assert torch.allclose(gb_sum, gb_ref, rtol=1e-4, atol=1e-4)


def test_sync_bn_whole_net_world_one_equals_plain():
    """With one rank the synchronised statistics ARE the local ones: the whole net with model._sync_bn set (moments pass +
    coefficient kernels with the global count) must reproduce the default path (statistics from the conv epilogues)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import asy_vrnet_amd as A
    from asy_vrnet_amd.parallel import SyncBatchNormStats
    outs = []
    for sync in (False, True):
        m = A.EfficientVRNet(4, 9, "nano", img_size=128).cuda().train()
        A.randomize_state_dict(m.state_dict(), seed=4)
        m._sync_bn = SyncBatchNormStats() if sync else None
        g = torch.Generator().manual_seed(2)
        x, r = torch.rand(2, 3, 128, 128, generator=g).cuda(), torch.rand(2, 4, 128, 128, generator=g).cuda()
        det, seg = m(x, r)
        loss = sum((d * d).mean() for d in det) + (seg * seg).mean()
        loss.backward()
        outs.append(([d.detach().clone() for d in det], seg.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None},
                     {k: b.clone() for k, b in m.named_buffers() if "running" in k}))
    (d0, s0, g0, b0), (d1, s1, g1, b1) = outs
    for a, b in zip(d0 + [s0], d1 + [s1]):
        assert torch.allclose(a, b, rtol=2e-4, atol=2e-5)
    for k in b0:
        assert torch.allclose(b0[k], b1[k], rtol=1e-5, atol=1e-6), k
    worst = max(((g0[k] - g1[k]).norm() / g0[k].norm().clamp_min(1e-12)).item() for k in g0)
    assert worst < 5e-3, worst
