"""End to end: the hot path's forward/backward with the reference's training loss (YOLOLoss SimOTA + focal + dice,
det + 5 seg, utils_fit.py:96-106), the fused SGD of train.py:460-473 and ModelEMA -- a few real training steps on a
fixed synthetic batch must drive the loss down, in fp32 and with bf16-operand convolutions."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_training_steps_reduce_the_loss(dtype):
    import asy_vrnet_amd as A
    from asy_vrnet_amd import losses, optim
    from oracle import loss_oracle as LO
    B, S, NC, NS = 4, 128, 4, 9
    model = A.EfficientVRNet(NC, NS, "nano", img_size=(S, S)).cuda().train()
    A.randomize_state_dict(model.state_dict(), seed=5)
    model.compute_dtype = dtype
    yl = losses.YOLOLoss(NC).cuda()
    lr, _ = optim.fit_lr(B, 1e-2, 1e-4, "sgd")
    opt = optim.build_optimizer(model, "sgd", lr * 4, 0.937, 5e-4)
    ema = optim.ModelEMA(model)
    x, r = A.synthetic_inputs(B, S, 9, "cuda")
    labels, pngs, seg_labels = LO.synthetic_targets(B, S, NC, NS, seed=4, empty=(2,))
    pngs, seg_labels, w = pngs.cuda(), seg_labels.cuda(), torch.ones(NS, device="cuda")
    hist = []
    for it in range(12):
        opt.zero_grad()
        det, seg = model(x, r)
        total, ldet, lseg = losses.training_loss(yl, det, seg, labels, pngs, seg_labels, w, NS, True, True)
        total.backward()
        opt.step()
        ema.update(model)
        hist.append(float(total.item()))
        assert np.isfinite(hist[-1]), hist
    assert min(hist[-3:]) < 0.9 * hist[0], hist            # the same batch every step: the loss must come down
    assert ema.updates == 12
    d = sum(float((p.detach() - q.detach()).abs().sum()) for p, q in zip(model.parameters(), ema.ema.parameters()))
    assert d > 0                                            # the EMA lags the live weights
    with torch.no_grad():                                   # the EMA copy is a working model of its own
        model.eval()
        ema.ema.compute_dtype = dtype
        de, se = ema.ema(x, r)
        assert torch.isfinite(se).all() and all(torch.isfinite(t).all() for t in de)
