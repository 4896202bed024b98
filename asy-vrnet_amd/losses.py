"""Training losses on the hot path's outputs, with the reference's interfaces:
`YOLOLoss(num_classes, fp16, strides)` (nets/yolo_training.py:60-97), `CE_Loss`, `Focal_Loss`, `Dice_loss`
(nets/deeplabv3_training.py:9-59) and `training_loss` = loss_det + 5 * loss_seg (utils/utils_fit.py:96-106).

Each call is a handful of HIP launches computing the value AND the gradient w.r.t. the head outputs, with no host
synchronisation (the reference loops over images and boxes in Python and calls .item() per box).  The returned loss
is an autograd node over the inputs; unlike the reference's YOLOLoss (get_output_and_grid :108-110) the det maps are
not modified in place.  No CPU fallback."""
import torch
import torch.nn as nn

from . import hip


def _pack_labels(labels, device):
    """list of (n_i, 5) [cx, cy, w, h, cls] -> (B, Gmax, 5) float32 + counts (B,) int32 on `device`.  No host
    synchronisation: the counts come from the shapes; labels that already live on the GPU (utils_fit.py:43 moves them
    there) are packed by device-side slice copies, host labels go through ONE pinned staging buffer."""
    B = len(labels)
    counts = [int(l.shape[0]) if l is not None and l.numel() else 0 for l in labels]
    G = max(counts) if counts else 0
    on_gpu = any(l is not None and l.is_cuda for l in labels)
    if on_gpu:
        packed = torch.zeros((B, max(G, 1), 5), dtype=torch.float32, device=device)
        for b, l in enumerate(labels):
            if counts[b]:
                packed[b, :counts[b]] = l.detach().to(device, torch.float32).reshape(-1, 5)
    else:
        host = torch.zeros((B, max(G, 1), 5), dtype=torch.float32).pin_memory()
        for b, l in enumerate(labels):
            if counts[b]:
                host[b, :counts[b]] = l.detach().to(torch.float32).reshape(-1, 5)
        packed = host.to(device, non_blocking=True)
    cnt = torch.tensor(counts, dtype=torch.int32).pin_memory().to(device, non_blocking=True)
    return packed, cnt, G


def _gpu_maps(inputs):
    outs = [t if (t.is_contiguous() and t.dtype == torch.float32) else t.contiguous().float() for t in inputs]
    if not outs or not all(o.is_cuda and o.dim() == 4 and o.shape[:2] == outs[0].shape[:2] for o in outs):
        raise RuntimeError("expects a list of (B, C, h, w) GPU tensors of one batch")
    return outs


def _is_one(g):
    """True when the upstream gradient is known WITHOUT a device sync to be the scalar 1 (a python number or a host
    tensor); a device tensor is multiplied in (no sync)."""
    if isinstance(g, (int, float)):
        return g == 1
    return (not g.is_cuda) and g.numel() == 1 and float(g) == 1.0


class _YoloLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, strides, labels, counts, max_gt, scale, *inputs):
        lv = _gpu_maps([i.detach() for i in inputs])
        need = any(i.requires_grad for i in inputs)
        grads = [torch.empty_like(t) for t in lv] if need else None
        out = torch.empty(5, dtype=torch.float32, device=lv[0].device)
        hip.yolo_loss(lv, grads, strides, labels, counts, max_gt, scale, out)
        ctx.grads = grads
        ctx.stats = out
        return out[0] * scale if scale != 1.0 else out[0].clone()

    @staticmethod
    def backward(ctx, g):
        gs = ctx.grads
        ctx.grads = None
        # the kernel already scaled the gradients by `scale`; `g` is the upstream factor (1 for a plain sum of losses)
        return (None, None, None, None, None) + tuple(gs if _is_one(g) else [x * g for x in gs])


class YOLOLoss(nn.Module):
    """nets/yolo_training.py:60-72.  `log_vars` is kept (it is in the reference's parameter list, :72) although the
    reference's active code never uses it (:184-194 are commented out there)."""

    def __init__(self, num_classes, fp16=False, strides=(8, 16, 32)):
        super().__init__()
        self.num_classes = num_classes
        self.strides = list(strides)
        self.fp16 = fp16
        self.log_vars = nn.Parameter(torch.zeros(3))
        self.last_stats = None          # device tensor [loss, num_fg, sum iou, sum obj, sum cls] of the last call

    def forward(self, inputs, labels=None, _scale=1.0):
        if len(inputs) != len(self.strides) or inputs[0].shape[1] != 5 + self.num_classes:
            raise RuntimeError(f"YOLOLoss: expects {len(self.strides)} maps with {5 + self.num_classes} channels")
        packed, counts, G = _pack_labels(labels, inputs[0].device)
        loss = _YoloLossFn.apply(self.strides, packed, counts, G, float(_scale), *inputs)
        return loss

    @torch.no_grad()
    def assignments(self, inputs, labels):
        """(fg_mask (B,A) bool, matched_gt (B,A) int, pred_iou (B,A), stats[5]): the SimOTA result, for inspection."""
        lv = _gpu_maps([i.detach() for i in inputs])
        packed, counts, G = _pack_labels(labels, lv[0].device)
        B, A = lv[0].shape[0], sum(t.shape[2] * t.shape[3] for t in lv)
        fg = torch.empty((B, A), dtype=torch.uint8, device=lv[0].device)
        mg = torch.empty((B, A), dtype=torch.int32, device=lv[0].device)
        pi = torch.empty((B, A), dtype=torch.float32, device=lv[0].device)
        out = torch.empty(5, dtype=torch.float32, device=lv[0].device)
        hip.yolo_loss(lv, None, self.strides, packed, counts, G, 1.0, out, fg, mg, pi)
        return fg.bool(), mg, pi, out


class _SegLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, png, onehot, weights, focal, dice, main, alpha, gamma, beta, smooth, scale):
        xs = x.detach()
        xs = xs if (xs.is_contiguous() and xs.dtype == torch.float32) else xs.contiguous().float()
        if not xs.is_cuda or xs.dim() != 4:
            raise RuntimeError("seg loss: expects (B, C, H, W) GPU logits")
        B, C, H, W = xs.shape
        if main and tuple(png.shape) != (B, H, W):
            raise RuntimeError("seg loss: logits must already be at label size (the reference's bilinear resize branch, "
                               "deeplabv3_training.py:12-13, never runs on this net: the seg head outputs full resolution)")
        if main:
            png = png.to(xs.device, torch.int64).contiguous()
        if dice:
            onehot = onehot.to(xs.device, torch.float32).contiguous()
            if tuple(onehot.shape) != (B, H, W, C + 1):
                raise RuntimeError("dice loss: seg_labels must be (B, H, W, C+1) one-hot")
        if weights is not None:
            weights = weights.to(xs.device, torch.float32).contiguous()
        dx = torch.empty_like(xs) if x.requires_grad else None
        out = torch.empty(3, dtype=torch.float32, device=xs.device)
        hip.seg_loss(xs, png if main else None, onehot if dice else None, weights if main else None,
                     (1 if focal else 0) if main else -1, dice, alpha, gamma, beta, smooth, scale, out, dx)
        ctx.dx = dx
        ctx.stats = out
        val = out[2] if (main and dice) else (out[0] if main else out[1])
        return val * scale if scale != 1.0 else val.clone()

    @staticmethod
    def backward(ctx, g):
        dx = ctx.dx
        ctx.dx = None
        return (dx * g,) + (None,) * 11


def _seg(inputs, png, onehot, weights, focal, dice, main=True, alpha=0.5, gamma=2.0, beta=1.0, smooth=1e-5, scale=1.0):
    return _SegLossFn.apply(inputs, png, onehot, weights, bool(focal), bool(dice), bool(main), float(alpha), float(gamma),
                            float(beta), float(smooth), float(scale))


def CE_Loss(inputs, target, cls_weights, num_classes=21):
    """deeplabv3_training.py:9-19 (ignore_index = num_classes = number of logit channels)."""
    if inputs.shape[1] != num_classes:
        raise RuntimeError("CE_Loss: num_classes must equal the logit channels (it is the ignore index)")
    return _seg(inputs, target, None, cls_weights, focal=False, dice=False)


def Focal_Loss(inputs, target, cls_weights, num_classes=21, alpha=0.5, gamma=2):
    """deeplabv3_training.py:22-38."""
    if inputs.shape[1] != num_classes:
        raise RuntimeError("Focal_Loss: num_classes must equal the logit channels (it is the ignore index)")
    if alpha is None:
        alpha = 1.0
    return _seg(inputs, target, None, cls_weights, focal=True, dice=False, alpha=alpha, gamma=gamma)


def Dice_loss(inputs, target, beta=1, smooth=1e-5):
    """deeplabv3_training.py:41-59; target = one-hot labels (B, H, W, C+1)."""
    return _seg(inputs, None, target, None, focal=False, dice=True, main=False, beta=beta, smooth=smooth)


def training_loss(yolo_loss, outputs, outputs_seg, targets, pngs, seg_labels, weights, num_class_seg, focal_loss=True,
                  dice_loss=True):
    """utils/utils_fit.py:96-106 (the active fp16 branch): loss_seg = Focal|CE (+ Dice); total = loss_det + 5 * loss_seg.
    The factor 5 is folded into the seg kernel's gradient.  Returns (total, loss_det, loss_seg)."""
    loss_det = yolo_loss(outputs, targets)
    seg5 = _seg(outputs_seg, pngs, seg_labels if dice_loss else None, weights, focal=focal_loss, dice=dice_loss, scale=5.0)
    return loss_det + seg5, loss_det, seg5 / 5.0


class _MeanSquareFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *tensors):
        ts = [t.contiguous() for t in tensors]
        ctx.save_for_backward(*ts)
        with torch.cuda.device(ts[0].device):
            return hip.mean_square(ts).reshape(())

    @staticmethod
    def backward(ctx, g):
        ts = ctx.saved_tensors
        grads = [torch.empty_like(t) for t in ts]
        with torch.cuda.device(ts[0].device):
            hip.mean_square_bwd(list(ts), g.reshape(1).contiguous().float(), grads)
        return tuple(grads)


def mean_square_loss(det, seg):
    """The synthetic scalar that drives a backward pass without the reference's real losses:
    L = sum_k mean(det_k^2) + mean(seg^2) -- value and gradient as `sum((d * d).mean() for d in det) + (seg * seg).mean()`
    and its autograd, in three launches (value: fp64 partials in a fixed order; gradient: (2 g / n_k) t_k) instead of ~27
    eager elementwise / reduce launches between the forward and the backward pass."""
    ts = list(det) + [seg]
    if not all(t.is_cuda and t.dtype == torch.float32 for t in ts):
        raise RuntimeError("mean_square_loss: fp32 tensors on a HIP device")
    return _MeanSquareFn.apply(*ts)
