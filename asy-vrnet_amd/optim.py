"""Training-step updates that consume the hot path's gradients, mirroring the reference's
interfaces: the three parameter groups and SGD/Adam of train.py:460-473, `ModelEMA` (nets/yolo_training.py:447-479),
`get_lr_scheduler` / `set_optimizer_lr` (nets/yolo_training.py:504-548) and the batch-size LR fit (train.py:451-455).

The arithmetic runs in libvrnet_hip.so: ONE multi-tensor launch per optimizer step and one per EMA update, instead
of several elementwise launches per tensor (887 state_dict entries at any phi).  No CPU fallback: parameters must
live on the GPU."""
import math
from copy import deepcopy
from functools import partial

import torch
import torch.nn as nn

from . import hip

CHUNK = 4096      # elements per workgroup of the multi-tensor kernels


# ------------------------------------------------------------------------------------------- parameter groups
def param_groups(model):
    """(pg0, pg1, pg2) exactly as train.py:460-467: pg2 = every module's `.bias` Parameter; pg0 = `.weight` of every
    BatchNorm2d or module whose qualified name contains "bn" (no decay); pg1 = every other `.weight` (weight decay).
    Parameters that are not called weight/bias (layer scales, sim_alpha/beta, SA c/s weights) are in no group and are
    therefore never updated -- that is the reference's behaviour, kept on purpose."""
    pg0, pg1, pg2 = [], [], []
    for k, v in model.named_modules():
        if hasattr(v, "bias") and isinstance(v.bias, nn.Parameter):
            pg2.append(v.bias)
        if isinstance(v, nn.BatchNorm2d) or "bn" in k:
            pg0.append(v.weight)
        elif hasattr(v, "weight") and isinstance(v.weight, nn.Parameter):
            pg1.append(v.weight)
    return pg0, pg1, pg2


def fit_lr(batch_size, init_lr, min_lr, optimizer_type):
    """train.py:451-455: scale the learning rates with batch_size / 64 inside per-optimizer limits."""
    nbs = 64
    lr_limit_max = 1e-3 if optimizer_type == "adam" else 5e-2
    lr_limit_min = 3e-4 if optimizer_type == "adam" else 5e-4
    init_fit = min(max(batch_size / nbs * init_lr, lr_limit_min), lr_limit_max)
    min_fit = min(max(batch_size / nbs * min_lr, lr_limit_min * 1e-2), lr_limit_max * 1e-2)
    return init_fit, min_fit


# ------------------------------------------------------------------------------------------- tensor tables
class _Table:
    """Device-side description of a list of equally-shaped tensor tuples for the multi-tensor kernels."""

    def __init__(self):
        self.key = None

    def build(self, roles):
        """roles: list (one per role) of lists of tensors.  Rebuilt only when an address changes."""
        key = tuple(t.data_ptr() for r in roles for t in r)
        if key == self.key:
            return
        first = roles[0]
        dev = first[0].device
        for r in roles:
            for a, b in zip(first, r):
                if not (b.is_cuda and b.dtype == torch.float32 and b.is_contiguous() and b.numel() == a.numel()):
                    raise RuntimeError("multi-tensor update needs contiguous fp32 GPU tensors of matching sizes")
        n = len(first)
        sizes = [t.numel() for t in first]
        ct, ci = [], []
        for i, s in enumerate(sizes):
            k = (s + CHUNK - 1) // CHUNK
            ct += [i] * k
            ci += list(range(k))
        self.n, self.n_chunks = n, len(ct)
        self.addrs = torch.tensor([t.data_ptr() for r in roles for t in r], dtype=torch.int64, device=dev)
        self.sizes = torch.tensor(sizes, dtype=torch.int64, device=dev)
        self.chunk_tensor = torch.tensor(ct, dtype=torch.int32, device=dev)
        self.chunk_index = torch.tensor(ci, dtype=torch.int32, device=dev)
        self.key = key


class _FusedOptimizer:
    """Minimal torch.optim-shaped front (param_groups, step, zero_grad, state_dict) over one multi-tensor launch."""

    def __init__(self, params, defaults):
        self.defaults = defaults
        self.param_groups = []
        self.state = {}
        self._table = _Table()
        self.add_param_group({"params": list(params)})

    def add_param_group(self, group):
        g = dict(self.defaults)
        g.update(group)
        g["params"] = list(g["params"])
        self.param_groups.append(g)

    def zero_grad(self, set_to_none=True):
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is not None:
                    if set_to_none:
                        p.grad = None
                    else:
                        p.grad.zero_()

    def _live(self):
        """(params, grads, per-tensor weight decay) of every parameter that has a gradient, in group order."""
        ps, gs, wd = [], [], []
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is None or p.numel() == 0:
                    continue
                ps.append(p.data)
                gs.append(p.grad.data if p.grad.is_contiguous() else p.grad.data.contiguous())
                wd.append(float(g["weight_decay"]))
        return ps, gs, wd

    def _uniform(self, name):
        vals = {g[name] if not isinstance(g[name], (tuple, list)) else tuple(g[name]) for g in self.param_groups}
        if len(vals) != 1:
            raise RuntimeError(f"fused optimizer: `{name}` must be equal in all parameter groups (the reference sets "
                               "one learning rate for all groups, nets/yolo_training.py:545-548)")
        return next(iter(vals))

    def state_dict(self):
        """torch.optim-compatible layout (per-parameter state indexed in group order)."""
        idx, packed, groups = 0, {}, []
        for g in self.param_groups:
            ids = []
            for p in g["params"]:
                if p in self.state:
                    packed[idx] = self.state[p]
                ids.append(idx)
                idx += 1
            groups.append({**{k: v for k, v in g.items() if k != "params"}, "params": ids})
        return {"state": packed, "param_groups": groups}

    def load_state_dict(self, sd):
        flat = [p for g in self.param_groups for p in g["params"]]
        for g, sg in zip(self.param_groups, sd["param_groups"]):
            g.update({k: v for k, v in sg.items() if k != "params"})
        self.state = {flat[int(i)]: {k: (v.to(flat[int(i)].device) if torch.is_tensor(v) else v) for k, v in st.items()}
                      for i, st in sd["state"].items()}
        self._table.key = None


class SGD(_FusedOptimizer):
    """torch.optim.SGD(params, lr, momentum, nesterov=True) as train.py:470 builds it."""

    def __init__(self, params, lr, momentum=0.0, weight_decay=0.0, nesterov=False):
        if nesterov and momentum <= 0:
            raise ValueError("Nesterov momentum requires a momentum")
        super().__init__(params, {"lr": lr, "momentum": momentum, "weight_decay": weight_decay, "nesterov": nesterov,
                                  "dampening": 0})

    @torch.no_grad()
    def step(self):
        ps, gs, wd = self._live()
        if not ps:
            return
        lr, mu, nest = self._uniform("lr"), self._uniform("momentum"), self._uniform("nesterov")
        bufs = []
        for p in ps:
            st = self.state.setdefault(self._owner(p), {})
            if "momentum_buffer" not in st:
                # torch copies d_p into a new buffer on a parameter's first step; mu * 0 + d_p is the same value
                # bit for bit, so new buffers start at zero and every tensor takes the same code path
                st["momentum_buffer"] = torch.zeros_like(p)
            bufs.append(st["momentum_buffer"])
        tab = self._table
        tab.build([ps, gs, bufs])
        if getattr(tab, "wd_key", None) != (tab.key, tuple(wd)):
            tab.wd = torch.tensor(wd, dtype=torch.float32, device=ps[0].device)
            tab.wd_key = (tab.key, tuple(wd))
        hip.mt_sgd(tab.addrs, tab.sizes, tab.chunk_tensor, tab.chunk_index, tab.wd, tab.n, tab.n_chunks, CHUNK, float(lr),
                   float(mu), bool(nest), False)

    def _owner(self, data):
        """Parameter object owning `data` (state is keyed by Parameter, like torch.optim)."""
        m = getattr(self, "_owners", None)
        if m is None or len(m) != sum(len(g["params"]) for g in self.param_groups):
            m = self._owners = {p.data_ptr(): p for g in self.param_groups for p in g["params"]}
        return m[data.data_ptr()]


class Adam(_FusedOptimizer):
    """torch.optim.Adam(params, lr, betas=(momentum, 0.999)) as train.py:469 builds it (eps 1e-8, no amsgrad)."""

    def __init__(self, params, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, {"lr": lr, "betas": tuple(betas), "eps": eps, "weight_decay": weight_decay})

    _owner = SGD._owner

    @torch.no_grad()
    def step(self):
        ps, gs, wd = self._live()
        if not ps:
            return
        lr, (b1, b2), eps = self._uniform("lr"), self._uniform("betas"), self._uniform("eps")
        by_step = {}
        for i, p in enumerate(ps):
            st = self.state.setdefault(self._owner(p), {})
            if "step" not in st:
                st["step"] = 0
                st["exp_avg"] = torch.zeros_like(p)
                st["exp_avg_sq"] = torch.zeros_like(p)
            st["step"] += 1
            by_step.setdefault(st["step"], []).append(i)
        for step, ids in by_step.items():           # one launch unless parameters joined at different times
            tab = self._table if len(by_step) == 1 else _Table()
            own = [self.state[self._owner(ps[i])] for i in ids]
            tab.build([[ps[i] for i in ids], [gs[i] for i in ids], [s["exp_avg"] for s in own], [s["exp_avg_sq"] for s in own]])
            wdt = torch.tensor([wd[i] for i in ids], dtype=torch.float32, device=ps[0].device)
            hip.mt_adam(tab.addrs, tab.sizes, tab.chunk_tensor, tab.chunk_index, wdt, tab.n, tab.n_chunks, CHUNK, float(lr),
                        float(b1), float(b2), float(eps), int(step))


def build_optimizer(model, optimizer_type, lr, momentum, weight_decay):
    """The optimizer of train.py:468-473 over the groups of train.py:460-467."""
    pg0, pg1, pg2 = param_groups(model)
    if optimizer_type == "adam":
        opt = Adam(pg0, lr, betas=(momentum, 0.999))
    elif optimizer_type == "sgd":
        opt = SGD(pg0, lr, momentum=momentum, nesterov=True)
    else:
        raise KeyError(optimizer_type)
    opt.add_param_group({"params": pg1, "weight_decay": weight_decay})
    opt.add_param_group({"params": pg2})
    return opt


# ------------------------------------------------------------------------------------------- EMA
def is_parallel(model):
    return type(model) in (nn.parallel.DataParallel, nn.parallel.DistributedDataParallel)


def de_parallel(model):
    return model.module if is_parallel(model) or hasattr(model, "module") and hasattr(model, "bucketer") else model


def copy_attr(a, b, include=(), exclude=()):
    for k, v in b.__dict__.items():
        if (len(include) and k not in include) or k.startswith("_") or k in exclude:
            continue
        setattr(a, k, v)


class ModelEMA:
    """nets/yolo_training.py:447-479: moving average of every floating entry of the state_dict (parameters AND
    buffers, i.e. BatchNorm running statistics too); decay ramps as decay * (1 - exp(-updates / tau))."""

    def __init__(self, model, decay=0.9999, tau=2000, updates=0):
        self.ema = deepcopy(de_parallel(model)).eval()
        self.updates = updates
        self.decay = lambda x: decay * (1 - math.exp(-x / tau))
        for p in self.ema.parameters():
            p.requires_grad_(False)
        self._table = _Table()

    def update(self, model):
        with torch.no_grad():
            self.updates += 1
            d = self.decay(self.updates)
            msd = de_parallel(model).state_dict()
            es, ms = [], []
            for k, v in self.ema.state_dict().items():
                if v.dtype.is_floating_point and v.numel() > 0:
                    es.append(v)
                    ms.append(msd[k].detach())
            self._table.build([es, ms])
            t = self._table
            hip.mt_ema(t.addrs, t.sizes, t.chunk_tensor, t.chunk_index, t.n, t.n_chunks, CHUNK, float(d))

    def update_attr(self, model, include=(), exclude=("process_group", "reducer")):
        copy_attr(self.ema, model, include, exclude)


# ------------------------------------------------------------------------------------------- LR schedule
def get_lr_scheduler(lr_decay_type, lr, min_lr, total_iters, warmup_iters_ratio=0.05, warmup_lr_ratio=0.1,
                     no_aug_iter_ratio=0.05, step_num=10):
    """nets/yolo_training.py:504-542: quadratic warm-up + cosine + flat tail ("cos"), or a 10-step geometric decay."""
    def warm_cos(lr, min_lr, total_iters, warmup_total_iters, warmup_lr_start, no_aug_iter, iters):
        if iters <= warmup_total_iters:
            return (lr - warmup_lr_start) * pow(iters / float(warmup_total_iters), 2) + warmup_lr_start
        if iters >= total_iters - no_aug_iter:
            return min_lr
        return min_lr + 0.5 * (lr - min_lr) * (
            1.0 + math.cos(math.pi * (iters - warmup_total_iters) / (total_iters - warmup_total_iters - no_aug_iter)))

    def step_lr(lr, decay_rate, step_size, iters):
        if step_size < 1:
            raise ValueError("step_size must above 1.")
        return lr * decay_rate ** (iters // step_size)

    if lr_decay_type == "cos":
        warmup_total_iters = min(max(warmup_iters_ratio * total_iters, 1), 3)
        warmup_lr_start = max(warmup_lr_ratio * lr, 1e-6)
        no_aug_iter = min(max(no_aug_iter_ratio * total_iters, 1), 15)
        return partial(warm_cos, lr, min_lr, total_iters, warmup_total_iters, warmup_lr_start, no_aug_iter)
    decay_rate = (min_lr / lr) ** (1 / (step_num - 1))
    return partial(step_lr, lr, decay_rate, total_iters / step_num)


def set_optimizer_lr(optimizer, lr_scheduler_func, epoch):
    lr = lr_scheduler_func(epoch)
    for param_group in optimizer.param_groups:
        param_group["lr"] = lr
