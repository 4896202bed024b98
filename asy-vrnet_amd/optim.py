"""Training-step updates that consume the hot path's gradients, mirroring the reference's
interfaces: the three parameter groups and SGD/Adam of train.py:460-473, `ModelEMA` (nets/yolo_training.py:447-479),
`get_lr_scheduler` / `set_optimizer_lr` (nets/yolo_training.py:504-548) and the batch-size LR fit (train.py:451-455).

The arithmetic runs in libvrnet_hip.so: ONE multi-tensor launch per optimizer step and one per EMA update, instead
of several elementwise launches per tensor (887 state_dict entries at any phi).  No CPU fallback: parameters must
live on the GPU."""
import math
from copy import deepcopy

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
    """Device-side description of K roles x n tensors for the multi-tensor kernels.  The chunk list depends only on
    the sizes and is built once per set of tensors; a step that merely sees new gradient ADDRESSES (autograd hands out
    fresh .grad tensors every backward) refreshes one row of the address table."""

    def __init__(self):
        self.sig = None

    def layout(self, sig, sizes, n_roles, device):
        if sig == self.sig:
            return False
        ct, ci = [], []
        for i, s in enumerate(sizes):
            k = (s + CHUNK - 1) // CHUNK
            ct += [i] * k
            ci += list(range(k))
        self.n, self.n_chunks, self.n_roles = len(sizes), len(ct), n_roles
        self.sizes = torch.tensor(sizes, dtype=torch.int64, device=device)
        self.chunk_tensor = torch.tensor(ct, dtype=torch.int32, device=device)
        self.chunk_index = torch.tensor(ci, dtype=torch.int32, device=device)
        self.addrs = torch.zeros((n_roles, self.n), dtype=torch.int64, device=device)
        self.rows = [None] * n_roles
        self.sig = sig
        return True

    def set_row(self, role, tensors):
        """Uploads the addresses of one role if they changed."""
        ptrs = [t.data_ptr() for t in tensors]
        if ptrs != self.rows[role]:
            # a FRESH pinned staging buffer per change: the caching host allocator keeps it alive until the async copy
            # has run, so a host that is a step ahead never overwrites addresses a queued copy has yet to read (eager
            # mode hands out new .grad tensors every backward; under a hipGraph the addresses never change)
            staged = torch.tensor(ptrs, dtype=torch.int64).pin_memory()
            self.addrs[role].copy_(staged, non_blocking=True)
            self.rows[role] = ptrs


def _check_tensors(ts, like=None):
    for i, t in enumerate(ts):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()) or (like is not None and t.numel() != like[i].numel()):
            raise RuntimeError("multi-tensor update needs contiguous fp32 GPU tensors of matching sizes")


class _FusedOptimizer:
    """Minimal torch.optim-shaped front (param_groups, step, zero_grad, state_dict) over one multi-tensor launch."""

    state_names = ()

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
        self._flat = None

    def zero_grad(self, set_to_none=True):
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is not None:
                    if set_to_none:
                        p.grad = None
                    else:
                        p.grad.zero_()

    def _live(self):
        """Parameters that have a gradient this step (group order), their gradients, and the table laid out for them."""
        if self._flat is None:
            self._flat = [(p, gi) for gi, g in enumerate(self.param_groups) for p in g["params"] if p.numel() > 0]
        live = [(p, gi) for p, gi in self._flat if p.grad is not None]
        if not live:
            return None
        ps = [p for p, _ in live]
        gs = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in ps]
        sig = (tuple(id(p) for p in ps), tuple(self.param_groups[gi]["weight_decay"] for _, gi in live))
        tab = self._table
        if tab.layout(sig, [p.numel() for p in ps], 2 + len(self.state_names), ps[0].device):
            _check_tensors(ps)
            tab.wd = torch.tensor(sig[1], dtype=torch.float32, device=ps[0].device)
            states = []
            for p in ps:
                st = self.state.setdefault(p, {})
                for n in self.state_names:
                    if n not in st:
                        st[n] = torch.zeros_like(p)
                states.append(st)
            tab.states = states
            tab.set_row(0, ps)
            for k, n in enumerate(self.state_names):
                tab.set_row(2 + k, [st[n] for st in states])
        _check_tensors(gs, ps)
        tab.set_row(1, gs)
        return tab

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
        state = {}
        for i, st in sd["state"].items():
            p, mine = flat[int(i)], {}
            for k, v in st.items():
                if k == "step":                       # torch.optim.Adam checkpoints carry a tensor step
                    mine[k] = int(v.item()) if torch.is_tensor(v) else int(v)
                elif v is None:                       # torch.optim.SGD before its first step: momentum_buffer None
                    mine[k] = torch.zeros_like(p)
                else:
                    mine[k] = v.to(p.device) if torch.is_tensor(v) else v
            state[p] = mine
        self.state = state
        self._table.sig = None


class SGD(_FusedOptimizer):
    """torch.optim.SGD(params, lr, momentum, nesterov=True) as train.py:470 builds it.  torch copies d_p into a new
    momentum buffer on a parameter's first step; mu * 0 + d_p is the same value bit for bit, so new buffers start at
    zero and every tensor takes one code path (parameters un-frozen later, train.py:584, simply join the table)."""

    state_names = ("momentum_buffer",)

    def __init__(self, params, lr, momentum=0.0, weight_decay=0.0, nesterov=False):
        if nesterov and momentum <= 0:
            raise ValueError("Nesterov momentum requires a momentum")
        super().__init__(params, {"lr": lr, "momentum": momentum, "weight_decay": weight_decay, "nesterov": nesterov,
                                  "dampening": 0})

    @torch.no_grad()
    def step(self):
        tab = self._live()
        if tab is None:
            return
        lr, mu, nest = self._uniform("lr"), self._uniform("momentum"), self._uniform("nesterov")
        hip.mt_sgd(tab.addrs, tab.sizes, tab.chunk_tensor, tab.chunk_index, tab.wd, tab.n, tab.n_chunks, CHUNK, float(lr),
                   float(mu), bool(nest), False)


class Adam(_FusedOptimizer):
    """torch.optim.Adam(params, lr, betas=(momentum, 0.999)) as train.py:469 builds it (eps 1e-8, no amsgrad).  One
    launch per distinct step count (parameters that joined later carry their own bias correction)."""

    state_names = ("exp_avg", "exp_avg_sq")

    def __init__(self, params, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, {"lr": lr, "betas": tuple(betas), "eps": eps, "weight_decay": weight_decay})

    @torch.no_grad()
    def step(self):
        tab = self._live()
        if tab is None:
            return
        lr, (b1, b2), eps = self._uniform("lr"), self._uniform("betas"), self._uniform("eps")
        nxt = [int(st.get("step", 0)) + 1 for st in tab.states]
        if len(set(nxt)) != 1:               # validated BEFORE any counter moves: an error leaves the state untouched
            raise RuntimeError("fused Adam: parameters with different step counts in one table (un-freeze with a new "
                               "optimizer, as train.py:575-590 does)")
        for st in tab.states:
            st["step"] = nxt[0]
        hip.mt_adam(tab.addrs, tab.sizes, tab.chunk_tensor, tab.chunk_index, tab.wd, tab.n, tab.n_chunks, CHUNK, float(lr),
                    float(b1), float(b2), float(eps), nxt[0])


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
_WRAPPERS = (nn.parallel.DataParallel, nn.parallel.DistributedDataParallel)


def is_parallel(model):
    """True for torch's own multi-device wrappers (interface of nets/yolo_training.py:430)."""
    return isinstance(model, _WRAPPERS) and type(model) in _WRAPPERS


def de_parallel(model):
    """The bare network behind torch's wrappers or this package's DataParallelVRNet (interface of yolo_training.py:435)."""
    ours = hasattr(model, "bucketer") and hasattr(model, "module")
    return model.module if (ours or is_parallel(model)) else model


def copy_attr(a, b, include=(), exclude=()):
    """Copies the public instance attributes of `b` onto `a` (interface of yolo_training.py:440): private names and
    `exclude` are skipped; a non-empty `include` is a whitelist."""
    wanted = set(include)
    skipped = set(exclude)
    for name in list(vars(b)):
        if name.startswith("_") or name in skipped or (wanted and name not in wanted):
            continue
        setattr(a, name, getattr(b, name))


class ModelEMA:
    """Moving average of every floating entry of the state_dict -- parameters AND buffers, i.e. BatchNorm running
    statistics too -- with the decay ramp decay * (1 - exp(-updates / tau)) (semantics of nets/yolo_training.py:447-479).
    One multi-tensor HIP launch per update (887 tensors); the (ema, model) tensor pairs are resolved once per model
    object; call `refresh()` after replacing parameter objects of the live model (load_state_dict copies in place and
    needs nothing)."""

    def __init__(self, model, decay=0.9999, tau=2000, updates=0):
        self.ema = deepcopy(de_parallel(model))
        self.ema.eval()
        self.ema.requires_grad_(False)
        self.updates = int(updates)
        self._decay_max, self._tau = float(decay), float(tau)
        self._table = _Table()
        self._pairs = None

    def decay(self, n):
        return self._decay_max * (1.0 - math.exp(-float(n) / self._tau))

    def refresh(self):
        self._pairs = None

    def update(self, model):
        with torch.no_grad():
            self.updates += 1
            d = self.decay(self.updates)
            model = de_parallel(model)
            if self._pairs is None or self._pairs[0] is not model:
                msd = model.state_dict()
                es, ms = [], []
                for k, v in self.ema.state_dict().items():
                    if v.dtype.is_floating_point and v.numel() > 0:
                        es.append(v)
                        ms.append(msd[k].detach())
                _check_tensors(es)
                _check_tensors(ms, es)
                self._pairs = (model, es, ms)
            _, es, ms = self._pairs
            t = self._table
            t.layout((id(model), len(es)), [e.numel() for e in es], 2, es[0].device)
            t.set_row(0, es)
            t.set_row(1, ms)
            hip.mt_ema(t.addrs, t.sizes, t.chunk_tensor, t.chunk_index, t.n, t.n_chunks, CHUNK, float(d))

    def update_attr(self, model, include=(), exclude=("process_group", "reducer")):
        copy_attr(self.ema, model, include, exclude)


# ------------------------------------------------------------------------------------------- LR schedule
class _CosineSchedule:
    """lr(it): quadratic warm-up from `start` to `peak` over `warm` iterations, half-cosine from `peak` down to `floor`,
    then flat at `floor` for the last `tail` iterations."""

    def __init__(self, peak, floor, total, warm, start, tail):
        self.peak, self.floor, self.total, self.warm, self.start, self.tail = peak, floor, total, warm, start, tail

    def __call__(self, it):
        if it <= self.warm:
            frac = it / float(self.warm)
            return self.start + (self.peak - self.start) * frac * frac
        if it >= self.total - self.tail:
            return self.floor
        phase = (it - self.warm) / (self.total - self.warm - self.tail)
        return self.floor + (self.peak - self.floor) * 0.5 * (1.0 + math.cos(math.pi * phase))


class _StepSchedule:
    """lr(it) = peak * rate ** floor(it / every)."""

    def __init__(self, peak, rate, every):
        if every < 1:
            raise ValueError("step schedule: fewer iterations than steps (step length < 1)")
        self.peak, self.rate, self.every = peak, rate, every

    def __call__(self, it):
        return self.peak * self.rate ** (it // self.every)


def get_lr_scheduler(lr_decay_type, lr, min_lr, total_iters, warmup_iters_ratio=0.05, warmup_lr_ratio=0.1,
                     no_aug_iter_ratio=0.05, step_num=10):
    """Epoch -> learning rate (interface and values of nets/yolo_training.py:504-542, pinned by
    tests/golden/optim_lr_schedules.json).  "cos": warm-up of clamp(ratio * total, 1, 3) epochs starting at
    max(ratio * lr, 1e-6), cosine, flat tail of clamp(ratio * total, 1, 15) epochs.  Anything else: `step_num`
    geometric steps from lr down to min_lr."""
    if lr_decay_type == "cos":
        warm = min(3, max(1, warmup_iters_ratio * total_iters))
        tail = min(15, max(1, no_aug_iter_ratio * total_iters))
        return _CosineSchedule(lr, min_lr, total_iters, warm, max(1e-6, warmup_lr_ratio * lr), tail)
    rate = (min_lr / lr) ** (1.0 / (step_num - 1))
    sched = _StepSchedule(lr, rate, total_iters / step_num)
    return sched


def set_optimizer_lr(optimizer, lr_scheduler_func, epoch):
    """Every parameter group gets the schedule's value for `epoch` (interface of yolo_training.py:544-548)."""
    value = lr_scheduler_func(epoch)
    for group in optimizer.param_groups:
        group["lr"] = value
    return value
