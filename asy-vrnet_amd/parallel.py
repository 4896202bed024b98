"""Data-parallel training of the hot path: one process per GPU, gradients averaged with RCCL
all-reduce over xGMI, overlapped with the hand-written backward.

Reference behaviour being replaced: torch DistributedDataParallel around EfficientVRNet
(train.py:367-368; DistributedSampler shards the minibatch, train.py:518-522).  Semantics kept:
BatchNorm statistics and the data_normal min/max are rank-local (sync_bn=False, train.py:51),
gradients are averaged over ranks.  Differences, by design for MI355X:

* gradients live in a few large flat buckets laid out in the order the backward pass produces
  them (head -> neck -> stage 3 ... -> input embeddings); the backward kernels write parameter
  gradients straight into bucket views (no copy into .grad, no autograd hooks);
* a bucket's all-reduce is enqueued the moment its last gradient kernel has been launched; RCCL
  runs it on its own stream, concurrent with the remaining backward kernels; the default bucket is
  32 MiB so that each collective is large enough to drive all 7 xGMI links of a GPU;
* parameters that receive no gradient (the six zero-sized ShuffleAttention(channel=3) entries -- the
  reference needs find_unused_parameters=True for them) are simply not part of any bucket;
* gradients do not accumulate across backward passes in this mode (each pass overwrites).
"""
import torch
import torch.distributed as dist


class GradBucketer:
    """Flat gradient buckets + ready-counting + async all-reduce.  Device and backend agnostic
    (tested with gloo on CPU); the HIP program calls `view()` / `mark_ready()`."""

    def __init__(self, params_in_backward_order, bucket_bytes=32 << 20, process_group=None, average=True):
        self.group = process_group
        self.average = average
        self.deferred = False      # True: no collective inside backward (HIP-graph capture); call allreduce_all()
        self.force_collective = False   # issue the collectives even with one rank (single-GPU test of the RCCL path)
        self.params = [p for p in params_in_backward_order if p.requires_grad and p.numel() > 0]
        self.bucket_of, self.views, self.buckets, self.pending0 = {}, {}, [], []
        cur, cur_bytes = [], 0
        groups = []
        for p in self.params:
            cur.append(p)
            cur_bytes += p.numel() * p.element_size()
            if cur_bytes >= bucket_bytes:
                groups.append(cur)
                cur, cur_bytes = [], 0
        if cur:
            groups.append(cur)
        # one contiguous arena, buckets are consecutive slices of it: the deferred (HIP-graph) mode reduces the whole
        # arena with a single collective, the overlapped mode one slice at a time (slices start 256-byte aligned)
        sizes = [-(-sum(p.numel() for p in g) // 64) * 64 for g in groups]
        ref = self.params[0] if self.params else None
        self.arena = torch.zeros(sum(sizes), dtype=ref.dtype, device=ref.device) if ref is not None else None
        base = 0
        for bi, g in enumerate(groups):
            n = sum(p.numel() for p in g)
            flat = self.arena[base:base + n]
            base += sizes[bi]
            off = 0
            for p in g:
                self.views[p] = flat[off:off + p.numel()].view_as(p)
                self.bucket_of[p] = bi
                off += p.numel()
            self.buckets.append(flat)
            self.pending0.append(len(g))
        self.reset()

    def reset(self):
        self.pending = list(self.pending0)
        self.seen = set()
        self.works = []

    def view(self, p):
        return self.views.get(p)

    def mark_ready(self, p):
        """Called once per parameter per backward, after its gradient kernel has been enqueued."""
        if p not in self.bucket_of or p in self.seen:
            return
        self.seen.add(p)
        bi = self.bucket_of[p]
        self.pending[bi] -= 1
        if self.pending[bi] == 0:
            self._launch(bi)

    def _launch(self, bi):
        if self.deferred:
            return
        if not (dist.is_available() and dist.is_initialized()):
            return
        if dist.get_world_size(self.group) == 1 and not self.force_collective:
            return
        flat = self.buckets[bi]
        if self.average and dist.get_backend(self.group) == "nccl":
            self.works.append((dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group, async_op=True), None))
        else:
            w = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self.works.append((w, flat if self.average else None))

    def finish(self):
        """Launches buckets whose parameters got no gradient this pass (their views are zero),
        waits for every collective and publishes the views as .grad."""
        for bi, n in enumerate(self.pending):
            if n > 0:
                for p in self.params:
                    if self.bucket_of[p] == bi and p not in self.seen:
                        self.views[p].zero_()
                self._launch(bi)
        ws = dist.get_world_size(self.group) if (dist.is_available() and dist.is_initialized()) else 1
        for w, scale in self.works:
            w.wait()
            if scale is not None:
                scale.div_(ws)
        for p in self.params:
            p.grad = self.views[p]
        self.reset()


def _allreduce_all(self):
    """Deferred mode: ONE collective over the whole gradient arena, after the captured step has been replayed (a
    single large message drives all xGMI links; seven 32 MiB ones would pay the ring latency seven times)."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    ws = dist.get_world_size(self.group)
    if ws == 1 and not self.force_collective:
        return
    if self.average and dist.get_backend(self.group) == "nccl":
        dist.all_reduce(self.arena, op=dist.ReduceOp.AVG, group=self.group)
    else:
        dist.all_reduce(self.arena, op=dist.ReduceOp.SUM, group=self.group)
        if self.average:
            self.arena.div_(ws)


GradBucketer.allreduce_all = _allreduce_all


def backward_param_order(model):
    """Parameters in the order the backward pass finishes them: reverse of the forward call order
    (nets/efficient_vrnet.py:24-27: backbone -> neck -> head)."""
    return list(reversed(list(model.parameters())))


class DataParallelVRNet(torch.nn.Module):
    """Drop-in for DistributedDataParallel(EfficientVRNet) on one node (one process per GPU)."""

    def __init__(self, module, bucket_bytes=32 << 20, process_group=None, force_collective=False):
        super().__init__()
        self.module = module
        self.bucketer = GradBucketer(backward_param_order(module), bucket_bytes, process_group)
        self.bucketer.force_collective = force_collective   # collectives even with one rank (single-GPU RCCL rehearsal)
        module._grad_bucketer = self.bucketer
        module._on_param_grad = self.bucketer.mark_ready
        if dist.is_available() and dist.is_initialized() and (dist.get_world_size(process_group) > 1 or force_collective):
            with torch.no_grad():                       # replicas start identical (DDP broadcasts at wrap time)
                for t in list(module.parameters()) + list(module.buffers()):
                    if t.numel():
                        dist.broadcast(t, src=0, group=process_group)
            if t.is_cuda:
                torch.cuda.synchronize(t.device)        # no collective left in flight when a HIP-graph capture starts

    def forward(self, x, x_radar):
        return self.module(x, x_radar)
