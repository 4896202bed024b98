"""Data-parallel training of the hot path: one process per GPU, gradients averaged with RCCL
all-reduce over xGMI, overlapped with the hand-written backward.

Reference behaviour being replaced: torch DistributedDataParallel around EfficientVRNet
(train.py:367-368; DistributedSampler shards the minibatch, train.py:518-522).  Semantics kept:
BatchNorm statistics and the data_normal min/max are rank-local (sync_bn=False, train.py:51),
gradients are averaged over ranks.  Differences, by design for MI355X:

* gradients live in ONE flat arena laid out in the order the backward pass *actually finishes* them: the first
  backward pass runs in recording mode (every parameter is stamped with the position of the top-level backward
  section that issued its last gradient kernel), then the arena is rebuilt in that order -- like DDP's bucket
  rebuild after the first iteration, but from the execution order of this program, where the image and the radar
  chain of a stage interleave, rather than from the registration order (train.py:367 + torch's reducer);
* the backward kernels write parameter gradients straight into arena views (no copy into .grad, no autograd
  hooks);
* eager steps: a bucket's all-reduce is enqueued the moment the section that completes it has been joined into the
  main stream; RCCL runs it on its own stream, concurrent with the remaining backward kernels (which keep their
  side-stream weight gradients); buckets are <= 32 MiB by default so that each collective is large enough to drive
  all 7 xGMI links of a GPU;
* captured steps (graph.GraphedStep): the backward is cut at up to `segments - 1` section boundaries into
  consecutive hipGraphs; the arena slice a segment completes is all-reduced while the next segment's graph replays
  (collectives stay outside the captured graphs);
* parameters that receive no gradient (the six zero-sized ShuffleAttention(channel=3) entries -- the reference
  needs find_unused_parameters=True for them) are not part of the arena; parameters frozen after wrapping
  (train.py:440 freezes the backbone behind the DDP wrap) keep their slot, are reduced as zeros and get .grad = None;
* gradients do not accumulate across backward passes in this mode (each pass overwrites).
"""
import contextlib
import hashlib

import torch
import torch.distributed as dist


def _dist_on(group):
    return dist.is_available() and dist.is_initialized()


class GradBucketer:
    """Flat gradient arena + ready-counting + async all-reduce.  Device and backend agnostic (tested with gloo on
    CPU); the HIP program calls `view()` and `mark_ready()`, the captured step `allreduce_segment()`."""

    def __init__(self, params_in_backward_order, bucket_bytes=32 << 20, process_group=None, average=True,
                 ready_pos=None, segments=3):
        self.group = process_group
        self.average = average
        self.bucket_bytes = bucket_bytes
        self.n_segments = max(1, segments)
        self.deferred = False      # True: no collective inside backward (HIP-graph capture); allreduce_segment() instead
        self.force_collective = False   # issue the collectives even with one rank (single-GPU test of the RCCL path)
        self.recording = ready_pos is None          # first pass: stamp parameters with their tape position
        self.params = [p for p in params_in_backward_order if p.numel() > 0]
        self._index = {p: i for i, p in enumerate(self.params)}     # rank-independent name of a parameter (registration order)
        self.ready_pos = dict(ready_pos) if ready_pos is not None else {}
        self._rec_order = {}
        self._layout()
        self.reset()

    # ---- layout ------------------------------------------------------------------------------------------------
    def _layout(self):
        """Arena order = self.params; segment boundaries at tape positions; buckets never straddle a segment."""
        self.cuts = self._choose_cuts()
        seg_of = {}
        for p in self.params:
            pos = self.ready_pos.get(p)
            seg_of[p] = len(self.cuts) if pos is None else sum(1 for c in self.cuts if pos < c)
        groups, seg_groups = [], []
        cur, cur_bytes, cur_seg = [], 0, None
        for p in self.params:
            sg = seg_of[p]
            if cur and (sg != cur_seg or cur_bytes >= self.bucket_bytes):
                groups.append(cur)
                seg_groups.append(cur_seg)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_seg = sg
            cur_bytes += p.numel() * p.element_size()
        if cur:
            groups.append(cur)
            seg_groups.append(cur_seg)
        # one contiguous arena, buckets are consecutive 256-byte aligned slices of it
        sizes = [-(-sum(p.numel() for p in g) // 64) * 64 for g in groups]
        ref = self.params[0] if self.params else None
        self.arena = torch.zeros(sum(sizes), dtype=ref.dtype, device=ref.device) if ref is not None else None
        self.bucket_of, self.views, self.buckets, self.pending0 = {}, {}, [], []
        self.bucket_segment = seg_groups
        self.segment_slices = {}
        base = 0
        for bi, g in enumerate(groups):
            n = sum(p.numel() for p in g)
            flat = self.arena[base:base + n]
            lo, hi = self.segment_slices.get(seg_groups[bi], (base, base))
            self.segment_slices[seg_groups[bi]] = (lo, base + sizes[bi])
            base += sizes[bi]
            off = 0
            for p in g:
                self.views[p] = flat[off:off + p.numel()].view_as(p)
                self.bucket_of[p] = bi
                off += p.numel()
            self.buckets.append(flat)
            self.pending0.append(len(g))

    TAIL_FRACTION = 0.05       # the last segment's all-reduce overlaps nothing: keep it below this share of the gradient bytes

    def _choose_cuts(self):
        """Tape positions (descending) at which a captured backward is cut.  The collective of segment k runs beside the
        replay of segment k + 1, the LAST segment's beside nothing -- so the last cut goes to the earliest section boundary
        behind which at most TAIL_FRACTION of the gradient bytes remain (the backward finishes with the high-resolution,
        few-parameter stages: embeddings, stages 0-1 and their fusion blocks are 3.6 % of the bytes at every width, and a
        third of the backward's run time is still ahead to hide the previous collective), and the sections in front of it
        are split into equal shares of the remaining bytes.  The granularity is the top-level tape closure: a backbone
        stage (both chains) is ONE section, so a rule that only balances bytes can strand stage 2 -- 28 % of the
        parameters -- in the last segment (round 4 did: tests/test_data_parallel_gloo.py::test_n8_plan_at_l).
        Without a boundary that leaves a small non-empty tail the segments are equal shares of everything."""
        if self.n_segments <= 1 or not self.ready_pos:
            return []
        per_pos = {}
        for p in self.params:
            pos = self.ready_pos.get(p)
            if pos is not None:
                per_pos[pos] = per_pos.get(pos, 0) + p.numel()
        total = sum(per_pos.values())
        order = sorted(per_pos, reverse=True)                 # the order the backward reaches them
        last_cut, head_total = None, total
        below = total
        for pos in order[:-1]:                                # cut candidates: BEFORE replaying closure pos - 1 (pos > min)
            below -= per_pos[pos]
            if 0 < below <= self.TAIL_FRACTION * total:
                last_cut, head_total = pos, total - below
                break
        n_head = self.n_segments - 1 if last_cut is not None else self.n_segments
        cuts, acc = [], 0
        for pos in order:
            if last_cut is not None and pos <= last_cut:
                break
            acc += per_pos[pos]
            if acc * n_head >= head_total and len(cuts) < n_head - 1 and pos > min(per_pos):
                cuts.append(pos)                              # cut BEFORE replaying closure pos - 1
                acc = 0
        if last_cut is not None:
            cuts.append(last_cut)
        return cuts

    def rebuild_from_recording(self):
        """After the recording pass: arena in execution order (parameters that never reported keep their relative order
        at the end -- they are reduced as zeros by finish()).  Gradient views change: call before capturing a graph."""
        seen = [p for p in self.params if p in self.ready_pos]
        order = sorted(range(len(seen)), key=lambda i: (-self.ready_pos[seen[i]], self._rec_order.get(seen[i], i)))
        rest = [p for p in self.params if p not in self.ready_pos]
        self.params = [seen[i] for i in order] + rest
        self.recording = False
        self._layout()
        self.reset()
        for p in self.params:
            p.grad = None
        self.check_layout_across_ranks()

    def layout_signature(self):
        """Hash of everything the collectives depend on: arena order (by registration index), sizes, cut positions and
        segment slices."""
        h = hashlib.sha256()
        h.update(repr([(self._index[p], p.numel()) for p in self.params]).encode())
        h.update(repr((list(self.cuts), sorted(self.segment_slices.items()), [b.numel() for b in self.buckets])).encode())
        return int.from_bytes(h.digest()[:7], "little")

    def check_layout_across_ranks(self):
        """Every rank derives the arena layout from ITS OWN recording pass; a rank that recorded something else (another
        frozen set, another order of completion) would reduce mismatched slices silently.  All ranks that REACH this point
        exchange the layout hash and raise together on a mismatch.  A rank whose first backward raised never gets here:
        its peers then wait in this all_gather until the process group's timeout fires -- pass `timeout=` to
        init_process_group for a bound (a rank that dies takes the job down through the launcher anyway)."""
        if not _dist_on(self.group) or dist.get_world_size(self.group) < 2:
            return
        dev = self.arena.device if self.arena is not None else "cpu"
        mine = torch.tensor([self.layout_signature()], dtype=torch.int64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(dist.get_world_size(self.group))]
        dist.all_gather(every, mine, group=self.group)
        sigs = [int(t.item()) for t in every]
        if len(set(sigs)) != 1:
            raise RuntimeError(f"data parallel: gradient-arena layouts differ between ranks (hashes {sigs}): the ranks did "
                               "not record the same backward pass (different frozen parameters? a failed first step?)")

    @contextlib.contextmanager
    def deferring(self):
        """Scope in which no collective is issued from inside the backward pass (HIP-graph capture: the collectives are
        issued between the captured segments instead).  Outside it the eager, overlapped mode is in force again."""
        old, self.deferred = self.deferred, True
        try:
            yield self
        finally:
            self.deferred = old

    def reset(self):
        self.pending = list(self.pending0)
        self.seen = set()
        self.works = []

    def view(self, p):
        return self.views.get(p)

    # ---- backward-time interface -------------------------------------------------------------------------------
    def mark_ready(self, p, tape_pos=None):
        """Called once per parameter per backward, after its gradient kernels have been enqueued and joined."""
        if p not in self.bucket_of or p in self.seen:
            return
        self.seen.add(p)
        if self.recording and tape_pos is not None:
            self.ready_pos[p] = tape_pos
            self._rec_order[p] = len(self._rec_order)
        bi = self.bucket_of[p]
        self.pending[bi] -= 1
        if self.pending[bi] == 0:
            self._launch(bi)

    def _collective_on(self):
        if not _dist_on(self.group):
            return False
        return dist.get_world_size(self.group) > 1 or self.force_collective

    def _reduce(self, flat):
        if self.average and dist.get_backend(self.group) == "nccl":
            return dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group, async_op=True), None
        w = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        return w, (flat if self.average else None)

    def _launch(self, bi):
        if self.deferred or self.recording or not self._collective_on():
            return
        self.works.append(self._reduce(self.buckets[bi]))

    def _wait_all(self):
        ws = dist.get_world_size(self.group) if _dist_on(self.group) else 1
        for w, scale in self.works:
            w.wait()
            if scale is not None:
                scale.div_(ws)
        self.works = []

    def finish(self):
        """Launches buckets whose parameters got no gradient this pass (their views are zeroed), waits for every
        collective and publishes the views as .grad (None for parameters that are frozen now)."""
        for bi, n in enumerate(self.pending):
            if n > 0:
                for p in self.params:
                    if self.bucket_of[p] == bi and p not in self.seen:
                        self.views[p].zero_()
                self._launch(bi)
        if self.recording and self._collective_on() and not self.deferred:
            self.works.append(self._reduce(self.arena))       # recording pass: one collective over everything
        self._wait_all()
        for p in self.params:
            p.grad = self.views[p] if p.requires_grad else None
        self.reset()

    # ---- captured steps ----------------------------------------------------------------------------------------
    def allreduce_segment(self, k):
        """Deferred mode: all-reduce of the arena slice that captured segment k completed; asynchronous (RCCL's
        stream waits for the work enqueued so far on the current stream, later replays overlap it)."""
        if not self._collective_on() or k not in self.segment_slices:
            return
        lo, hi = self.segment_slices[k]
        self.works.append(self._reduce(self.arena[lo:hi]))

    def allreduce_all(self):
        """Deferred mode without cuts: ONE collective over the whole arena after the replay."""
        if self._collective_on():
            self.works.append(self._reduce(self.arena))
        self._wait_all()

    def wait(self):
        self._wait_all()


def backward_param_order(model):
    """First guess, used for the recording pass only: reverse of the forward registration order
    (nets/efficient_vrnet.py:24-27: backbone -> neck -> head)."""
    return list(reversed(list(model.parameters())))


class SyncBatchNormStats:
    """The collective step of synchronised BatchNorm (the reference's `sync_bn` option, train.py:356-357:
    torch.nn.SyncBatchNorm.convert_sync_batchnorm): per-channel sums over the samples of ALL ranks.  program.bn_forward /
    bn_backward hand their per-sample fp64 moments (B, C, 2) to `total()`; what comes back is the (1, C, 2) sum over
    samples and ranks, from which the usual coefficient kernels take mean / variance (forward) and the two gradient sums
    (backward) with the GLOBAL element count `count(B * HW)`.  The parameter gradients keep the LOCAL sums, as
    SyncBatchNorm does (the gradient all-reduce averages them afterwards).  Every rank issues the same sequence of
    collectives: the program's launch order is a function of the model only."""

    def __init__(self, process_group=None):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if _dist_on(process_group) else 1
        self.batch_total = None

    def begin_forward(self, batch_local, device):
        """Once per forward pass: the number of samples over ALL ranks (torch.nn.SyncBatchNorm gathers the per-rank counts
        too).  The ranks need not hold equal batches -- the reference's loaders drop the last batch (utils/dataloader.py,
        drop_last=True), but an evaluation loop or a resumed run may not.  One tiny collective and one host read per
        forward pass, issued by every rank unconditionally."""
        if self.world > 1:
            t = torch.tensor([batch_local], dtype=torch.int64, device=device)
            dist.all_reduce(t, group=self.group)
            self.batch_total = int(t.item())
        else:
            self.batch_total = batch_local

    def count(self, batch_local, per_sample, batch_total=None):
        """Global element count per channel of a map with `per_sample` pixels per sample.  batch_total: the global sample
        count of the FORWARD pass the caller belongs to (program.RT.sync_batch_total, captured right after begin_forward):
        a backward pass that runs after another forward with a different batch (gradient accumulation with a short last
        micro-batch, two models sharing this object) must not normalise with the newer count."""
        total = batch_total if batch_total is not None else \
            (self.batch_total if self.batch_total is not None else batch_local * self.world)
        return total * per_sample

    def total(self, mom):
        tot = mom.sum(0, keepdim=True)
        if self.world > 1:
            dist.all_reduce(tot, group=self.group)
        return tot


class DataParallelVRNet(torch.nn.Module):
    """Drop-in for DistributedDataParallel(EfficientVRNet) on one node (one process per GPU)."""

    def __init__(self, module, bucket_bytes=32 << 20, process_group=None, force_collective=False, segments=3,
                 broadcast_buffers=False, sync_bn=False):
        """broadcast_buffers: the reference's DDP default (train.py:367-368 leaves broadcast_buffers=True): rank 0's
        BatchNorm running statistics overwrite the other ranks' before every training forward.  Off by default here:
        rank-0 checkpoints (the only ones the reference writes, utils_fit.py:213-225) are identical either way, ranks
        != 0 then evaluate with their own running statistics, and the step saves a collective.  The constant position
        buffers (fea_pos, fea_pos_r: 2 x 524 288 floats) are never re-sent."""
        super().__init__()
        self.module = module
        self.group = process_group
        self.broadcast_buffers = broadcast_buffers
        self.bucketer = GradBucketer(backward_param_order(module), bucket_bytes, process_group, segments=segments)
        self.bucketer.force_collective = force_collective   # collectives even with one rank (single-GPU RCCL rehearsal)
        module._grad_bucketer = self.bucketer
        # sync_bn: BatchNorm batch statistics over the samples of all ranks (the reference's opt-in `sync_bn`; its default and
        # this wrapper's is rank-local statistics).  Two small all-reduces per BatchNorm and direction inside the forward /
        # backward program: eager steps only -- graph.GraphedStep refuses a model with it (no collective inside a capture).
        module._sync_bn = SyncBatchNormStats(process_group) if sync_bn else None
        if _dist_on(process_group) and (dist.get_world_size(process_group) > 1 or force_collective):
            with torch.no_grad():                       # replicas start identical (DDP broadcasts at wrap time)
                for t in list(module.parameters()) + list(module.buffers()):
                    if t.numel():
                        dist.broadcast(t, src=0, group=process_group)
            if t.is_cuda:
                torch.cuda.synchronize(t.device)        # no collective left in flight when a HIP-graph capture starts

    def forward(self, x, x_radar):
        if self.bucketer.recording and self.bucketer.ready_pos and torch.is_grad_enabled():
            self.finalize_layout()                      # the previous backward was the recording pass
        if self.broadcast_buffers and self.module.training and torch.is_grad_enabled():
            self.sync_buffers()
        return self.module(x, x_radar)

    def sync_buffers(self):
        """Rank 0's non-constant buffers to every rank: one broadcast per dtype (flattened), not one per buffer."""
        if not self.bucketer._collective_on():
            return
        by_dtype = {}
        for name, b in self.module.named_buffers():
            if b.numel() and not name.endswith(("fea_pos", "fea_pos_r")):
                by_dtype.setdefault(b.dtype, []).append(b)
        with torch.no_grad():
            for bufs in by_dtype.values():
                flat = torch.cat([b.reshape(-1) for b in bufs])
                dist.broadcast(flat, src=0, group=self.group)
                off = 0
                for b in bufs:
                    b.copy_(flat[off:off + b.numel()].view_as(b))
                    off += b.numel()

    def finalize_layout(self):
        """Call after the first backward pass (the recording pass): rebuilds the arena in execution order.  Done
        automatically by the next forward if omitted."""
        if self.bucketer.recording and self.bucketer.ready_pos:
            self.bucketer.rebuild_from_recording()
