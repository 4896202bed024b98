"""N > 1 path on CPU: the gradient arena / execution-order recording / segment and bucket bookkeeping of
parallel.GradBucketer with the gloo backend, world_size 2 (the HIP kernels themselves need a GPU; what is
distributed is covered here)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _execution_like_order(params):
    """A backward 'execution order' that differs from the registration order the way the real program's does: the two
    halves of the parameter list (think image / radar chain) interleave, and tape positions descend."""
    n = len(params)
    a, b = params[: n // 2], params[n // 2:]
    order = []
    for i in range(max(len(a), len(b))):
        if i < len(b):
            order.append(b[i])
        if i < len(a):
            order.append(a[i])
    pos = {p: 40 - (40 * i) // len(order) for i, p in enumerate(order)}     # 41 "top-level closures", replayed 40 .. 0
    return order, pos


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from asy_vrnet_amd.parallel import GradBucketer, backward_param_order
    import asy_vrnet_amd as A
    torch.manual_seed(0)
    model = A.EfficientVRNet(4, 9, "nano", img_size=64)
    bk = GradBucketer(backward_param_order(model), bucket_bytes=1 << 20, segments=3)
    ok = bk.recording and len(bk.cuts) == 0
    # ---- pass 1 = recording pass: gradients are produced in an order that is NOT the registration order; the last 7
    # parameters of that order get no gradient (unused-parameter path); ONE collective over the arena
    order, pos = _execution_like_order(bk.params)
    used = order[:-7]
    val = {p: 1 + i % 5 for i, p in enumerate(order)}
    for p in used:
        g = bk.view(p)
        assert g.shape == p.shape and g.data_ptr() != p.data_ptr()
        g.fill_(float(rank + 1) * val[p])
        bk.mark_ready(p, pos[p])
    bk.finish()
    mean = sum(range(1, world + 1)) / world
    for p in order:
        ok = ok and torch.allclose(p.grad, torch.full_like(p, val[p] * mean if p in set(used) else 0.0))
    ok = ok and all(p.numel() > 0 for p in bk.params)            # zero-sized parameters are never part of the arena
    # ---- rebuild: arena in recorded execution order, cut into 3 segments at tape positions, buckets inside segments
    bk.rebuild_from_recording()
    ok = ok and not bk.recording and bk.params[:len(used)] == used and len(bk.cuts) == 2 and bk.cuts[0] > bk.cuts[1] > 0
    ok = ok and all(p.grad is None for p in bk.params)
    seg_of = lambda p: sum(1 for c in bk.cuts if pos[p] < c)
    for p in used:
        lo, hi = bk.segment_slices[seg_of(p)]
        off = (bk.view(p).data_ptr() - bk.arena.data_ptr()) // 4
        ok = ok and lo <= off and off + p.numel() <= hi
    ok = ok and len(bk.buckets) > 3 and all(b.numel() * 4 <= (1 << 20) + 4 * max(p.numel() for p in bk.params) for b in bk.buckets)
    sizes = [sum(p.numel() for p in used if seg_of(p) == k) for k in range(3)]
    ok = ok and sizes[0] * 3 >= sum(sizes) and sizes[2] > 0          # the last, un-overlapped segment is the remainder
    # ---- pass 2 = eager overlapped mode: a bucket's collective starts when its last parameter is marked ready
    launched_early = False
    for i, p in enumerate(used):
        bk.view(p).fill_(float(rank))
        bk.mark_ready(p, pos[p])
        launched_early = launched_early or (i < len(used) - 1 and len(bk.works) > 0)
    ok = ok and launched_early
    bk.finish()
    ok = ok and all(torch.allclose(p.grad, torch.full_like(p, (world - 1) / 2)) for p in used)
    ok = ok and all(float(p.grad.abs().max()) == 0.0 for p in order[-7:])
    # frozen after wrapping (train.py:440): slot kept, .grad None
    used[3].requires_grad_(False)
    for p in used:
        bk.view(p).fill_(1.0)
        bk.mark_ready(p, pos[p])
    bk.finish()
    ok = ok and used[3].grad is None and used[4].grad is not None
    used[3].requires_grad_(True)
    # ---- pass 3 = captured-step mode: nothing is sent during the pass; segment k's slice is reduced after "graph k",
    # while the later segments are still being written (here: written afterwards, which must not disturb segment k)
    bk.deferred = True
    by_seg = {k: [p for p in used if seg_of(p) == k] for k in range(3)}
    for k in range(3):
        for p in by_seg[k]:
            bk.view(p).fill_(float(rank + 1) * val[p])
            bk.mark_ready(p, pos[p])
        ok = ok and not bk.works if k == 0 else ok
        bk.allreduce_segment(k)
    bk.wait()
    ok = ok and all(torch.allclose(bk.view(p), torch.full_like(p, val[p] * mean)) for p in used)
    bk.reset()
    # single-graph variant: one collective over the whole arena
    for p in used:
        bk.view(p).fill_(float(rank + 1))
        bk.mark_ready(p, pos[p])
    bk.allreduce_all()
    bk.reset()
    ok = ok and all(torch.allclose(bk.view(p), torch.full_like(p, mean)) for p in used)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_bucketed_allreduce_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(ok for _, ok in res), res


def test_first_guess_order_is_reverse_forward():
    import asy_vrnet_amd as A
    from asy_vrnet_amd.parallel import backward_param_order
    m = A.EfficientVRNet(4, 9, "nano", img_size=64)
    names = {id(p): k for k, p in m.named_parameters()}
    order = [names[id(p)] for p in backward_param_order(m)]
    assert order[0].startswith("head.") and order[-1].startswith("backbone.backbone.image_initial")


def _worker_hardening(rank, world, port, q):
    """deferring() scope, layout-hash exchange, the captured step's segment / all-reduce sequencing (stub graphs) and the
    optional buffer broadcast -- world size 2, gloo."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from asy_vrnet_amd.parallel import GradBucketer, DataParallelVRNet, backward_param_order
    from asy_vrnet_amd.graph import replay_segments
    import asy_vrnet_amd as A
    torch.manual_seed(0)
    model = A.EfficientVRNet(4, 9, "nano", img_size=64)
    mean = sum(range(1, world + 1)) / world
    ok = True
    # ---- recording pass, identical on both ranks -> rebuild passes the cross-rank layout check
    bk = GradBucketer(backward_param_order(model), bucket_bytes=1 << 20, segments=3)
    order, pos = _execution_like_order(bk.params)
    for p in order:
        bk.view(p).fill_(1.0)
        bk.mark_ready(p, pos[p])
    bk.finish()
    bk.rebuild_from_recording()
    sig = bk.layout_signature()
    ok = ok and len(bk.cuts) == 2
    # ----  a captured step exists, then an EAGER step runs: the bucketer must reduce again outside deferring()
    with bk.deferring():
        for p in order:
            bk.view(p).fill_(float(rank + 1))
            bk.mark_ready(p, pos[p])
        ok = ok and not bk.works                       # nothing is sent from inside a capture
        bk.finish()
    ok = ok and not bk.deferred
    for p in order:
        bk.view(p).fill_(float(rank + 1))
        bk.mark_ready(p, pos[p])
    ok = ok and len(bk.works) > 0                       # eager mode again: bucket collectives in flight during the pass
    bk.finish()
    ok = ok and all(torch.allclose(p.grad, torch.full_like(p, mean)) for p in order)
    # ---- captured step with stub graphs: "graph k" writes the gradients of segment k; the slice of segment k is reduced
    # right behind it, before graph k + 1 has written anything
    seg_of = lambda p: sum(1 for c in bk.cuts if pos[p] < c)
    log = []

    class StubGraph:
        def __init__(self, k):
            self.k = k

        def replay(self):
            log.append(("replay", self.k))
            for p in order:
                if seg_of(p) == self.k:
                    bk.view(p).fill_(float(rank + 1) * (self.k + 1))

    orig = bk.allreduce_segment

    def spy(k):
        log.append(("allreduce", k))
        orig(k)
    bk.allreduce_segment = spy
    bk.arena.zero_()
    replay_segments([StubGraph(k) for k in range(3)], bk)
    ok = ok and log == [("replay", 0), ("allreduce", 0), ("replay", 1), ("allreduce", 1), ("replay", 2), ("allreduce", 2)]
    ok = ok and all(torch.allclose(bk.view(p), torch.full_like(p, mean * (seg_of(p) + 1))) for p in order) and not bk.works
    bk.arena.zero_()
    replay_segments([StubGraph(0)], bk)                # single-graph variant: ONE collective over the arena afterwards
    ok = ok and all(torch.allclose(bk.view(p), torch.full_like(p, mean if seg_of(p) == 0 else 0.0)) for p in order)
    # ---- a rank that recorded a different backward: every rank raises instead of reducing mismatched slices
    bk2 = GradBucketer(backward_param_order(model), bucket_bytes=1 << 20, segments=3)
    order2 = order if rank == 0 else order[1:] + order[:1]
    for i, p in enumerate(order2):
        bk2.view(p).fill_(1.0)
        bk2.mark_ready(p, 40 - (40 * i) // len(order2))
    bk2.finish()
    try:
        bk2.rebuild_from_recording()
        ok = False
    except RuntimeError as e:
        ok = ok and "layouts differ" in str(e)
    ok = ok and bk2.layout_signature() != sig if rank == 1 else ok
    # ---- optional buffer broadcast (the reference's DDP default): BatchNorm statistics follow rank 0, fea_pos is not sent
    net = DataParallelVRNet(model, broadcast_buffers=True)
    bn = model.head.stems[0].bn
    with torch.no_grad():
        bn.running_mean.fill_(float(rank + 5))
        bn.num_batches_tracked.fill_(rank + 7)
        model.backbone.backbone.fea_pos.fill_(float(rank))
    net.sync_buffers()
    ok = ok and float(bn.running_mean[0]) == 5.0 and int(bn.num_batches_tracked) == 7
    ok = ok and float(model.backbone.backbone.fea_pos.flatten()[0]) == float(rank)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_data_parallel_hardening_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_hardening, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(ok for _, ok in res), res


def _worker_sync_bn(rank, world, port, q):
    """parallel.SyncBatchNormStats (the reference's `sync_bn` option): the (1, C, 2) totals every rank gets back are the sums
    over the samples of ALL ranks, and mean / variance taken from them with the global count equal the statistics of the
    concatenated batch (what torch.nn.SyncBatchNorm normalises with)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from asy_vrnet_amd.parallel import SyncBatchNormStats, DataParallelVRNet
    import asy_vrnet_amd as A
    C, HW = 5, 7
    Bs = [3, 2]                                                            # UNEVEN local batches (an odd last batch)
    xs = [torch.randn(Bs[r], HW, C, dtype=torch.float64, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
    x, B = xs[rank], Bs[rank]
    mom = torch.stack([x.sum(1), (x * x).sum(1)], -1)                      # (B, C, 2): what hip.moments hands over
    sb = SyncBatchNormStats()
    sb.begin_forward(B, "cpu")                                             # the per-rank sample counts are summed too
    tot = sb.total(mom)
    full = torch.cat(xs, 0)
    n = sb.count(B, HW)
    ok = tot.shape == (1, C, 2) and n == sum(Bs) * HW
    mean = tot[0, :, 0] / n
    var = tot[0, :, 1] / n - mean * mean
    ok = ok and torch.allclose(mean, full.mean((0, 1)), atol=1e-12) and torch.allclose(var, full.var((0, 1), unbiased=False), atol=1e-12)
    # a backward pass keeps the count of ITS forward (captured by the program right after begin_forward), whatever a later
    # forward with another batch left in the object (gradient accumulation with a short last micro-batch; round-4 ADVICE)
    captured = sb.batch_total
    sb.begin_forward(B + 1, "cpu")
    ok = ok and sb.count(B, HW, captured) == sum(Bs) * HW and sb.count(B + 1, HW) == (sum(Bs) + world) * HW
    # the wrapper option: sets the stats object on the module; the captured step refuses it
    model = A.EfficientVRNet(4, 9, "nano", img_size=64)
    net = DataParallelVRNet(model, sync_bn=True)
    ok = ok and isinstance(model._sync_bn, SyncBatchNormStats) and model._sync_bn.world == world
    ok = ok and DataParallelVRNet(A.EfficientVRNet(4, 9, "nano", img_size=64))._modules["module"]._sync_bn is None
    try:
        from asy_vrnet_amd.graph import GraphedStep
        GraphedStep(net, lambda det, seg: seg.sum(), 1, 64, "cpu")
        ok = False
    except RuntimeError as e:
        ok = ok and "synchronised BatchNorm" in str(e)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_sync_batch_norm_statistics_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_sync_bn, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(ok for _, ok in res), res


def test_stock_ddp_wrapper_is_found_on_the_call_stack():
    """program._ddp_wrapper_of: the guard behind the stock-DistributedDataParallel path (train.py:367-368).  A module whose
    forward asks "is a DDP wrapper calling me?" gets the wrapper when called through it -- also as a submodule of the
    wrapped module -- and None when called directly.  (The routing of the gradients through torch.autograd that the guard
    switches on needs the kernels: tests/test_net_parity.py::test_stock_distributed_data_parallel_single_rank.)"""
    from asy_vrnet_amd.program import _ddp_wrapper_of
    from torch.nn.parallel import DistributedDataParallel as DDP

    class Probe(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.ones(3))
            self.seen = "unset"

        def forward(self, x):
            self.seen = _ddp_wrapper_of(self)
            return x * self.w

    class Outer(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.inner = Probe()

        def forward(self, x):
            return self.inner(x) + 1

    own = not dist.is_initialized()
    if own:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
        dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        m = Probe()
        m(torch.zeros(3))
        assert m.seen is None
        ddp = DDP(m, find_unused_parameters=True)
        ddp(torch.zeros(3)).sum().backward()
        assert m.seen is ddp
        m(torch.zeros(3))
        assert m.seen is None                       # called directly again: no wrapper on the stack
        o = Outer()
        ddp2 = DDP(o)
        ddp2(torch.zeros(3)).sum().backward()
        assert o.inner.seen is ddp2                 # found as a submodule of the wrapped module
        other = Probe()
        DDP(Outer())                                # an unrelated wrapper elsewhere does not count
        other(torch.zeros(3))
        assert other.seen is None
        # 24 nested wrapper modules (72+ Python frames between DistributedDataParallel.forward and the probe): round 5 walked
        # 16 frames and would have missed the wrapper here -- the whole stack is walked now
        class Shell(torch.nn.Module):
            def __init__(self, inner):
                super().__init__()
                self.inner = inner

            def forward(self, x):
                return self.inner(x)

        deep = Probe()
        nest = deep
        for _ in range(24):
            nest = Shell(nest)
        ddp3 = DDP(nest)
        ddp3(torch.zeros(3)).sum().backward()
        assert deep.seen is ddp3
    finally:
        if own:
            dist.destroy_process_group()


def test_n8_plan_at_l():
    """The segment / bucket plan an 8-GPU job would run at phi = l (BASELINE configs[3]), built on the CPU from the execution
    order a real backward pass recorded (tests/golden/dp_ready_pos.json, written on an MI355X by tools/make_golden_dp_plan.py;
    tests/test_net_parity.py::test_recorded_order_fixture_is_current keeps it current).  The collective of the LAST captured
    segment overlaps nothing, so it must be small: <= 5 % of the gradient bytes (round 4's byte-balancing rule stranded stage
    2 there: 31 %).  Also: three segments, every bucket inside one segment and <= bucket_bytes + one parameter, the arena
    covered exactly once, the head first and the embeddings last."""
    import json
    import asy_vrnet_amd as A
    from asy_vrnet_amd.parallel import GradBucketer, backward_param_order
    pos = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dp_ready_pos.json")))
    for phi in ("l", "nano"):
        with torch.device("meta"):                           # layout only: no 200 MB arena in the CPU suite
            m = A.EfficientVRNet(4, 9, phi)
        names = dict(m.named_parameters())
        assert set(pos) == {k for k, p in names.items() if p.numel()}, "fixture and parameter set differ"
        bk = GradBucketer(backward_param_order(m), bucket_bytes=32 << 20, segments=3)
        for k, (tape_pos, order) in pos.items():             # what mark_ready() stamps during the recording pass
            bk.ready_pos[names[k]] = tape_pos
            bk._rec_order[names[k]] = order
        bk.rebuild_from_recording()
        total = sum(p.numel() for p in bk.params)
        assert len(bk.cuts) == 2 and sorted(bk.segment_slices) == [0, 1, 2]
        share = {k: sum(p.numel() for p in bk.params if bk.bucket_segment[bk.bucket_of[p]] == k) / total for k in range(3)}
        assert 0 < share[2] <= 0.05, share                   # the un-overlapped collective
        assert share[0] >= 0.25 and share[1] >= 0.25, share  # the two overlapped ones carry the bytes (section granularity: a stage is one)
        # slices: consecutive, cover the arena, one segment per bucket, bucket size bounded
        lo = 0
        for k in range(3):
            assert bk.segment_slices[k][0] == lo
            lo = bk.segment_slices[k][1]
        assert lo == bk.arena.numel()
        biggest = max(p.numel() for p in bk.params) * 4
        for bi, flat in enumerate(bk.buckets):
            assert flat.numel() * 4 <= (32 << 20) + biggest
            assert len({bk.bucket_segment[bk.bucket_of[p]] for p in bk.params if bk.bucket_of[p] == bi}) == 1
        # (round-5 review, item 9) a segment bigger than one bucket is itself cut into <= 32 MiB buckets that complete one after
        # another in execution order: in eager mode the all-reduce of its first bucket starts while the rest of the segment is
        # still being computed, and under capture the collective behind graph k runs bucket by bucket
        for k in range(3):
            seg = [bi for bi in range(len(bk.buckets)) if bk.bucket_segment[bi] == k]
            seg_bytes = sum(bk.buckets[bi].numel() * 4 for bi in seg)
            assert seg == list(range(seg[0], seg[-1] + 1)), "a segment's buckets are consecutive"
            assert len(seg) >= -(-seg_bytes // ((32 << 20) + biggest)), (k, seg_bytes, len(seg))
            if seg_bytes > (32 << 20) + biggest:
                assert len(seg) >= 2, (k, seg_bytes, len(seg))
            done_at = [max(bk._rec_order[p] for p in bk.params if bk.bucket_of[p] == bi) for bi in seg]
            assert done_at == sorted(done_at), "buckets of a segment must complete in arena order"
        if phi == "l":      # 200 MB of gradients: the two overlapped segments hold several buckets each or one
            nb = [sum(1 for bi in range(len(bk.buckets)) if bk.bucket_segment[bi] == k) for k in range(3)]
            assert nb[0] + nb[1] >= 6 and max(nb[0], nb[1]) >= 3, nb
        inv = {p: k for k, p in names.items()}
        assert inv[bk.params[0]].startswith("head.") and "initial" in inv[bk.params[-1]] or "patch_embed" in inv[bk.params[-1]] \
            or "enhance" in inv[bk.params[-1]], inv[bk.params[-1]]
        # ring all-reduce over 8 ranks at ~150 GB/s per xGMI link and direction: wire time of the exposed collective
        exposed_ms = 2 * 7 / 8 * share[2] * total * 4 / 150e9 * 1e3
        assert exposed_ms < (0.15 if phi == "l" else 0.02), exposed_ms
