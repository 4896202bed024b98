"""N > 1 path on CPU: the gradient bucketing / overlap bookkeeping of parallel.GradBucketer with the
gloo backend, world_size 2 (the HIP kernels themselves need a GPU; what is distributed is covered here)."""
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


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from asy_vrnet_amd.parallel import GradBucketer, backward_param_order
    import asy_vrnet_amd as A
    torch.manual_seed(0)
    model = A.EfficientVRNet(4, 9, "nano", img_size=64)
    order = backward_param_order(model)
    bk = GradBucketer(order, bucket_bytes=1 << 20)
    assert len(bk.buckets) > 3
    # the backward pass writes each gradient into its bucket view, then marks it ready, in backward order;
    # leave the last 7 parameters without a gradient (unused-parameter path)
    used = bk.params[:-7]
    for i, p in enumerate(used):
        g = bk.view(p)
        assert g.shape == p.shape and g.data_ptr() != p.data_ptr()
        g.fill_(float(rank + 1) * (1 + i % 5))
        bk.mark_ready(p)
    bk.finish()
    ok = True
    for i, p in enumerate(bk.params):
        expect = (1 + i % 5) * (sum(range(1, world + 1)) / world) if i < len(used) else 0.0
        ok = ok and torch.allclose(p.grad, torch.full_like(p, expect))
    # zero-sized parameters are never bucketed
    ok = ok and all(p.numel() > 0 for p in bk.params)
    # second pass reuses the buckets
    for p in bk.params:
        bk.view(p).fill_(float(rank))
        bk.mark_ready(p)
    bk.finish()
    ok = ok and all(torch.allclose(p.grad, torch.full_like(p, (world - 1) / 2)) for p in bk.params)
    # deferred mode (HIP-graph step): nothing is sent during the pass, one collective over the whole arena afterwards
    bk.deferred = True
    for i, p in enumerate(bk.params):
        bk.view(p).fill_(float(rank + 1) * (1 + i % 3))
        bk.mark_ready(p)
    ok = ok and not bk.works
    bk.allreduce_all()
    bk.reset()
    ok = ok and all(torch.allclose(bk.view(p), torch.full_like(p, (1 + i % 3) * 1.5)) for i, p in enumerate(bk.params))
    ok = ok and all(bk.view(p).data_ptr() >= bk.arena.data_ptr() for p in bk.params)
    q.put((rank, ok))
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


def test_bucket_order_is_reverse_forward():
    import asy_vrnet_amd as A
    from asy_vrnet_amd.parallel import backward_param_order
    m = A.EfficientVRNet(4, 9, "nano", img_size=64)
    names = {id(p): k for k, p in m.named_parameters()}
    order = [names[id(p)] for p in backward_param_order(m)]
    assert order[0].startswith("head.") and order[-1].startswith("backbone.backbone.image_initial")
