"""HIP-graph capture of one training step (forward + backward) of the hot path.

The step is ~2000 kernel launches issued from Python; eager execution is launch-bound for the
narrow widths.  Shapes, buffers and the launch sequence are static (no host sync anywhere on the
path -- the reference's `if d_min < 0` sync, vr_coc.py:61, is dead code and dropped), so the whole step
is captured once into a hipGraph and replayed: one graph launch per step.
"""
import torch


class GraphedStep:
    """step(x, x_radar) -> loss tensor; parameter .grad tensors are static and rewritten by each replay.

    `net` is EfficientVRNet or parallel.DataParallelVRNet.  With data parallelism the captured graph
    writes the gradient buckets and the RCCL all-reduce of all buckets is issued right after the replay
    (a collective inside a captured graph is avoided on purpose)."""

    def __init__(self, net, loss_fn, batch, size, device, warmup=2):
        self.net, self.loss_fn = net, loss_fn
        self.model = getattr(net, "module", net)
        self.bucketer = getattr(self.model, "_grad_bucketer", None)
        self.x = torch.zeros((batch, 3, size, size), device=device)
        self.r = torch.zeros((batch, 4, size, size), device=device)
        if self.bucketer is not None:
            self.bucketer.deferred = True
        cur = torch.cuda.current_stream(device)
        side = torch.cuda.Stream(device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):          # allocates workspaces / caches outside the capture
                self._run()
        cur.wait_stream(side)
        torch.cuda.synchronize(device)
        if self.bucketer is None:
            self.model.zero_grad(set_to_none=True)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self._run()

    def _run(self):
        det, seg = self.net(self.x, self.r)
        loss = self.loss_fn(det, seg)
        loss.backward()
        return loss.detach()

    def __call__(self, x, x_radar):
        self.x.copy_(x, non_blocking=True)
        self.r.copy_(x_radar, non_blocking=True)
        self.graph.replay()
        if self.bucketer is not None:
            self.bucketer.allreduce_all()
        return self.loss
