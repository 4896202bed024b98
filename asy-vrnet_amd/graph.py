"""HIP-graph capture of one training step (forward + backward) of the hot path.

The step is ~2000 kernel launches issued from Python; eager execution is launch-bound for the
narrow widths.  Shapes, buffers and the launch sequence are static (no host sync anywhere on the
path -- the reference's `if d_min < 0` sync, vr_coc.py:61, is dead code and dropped), so the step
is captured once and replayed.

Single GPU: one hipGraph per step.  Data parallel (the reference: DDP's reducer overlapping the
bucket all-reduces with autograd, train.py:367-368): the backward is cut at section boundaries into
2-3 consecutive hipGraphs that share one memory pool; after segment k has been enqueued, the RCCL
all-reduce of the gradient-arena slice it completed is issued (RCCL's stream waits for the work
enqueued so far) and overlaps the replay of segment k+1.  Collectives stay outside the captured
graphs.  Any change of shape (last batch of an epoch, validation loop) needs its own GraphedStep;
the plain `model(x, r)` path stays available for those.
"""
import torch

from . import program


def replay_segments(graphs, bucketer):
    """One captured step: replays the segment graphs in order; with a data-parallel bucketer, the all-reduce of the arena
    slice segment k completed is issued right behind graph k (it runs beside the replay of k + 1), and the step ends when
    every collective has finished.  `graphs`: objects with .replay()."""
    if bucketer is None:
        graphs[0].replay()
    elif len(graphs) == 1:
        graphs[0].replay()
        bucketer.allreduce_all()
    else:
        for k, g in enumerate(graphs):
            g.replay()
            bucketer.allreduce_segment(k)
        bucketer.wait()


class GraphedStep:
    """step(x, x_radar) -> loss tensor; parameter .grad tensors are static and rewritten by each replay.

    `net` is EfficientVRNet or parallel.DataParallelVRNet; loss_fn(det list, seg) -> scalar tensor.
    Construction runs `warmup` (at least two) eager passes on zero inputs (workspaces, caches, with data parallelism the recording pass
    and its collectives -- so every rank must construct it) and then captures; the module's buffers (BatchNorm running
    statistics, num_batches_tracked) are restored afterwards, so building a GraphedStep on a loaded checkpoint leaves
    the checkpoint's statistics untouched."""

    def __init__(self, net, loss_fn, batch, size, device, warmup=2, segments=None):
        self.net, self.loss_fn = net, loss_fn
        self.model = getattr(net, "module", net)
        self.bucketer = getattr(self.model, "_grad_bucketer", None)
        if getattr(self.model, "_sync_bn", None) is not None:
            raise RuntimeError("GraphedStep: synchronised BatchNorm issues collectives inside the forward / backward program; "
                               "capture is for rank-local BatchNorm statistics (the default) -- run sync_bn steps eagerly")
        self.device = torch.device(device)
        h, w = (size, size) if isinstance(size, int) else size          # size: a side, or (H, W) for rectangular inputs
        self.x = torch.zeros((batch, 3, h, w), device=device)
        self.r = torch.zeros((batch, 4, h, w), device=device)
        cur = torch.cuda.current_stream(device)
        saved = [(b, b.detach().clone()) for b in self.model.buffers() if b.numel()]
        self.stream = torch.cuda.Stream(device)          # warm-up AND capture run here: scratch arenas (hip.Workspace is
        self.stream.wait_stream(cur)                     # keyed by stream) exist before the capture and belong to no graph pool
        with torch.cuda.stream(self.stream):
            # at least two passes: the first one's BACKWARD registers derived caches (the data-gradient weight planes) whose
            # table is rebuilt -- a host-to-device copy -- by the next forward; that must not be the captured one
            for _ in range(max(2, warmup)):              # allocates workspaces / caches outside the capture
                self._eager()
                if self.bucketer is not None and self.bucketer.recording:
                    self.bucketer.rebuild_from_recording()      # arena in execution order before anything is captured
        cur.wait_stream(self.stream)
        torch.cuda.synchronize(device)
        with torch.no_grad():
            for b, old in saved:                         # the warm-up's zero-input statistics must not leak into the model
                b.copy_(old)
        if self.bucketer is None:
            self.model.zero_grad(set_to_none=True)
            cuts = []
        else:
            cuts = list(self.bucketer.cuts) if (segments is None or segments > 1) else []
        self.graphs = []
        if self.bucketer is None:
            self._capture(cuts)
        else:
            # no collective may be issued from inside the captured backward; OUTSIDE the capture the bucketer is back in
            # its eager, overlapped mode, so a plain `net(x, r)` + backward (odd last batch, validation) still reduces
            with self.bucketer.deferring():
                self._capture(cuts)

    def _eager(self):
        det, seg = self.net(self.x, self.r)
        loss = self.loss_fn(det, seg)
        loss.backward()
        return loss.detach()

    def _capture(self, cuts):
        """cuts: descending tape positions; segment 0 = forward + loss + backward down to cuts[0], ..."""
        model = self.model
        pool = None
        state = {}

        def seg0():
            rt, inputs, dets, seg = program.forward_pass(model, self.x, self.r, record=True)
            leaves = [d.requires_grad_(True) for d in dets] + [seg.requires_grad_(True)]
            loss = self.loss_fn(leaves[:3], leaves[3])
            grads = torch.autograd.grad(loss, leaves, allow_unused=True)
            program.backward_begin(rt, grads[:3], grads[3])
            state.update(rt=rt, inputs=inputs, n=len(rt.tape))
            self.loss = loss.detach()

        bounds = None
        for k in range(len(cuts) + 1):
            g = torch.cuda.CUDAGraph()
            # thread_local: with a process group alive, RCCL's watchdog thread polls its work events (hipEventQuery) at any
            # moment; under the default "global" capture mode such a call from ANOTHER thread invalidates the capture
            # (hipErrorStreamCaptureInvalidated) and raises in the watchdog, which aborts the process at exit
            with torch.cuda.graph(g, pool=pool, stream=self.stream, capture_error_mode="thread_local"):
                if k == 0:
                    seg0()
                    n = state["n"]
                    bounds = [n] + [c for c in cuts if 0 < c < n] + [0]
                rt = state["rt"]
                if k + 1 < len(bounds):
                    program.backward_range(rt, bounds[k + 1], bounds[k])
                if k + 2 == len(bounds):
                    program.backward_end(rt, model, state["inputs"], (False, False))
                else:
                    program.backward_cut(rt, final=False)
            pool = g.pool()
            self.graphs.append(g)
            if k + 2 >= len(bounds):
                break

    def __call__(self, x, x_radar):
        self.x.copy_(x, non_blocking=True)
        self.r.copy_(x_radar, non_blocking=True)
        if getattr(self.net, "broadcast_buffers", False) and self.model.training:
            self.net.sync_buffers()
        replay_segments(self.graphs, self.bucketer)
        return self.loss
