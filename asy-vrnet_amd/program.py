"""Forward / backward program of the fusion hot path over the C-ABI kernels (hip.py).

Host-side mirror of the reference call tree (nets/efficient_vrnet.py:24-27 ->
neck/coc_fpn_dual.py:184-224 -> backbone/fusion/vr_coc.py:575-704 -> head/decouplehead.py:42-88).
Activations are NHWC fp32 tensors that live on the GPU for the whole step; torch only allocates
them.  The backward pass is hand-written: every forward block pushes one closure on a tape, the
tape is replayed in reverse, gradients are accumulated in place by the kernels (`accumulate`
flags) and parameter gradients are written once per parameter.  A single autograd.Function
exposes the whole network to torch.autograd, so `loss.backward()`, optimizers, EMA deep-copies
and state_dict work exactly as with the reference module.
"""
import os
import weakref

import torch

from . import hip
from . import modules as M


class Act:
    """NHWC activation: `t` is a (B,H,W,C) tensor or channel-slice view; row stride `ld` floats.

    Two-stream buffers: the image and the radar stream of a backbone stage live in ONE (2B,H,W,C) tensor (image
    samples first) so that a stage's ClusterBlocks run as one launch per layer; `half(k)` is the Act of one stream --
    a view whose gradient is the matching half of the parent's gradient buffer (`written[k]`: that half holds data)."""
    __slots__ = ("t", "B", "H", "W", "C", "ld", "grad", "need_grad", "pairs", "colpart", "colchunks", "parent", "slot", "written",
                 "_halves", "gradp", "want_gradp", "sa", "__weakref__")

    def __init__(self, t, need_grad=True):
        self.t = t
        self.pairs = None       # (fp64 (sum, sumsq) pairs, pairs per sample) emitted by the conv that produced `t`
        self.colpart = None     # fp64 per-channel (sum, sumsq) partials per 32-row tile emitted by that conv (BatchNorm)
        self.colchunks = 0      # > 0: `colpart` is [colchunks][C][2] over ALL rows (left by a fused kernel of csrc/fusion.hip)
        self.B, self.H, self.W, self.C = t.shape
        self.ld = t.stride(2)
        self.grad = None
        self.gradp = None       # hip.Planes copy of `grad` (valid only while grad is the buffer it was written with)
        self.want_gradp = 0     # np of the plane copy the consumer of d(act) would like to receive (0: none)
        self.need_grad = need_grad
        self.parent, self.slot, self.written, self._halves = None, 0, None, None

    def half(self, k):
        """The Act of stream k.  A half references its parent; the parent only remembers its halves WEAKLY (a strong link
        both ways would be a reference cycle that keeps the buffer alive until Python's cycle collector runs).  A half
        carries no state of its own -- its gradient is the parent's (`written[k]`) -- so one that died is simply rebuilt."""
        if self._halves is None:
            assert self.B % 2 == 0 and self.ld == self.C
            self._halves = [None, None]
        ref = self._halves[k]
        a = ref() if ref is not None else None
        if a is None:
            n = self.B // 2
            a = Act(self.t[k * n:(k + 1) * n], self.need_grad)
            a.parent, a.slot = self, k
            self._halves[k] = weakref.ref(a)
        return a

    def halves(self):
        return self.half(0), self.half(1)

    @property
    def HW(self):
        return self.H * self.W

    @property
    def rows(self):
        return self.B * self.H * self.W


ASIDE_LAG = int(os.environ.get("VRNET_ASIDE_LAG", "1"))           # tape closures a main-chain weight gradient may trail by
_SERIAL_SITES = int(os.environ.get("VRNET_SERIAL_SITES", "0"), 0)   # diagnostic: parallel sections (bit = site id) run serially
FUSED_MLP_MIN_ROWS = int(os.environ.get("VRNET_FUSED_MLP_MIN_ROWS", "16384"))      # (diagnostic override)
WGRAD_STREAMS = int(os.environ.get("VRNET_WGRAD_STREAMS", "2"))   # side streams for weight gradients (measured: 2 < 1 < 4 ms/step)


class RT:
    """Per-call runtime: mode, tape, parameter-gradient table, packed-weight cache."""

    def __init__(self, device, training, record):
        self.device, self.training, self.record = device, training, record
        self.tape = []
        self.pgrads = {}
        self.packed = {}
        self.packed_t = {}
        self.bf16 = False           # dense convs with bf16-rounded operands on the bf16 MFMA (model.compute_dtype)
        self.fp32_precision = 2     # fp32 layers: 2 = six-bf16-product kernels where available, 0 = fp32 MFMA only
        self.fused_mlp = True       # Mlp of a ClusterBlock as one kernel per direction where the library has one
        self.wplanes = None         # WeightPlanes: pre-split weights for the x6 kernels (model.weight_planes, default on)
        self.bn_colstats = True     # BatchNorm batch statistics from the producing conv's epilogue (no pass over z)
        self.pnp = 0                # plane GEMMs (csrc/pgemm.hip) in the ClusterBlocks: 0 off, 3 fp32 values as three bf16 planes,
                                    # 1 bf16 tensors (compute_dtype "bf16"); pg_fwd / pg_wgrad: which GEMM kinds take them
        self.pg_fwd, self.pg_wgrad = True, True
        self.pweights = None        # PlaneWeights: the weights of the plane GEMMs, split once per forward
        self.forced_idx = None      # {block name: (B,H,W,E) uint8}: Cluster assignments to replay instead of the arg-max (model.forced_idx_maps)
        self.sync_bn = None         # parallel.SyncBatchNormStats: BatchNorm statistics over all ranks (model._sync_bn)
        self.consts = {}
        self.idx_maps = {}
        self.relu_masks = None      # {BatchNorm module: ReLU output Act} when model.record_relu_masks (parity tests)
        self.on_param_grad = None
        self.bucketer = None        # parallel.GradBucketer: gradients are written into its flat buckets
        self.det_grads, self.seg_grad = (None, None, None), None
        self._depth = 0
        self._chain = "main"        # logical chain id: "main" or (depth, branch index)
        self._aside_open = None     # aside work of the tape closure being replayed
        self._aside_batches = []    # closed batches not yet joined: (events, tensors kept alive, parameters reported)
        self._deferred_wgrads = []
        self._started_wgrads = []   # weight gradients issued early on the side streams, not yet joined (start_deferred_wgrads)
        self.ready = None           # with a bucketer: parameters whose gradient kernels were issued, not yet handed over
        self.tape_pos = 0           # index of the top-level tape closure being replayed
        self.concurrent = True      # fork independent chains (image / radar, seg / det, head levels) onto side streams
        self.pair_streams = False   # True: image + radar chain of a backbone stage as ONE two-stream batch (one launch per layer)
        self.pending_ab = []        # deferred (d alpha, d beta) reductions of the Cluster modules of the section being replayed
        self.sync_batch_total = None   # SyncBatchNorm: global sample count of THIS forward pass
        self.overlap_fusion = True
        self.mlp_recompute = "auto"  # fused Mlp backward recomputes the pre-activation instead of reading a stored one
        self.fused_fusion = True    # the fused passes of csrc/fusion.hip in the fusion blocks
        self.fused_upsample = True  # CoCUpsample: BatchNorm + ReLU applied on the taps of the bilinear gather (K11)
        self.early_wgrads = 2       # 1 = behind every section: measured (round 5, same call): 25.87-26.06 ms with it against 25.75-25.92 without -- the weight
                                    # gradients then contend with the small kernels of the critical chain they were meant to fill
        self.stamps = None          # diagnostic: (int64 buffer, [names]) -- rt.stamp(name) writes the device clock in stream order
        self.via_autograd = False   # parameter gradients go back through torch.autograd (stock DistributedDataParallel)

    def stamp(self, name):
        """Diagnostic (model.debug_stamps): the device clock at this point of the current stream (tools/debug/section_stamps.py)."""
        if self.stamps is None:
            return
        buf, names = self.stamps
        if len(names) < buf.numel():
            hip.clock_stamp(buf, len(names))
            names.append(name)

    # ---- fork / join of independent chains -------------------------------------------------------------
    _side_streams = {}
    _aux_streams = {}

    def _streams(self, n):
        """n side streams for the current nesting depth (nested sections get their own streams)."""
        pool = RT._side_streams.setdefault(self.device, [])
        lo = self._depth * 8
        while len(pool) < lo + n:
            pool.append(torch.cuda.Stream(self.device))      # (the longer chain on a high-priority stream: no effect, round 5)
        return pool[lo:lo + n]

    # ---- parameter-gradient kernels off the critical path ---------------------------------------------------
    # hipGraph capture on ROCm 7 only accepts a STAR of streams around the capturing stream: a dependency between two
    # side streams crashes hipStreamEndCapture (tools/micro/graph_fork_probe.py).  Hence: on the main chain a weight
    # gradient forks an auxiliary stream directly; inside a forked chain it is deferred and all deferred weight
    # gradients of the section run, mutually concurrent, on the section's side streams once the chains have joined.
    # The main-chain joins LAG by one tape closure: the weight gradients of block k run beside the data gradients of
    # block k+1 and are joined (their operands released, their parameters reported ready) after block k+1.
    def aside(self, fn, keep):
        """`fn` launches weight / parameter-gradient kernels whose results no later backward kernel reads;
        `keep`: tensors it reads (held until it has been ordered before their release)."""
        if not self.concurrent:
            fn()
            return
        if self._chain != "main":
            self._deferred_wgrads.append((fn, keep))
            return
        cur = torch.cuda.current_stream(self.device)
        pool = RT._aux_streams.setdefault(self.device, [])
        while len(pool) < WGRAD_STREAMS:
            pool.append(torch.cuda.Stream(self.device))
        ent = self._aside_open
        if ent is None:
            ent = self._aside_open = {"streams": {}, "keep": [], "n": 0}
        aux = pool[ent["n"] % len(pool)]
        ent["n"] += 1
        aux.wait_stream(cur)
        with torch.cuda.stream(aux):
            fn()
        ent["streams"][aux] = True
        ent["keep"].extend(keep)
        ent["keep"].append(fn)      # the closure too: it may own scratch buffers its kernels are still using

    def join_aside(self, lag=1):
        """End of a top-level tape closure: closes the closure's batch of aside work (events on the auxiliary streams it
        used, the parameters it reported) and joins the batches older than `lag` closures into the current stream.
        Returns the parameters whose gradient kernels are now ordered before the current stream."""
        if self._chain != "main":
            return []
        ent, self._aside_open = self._aside_open, None
        fresh = list(self.ready) if self.ready is not None else []
        if self.ready is not None:
            self.ready.clear()          # in place: rt.on_param_grad is this list's bound append
        evs = []
        if ent is not None:
            for st in ent["streams"]:
                ev = torch.cuda.Event()
                ev.record(st)
                evs.append(ev)
        self._aside_batches.append((evs, ent["keep"] if ent is not None else [], fresh))
        done = []
        cur = torch.cuda.current_stream(self.device)
        while len(self._aside_batches) > lag:
            evs, keep, ready = self._aside_batches.pop(0)
            for ev in evs:
                cur.wait_event(ev)
            keep.clear()
            done.extend(ready)
        return done

    def _launch_deferred_wgrads(self, cur, streams):
        """Issues the pending deferred weight gradients round-robin on `streams` (already forked from `cur`)."""
        work, self._deferred_wgrads = self._deferred_wgrads, []
        # (newest first -- the operands most likely still in the Infinity Cache -- measured neutral: 24.70-24.93 vs 24.75-24.93 ms)
        for i, (fn, _) in enumerate(work):
            with torch.cuda.stream(streams[i % len(streams)]):
                fn()
        return work          # keeps the tensors alive until the caller has joined the streams

    def start_deferred_wgrads(self, nstreams=0):
        """(round 5) Issues the weight gradients a parallel section deferred NOW, on the weight-gradient side streams, without
        waiting for them: they run beside whatever the main chain does next (the ImageEnhanceByRadar backward between two
        sections -- ~30 small launches during which the chip was idle) instead of waiting for the next parallel section to take
        them along.  Joined by join_started_wgrads() (the next section's backward / a cut); the parameters they report become
        ready only then."""
        if not self._deferred_wgrads or not self.concurrent or self._chain != "main":
            return
        cur = torch.cuda.current_stream(self.device)
        streams = self._streams(8)[4:4 + max(WGRAD_STREAMS, nstreams)]
        for st in streams:
            st.wait_stream(cur)
        mark = len(self.ready) if self.ready is not None else 0
        work = self._launch_deferred_wgrads(cur, streams)
        fresh = []
        if self.ready is not None:
            fresh = self.ready[mark:]
            del self.ready[mark:]
        self._started_wgrads.append((streams, work, fresh))

    def join_started_wgrads(self):
        if not self._started_wgrads:
            return
        cur = torch.cuda.current_stream(self.device)
        for streams, work, fresh in self._started_wgrads:
            for st in streams:
                cur.wait_stream(st)
            work.clear()
            if self.ready is not None:
                self.ready.extend(fresh)
        self._started_wgrads = []

    def flush_deferred_wgrads(self):
        self.join_started_wgrads()
        if not self._deferred_wgrads:
            return
        cur = torch.cuda.current_stream(self.device)
        streams = self._streams(8)[4:4 + WGRAD_STREAMS]
        for st in streams:
            st.wait_stream(cur)
        work = self._launch_deferred_wgrads(cur, streams)
        for st in streams:
            cur.wait_stream(st)
        work.clear()

    def parallel(self, fns, site=None):
        """Runs independent chains `fns` (callables issuing kernels) on forked HIP streams and joins them.
        Each chain records its backward closures on its own sub-tape; one closure on the main tape replays the
        sub-tapes concurrently.  Allocator safety: inside a chain the side stream is torch's current stream, so
        its temporaries live in that stream's pool; buffers crossing the fork / join are ordered by the events
        (wait_stream) on both sides.  Under hipGraph capture the fork / join become graph edges."""
        if not self.concurrent or len(fns) < 2 or self._depth > 0:      # no nested forks (star topology only)
            return [fn() for fn in fns]
        if _SERIAL_SITES and site is not None and (_SERIAL_SITES >> site) & 1:      # diagnostic: VRNET_SERIAL_SITES bit mask
            return [fn() for fn in fns]
        cur = torch.cuda.current_stream(self.device)
        streams = self._streams(len(fns))
        main_tape, outs, subtapes = self.tape, [], []
        self._depth += 1
        outer_chain = self._chain
        for bi, (st, fn) in enumerate(zip(streams, fns)):
            st.wait_stream(cur)
            self.tape = []
            self._chain = (self._depth, bi)
            with torch.cuda.stream(st):
                self.stamp(f"  fwd site {site} chain {bi} start")
                outs.append(fn())
                self.stamp(f"  fwd site {site} chain {bi} end")
            subtapes.append(self.tape)
        self._chain = outer_chain
        self._depth -= 1
        self.tape = main_tape
        for st in streams:
            cur.wait_stream(st)
        if self.record:
            def bwd():
                cur_b = torch.cuda.current_stream(self.device)
                self.join_started_wgrads()      # (same side streams; their work queues behind what was started early)
                # weight gradients deferred by the PREVIOUS section run beside this section's data-gradient chains
                wstreams = self._streams(8)[4:4 + WGRAD_STREAMS] if self._deferred_wgrads else []
                for st in wstreams:
                    st.wait_stream(cur_b)
                held = self._launch_deferred_wgrads(cur_b, wstreams) if wstreams else []
                self._depth += 1
                outer = self._chain
                for bi, (st, sub) in enumerate(zip(streams, subtapes)):
                    st.wait_stream(cur_b)
                    self._chain = (self._depth, bi)
                    with torch.cuda.stream(st):
                        self.stamp(f"  bwd site {site} chain {bi} start")
                        for f in reversed(sub):
                            f()
                        self.stamp(f"  bwd site {site} chain {bi} end")
                self._chain = outer
                self._depth -= 1
                # (round 5, measured neutral and not kept: joining the weight-gradient streams only at the next section -- the
                # join of the stage-1 section waits 0.45 ms for stage 2's weight gradients, but the step is bound by the work,
                # not by that wait: 24.44 vs 24.47 ms; a chain of the last section issuing its own weight gradients: +0.1 ms)
                for st in list(streams) + list(wstreams):
                    cur_b.wait_stream(st)
                held.clear()
            bwd.is_parallel = True
            main_tape.append(bwd)
        return outs

    def new(self, B, H, W, C, need_grad=True):
        return Act(torch.empty((B, H, W, C), dtype=torch.float32, device=self.device), need_grad)

    def new_pair(self, B2, H, W, C):
        """Two-stream buffer (2B samples, image stream first).  Its gradient buffer is allocated HERE, on the main stream:
        the two halves are written by different chains on different side streams (seg / det branch of the neck), and a
        buffer allocated lazily inside one chain would come from that stream's allocator pool -- where it may alias a
        tensor the OTHER chain is not ordered against."""
        a = self.new(B2, H, W, C)
        if self.record:
            a.grad = self.buf(B2, H, W, C)
            a.written = [False, False]
        return a

    def release(self):
        """End of the call's life (after the backward pass, or after a forward that recorded nothing)."""
        self.tape = None
        self.pgrads.clear()
        self._aside_batches = []
        self._deferred_wgrads = []
        self._started_wgrads = []
        self.pending_ab = []

    def flush_cluster_ab(self):
        """ONE launch finishes (d alpha, d beta) of every Cluster module whose backward kernel ran since the last flush (they
        left per-workgroup partials in buffers of their own): called on the main stream once the section's chains have joined,
        instead of one finishing launch on the chain behind each of the 27 Cluster backward kernels."""
        if not self.pending_ab:
            return
        work, self.pending_ab = self.pending_ab, []
        hip.cluster_ab_reduce_multi([(ws, nb, ga, gb, acc) for ws, nb, ga, gb, acc, _ in work])
        if self.on_param_grad:
            for *_, tm in work:
                self.on_param_grad(tm.sim_alpha)
                self.on_param_grad(tm.sim_beta)

    def buf(self, *shape, dtype=torch.float32):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def const(self, n, value):
        key = (n, value)
        c = self.consts.get(key)
        if c is None:
            c = self.buf(n)
            hip.fill_(c, value)
            self.consts[key] = c
        return c

    def weight(self, conv):
        """[kh*kw][Cout][Cin] packing of a dense conv weight (the OIHW tensor itself for 1x1)."""
        w = conv.weight
        co, ci, kh, kw = w.shape
        if kh * kw == 1:
            return w
        p = self.packed.get(conv)
        if p is None:
            p = self.buf(kh * kw, co, ci)
            hip.pack_weight(w, p, co, ci, kh, kw)
            self.packed[conv] = p
        return p

    def mlp_packs(self, mlp, C, hid, pmlp, rc=False):
        """(forward, backward) weight planes of a fused Mlp (hip.mlp_pack).
        rc: the backward pack of the kernel that recomputes the pre-activation (hip.mlp_pack_rc)."""
        if rc:
            fwd, _ = hip.mlp_pack(mlp.fc1.weight, mlp.fc2.weight, C, hid, pmlp, want_bwd=False)
            return fwd, (hip.mlp_pack_rc(mlp.fc1.weight, mlp.fc2.weight, C, hid, pmlp) if self.record else None)
        return hip.mlp_pack(mlp.fc1.weight, mlp.fc2.weight, C, hid, pmlp, want_bwd=self.record)

    def mlp_rc(self, C, hid, bf16_tensors):
        """Whether a fused Mlp block runs without a stored pre-activation (model.mlp_recompute: "auto" = where it measured
        faster -- the 64-channel stage in fp32, where the backward is HBM-bound; both fused stages with bf16 tensors, where the
        extra GEMM is one product instead of six)."""
        mode = self.mlp_recompute
        if not mode or not self.record or not hip.mlp_rc_ok(C, hid):
            return False
        if mode == "auto":
            return bool(bf16_tensors) or C <= 64
        return True

    def prec_fwd(self, lda, ci, co):
        """precision flag of a forward conv launch: 1 = bf16-rounded operands (compute_dtype "bf16"); 2 = fp32 products
        as six exact bf16 x bf16 products where the library has a kernel for the shape, the fp32 MFMA otherwise
        (compute_dtype "f32", the default); 0 = fp32 MFMA only (compute_dtype "f32-mfma")."""
        if self.bf16:
            return 1 if hip.bf16_conv_ok(lda, ci, co, 0) else self.fp32_precision
        return self.fp32_precision

    def prec_mlp(self, C, hid, rows, HW):
        """precision flag of the fused fc1 -> GELU -> fc2 kernels (hip.mlp_fwd / mlp_bwd) for a block, 0 = not fused:
        2 = x6 (compute_dtype "f32"), 1 = bf16-rounded operands (compute_dtype "bf16"); "f32-mfma" keeps the two convs."""
        # (rows: one workgroup per 128 pixels -- below 16 384 rows the launch leaves most CUs empty: at phi = nano the fused kernels
        # ran the 16 x 16 / 32 x 32 stages (C = 128 / 64) at 7.6 TFLOP/s, 71 us for a 0.5 GFLOP GEMM pair; two tile launches are faster
        # there: 11.50 -> 11.24 ms per step, profiles/r06_nano_variants.txt.  phi = s / m / l have C = 64 / 128 only on the big maps)
        if not self.fused_mlp or HW % 32 or rows < FUSED_MLP_MIN_ROWS or not hip.mlp_fused_ok(C, hid, rows):
            return 0
        if self.bf16:
            return 1
        return 2 if self.fp32_precision == 2 else 0

    def planes(self, w, mode, J, K, rows, kscale=None):
        """bf16 planes of a 1x1 weight for a launch of `rows` GEMM rows and J columns (None: the launch does not run on
        an x6 tile kernel, or the planes are switched off).  mode 0: w is [J = Cout][K = Cin]; mode 1: [K = Cout][J = Cin]."""
        if self.wplanes is None or self.bf16 or self.fp32_precision != 2 or K % 16 or not hip.conv2d_dma_plan(rows, J, K)[0]:
            return None
        sj, sk = (K, 1) if mode == 0 else (1, J)
        return self.wplanes.get((id(w), mode), w, J, K, sj, sk, kscale)

    def prec_wgrad(self, ldx, lddy, ci, co):
        if self.bf16 and hip.bf16_wgrad_ok(ldx, lddy, ci, co):
            return 1
        return self.fp32_precision

    def dgrad_operands(self, key, w_oihw, w_packed, co, ci, kh, kw, kscale, lddy, rows=None):
        """(weights, kscale, precision) of a data-gradient launch: the bf16 path contracts over Cout with the
        TRANSPOSED pack [t][Cin][Cout], into which kscale is folded."""
        if self.bf16 and hip.bf16_conv_ok(lddy, ci, co, 1):
            if rows is not None and (kscale is None or co <= 1024) and hip.conv2d_dma_tile(rows, ci):
                return w_packed, kscale, 3        # LDS-DMA tile kernel: standard weight layout, kscale applied in the kernel
            ck = (key, None if kscale is None else kscale.data_ptr())
            wt = self.packed_t.get(ck)
            if wt is None:
                wt = self.buf(kh * kw, ci, co)
                hip.pack_weight_t(w_oihw, kscale, wt, co, ci, kh, kw)
                self.packed_t[ck] = wt
            return wt, None, 1
        return w_packed, kscale, self.fp32_precision

    def pgrad(self, param):
        """(gradient buffer, accumulate flag) for a parameter; None if it needs no gradient."""
        if not param.requires_grad or param.numel() == 0:
            return None, 0
        g = self.pgrads.get(param)
        if g is None:
            g = self.bucketer.view(param) if self.bucketer is not None else None
            if g is None:
                g = torch.empty_like(param)
            self.pgrads[param] = g
            return g, 0
        return g, 1

    def push(self, fn):
        if self.record:
            self.tape.append(fn)

    def grad_target(self, act):
        """(buffer, accumulate) to write d(act) into."""
        par = act.parent
        if par is not None:                       # one stream of a two-stream buffer: its half of the parent's gradient
            if par.grad is None:
                par.grad = self.buf(par.B, par.H, par.W, par.C)
                par.written = [False, False]
            elif par.written is None:
                par.written = [True, True]
            n = act.B
            acc = 1 if par.written[act.slot] else 0
            par.written[act.slot] = True
            return par.grad[act.slot * n:(act.slot + 1) * n], acc
        if act.grad is None:
            act.grad = self.buf(act.B, act.H, act.W, act.C)
            act.written = None
            return act.grad, 0
        if act.written == [False, False]:     # pre-allocated two-stream buffer, nothing in it yet
            act.written = None
            return act.grad, 0
        _complete(act)
        act.gradp = None
        return act.grad, 1

    def give_grad(self, act, g, planes=None):
        """Hands an owned, contiguous gradient buffer to `act` (planes: a bf16-plane copy of it, kept only when `g` becomes
        the gradient as it is)."""
        if not act.need_grad:
            return
        if planes is not None and act.parent is None and (act.grad is None or act.written == [False, False]):
            act.gradp = planes
        if act.parent is not None:
            buf, acc = self.grad_target(act)
            if acc:
                hip.add_(buf, g)
            else:
                hip.copy_channels(g, act.C, 1, buf, act.C, 1, act.rows, act.C)
            return
        if act.grad is None or act.written == [False, False]:
            act.grad = g                      # (a pre-allocated two-stream gradient buffer nobody has written: replace it)
            act.written = None
        else:
            _complete(act)
            act.gradp = None
            hip.add_(act.grad, g)


def _complete(act):
    """A two-stream gradient buffer of which only one half has been written so far: zero the other half."""
    w = act.written
    if w is not None:
        n = act.B // 2
        for k in range(2):
            if not w[k]:
                hip.fill_(act.grad[k * n:(k + 1) * n], 0.0)
        act.written = None


def take_grad(act):
    par = act.parent
    if par is not None:
        if par.grad is None or (par.written is not None and not par.written[act.slot]):
            return None
        n = act.B
        return par.grad[act.slot * n:(act.slot + 1) * n]
    g = act.grad
    if g is not None:
        if act.written == [False, False]:     # pre-allocated two-stream buffer that received no gradient
            act.grad = None
            return None
        _complete(act)
    act.grad = None
    return g


def take_gradp(act):
    """The plane copy of the gradient take_grad just returned (None: there is none)."""
    gp, act.gradp = act.gradp, None
    return gp


def _pair(m):
    """(first, second or None) of a module / tensor or a pair of them (two-stream launch)."""
    return (m[0], m[1]) if isinstance(m, tuple) else (m, None)


def _attr(ms, name):
    a, b = _pair(ms)
    return (getattr(a, name), getattr(b, name)) if b is not None else getattr(a, name)


# ----------------------------------------------------------------------------------------- conv helpers
def conv_geom(x, conv):
    co, ci, kh, kw = conv.weight.shape
    s, p, d = conv.stride[0], conv.padding[0], conv.dilation[0]
    OH = (x.H + 2 * p - d * (kh - 1) - 1) // s + 1
    OW = (x.W + 2 * p - d * (kw - 1) - 1) // s + 1
    return co, ci, kh, kw, s, p, d, OH, OW


def _pair_kw(rt, x, convs, w_of, bias=True, res_scale=None, kscale=None):
    """Keyword arguments of a two-stream conv2d launch (empty for a single module): rows of the first stream and
    the second stream's parameter set."""
    c0, c1 = _pair(convs)
    if c1 is None:
        return {}
    rs0, rs1 = _pair(res_scale) if res_scale is not None else (None, None)
    ks0, ks1 = _pair(kscale) if kscale is not None else (None, None)
    assert x.B % 2 == 0 and (x.B // 2) * x.H * x.W % 128 == 0
    return dict(w2=w_of(c1), bias2=c1.bias if bias else None, res_scale2=rs1, kscale2=ks1)


def conv_call(rt, x, conv, out, act=0, ypre=None, res=None, res_scale=None, nchw=None, bias=True, stats=False, bn_stats=False):
    """out: Act (NHWC target) or, with nchw=(tensor, ctot, coff), a channel range of an NCHW tensor.
    conv / res_scale may be pairs (two-stream launch over a (2B,...) input: first half of the rows = first module)."""
    c0, c1 = _pair(conv)
    co, ci, kh, kw, s, p, d, OH, OW = conv_geom(x, c0)
    b = c0.bias if bias else None
    if nchw is None:
        pairs, per = hip.conv_stats_buffer(x.B, OH * OW, co, x.t.device) if stats else (None, 0)
        kw2 = {}
        if bn_stats and rt.bn_colstats and c1 is None and rt.training and hip.colstats_ok(OH * OW, co, out.ld, x.ld) and out.t.data_ptr() % 16 == 0 \
                and x.t.data_ptr() % 16 == 0 and ci % 4 == 0:
            # train-mode BatchNorm behind this conv: its per-channel batch sums come out of the epilogue (no pass over z)
            out.colpart, _ = hip.colstats_buffers(x.B, OH * OW, co, x.t.device)
            kw2["colstats"] = (out.colpart, None, 0, None, None)
        if c1 is not None:
            kw2 = _pair_kw(rt, x, conv, rt.weight, bias, res_scale)
            kw2["pair_rows"] = (x.B // 2) * OH * OW
        if c1 is None and kh == 1 and kw == 1 and s == 1 and p == 0:
            kw2["w_planes"] = rt.planes(c0.weight, 0, co, ci, x.B * OH * OW)
        hip.conv2d(x.t, x.ld, rt.weight(c0), b, out.t, out.ld, x.B, x.H, x.W, ci, OH, OW, co, kh, kw, s, p, d,
                   mode=0, act=act, ypre=None if ypre is None else ypre.t, ldypre=0 if ypre is None else ypre.ld,
                   res=None if res is None else res.t, ldres=0 if res is None else res.ld,
                   res_scale=_pair(res_scale)[0] if res_scale is not None else None,
                   stats=pairs, precision=rt.prec_fwd(x.ld, ci, co), **kw2)
        if pairs is not None:       # statistics of the stored output: the consumer's GroupNorm skips its moments pass
            out.pairs = (pairs, per)
    else:
        assert c1 is None
        t, ctot, coff = nchw
        hip.conv2d(x.t, x.ld, rt.weight(c0), b, t, 0, x.B, x.H, x.W, ci, OH, OW, co, kh, kw, s, p, d, mode=0, act=act,
                   out_nchw=1, out_ctot=ctot, out_coff=coff, precision=rt.prec_fwd(x.ld, ci, co))


def conv_backward(rt, x, conv, dy, lddy, kscale=None, aux=None, row_scale=None, skip_bias=False, dx_to=None,
                  defer_ok=True, ls_grad=None, no_dx=False):
    """Gradients of y = conv(x): weight/bias into the parameter table, dx accumulated into x.grad
    (or written to the Act `dx_to`).  dy: tensor whose data_ptr is the (0,0) element, row stride lddy.
    conv / kscale / row_scale / ls_grad may be pairs (two-stream launch).  ls_grad: the layer-scale parameter(s) behind
    this (1x1) conv: their gradient comes out of the weight-gradient slabs (hip.conv2d_wgrad, dls).
    no_dx: parameter gradients only (the fused Mlp kernel has produced the data gradient)."""
    c0, c1 = _pair(conv)
    co, ci, kh, kw, s, p, d, OH, OW = conv_geom(x, c0)
    gw, accw = rt.pgrad(c0.weight)
    gb, accb = (None, 0) if (c0.bias is None or skip_bias) else rt.pgrad(c0.bias)
    gw2 = gb2 = None
    if c1 is not None:
        gw2, accw2 = rt.pgrad(c1.weight)
        gb2, accb2 = (None, 0) if (c1.bias is None or skip_bias) else rt.pgrad(c1.bias)
        assert (gw is None) == (gw2 is None) and (gb is None) == (gb2 is None) and (gw is None or accw == accw2), \
            "the two streams of a stage must be frozen / trained together"
    rs0, rs1 = _pair(row_scale) if row_scale is not None else (None, None)
    kwl = {}
    l0 = l1 = None
    if ls_grad is not None:
        l0, l1 = _pair(ls_grad)
        gl, accl = rt.pgrad(l0)
        if gl is not None:
            assert gw is not None and accl == accw and (c0.bias is None or gb is not None), \
                "a layer scale is trained together with the conv in front of it"
            kwl = dict(w=c0.weight, bias=c0.bias, dls=gl)
            if l1 is not None:
                gl2, _ = rt.pgrad(l1)
                kwl.update(w2=c1.weight, bias2=c1.bias, dls2=gl2)
    if gw is not None or gb is not None:
        assert gb is None or gw is not None
        assert gb is None or accb == accw

        def wgrad():
            hip.conv2d_wgrad(x.t, x.ld, dy, lddy, gw, gb, rs0, x.B, x.H, x.W, ci, OH, OW, co, kh, kw, s, p, d,
                             accumulate=accw, precision=rt.prec_wgrad(x.ld, lddy, ci, co), dw2=gw2, dbias2=gb2,
                             row_scale2=rs1, **kwl)
            if rt.on_param_grad:
                for cv, g, l in ((c0, gb, l0), (c1, gb2, l1)):
                    if cv is not None:
                        rt.on_param_grad(cv.weight)
                        if g is not None:
                            rt.on_param_grad(cv.bias)
                        if l is not None and kwl:
                            rt.on_param_grad(l)
        if defer_ok:
            rt.aside(wgrad, (x.t, dy))
        else:                    # dy is updated in place later in this closure: the weight gradient must read it now
            wgrad()
    target = dx_to if dx_to is not None else (x if x.need_grad else None)
    if target is not None and not no_dx:
        if dx_to is not None:
            buf, acc, ld = dx_to.t, 0, dx_to.ld
        else:
            buf, acc = rt.grad_target(x)
            ld = x.C
        ks0, ks1 = _pair(kscale) if kscale is not None else (None, None)
        rows = x.B * x.H * x.W if not isinstance(conv, tuple) or (x.B // 2) * x.H * x.W % 128 == 0 else None
        wd, ks, prec = rt.dgrad_operands(c0, c0.weight, rt.weight(c0), co, ci, kh, kw, ks0, lddy, rows)
        kw2 = {}
        if c1 is not None:
            wd1, ks1, prec1 = rt.dgrad_operands(c1, c1.weight, rt.weight(c1), co, ci, kh, kw, ks1, lddy, rows)
            assert prec1 == prec
            kw2 = dict(pair_rows=(x.B // 2) * x.H * x.W, w2=wd1, kscale2=ks1)
        if c1 is None and kh == 1 and kw == 1 and s == 1 and p == 0 and prec == 2:
            kw2["w_planes"] = rt.planes(c0.weight, 1, ci, co, x.B * x.H * x.W, kscale=ks)
        hip.conv2d(dy, lddy, wd, None, buf, ld, x.B, x.H, x.W, ci, OH, OW, co, kh, kw, s, p, d, mode=1,
                   kscale=ks, aux=None if aux is None else aux.t, ldaux=0 if aux is None else aux.ld, accumulate=acc,
                   precision=prec, **kw2)


def simple_conv(rt, x, conv, out=None):
    """Plain conv with bias (PointRecuder.proj): recorded on the tape.  conv may be a pair (two-stream launch)."""
    co, ci, kh, kw, s, p, d, OH, OW = conv_geom(x, _pair(conv)[0])
    if out is not None:
        y = out
    else:
        y = rt.new_pair(x.B, OH, OW, co) if isinstance(conv, tuple) else rt.new(x.B, OH, OW, co)
    conv_call(rt, x, conv, y)

    def bwd():
        g = take_grad(y)
        if g is None:
            return
        conv_backward(rt, x, conv, g, y.C)
    rt.push(bwd)
    return y


# ----------------------------------------------------------------------------------------- norms
def bn_fwd_coef(rt, z, bn):
    """Coefficients of y = A (z - S) + D for BatchNorm `bn` over z (train mode: batch statistics, running statistics updated --
    from the producer's column partials when it left any, else by a moments pass; eval mode: running statistics).
    Returns (A, D, S, ctx)."""
    B, HW, C = z.B, z.HW, z.C
    A, D, S, ms = rt.buf(C), rt.buf(C), rt.buf(C), rt.buf(C, 2)
    if rt.training and rt.sync_bn is None and z.colpart is not None and z.colchunks:
        hip.bn_coef_fwd_from_chunks(z.colpart, z.colchunks, B * HW, bn.weight, bn.bias, bn.eps, bn.momentum, bn.running_mean,
                                    bn.running_var, bn.num_batches_tracked, C, A, D, S, ms)
        z.colpart, z.colchunks = None, 0
    elif rt.training and rt.sync_bn is not None:
        # synchronised BatchNorm: per-sample moments -> sum over samples and ranks -> coefficients with the global count
        tot = rt.sync_bn.total(hip.moments(z.t, z.ld, B, HW, C))
        hip.bn_coef_fwd(tot, bn.weight, bn.bias, bn.eps, bn.momentum, bn.running_mean, bn.running_var,
                        bn.num_batches_tracked, True, 1, rt.sync_bn.count(B, HW, rt.sync_batch_total), C, A, D, S, ms)
        z.colpart = None
    elif rt.training and z.colpart is not None:
        hip.bn_coef_fwd_from_partials(z.colpart, bn.weight, bn.bias, bn.eps, bn.momentum, bn.running_mean, bn.running_var,
                                      bn.num_batches_tracked, B, HW, C, A, D, S, ms)
        z.colpart = None
    elif rt.training:
        hip.bn_stats_fwd(z.t, z.ld, bn.weight, bn.bias, bn.eps, bn.momentum, bn.running_mean, bn.running_var,
                         bn.num_batches_tracked, B, HW, C, A, D, S, ms)
    else:
        hip.bn_coef_fwd(None, bn.weight, bn.bias, bn.eps, bn.momentum, bn.running_mean, bn.running_var,
                        bn.num_batches_tracked, False, B, HW, C, A, D, S, ms)
        z.colpart, z.colchunks = None, 0
    return A, D, S, ms


def bn_forward(rt, z, bn, relu, out=None, residual=None):
    """y = [relu](BN(z)) [+ residual].  Returns (y, ctx) where ctx feeds bn_backward."""
    B, HW, C = z.B, z.HW, z.C
    A, D, S, ms = bn_fwd_coef(rt, z, bn)
    y = out if out is not None else rt.new(z.B, z.H, z.W, C)
    hip.affine(y.t, y.ld, B, HW, C, x1=z.t, ld1=z.ld, A=A, D1=D, S1=S, pre=1 if relu else 0,
               x2=None if residual is None else residual.t, ld2=0 if residual is None else residual.ld)
    if relu and rt.relu_masks is not None:
        rt.relu_masks[bn] = y
    if relu:
        ms.fwd_coef = (A, D, S)      # bn_backward recomputes the ReLU mask from z with these instead of reading y
    return y, ms


def bn_bwd_coef(rt, bn, z, ms, dy, lddy):
    """Coefficients (A, E, D, S) of dz = A dy + E (z - S) + D for y = BN(z) (no ReLU) and its parameter gradients: the moments
    pass + the coefficient kernel of bn_backward without the apply launch (the caller's fused kernel applies them)."""
    B, HW, C = z.B, z.HW, z.C
    (gw, gb), accw = _pgrads_or_scratch(rt, (bn.weight, bn.bias), (C, C))
    A, E, D, S = rt.buf(C), rt.buf(C), rt.buf(C), rt.buf(C)
    hip.bn_stats_bwd(dy, lddy, z.t, z.ld, None, 0, ms, bn.weight, rt.training, B, HW, C, A, E, D, S, gw, gb, accw)
    if rt.on_param_grad:
        rt.on_param_grad(bn.weight)
        rt.on_param_grad(bn.bias)
    return A, E, D, S


def bn_backward(rt, bn, z, ms, dy, lddy, mask=None, dz_out=None):
    """dz of y = [relu](BN(z)); mask = the ReLU output (None: no ReLU).  Returns contiguous dz tensor."""
    B, HW, C = z.B, z.HW, z.C
    gw, accw = rt.pgrad(bn.weight)
    gb, accb = rt.pgrad(bn.bias)
    if gw is None:
        gw, accw = rt.buf(C), 0
    if gb is None:
        gb = rt.buf(C)
    A, E, D, S = rt.buf(C), rt.buf(C), rt.buf(C), rt.buf(C)
    if rt.training and rt.sync_bn is not None:
        # the two gradient sums over all ranks for dz; d gamma / d beta from the LOCAL sums (as SyncBatchNorm: the gradient
        # all-reduce averages them afterwards)
        mom2 = hip.moments(dy, lddy, B, HW, C, x2=z.t, ldx2=z.ld, mask=None if mask is None else mask.t,
                           ldm=0 if mask is None else mask.ld)
        tot2 = rt.sync_bn.total(mom2)
        hip.bn_coef_bwd(tot2, ms, bn.weight, True, 1, rt.sync_bn.count(B, HW, rt.sync_batch_total), C, A, E, D, S, rt.buf(C), rt.buf(C), 0)
        hip.bn_coef_bwd(mom2, ms, bn.weight, True, B, HW, C, rt.buf(C), rt.buf(C), rt.buf(C), rt.buf(C), gw, gb, accw)
    else:
        fwd = getattr(ms, "fwd_coef", None) if (mask is not None and BN_ZMASK) else None
        if fwd is not None:
            # y = ReLU(BN(z)): neither pass reads y -- two tensor reads each where there were three
            hip.bn_stats_bwd_zmask(dy, lddy, z.t, z.ld, fwd, ms, bn.weight, rt.training, B, HW, C, A, E, D, S, gw, gb, accw)
            if rt.on_param_grad:
                rt.on_param_grad(bn.weight)
                rt.on_param_grad(bn.bias)
            dz = dz_out if dz_out is not None else rt.buf(z.B, z.H, z.W, C)
            hip.bn_apply_bwd_zmask(dy, lddy, z.t, z.ld, fwd, A, E, D, S, dz, C, B, HW, C)
            return dz
        hip.bn_stats_bwd(dy, lddy, z.t, z.ld, None if mask is None else mask.t, 0 if mask is None else mask.ld, ms, bn.weight,
                         rt.training, B, HW, C, A, E, D, S, gw, gb, accw)
    if rt.on_param_grad:
        rt.on_param_grad(bn.weight)
        rt.on_param_grad(bn.bias)
    dz = dz_out if dz_out is not None else rt.buf(z.B, z.H, z.W, C)
    hip.affine(dz, C, B, HW, C, x1=dy, ld1=lddy, A=A, pre=2 if mask is not None else 0,
               masky=None if mask is None else mask.t, ldm=0 if mask is None else mask.ld, x2=z.t, ld2=z.ld, E=E, D2=D,
               S2=S)
    return dz


def gn_forward(rt, x, gn):
    """GroupNorm(1, C); gn may be a pair of modules (two-stream launch: second half of the samples = second module)."""
    g0, g1 = _pair(gn)
    B, HW, C = x.B, x.HW, x.C
    ms = rt.buf(B, 2)
    kw2 = {} if g1 is None else dict(gamma2=g1.weight, beta2=g1.bias)
    if x.pairs is not None and g1 is None and hip.gn_apply_ok(C, x.ld):
        # statistics from the producer's epilogue AND the coefficient step inside the apply kernel: ONE launch on the chain
        y = rt.new(x.B, x.H, x.W, C)
        hip.gn_apply_fwd(x.t, x.ld, x.pairs[0], x.pairs[1], g0.weight, g0.bias, g0.eps, B, HW, C, y.t, C, ms)
        return y, ms
    A, D, S = rt.buf(B, C), rt.buf(B, C), rt.buf(B, C)
    if x.pairs is not None:
        hip.gn_coef_from_pairs(x.pairs[0], x.pairs[1], g0.weight, g0.bias, g0.eps, B, HW, C, A, D, S, ms, **kw2)
    else:
        hip.gn_stats_fwd(x.t, x.ld, g0.weight, g0.bias, g0.eps, B, HW, C, A, D, S, ms, **kw2)
    y = rt.new(x.B, x.H, x.W, C)
    hip.affine(y.t, C, B, HW, C, x1=x.t, ld1=x.ld, A=A, D1=D, S1=S, bstride=C)
    return y, ms


def _pgrads_or_scratch(rt, params, sizes):
    """Gradient buffers for a list of parameters (scratch where a parameter needs no gradient) and ONE accumulate flag."""
    outs, acc = [], 0
    for prm, n in zip(params, sizes):
        g, a = rt.pgrad(prm)
        if g is None:
            g, a = rt.buf(n), 0
        else:
            acc = a
        outs.append(g)
    return outs, acc


def gn_backward(rt, gn, x, ms, dy, out, accumulate=0, add=None):
    """out = dx of y = GN(x) given contiguous dy [+ out (accumulate) | + add (another contiguous tensor)].
    (Round 3 could also take the moments from the epilogue of the data-gradient conv that produced dy -- vrnet_conv_colstats with
    x2 / gamma, vrnet_gn_apply_bwd_from_partials: measured neutral-to-negative in the step in rounds 3-5, never on by default; the
    program path was removed in round 6, the library entry points and their tests remain.)"""
    g0, g1 = _pair(gn)
    B, HW, C = x.B, x.HW, x.C
    if g1 is None and hip.gn_apply_ok(C, x.ld) and (not accumulate or add is None):
        # two launches (moments; apply + parameter gradients) where moments, reduce, coefficients and affine were four
        (gw, gb), accw = _pgrads_or_scratch(rt, (g0.weight, g0.bias), (C, C))
        hip.gn_apply_bwd(dy, C, x.t, x.ld, ms, g0.weight, B, HW, C, out, C, gw, gb, accw, add=out if accumulate else add,
                         ldadd=C if (accumulate or add is not None) else 0)
        if rt.on_param_grad:
            rt.on_param_grad(g0.weight)
            rt.on_param_grad(g0.bias)
        return
    mom2 = hip.moments(dy, C, B, HW, C, x2=x.t, ldx2=x.ld)
    A, E, D, S = rt.buf(B, C), rt.buf(B, C), rt.buf(B, C), rt.buf(B, C)
    (gw, gb), accw = _pgrads_or_scratch(rt, (g0.weight, g0.bias), (C, C))
    kw2 = {}
    if g1 is not None:
        (gw2, gb2), accw2 = _pgrads_or_scratch(rt, (g1.weight, g1.bias), (C, C))
        assert accw2 == accw
        kw2 = dict(gamma2=g1.weight, dgamma2=gw2, dbeta2=gb2)
    hip.gn_coef_bwd(mom2, ms, g0.weight, B, HW, C, A, E, D, S, gw, gb, accw, **kw2)
    if rt.on_param_grad:
        for g in (g0, g1):
            if g is not None:
                rt.on_param_grad(g.weight)
                rt.on_param_grad(g.bias)
    hip.affine(out, C, B, HW, C, x1=dy, ld1=C, A=A, x2=x.t, ld2=x.ld, E=E, D2=D, S2=S, bstride=C, accumulate=accumulate,
               add=add, ldadd=0 if add is None else C)


# ----------------------------------------------------------------------------------------- BaseConv
def base_conv(rt, x, m, out=None):
    """BaseConv (normal_conv.py:36-49): [dw3x3 ->] conv -> BN -> ReLU."""
    if m.ds_conv:
        return ds_base_conv(rt, x, m, out)
    conv = m.conv
    co, ci, kh, kw, s, p, d, OH, OW = conv_geom(x, conv)
    z = rt.new(x.B, OH, OW, co)
    conv_call(rt, x, conv, z, bias=False, bn_stats=True)
    y, ms = bn_forward(rt, z, m.bn, relu=True, out=out)

    def bwd():
        g = take_grad(y)
        if g is None:
            return
        dz = bn_backward(rt, m.bn, z, ms, g, y.C, mask=y)
        conv_backward(rt, x, conv, dz, co)
    rt.push(bwd)
    return y


def ds_base_conv(rt, x, m, out=None):
    """BaseConv(ds_conv=True): depthwise 3x3 (no bias) -> pointwise 1x1 (no bias) -> BN -> ReLU."""
    dconv, pconv = m.conv.dconv, m.conv.pconv
    C = x.C
    t = rt.new(x.B, x.H, x.W, C)
    hip.dwconv3x3(x.t, x.ld, dconv.weight, t.t, C, x.B, x.H, x.W, C)
    z = rt.new(x.B, x.H, x.W, pconv.weight.shape[0])
    conv_call(rt, t, pconv, z, bias=False, bn_stats=True)
    y, ms = bn_forward(rt, z, m.bn, relu=True, out=out)

    def bwd():
        g = take_grad(y)
        if g is None:
            return
        dz = bn_backward(rt, m.bn, z, ms, g, y.C, mask=y)
        conv_backward(rt, t, pconv, dz, z.C)
        dt = take_grad(t)
        gw, acc = rt.pgrad(dconv.weight)
        if gw is not None:
            def dw_wgrad(acc=acc):   # (round 5: off the chain like every other weight gradient -- two launches, ~21 us per layer,
                hip.dwconv3x3_wgrad(x.t, x.ld, dt, C, gw, x.B, x.H, x.W, C, accumulate=acc)      # that the data gradient below
                if rt.on_param_grad:                                                            # used to queue behind)
                    rt.on_param_grad(dconv.weight)
            rt.aside(dw_wgrad, (x.t, dt))
        if x.need_grad:
            buf, acc = rt.grad_target(x)
            hip.dwconv3x3(dt, C, dconv.weight, buf, C, x.B, x.H, x.W, C, flip=1, accumulate=acc)
    rt.push(bwd)
    return y


# ----------------------------------------------------------------------------------------- ClusterBlock
def cluster_block(rt, x, m, name=None):
    """ClusterBlock.forward (vr_coc.py:264-271): x + ls1*Cluster(GN(x)), then + ls2*Mlp(GN(.)).
    m / name may be pairs: the image and the radar block of a backbone stage (vr_coc.py:589-600) as ONE launch per
    layer over a (2B,H,W,C) two-stream buffer, each half with its own parameters."""
    m0, m1 = _pair(m)
    paired = m1 is not None
    tm, mlp = _attr(m, "token_mixer"), _attr(m, "mlp")
    tm0, tm1 = _pair(tm)
    mlp0 = _pair(mlp)[0]
    B, H, W, C = x.B, x.H, x.W, x.C
    E, Dh, fold = tm0.heads, tm0.head_dim, tm0.fold
    ED = E * Dh
    rows_half = (B // 2) * H * W
    if rt.pnp and not paired and C % 8 == 0 and ED % 8 == 0:
        hid_ = mlp0.fc1.weight.shape[0]
        pmlp_ = rt.prec_mlp(C, hid_, B * H * W, H * W)
        plan = _planes_plan(rt, B * H * W, C, ED, hid_, bool(pmlp_))
        if any(any(v) for v in plan.values()) or (pmlp_ and rt.pnp == 1 and rt.pg_wgrad):
            return cluster_block_planes(rt, x, m0, name, plan, pmlp_)
    wcat, bcat = tm0._fused_qkv                                  # [fc1 ; fc_v]: one GEMM, f | v side by side
    kwq = dict(pair_rows=rows_half, w2=tm1._fused_qkv[0], bias2=tm1._fused_qkv[1]) if paired else {}
    fv = rt.new(B, H, W, 2 * ED)
    if not paired:
        kwq["w_planes"] = rt.planes(wcat, 0, 2 * ED, C, B * H * W)
    xn, ms1 = gn_forward(rt, x, _attr(m, "norm1"))
    hip.conv2d(xn.t, xn.ld, wcat, bcat, fv.t, 2 * ED, B, H, W, C, H, W, 2 * ED, 1, 1, 1, 0, 1, mode=0,
               precision=rt.prec_fwd(xn.ld, C, 2 * ED), **kwq)
    f_t, v_t = fv.t, fv.t[..., ED:]
    o = rt.new(B, H, W, ED)
    idx = rt.buf(B, H, W, E, dtype=torch.uint8)
    big = (H // max(fold, 1)) * (W // max(fold, 1)) > 256       # streaming kernel keeps the similarity map
    kwc = dict(alpha2=tm1.sim_alpha, beta2=tm1.sim_beta) if paired else {}
    if rt.forced_idx is not None and not paired and name is not None and name in rt.forced_idx:
        idx.copy_(rt.forced_idx[name])                          # teacher-forced assignment (model.forced_idx_maps: parity tests)
        kwc["forced"] = True
    # regions of more than 256 points (streaming kernel): the forward keeps its similarity map and per-region state, the
    # backward then skips the two passes that would recompute them
    wgt = rt.buf(B, H, W, E) if big else None
    nst = hip.cluster_state_floats(B, H, W, E, fold) if (big and not paired and rt.record) else 0
    cstate = rt.buf(nst) if nst else None
    if cstate is not None:
        kwc["state"] = cstate
    hip.cluster_fwd(f_t, v_t, 2 * ED, tm0.sim_alpha, tm0.sim_beta, o.t, ED, idx, wgt, B, H, W, E, Dh, fold, **kwc)
    if name is not None:
        n0, n1 = _pair(name)
        if paired:
            rt.idx_maps[n0], rt.idx_maps[n1] = idx[:B // 2], idx[B // 2:]
        else:
            rt.idx_maps[n0] = idx
    ls1, ls2 = _attr(m, "layer_scale_1"), _attr(m, "layer_scale_2")
    x1 = rt.new(B, H, W, C)
    conv_call(rt, o, _attr(tm, "fc2"), x1, res=x, res_scale=ls1, stats=True)
    hid = mlp0.fc1.weight.shape[0]
    x2 = rt.new(B, H, W, C)
    pmlp = 0 if paired else rt.prec_mlp(C, hid, B * H * W, H * W)
    mlp_rc = bool(pmlp) and pmlp == 2 and rt.mlp_rc(C, hid, False)
    u = rt.new(B, H, W, hid, need_grad=False) if (rt.record and not mlp_rc) else None
    xn2, ms2 = gn_forward(rt, x1, _attr(m, "norm2"))
    if pmlp:
        # fc1 -> GELU -> fc2 (+ layer-scale residual, + GroupNorm statistics of the output) as ONE kernel: the hidden
        # activation never reaches HBM; only the pre-activation is stored, for the backward pass -- or (mlp_rc) nothing at all:
        # the backward kernel recomputes it from xn2
        packs = rt.mlp_packs(mlp0, C, hid, pmlp, rc=mlp_rc)
        pairs, per = hip.conv_stats_buffer(B, H * W, C, x.t.device)
        hip.mlp_fwd(xn2.t, xn2.ld, packs[0], mlp0.fc1.bias, mlp0.fc2.bias, x1.t, x1.ld, ls2, x2.t, C,
                    None if u is None else u.t, hid, pairs, B * H * W, C, hid, pmlp)
        if pairs is not None:
            x2.pairs = (pairs, per)
        h = None
    else:
        h = rt.new(B, H, W, hid)
        conv_call(rt, xn2, _attr(mlp, "fc1"), h, act=2, ypre=u)
        conv_call(rt, h, _attr(mlp, "fc2"), x2, res=x1, res_scale=ls2, stats=True)

    def bwd():
        dx2 = take_grad(x2)
        if dx2 is None:
            return
        # Nothing below updates a gradient map in place, so EVERY weight gradient of the block runs beside the data
        # gradients on the auxiliary streams.  The layer-scale gradients come out of the fc2 weight-gradient slabs
        # (conv_backward, ls_grad): the branch outputs are not stored and no (dx, branch) moments pass exists.
        # ---- MLP branch
        du = rt.new(B, H, W, hid)
        dxn2 = rt.new(B, H, W, C)
        if pmlp:
            # one kernel: d(pre-activation) and the recomputed activation are written once for the two weight gradients,
            # which run beside the rest of the block's backward like every other weight gradient
            hb = rt.new(B, H, W, hid, need_grad=False)
            if mlp_rc:
                hip.mlp_bwd_rc(dx2, C, ls2, packs[1], xn2.t, xn2.ld, mlp0.fc1.bias, hb.t, hid, du.t, hid, dxn2.t, C, B * H * W, C, hid, 2)
            else:
                hip.mlp_bwd(dx2, C, ls2, packs[1], u.t, hid, hb.t, hid, du.t, hid, dxn2.t, C, B * H * W, C, hid, pmlp)
            conv_backward(rt, hb, mlp0.fc2, dx2, C, row_scale=ls2, ls_grad=ls2, no_dx=True)
            conv_backward(rt, xn2, mlp0.fc1, du.t, hid, no_dx=True)
        else:
            conv_backward(rt, h, _attr(mlp, "fc2"), dx2, C, kscale=ls2, aux=u, row_scale=ls2, dx_to=du, ls_grad=ls2)
            # (GroupNorm-backward moments from this conv's epilogue: built in round 3, neutral in rounds 3-5, removed in round 6)
            conv_backward(rt, xn2, _attr(mlp, "fc1"), du.t, hid, dx_to=dxn2)
        dx1 = rt.buf(B, H, W, C)
        gn_backward(rt, _attr(m, "norm2"), x1, ms2, dxn2.t, dx1, add=dx2)   # dx1 = dx2 + d(GN -> Mlp branch)
        # ---- Cluster branch
        do = rt.new(B, H, W, ED)
        conv_backward(rt, o, _attr(tm, "fc2"), dx1, C, kscale=ls1, row_scale=ls1, dx_to=do, ls_grad=ls1)
        dfv = rt.new(B, H, W, 2 * ED)
        (ga, gb), acca = _pgrads_or_scratch(rt, (tm0.sim_alpha, tm0.sim_beta), (1, 1))
        kwb = {}
        if paired:
            (ga2, gb2), acca2 = _pgrads_or_scratch(rt, (tm1.sim_alpha, tm1.sim_beta), (1, 1))
            assert acca2 == acca
            kwb = dict(alpha2=tm1.sim_alpha, beta2=tm1.sim_beta, dalpha2=ga2, dbeta2=gb2)
        if cstate is not None:
            kwb["saved"] = (wgt, cstate)
        if paired:
            hip.cluster_bwd(f_t, v_t, 2 * ED, tm0.sim_alpha, tm0.sim_beta, idx, do.t, ED, dfv.t, dfv.t[..., ED:], 2 * ED, ga, gb,
                            acca, B, H, W, E, Dh, fold, **kwb)
            if rt.on_param_grad:
                for t in (tm0, tm1):
                    rt.on_param_grad(t.sim_alpha)
                    rt.on_param_grad(t.sim_beta)
        else:       # the two scalars are finished by the section's ONE rt.flush_cluster_ab() launch
            ws_ab = hip.cluster_bwd(f_t, v_t, 2 * ED, tm0.sim_alpha, tm0.sim_beta, idx, do.t, ED, dfv.t, dfv.t[..., ED:], 2 * ED,
                                    None, None, 0, B, H, W, E, Dh, fold, **kwb)
            rt.pending_ab.append((ws_ab, B * E * max(fold, 1) ** 2, ga, gb, acca, tm0))
        _fused_qkv_wgrad(rt, tm, xn, dfv)
        dxn = rt.new(B, H, W, C)                                 # d xn = [df | dv] . [fc1 ; fc_v]: one data-gradient GEMM
        wd, _, prec = rt.dgrad_operands(tm0, wcat, wcat, 2 * ED, C, 1, 1, None, 2 * ED, B * H * W)
        kwd = {}
        if paired:
            wd1, _, _ = rt.dgrad_operands(tm1, tm1._fused_qkv[0], tm1._fused_qkv[0], 2 * ED, C, 1, 1, None, 2 * ED, B * H * W)
            kwd = dict(pair_rows=rows_half, w2=wd1)
        if not paired and prec == 2:
            kwd["w_planes"] = rt.planes(wcat, 1, C, 2 * ED, B * H * W)
        hip.conv2d(dfv.t, 2 * ED, wd, None, dxn.t, C, B, H, W, C, H, W, 2 * ED, 1, 1, 1, 0, 1, mode=1, precision=prec, **kwd)
        dx = rt.buf(B, H, W, C)
        gn_backward(rt, _attr(m, "norm1"), x, ms1, dxn.t, dx, add=dx1)     # dx = dx1 + d(GN -> Cluster branch)
        rt.give_grad(x, dx)
    rt.push(bwd)
    return x2


def _planes_plan(rt, M, C, ED, hid, fused_mlp):
    """Which GEMMs of a ClusterBlock of M pixels run on plane operands: {conv: (forward, data gradient, weight gradient)},
    conv in fcfv / proj / fc1 / fc2 (rows x columns x contraction of each product; the Mlp convs only when the block has no
    fused Mlp kernel)."""
    def gemm(cols, K):
        return rt.pg_fwd and M <= PG_MAX_ROWS and hip.gemm_planes_ok(M, cols, K)

    def wg(ci, co):
        return rt.pg_wgrad and M <= PG_MAX_ROWS and hip.wgrad_planes_ok(M, ci, co)
    plan = {"fcfv": (gemm(2 * ED, C), gemm(C, 2 * ED), wg(C, 2 * ED)),
            "proj": (gemm(C, ED), gemm(ED, C), wg(ED, C))}
    if not fused_mlp:
        plan["fc1"] = (gemm(hid, C), gemm(C, hid), wg(C, hid))
        plan["fc2"] = (gemm(C, hid), gemm(hid, C), wg(hid, C))
    else:
        plan["fc1"] = plan["fc2"] = (False, False, False)
    return plan


def cluster_block_planes(rt, x, m, name, plan, pmlp):
    """cluster_block for ONE stream with the GEMMs `plan` selects on plane operands (csrc/pgemm.hip): the tensors that are
    only ever GEMM operands -- both GroupNorm outputs, the Cluster output, the Mlp hidden activation, and on the way back the
    block's incoming gradient, d(pre-activation), d(residual) and [df | dv] -- are written as bf16 planes by the kernels that
    produce them (np = 3: the fp32 value, exactly; np = 1: bf16 tensors) and the GEMMs split nothing.  A tensor keeps its fp32
    form only while some consumer still wants it (a GEMM kind the plan leaves on the in-kernel-split kernels)."""
    np_ = rt.pnp
    pw = rt.pweights
    tm, mlp = m.token_mixer, m.mlp
    B, H, W, C = x.B, x.H, x.W, x.C
    E, Dh, fold = tm.heads, tm.head_dim, tm.fold
    ED, M, HW = E * Dh, B * H * W, H * W
    hid = mlp.fc1.weight.shape[0]
    wcat, bcat = tm._fused_qkv
    ls1, ls2 = m.layer_scale_1, m.layer_scale_2
    P = lambda c: hip.Planes.empty(np_, (B, H, W, c), x.t.device)
    f32 = lambda c, ng=True: rt.new(B, H, W, c, need_grad=ng)
    fcfv, proj, fc1, fc2 = plan["fcfv"], plan["proj"], plan["fc1"], plan["fc2"]
    rec = rt.record

    def stats_buf(c):
        return hip.conv_stats_buffer(B, HW, c, x.t.device)

    # ---- GroupNorm 1 -> [fc1 | fc_v]
    xn_p = P(C) if (fcfv[0] or (rec and fcfv[2])) else None
    xn_f = f32(C) if (not fcfv[0] or (rec and not fcfv[2])) else None
    ms1 = rt.buf(B, 2)
    if x.pairs is not None and hip.gn_apply_ok(C, x.ld):
        hip.gn_apply_fwd(x.t, x.ld, x.pairs[0], x.pairs[1], m.norm1.weight, m.norm1.bias, m.norm1.eps, B, HW, C,
                         None if xn_f is None else xn_f.t, C, ms1, planes=xn_p)
    else:      # no tile statistics from the producer: the three-launch form, then one conversion pass
        xn_f, ms1 = gn_forward(rt, x, m.norm1)
        if xn_p is not None:
            hip.planes_from_f32(xn_f.t, C, M, C, xn_p)
    # (f | v and d(out) stay fp32 in bf16 mode too: bf16 tensors for them -- the Cluster kernels take either,
    # vrnet_cluster_*_planes_f32(in_bf16) -- measured 34.0 against 33.7 ms per step at batch 16: those kernels are not bound
    # by the bytes they read)
    fv = f32(2 * ED)
    fv_t = fv.t
    if fcfv[0]:
        hip.gemm_planes(xn_p, pw.fwd(wcat), M, 2 * ED, C, bias=bcat, y=fv.t, ldy=2 * ED)
    else:
        hip.conv2d(xn_f.t, C, wcat, bcat, fv.t, 2 * ED, B, H, W, C, H, W, 2 * ED, 1, 1, 1, 0, 1, mode=0,
                   precision=rt.prec_fwd(C, C, 2 * ED), w_planes=rt.planes(wcat, 0, 2 * ED, C, M))
    # ---- Cluster core -> proj
    f_t, v_t = fv_t, fv_t[..., ED:]
    o_p = P(ED) if (proj[0] or (rec and proj[2])) else None
    o = f32(ED) if (not proj[0] or (rec and not proj[2])) else None      # the fp32 form only while a consumer still wants it
    idx = rt.buf(B, H, W, E, dtype=torch.uint8)
    big = (H // max(fold, 1)) * (W // max(fold, 1)) > 256
    forced = rt.forced_idx is not None and name is not None and name in rt.forced_idx
    if forced:
        idx.copy_(rt.forced_idx[name])
    wgt = rt.buf(B, H, W, E) if big else None
    nst = hip.cluster_state_floats(B, H, W, E, fold) if (big and rec) else 0
    cstate = rt.buf(nst) if nst else None
    hip.cluster_fwd(f_t, v_t, 2 * ED, tm.sim_alpha, tm.sim_beta, None if o is None else o.t, ED, idx, wgt, B, H, W, E, Dh, fold,
                    forced=forced, planes=o_p, state=cstate)
    if name is not None:
        rt.idx_maps[name] = idx
    x1 = f32(C)
    pr1, per1 = stats_buf(C)
    if proj[0]:
        hip.gemm_planes(o_p, pw.fwd(tm.fc2.weight), M, C, ED, bias=tm.fc2.bias, y=x1.t, ldy=C, res=x.t, ldres=x.ld, res_scale=ls1,
                        stats=pr1, stats_hw=HW)
        if pr1 is not None:
            x1.pairs = (pr1, per1)
    else:
        conv_call(rt, o, tm.fc2, x1, res=x, res_scale=ls1, stats=True)
    # ---- GroupNorm 2 -> Mlp
    # fused Mlp kernels on bf16 tensors (compute_dtype "bf16"): the hidden-sized tensors -- u, and on the way back h and du, the
    # operands of the two weight gradients -- are bf16 in HBM (hip.mlp_fwd / mlp_bwd precision 4) and the weight gradients take
    # them as they are (hip.wgrad_planes, np = 1)
    mlp_hb = bool(pmlp) and np_ == 1 and rt.pg_wgrad and hip.wgrad_planes_ok(M, hid, C) and hip.wgrad_planes_ok(M, C, hid)
    mlp_rc = mlp_hb and rt.mlp_rc(C, hid, True)          # ... and u not stored at all: the backward kernel recomputes it from xn2
    u_b = torch.empty((B, H, W, hid), dtype=torch.bfloat16, device=x.t.device) if (mlp_hb and rec and not mlp_rc) else None
    # two-launch Mlp on bf16 tensors with fc1 forward and fc2 data gradient both on plane GEMMs: u (the GELU' argument) as a
    # bf16 tensor too, as the fused kernels keep it at precision 4
    u_half = np_ == 1 and not pmlp and fc1[0] and fc2[1]
    if not rec:
        u = None
    elif mlp_hb:
        u = True
    elif u_half:
        u = Act(torch.empty((B, H, W, hid), dtype=torch.bfloat16, device=x.t.device), False)
    else:
        u = f32(hid, False)
    x2 = f32(C)
    xn2_p = P(C) if (fc1[0] or (rec and (fc1[2] or mlp_hb))) else None
    xn2_f = f32(C) if (pmlp or not fc1[0] or (rec and not fc1[2])) else None
    ms2 = rt.buf(B, 2)
    if x1.pairs is not None and hip.gn_apply_ok(C, x1.ld):
        hip.gn_apply_fwd(x1.t, x1.ld, x1.pairs[0], x1.pairs[1], m.norm2.weight, m.norm2.bias, m.norm2.eps, B, HW, C,
                         None if xn2_f is None else xn2_f.t, C, ms2, planes=xn2_p)
    else:
        xn2_f, ms2 = gn_forward(rt, x1, m.norm2)
        if xn2_p is not None:
            hip.planes_from_f32(xn2_f.t, C, M, C, xn2_p)
    h_f = h_p = packs = None
    if pmlp:
        packs = rt.mlp_packs(mlp, C, hid, pmlp, rc=mlp_rc)
        pairs, per = stats_buf(C)
        hip.mlp_fwd(xn2_f.t, C, packs[0], mlp.fc1.bias, mlp.fc2.bias, x1.t, x1.ld, ls2, x2.t, C,
                    None if (u is None or mlp_rc) else (u_b if mlp_hb else u.t), hid, pairs, M, C, hid, 4 if mlp_hb else pmlp)
        if pairs is not None:
            x2.pairs = (pairs, per)
    else:
        h_p = P(hid) if (fc2[0] or (rec and fc2[2])) else None
        h_f = f32(hid) if (not fc2[0] or (rec and not fc2[2])) else None
        if fc1[0]:
            hip.gemm_planes(xn2_p, pw.fwd(mlp.fc1.weight), M, hid, C, bias=mlp.fc1.bias, y=None if h_f is None else h_f.t, ldy=hid,
                            yp=h_p, act=2, ypre=None if u is None else u.t, ldypre=hid)
        else:
            conv_call(rt, xn2_f, mlp.fc1, h_f, act=2, ypre=u)
            if h_p is not None:
                hip.planes_from_f32(h_f.t, hid, M, hid, h_p)
        pr2, per2 = stats_buf(C)
        if fc2[0]:
            hip.gemm_planes(h_p, pw.fwd(mlp.fc2.weight), M, C, hid, bias=mlp.fc2.bias, y=x2.t, ldy=C, res=x1.t, ldres=C,
                            res_scale=ls2, stats=pr2, stats_hw=HW)
            if pr2 is not None:
                x2.pairs = (pr2, per2)
        else:
            conv_call(rt, h_f, mlp.fc2, x2, res=x1, res_scale=ls2, stats=True)
    # the gradient this block would like to receive for x2 as planes: the dy operand of fc2's data / weight gradient
    x2.want_gradp = np_ if ((not pmlp and (fc2[1] or fc2[2])) or mlp_hb) else 0
    want_dxp = x.want_gradp == np_          # whoever produced x wants ITS incoming gradient as planes

    def planes_of(t2d, c, have):
        """plane copy of a contiguous fp32 (M, c) tensor when no producer wrote one."""
        if have is not None:
            return have
        pl = P(c)
        hip.planes_from_f32(t2d, c, M, c, pl)
        return pl

    def wgrad(xf, xp, dyf, dyp, conv, cin, cout, use_planes, row_scale=None, ls=None):
        """weight (+ bias, + layer-scale) gradient of a 1x1 conv, off the critical path."""
        gw, accw = rt.pgrad(conv.weight)
        gb, _ = (None, 0) if conv.bias is None else rt.pgrad(conv.bias)
        gl = None
        if ls is not None:
            gl, _ = rt.pgrad(ls)
        if gw is None:
            return
        if use_planes:
            def run():
                hip.wgrad_planes(xp, dyp, M, cin, cout, gw, gb, row_scale, accumulate=accw,
                                 w=conv.weight if gl is not None else None, bias=conv.bias if gl is not None else None, dls=gl)
                if rt.on_param_grad:
                    rt.on_param_grad(conv.weight)
                    if gb is not None:
                        rt.on_param_grad(conv.bias)
                    if gl is not None:
                        rt.on_param_grad(ls)
            rt.aside(run, (xp.t, dyp.t))
        else:
            kwl = dict(w=conv.weight, bias=conv.bias, dls=gl) if gl is not None else {}

            def run():
                hip.conv2d_wgrad(xf, cin, dyf, cout, gw, gb, row_scale, B, H, W, cin, H, W, cout, 1, 1, 1, 0, 1, accumulate=accw,
                                 precision=rt.prec_wgrad(cin, cout, cin, cout), **kwl)
                if rt.on_param_grad:
                    rt.on_param_grad(conv.weight)
                    if gb is not None:
                        rt.on_param_grad(conv.bias)
                    if gl is not None:
                        rt.on_param_grad(ls)
            rt.aside(run, (xf, dyf))

    def bwd():
        dx2 = take_grad(x2)
        dx2_p = take_gradp(x2)
        if dx2 is None:
            return
        if dx2_p is not None and dx2_p.np != np_:
            dx2_p = None
        # ---- MLP branch
        dxn2 = f32(C)
        if mlp_hb:
            hb_p, du_p = P(hid), P(hid)
            if mlp_rc:
                hip.mlp_bwd_rc(dx2, C, ls2, packs[1], xn2_f.t, C, mlp.fc1.bias, hb_p.t[0], hid, du_p.t[0], hid, dxn2.t, C, M, C, hid, 4)
            else:
                hip.mlp_bwd(dx2, C, ls2, packs[1], u_b, hid, hb_p.t[0], hid, du_p.t[0], hid, dxn2.t, C, M, C, hid, 4)
            dx2_p = planes_of(dx2, C, dx2_p)
            wgrad(None, hb_p, dx2, dx2_p, mlp.fc2, hid, C, True, row_scale=ls2, ls=ls2)
            wgrad(None, xn2_p, None, du_p, mlp.fc1, C, hid, True)
        elif pmlp:
            du = f32(hid)
            hb = f32(hid, False)
            hip.mlp_bwd(dx2, C, ls2, packs[1], u.t, hid, hb.t, hid, du.t, hid, dxn2.t, C, M, C, hid, pmlp)
            conv_backward(rt, hb, mlp.fc2, dx2, C, row_scale=ls2, ls_grad=ls2, no_dx=True)
            conv_backward(rt, xn2_f, mlp.fc1, du.t, hid, no_dx=True)
        else:
            if fc2[1] or fc2[2]:
                dx2_p = planes_of(dx2, C, dx2_p)
            du_p = P(hid) if (fc1[1] or fc1[2]) else None
            du_f = f32(hid) if (not fc1[1] or not fc1[2] or not fc2[1]) else None
            if fc2[1]:      # du = (dx2 . (ls2 W2)) * gelu'(u)
                hip.gemm_planes(dx2_p, pw.dgrad(mlp.fc2.weight, ls2), M, hid, C, y=None if du_f is None else du_f.t, ldy=hid,
                                yp=du_p, aux=u.t, ldaux=hid)
            else:
                wd, ks, prec = rt.dgrad_operands(mlp.fc2, mlp.fc2.weight, mlp.fc2.weight, C, hid, 1, 1, ls2, C, M)
                hip.conv2d(dx2, C, wd, None, du_f.t, hid, B, H, W, hid, H, W, C, 1, 1, 1, 0, 1, mode=1, kscale=ks, aux=u.t, ldaux=hid,
                           precision=prec, w_planes=rt.planes(mlp.fc2.weight, 1, hid, C, M, kscale=ks) if prec == 2 else None)
                if du_p is not None:
                    hip.planes_from_f32(du_f.t, hid, M, hid, du_p)
            wgrad(None if h_f is None else h_f.t, h_p, dx2, dx2_p, mlp.fc2, hid, C, fc2[2], row_scale=ls2, ls=ls2)
            if fc1[1]:
                hip.gemm_planes(du_p, pw.dgrad(mlp.fc1.weight), M, C, hid, y=dxn2.t, ldy=C)
            else:
                wd, ks, prec = rt.dgrad_operands(mlp.fc1, mlp.fc1.weight, mlp.fc1.weight, hid, C, 1, 1, None, hid, M)
                hip.conv2d(du_f.t, hid, wd, None, dxn2.t, C, B, H, W, C, H, W, hid, 1, 1, 1, 0, 1, mode=1, precision=prec,
                           w_planes=rt.planes(mlp.fc1.weight, 1, C, hid, M) if prec == 2 else None)
            wgrad(None if xn2_f is None else xn2_f.t, xn2_p, None if du_f is None else du_f.t, du_p, mlp.fc1, C, hid, fc1[2])
        # dx1 = dx2 + d(GN -> Mlp branch), also as planes when proj's gradients take them
        dx1 = rt.buf(B, H, W, C)
        dx1_p = P(C) if (proj[1] or proj[2]) else None
        (gw, gb), accw = _pgrads_or_scratch(rt, (m.norm2.weight, m.norm2.bias), (C, C))
        hip.gn_apply_bwd(dxn2.t, C, x1.t, x1.ld, ms2, m.norm2.weight, B, HW, C, dx1, C, gw, gb, accw, add=dx2, ldadd=C, planes=dx1_p)
        if rt.on_param_grad:
            rt.on_param_grad(m.norm2.weight)
            rt.on_param_grad(m.norm2.bias)
        # ---- Cluster branch
        do = f32(ED)
        do_t = do.t
        if proj[1]:
            hip.gemm_planes(dx1_p, pw.dgrad(tm.fc2.weight, ls1), M, ED, C, y=do.t, ldy=ED)
        else:
            wd, ks, prec = rt.dgrad_operands(tm.fc2, tm.fc2.weight, tm.fc2.weight, C, ED, 1, 1, ls1, C, M)
            hip.conv2d(dx1, C, wd, None, do.t, ED, B, H, W, ED, H, W, C, 1, 1, 1, 0, 1, mode=1, kscale=ks, precision=prec,
                       w_planes=rt.planes(tm.fc2.weight, 1, ED, C, M, kscale=ks) if prec == 2 else None)
        wgrad(None if o is None else o.t, o_p, dx1, dx1_p, tm.fc2, ED, C, proj[2], row_scale=ls1, ls=ls1)
        dfv_p = P(2 * ED) if (fcfv[1] or fcfv[2]) else None
        dfv = f32(2 * ED) if (not fcfv[1] or not fcfv[2]) else None
        (ga, gb_), acca = _pgrads_or_scratch(rt, (tm.sim_alpha, tm.sim_beta), (1, 1))
        ws_ab = hip.cluster_bwd(f_t, v_t, 2 * ED, tm.sim_alpha, tm.sim_beta, idx, do_t, ED, None if dfv is None else dfv.t,
                                None if dfv is None else dfv.t[..., ED:], 2 * ED, None, None, 0, B, H, W, E, Dh, fold, planes=dfv_p,
                                saved=None if cstate is None else (wgt, cstate))
        rt.pending_ab.append((ws_ab, B * E * max(fold, 1) ** 2, ga, gb_, acca, tm))    # finished by rt.flush_cluster_ab()
        if fcfv[2]:
            _fused_qkv_wgrad(rt, tm, xn_p, dfv_p, planes=True)
        else:
            _fused_qkv_wgrad(rt, tm, xn_f, dfv)
        dxn = f32(C)
        if fcfv[1]:
            hip.gemm_planes(dfv_p, pw.dgrad(wcat), M, C, 2 * ED, y=dxn.t, ldy=C)
        else:
            wd, _, prec = rt.dgrad_operands(tm, wcat, wcat, 2 * ED, C, 1, 1, None, 2 * ED, M)
            hip.conv2d(dfv.t, 2 * ED, wd, None, dxn.t, C, B, H, W, C, H, W, 2 * ED, 1, 1, 1, 0, 1, mode=1, precision=prec,
                       w_planes=rt.planes(wcat, 1, C, 2 * ED, M) if prec == 2 else None)
        dx = rt.buf(B, H, W, C)
        dx_p = P(C) if want_dxp else None
        (gw, gb), accw = _pgrads_or_scratch(rt, (m.norm1.weight, m.norm1.bias), (C, C))
        if hip.gn_apply_ok(C, x.ld):
            hip.gn_apply_bwd(dxn.t, C, x.t, x.ld, ms1, m.norm1.weight, B, HW, C, dx, C, gw, gb, accw, add=dx1, ldadd=C, planes=dx_p)
            if rt.on_param_grad:
                rt.on_param_grad(m.norm1.weight)
                rt.on_param_grad(m.norm1.bias)
        else:
            gn_backward(rt, m.norm1, x, ms1, dxn.t, dx, add=dx1)
            dx_p = None
        rt.give_grad(x, dx, planes=dx_p)
    rt.push(bwd)
    return x2


def _fused_qkv_wgrad(rt, tm, xn, dfv, planes=False):
    """(planes: xn and dfv are hip.Planes -- the plane weight gradient, single stream.)
    Weight / bias gradients of fc1 and fc_v as one GEMM into [2ED, C] | [2ED] (per stream), deferred off the critical
    path like every other weight gradient.  The GEMM writes the four parameter gradients IN PLACE when their buffers are
    adjacent -- always without a bucketer (they are handed out as views of one buffer), and with the execution-order
    arena of parallel.GradBucketer, where the four are reported ready back to back; otherwise (the bucketer's recording
    pass) through a scratch matrix and four strided copies."""
    tms = [t for t in _pair(tm) if t is not None]
    ed, c = tms[0].fc1.weight.shape[0], tms[0].fc1.weight.shape[1]
    if planes:
        B, H, W = xn.t.shape[1:4]
    else:
        B, H, W = xn.B, xn.H, xn.W
    plans = []
    for t in tms:
        prms = (t.fc1.weight, t.fc_v.weight, t.fc1.bias, t.fc_v.bias)
        if not any(p.requires_grad for p in prms):
            plans.append(None)
            continue
        fresh = rt.bucketer is None and all(p.requires_grad and p not in rt.pgrads for p in prms)
        if fresh:
            gw, gbias = rt.buf(2 * ed, c), rt.buf(2 * ed)
            rt.pgrads[prms[0]], rt.pgrads[prms[1]] = gw[:ed].view_as(prms[0]), gw[ed:].view_as(prms[1])
            rt.pgrads[prms[2]], rt.pgrads[prms[3]] = gbias[:ed], gbias[ed:]
            plans.append((gw, gbias, 0, None))
            continue
        got = [rt.pgrad(p) for p in prms]
        (g0, a0), (g1, a1), (g2, a2), (g3, a3) = got
        adjacent = all(g is not None for g, _ in got) and a0 == a1 == a2 == a3 and \
            g1.data_ptr() == g0.data_ptr() + 4 * g0.numel() and g3.data_ptr() == g2.data_ptr() + 4 * g2.numel()
        if adjacent:
            gw = torch.as_strided(g0, (2 * ed, c), (c, 1))
            gbias = torch.as_strided(g2, (2 * ed,), (1,))
            plans.append((gw, gbias, a0, None))
        else:
            plans.append((rt.buf(2 * ed, c), rt.buf(2 * ed), 0, [(p, g, a) for p, (g, a) in zip(prms, got)]))
    if all(pl is None for pl in plans):
        return
    assert all(pl is not None for pl in plans) and len({pl[2] for pl in plans}) == 1, \
        "the two streams of a stage must be frozen / trained together"

    def wgrad():
        kw2 = dict(dw2=plans[1][0], dbias2=plans[1][1]) if len(plans) == 2 else {}
        if planes:
            hip.wgrad_planes(xn, dfv, B * H * W, c, 2 * ed, plans[0][0], plans[0][1], None, accumulate=plans[0][2])
        else:
            hip.conv2d_wgrad(xn.t, xn.ld, dfv.t, 2 * ed, plans[0][0], plans[0][1], None, B, H, W, c, H, W, 2 * ed, 1, 1, 1, 0, 1,
                             accumulate=plans[0][2], precision=rt.prec_wgrad(xn.ld, 2 * ed, c, 2 * ed), **kw2)
        for t, (gw, gbias, _, scatter) in zip(tms, plans):
            if scatter is not None:
                for (prm, g, acc), (src, width) in zip(scatter, ((gw, c), (gw[ed:], c), (gbias, 1), (gbias[ed:], 1))):
                    if g is not None:
                        hip.copy_channels(src, width, 1, g, width, 1, ed, width, accumulate=acc)
            if rt.on_param_grad:
                for prm in (t.fc1.weight, t.fc_v.weight, t.fc1.bias, t.fc_v.bias):     # back to back: adjacent in the arena
                    rt.on_param_grad(prm)
    rt.aside(wgrad, (xn.t, dfv.t))


# ----------------------------------------------------------------------------------------- fusion blocks
def _fusion_chunks(rt, *acts):
    """Chunk count of the fused fusion-block kernels (csrc/fusion.hip) for contiguous maps of one shape, 0: not applicable."""
    a = acts[0]
    if not rt.fused_fusion or rt.sync_bn is not None or any(t.ld != t.C or t.C != a.C or t.rows != a.rows for t in acts):
        return 0
    return hip.fusion_chunks(a.rows * a.C, a.C)


def image_enhance(rt, x, r, m, out=None):
    """ImageEnhanceByRadar.forward (vr_coc.py:312-316): BN((1 + minmax(ReLU(BN(conv3x3(r))))) * x).
    out: Act to write the result into (one stream's half of the next two-stream buffer).
    Round 5 (csrc/fusion.hip): the BatchNorm + ReLU apply leaves the (min, max) partials, the gain kernel the column statistics
    of its output, the backward of `norm` the four sums of the gain's backward, and the gain's backward the moments of bn1's --
    6 + 7 launches where the unfused form (kept for maps the fused kernels do not take: 3 channels at the input level) has 8 + 9."""
    conv, bn1 = m.radar_projection.conv, m.radar_projection.bn
    B, H, W, C = x.B, x.H, x.W, x.C
    z = rt.new(B, H, W, C)
    conv_call(rt, r, conv, z, bias=False, bn_stats=True)
    assert x.ld == C, "image map must be contiguous"
    n = x.rows * C
    nch = _fusion_chunks(rt, x, z)
    mm = rt.buf(2)
    t = rt.new(B, H, W, C)
    if nch:
        A, D, S, ms1 = bn_fwd_coef(rt, z, bn1)
        p = rt.new(B, H, W, C)
        nfold = hip.fusion_fold_chunks(n, C)
        mmpart = rt.buf(nfold, 2)
        hip.bn_relu_minmax(z.t, A, D, S, p.t, n, C, mmpart)
        ms1.fwd_coef = (A, D, S)
        if rt.relu_masks is not None:
            rt.relu_masks[bn1] = p
        colpart = rt.buf(nch, C, 2, dtype=torch.float64)
        hip.enhance_stats(p.t, x.t, mmpart, nfold, mm, t.t, n, C, colpart)
        if rt.training:
            t.colpart, t.colchunks = colpart, nch          # BatchNorm `norm`: coefficients straight from these
    else:
        p, ms1 = bn_forward(rt, z, bn1, relu=True)
        hip.enhance_fwd(p.t, x.t, mm, t.t, n)           # min / max of p and the gain in two launches (three before round 5)
    y, ms2 = bn_forward(rt, t, m.norm, relu=False, out=out)

    def bwd():
        g = take_grad(y)
        if g is None:
            return
        if x.need_grad:
            dxb, acc = rt.grad_target(x)
        else:
            dxb, acc = rt.buf(B, H, W, C), 0
        dp = rt.buf(B, H, W, C)
        fwd = getattr(ms1, "fwd_coef", None)
        if nch and BN_ZMASK and fwd is not None and g.is_contiguous():
            A2, E2, D2, S2 = bn_bwd_coef(rt, m.norm, t, ms2, g, C)
            dt = rt.buf(B, H, W, C)
            nf = hip.fusion_fold_chunks(n, C)
            sums4 = rt.buf(nf, 4, dtype=torch.float64)
            hip.bn_bwd_enhance(g, t.t, A2, E2, D2, S2, x.t, p.t, mm, dt, n, C, sums4)
            colpart = rt.buf(nch, C, 2, dtype=torch.float64)
            hip.enhance_bwd_stats(dt, x.t, p.t, mm, sums4, nf, z.t, fwd[0], fwd[1], fwd[2], dxb, dp, n, C, acc, colpart)
            (gw, gb), accw = _pgrads_or_scratch(rt, (bn1.weight, bn1.bias), (C, C))
            A1, E1, D1, S1 = rt.buf(C), rt.buf(C), rt.buf(C), rt.buf(C)
            hip.bn_coef_bwd_from_chunks(colpart, nch, x.rows, ms1, bn1.weight, rt.training, C, A1, E1, D1, S1, gw, gb, accw)
            if rt.on_param_grad:
                rt.on_param_grad(bn1.weight)
                rt.on_param_grad(bn1.bias)
            dz = rt.buf(B, H, W, C)
            hip.bn_apply_bwd_zmask(dp, C, z.t, z.ld, fwd, A1, E1, D1, S1, dz, C, B, H * W, C)
        else:
            dt = bn_backward(rt, m.norm, t, ms2, g, C)
            hip.enhance_bwd(dt, x.t, p.t, mm, dxb, dp, n, accumulate_dx=acc)
            dz = bn_backward(rt, bn1, z, ms1, dp, C, mask=p)
        conv_backward(rt, r, conv, dz, C)
    rt.push(bwd)
    return y


def shuffle_attention(rt, x, m, defer_apply=False):
    """ShuffleAttention.forward (shuffle_attention.py:48-72) on a contiguous map.
    defer_apply: only the gate coefficients are computed; the returned Act has NO tensor (a shape and a gradient slot) and
    carries the coefficients as `.sa = (P, Q, Mn)` -- the caller's fused kernel (hip.sa_cat_sums) applies the gate while it
    writes the consumer's tensor.  The backward closure is the same either way (it never reads the output)."""
    B, HW, C, G = x.B, x.HW, x.C, m.G
    params = [t.reshape(-1) for t in (m.cweight, m.cbias, m.sweight, m.sbias, m.gn.weight, m.gn.bias)]
    mom = hip.moments(x.t, x.ld, B, HW, C)
    P, Q, Mn = rt.buf(B, C), rt.buf(B, C), rt.buf(B, C)
    hip.sa_coef_fwd(mom, *params, B, HW, C, G, P, Q, Mn)
    if defer_apply:
        y = Act(torch.empty((x.B, x.H, x.W, C), device="meta"))
        y.sa = (P, Q, Mn)
    else:
        y = rt.new(x.B, x.H, x.W, C)
        hip.sa_apply(x.t, x.ld, P, Q, Mn, y.t, C, B, HW, C)

    def bwd():
        g = take_grad(y)
        if g is None:
            return
        plist = (m.cweight, m.cbias, m.sweight, m.sbias, m.gn.weight, m.gn.bias)
        grads, acc = [], 0
        for prm in plist:
            gp, a = rt.pgrad(prm)
            if gp is None:
                gp, a = rt.buf(prm.numel()), 0
            grads.append(gp)
            acc = a
        if x.need_grad:
            dxb, accx = rt.grad_target(x)
        else:
            dxb, accx = rt.buf(x.B, x.H, x.W, C), 0
        hip.sa_bwd(g, C, x.t, x.ld, P, Q, Mn, mom, params, dxb, C, grads, rt.buf(2, B, C), B, HW, C, G, accx, acc)
        if rt.on_param_grad:
            for prm in plist:
                rt.on_param_grad(prm)
    rt.push(bwd)
    return y


def cat2(rt, a, b, interleave, written=None):
    """torch.cat([a, b], 1) [+ 2-group channel shuffle when both halves have equal width]
    (vr_coc.py:70-80, coc_fpn_dual.py:120-130): one launch into one buffer, one for the adjoint.
    written: the concatenated Act, already filled by a fused kernel (only the adjoint is recorded)."""
    B, H, W = a.B, a.H, a.W
    Ct = a.C + b.C
    out = written if written is not None else rt.new(B, H, W, Ct)
    rows = a.rows
    il = bool(interleave and Ct % 2 == 0)
    if il:
        assert a.C == b.C
    if written is None:
        hip.cat2(a.t, a.ld, a.C, b.t, b.ld, b.C, out.t, Ct, rows, il)           # one launch (two strided copies before)

    def bwd():
        g = take_grad(out)
        if g is None:
            return
        ba, acca = rt.grad_target(a) if a.need_grad else (None, 0)
        bb, accb = rt.grad_target(b) if b.need_grad else (None, 0)
        if ba is not None or bb is not None:
            hip.cat2(ba, a.C, a.C, bb, b.C, b.C, g, Ct, rows, il, dir=1, accumulate_a=acca, accumulate_b=accb)
    rt.push(bwd)
    return out


def eca(rt, x, m, mom=None):
    """eca_block.forward (eca.py:16-22).  mom: the (B, C, 2) channel sums of x when a fused producer already has them."""
    B, HW, C = x.B, x.HW, x.C
    k = m.kernel_size
    wk = m.conv.weight.reshape(-1)
    if mom is None:
        mom = hip.moments(x.t, x.ld, B, HW, C)
    gate = rt.buf(B, C)
    hip.eca_coef_fwd(mom, wk, k, B, HW, C, gate)
    y = rt.new(x.B, x.H, x.W, C)
    hip.affine(y.t, C, B, HW, C, x1=x.t, ld1=x.ld, A=gate, bstride=C)

    def bwd():
        g = take_grad(y)
        if g is None:
            return
        mom2 = hip.moments(g, C, B, HW, C, x2=x.t, ldx2=x.ld)
        Fc = rt.buf(B, C)
        gw, acc = rt.pgrad(m.conv.weight)
        if gw is None:
            gw, acc = rt.buf(k), 0
        hip.eca_coef_bwd(mom2, mom, gate, wk, k, B, HW, C, Fc, None, 0)

        def kernel_weight_grad():        # (13 us that no later backward kernel waits for: off the chain, like a weight gradient)
            hip.eca_coef_bwd(mom2, mom, gate, wk, k, B, HW, C, None, gw, acc)
            if rt.on_param_grad:
                rt.on_param_grad(m.conv.weight)
        rt.aside(kernel_weight_grad, (mom2, mom, gate))
        if x.need_grad:
            buf, accx = rt.grad_target(x)
            hip.affine(buf, C, B, HW, C, x1=g, ld1=C, A=gate, D2=Fc, bstride=C, accumulate=accx)
    rt.push(bwd)
    return y


def radar_enhance(rt, x, r, m, out=None):
    """RadarEnhanceByImage.forward (vr_coc.py:331-359).  out: as in image_enhance."""
    C0 = x.C
    if not m.initial and rt.fused_fusion and C0 == r.C and C0 % 4 == 0 and C0 <= 512 and r.ld % 2 == 0 and r.t.data_ptr() % 8 == 0:
        # attention apply + concat + channel shuffle + the ECA gate's channel sums in ONE launch (csrc/stream_ops.hip,
        # sa_cat_sums_kernel): the attention's output is never stored, the 2 C-wide tensor is written once and not re-read
        a = shuffle_attention(rt, x, m.image_attn, defer_apply=True)
        cat = rt.new(x.B, x.H, x.W, 2 * C0)
        mom_cat = hip.sa_cat_sums(x.t, x.ld, a.sa[0], a.sa[1], a.sa[2], r.t, r.ld, cat.t, 2 * C0, x.B, x.HW, C0)
        u = eca(rt, cat2(rt, a, r, interleave=True, written=cat), m.channel_attn, mom=mom_cat)
    else:
        a = x if m.initial else shuffle_attention(rt, x, m.image_attn)
        u = eca(rt, cat2(rt, a, r, interleave=True), m.channel_attn)
    conv, bn1 = m.inverse_projection.conv, m.inverse_projection.bn
    B, H, W, C = r.B, r.H, r.W, r.C
    z = rt.new(B, H, W, C)
    conv_call(rt, u, conv, z, bias=False, bn_stats=True)
    nch = _fusion_chunks(rt, r, z)
    if BN_ZMASK and rt.relu_masks is None and rt.sync_bn is None:      # (synchronised BatchNorm reads the ReLU output as its mask)
        # s = ReLU(BN(z)) + r in ONE apply launch; the ReLU output itself is never stored: the backward pass recomputes its
        # mask from z with the forward coefficients (bn_backward, zmask form).  (round 5: -1 launch, -2 tensor passes per level)
        if nch and rt.training:
            # ... which also leaves the column statistics of s: BatchNorm `norm` needs no pass of its own over it
            A1, D1, S1, ms1 = bn_fwd_coef(rt, z, bn1)
            s = rt.new(B, H, W, C)
            s.colpart, s.colchunks = rt.buf(nch, C, 2, dtype=torch.float64), nch
            hip.bn_relu_res_stats(z.t, A1, D1, S1, r.t, s.t, r.rows * C, C, s.colpart)
            ms1.fwd_coef = (A1, D1, S1)
        else:
            s, ms1 = bn_forward(rt, z, bn1, relu=True, residual=r)
        q = s                        # (stands in for "there is a ReLU": the zmask backward never reads it)
    else:
        q, ms1 = bn_forward(rt, z, bn1, relu=True)
        s = rt.new(B, H, W, C)
        hip.affine(s.t, C, B, H * W, C, x1=q.t, ld1=C, x2=r.t, ld2=r.ld)
    y, ms2 = bn_forward(rt, s, m.norm, relu=False, out=out)

    def bwd():
        g = take_grad(y)
        if g is None:
            return
        fwd = getattr(ms1, "fwd_coef", None)
        if nch and BN_ZMASK and fwd is not None and q is s and g.is_contiguous():
            # the backward apply of `norm` also leaves the moments of bn1's backward (csrc/fusion.hip): one pass over ds less
            A2, E2, D2, S2 = bn_bwd_coef(rt, m.norm, s, ms2, g, C)
            ds = rt.buf(B, H, W, C)
            colpart = rt.buf(nch, C, 2, dtype=torch.float64)
            hip.bn_bwd_next_stats(g, s.t, A2, E2, D2, S2, z.t, fwd, ds, r.rows * C, C, colpart)
            (gw, gb), accw = _pgrads_or_scratch(rt, (bn1.weight, bn1.bias), (C, C))
            A1, E1, D1, S1 = rt.buf(C), rt.buf(C), rt.buf(C), rt.buf(C)
            hip.bn_coef_bwd_from_chunks(colpart, nch, r.rows, ms1, bn1.weight, rt.training, C, A1, E1, D1, S1, gw, gb, accw)
            if rt.on_param_grad:
                rt.on_param_grad(bn1.weight)
                rt.on_param_grad(bn1.bias)
            dz = rt.buf(B, H, W, C)
            hip.bn_apply_bwd_zmask(ds, C, z.t, z.ld, fwd, A1, E1, D1, S1, dz, C, B, H * W, C)
        else:
            ds = bn_backward(rt, m.norm, s, ms2, g, C)
            dz = bn_backward(rt, bn1, z, ms1, ds, C, mask=q)
        rt.give_grad(r, ds)                                       # long residual path (+ radar_map)
        conv_backward(rt, u, conv, dz, C)
    rt.push(bwd)
    return y


# ----------------------------------------------------------------------------------------- neck pieces
def coc_upsample(rt, x, m, nchw_out=None):
    """CoCUpsample.forward (coc_fpn_dual.py:24-26): BaseConv 1x1 -> bilinear, align_corners=True."""
    bc = m.upsample[0]
    s = m.scale
    if rt.fused_upsample and not bc.ds_conv and BN_ZMASK and rt.relu_masks is None and rt.sync_bn is None:
        # (round 5, K11) conv -> [BatchNorm + ReLU on the taps of the bilinear gather]: the low-resolution activation is never
        # stored (the backward needs z and the forward coefficients only: the ReLU mask is recomputed from them)
        conv, bn = bc.conv, bc.bn
        co, ci, kh, kw, st, pd, dl, OH, OW = conv_geom(x, conv)
        z = rt.new(x.B, OH, OW, co)
        conv_call(rt, x, conv, z, bias=False, bn_stats=True)
        cA, cD, cS, ms = bn_fwd_coef(rt, z, bn)
        ms.fwd_coef = (cA, cD, cS)
        B, H, W, C = z.B, z.H, z.W, co
        to_nchw = nchw_out is not None      # (the closure below must not capture the output tensor itself: output -> autograd
        hi = None if to_nchw else rt.new(B, H * s, W * s, C)      #  node -> tape -> closure -> output would be a cycle)
        if to_nchw:
            hip.bn_relu_upsample(z.t, C, cA, cD, cS, nchw_out, 0, B, H, W, C, s, out_nchw=1)
        else:
            hip.bn_relu_upsample(z.t, C, cA, cD, cS, hi.t, C, B, H, W, C, s)

        def bwd_fused():
            g = rt.seg_grad if to_nchw else take_grad(hi)
            if g is None:
                return
            dlo = rt.buf(B, H, W, C)
            if to_nchw:
                hip.upsample_bwd(g.contiguous(), 0, 1, dlo, C, B, H, W, C, s)
            else:
                hip.upsample_bwd(g, C, 0, dlo, C, B, H, W, C, s)
            dz = bn_backward(rt, bn, z, ms, dlo, C, mask=z)      # (mask: only "there is a ReLU" on this path)
            conv_backward(rt, x, conv, dz, co)
        rt.push(bwd_fused)
        return z if to_nchw else hi
    lo = base_conv(rt, x, bc)
    B, H, W, C = lo.B, lo.H, lo.W, lo.C
    if nchw_out is not None:
        hip.upsample(lo.t, C, nchw_out, 0, B, H, W, C, s, out_nchw=1)

        def bwd_nchw():                                            # gradient of the NCHW seg logits (rt.seg_grad)
            g = rt.seg_grad
            if g is None:
                return
            buf, acc = rt.grad_target(lo)
            hip.upsample_bwd(g.contiguous(), 0, 1, buf, C, B, H, W, C, s, accumulate=acc)
        rt.push(bwd_nchw)
        return lo
    hi = rt.new(B, H * s, W * s, C)
    hip.upsample(lo.t, C, hi.t, C, B, H, W, C, s)

    def bwd():
        g = take_grad(hi)
        if g is None:
            return
        buf, acc = rt.grad_target(lo)
        hip.upsample_bwd(g, C, 0, buf, C, B, H, W, C, s, accumulate=acc)
    rt.push(bwd)
    return hi


def coc_conv(rt, x, m, name=None):
    """CoC_Conv.forward (coc_fpn_dual.py:36-39)."""
    return base_conv(rt, cluster_block(rt, x, m.coc, name), m.conv_att)


def aspp(rt, x, m):
    """ASPP.forward (coc_fpn_dual.py:79-104); the five branches write channel slices of one buffer."""
    B, H, W, C = x.B, x.H, x.W, x.C
    HW = H * W
    cat = rt.new(B, H, W, 5 * C)

    def conv_branch(k, br):
        conv, bn = br[0], br[1]
        z = rt.new(B, H, W, C)
        conv_call(rt, x, conv, z, bn_stats=True)
        sl = Act(cat.t[..., k * C:(k + 1) * C])
        _, ms = bn_forward(rt, z, bn, relu=True, out=sl)
        return (conv, bn, z, sl, ms)

    def pool_branch():      # global-average branch: (B,1,1,C) "pixels"
        gm = rt.new(B, 1, 1, C)
        hip.moments_to_float(hip.moments(x.t, x.ld, B, HW, C), gm.t, B * C, 1.0 / HW)
        z5 = rt.new(B, 1, 1, C)
        conv_call(rt, gm, m.branch5_conv, z5)
        q5, ms5 = bn_forward(rt, z5, m.branch5_bn, relu=True)
        hip.affine(cat.t[..., 4 * C:], 5 * C, B, HW, C, D2=q5.t, bstride=C)   # bilinear from 1x1, align_corners: constant
        return gm, z5, q5, ms5
    record, rt.record = rt.record, False          # one hand-written closure below covers all five branches
    res = rt.parallel([(lambda k=k, br=br: conv_branch(k, br))
                       for k, br in enumerate((m.branch1, m.branch2, m.branch3, m.branch4))] + [pool_branch], site=6)
    rt.record = record
    saved, (gm, z5, q5, ms5) = res[:4], res[4]
    convc, bnc = m.conv_cat[0], m.conv_cat[1]
    zc = rt.new(B, H, W, C)
    conv_call(rt, cat, convc, zc, bn_stats=True)
    y, msc = bn_forward(rt, zc, bnc, relu=True)

    def bwd():
        g = take_grad(y)
        if g is None:
            return
        dzc = bn_backward(rt, bnc, zc, msc, g, C, mask=y)
        conv_backward(rt, cat, convc, dzc, C)
        dcat = take_grad(cat)
        for k, (conv, bn, z, sl, ms) in enumerate(saved):
            dz = bn_backward(rt, bn, z, ms, dcat[..., k * C:], 5 * C, mask=sl)
            conv_backward(rt, x, conv, dz, C)
        dq5 = rt.buf(B, 1, 1, C)
        hip.moments_to_float(hip.moments(dcat[..., 4 * C:], 5 * C, B, HW, C), dq5, B * C, 1.0)
        dz5 = bn_backward(rt, m.branch5_bn, z5, ms5, dq5, C, mask=q5)
        conv_backward(rt, gm, m.branch5_conv, dz5, C)
        dgm = take_grad(gm)
        if x.need_grad:
            buf, acc = rt.grad_target(x)
            scaled = rt.buf(B, C)
            hip.affine(scaled, C, B, 1, C, x1=dgm, ld1=C, A=rt.const(C, 1.0 / HW))
            hip.affine(buf, C, B, HW, C, D2=scaled, bstride=C, accumulate=acc)
    rt.push(bwd)
    return y


# ----------------------------------------------------------------------------------------- assembly
def backbone_forward(rt, bb, x, r):
    """VRCoC.forward_embeddings + forward_tokens (vr_coc.py:575-675)."""
    B, H, W = x.B, x.H, x.W
    x0, r0 = x, r
    x, r = rt.parallel([lambda: simple_conv(rt, x0, bb.image_initial.proj),
                        lambda: simple_conv(rt, r0, bb.radar_initial.proj)], site=0)
    x = image_enhance(rt, x, r, bb.image_enhance_by_radar1)
    overlapped = rt.concurrent and not rt.pair_streams and rt.overlap_fusion
    if not overlapped:
        r = radar_enhance(rt, x, r, bb.radar_enhance_by_image1)
    if tuple(bb.fea_pos.shape[:2]) != (H, W):
        raise RuntimeError(f"input {H}x{W} does not match fea_pos {tuple(bb.fea_pos.shape[:2])}: "
                           "construct EfficientVRNet(..., img_size=(H, W))")
    def embed(act, pe, out=None):
        """cat([x, fea_pos]) -> 4x4/s4 PointRecuder (vr_coc.py:583-586, 99-102) as patch gather + one plain GEMM:
        the concat is never materialised and the projection runs on the vector path with K = 16*(C+2)."""
        conv = pe.proj
        co, ci, kh, kw = conv.weight.shape
        C, CP, k = act.C, ci - act.C, kh
        assert kh == kw == conv.stride[0] and conv.padding[0] == 0 and CP == 2 and H % k == 0 and W % k == 0
        OH, OW, KT = H // k, W // k, k * k * ci
        patches = rt.new(B, OH, OW, KT)
        hip.patch_gather(act.t, act.ld, bb.fea_pos, patches.t, B, H, W, C, CP, k)   # :585 uses fea_pos for both streams
        w2 = rt.buf(co, KT)                                       # OHWI: [n][(ky,kx)][c]
        hip.weight_ohwi(conv.weight, w2, co, ci, kh, kw, 0)
        y = out if out is not None else rt.new(B, OH, OW, co)
        hip.conv2d(patches.t, KT, w2, conv.bias, y.t, y.ld, B, OH, OW, KT, OH, OW, co, 1, 1, 1, 0, 1, mode=0,
                   precision=rt.prec_fwd(KT, KT, co))

        def bwd(act=act, patches=patches, w2=w2, y=y):
            g = take_grad(y)
            if g is None:
                return
            gw, accw = rt.pgrad(conv.weight)
            gb, accb = rt.pgrad(conv.bias)
            if gw is not None:
                assert gb is None or accb == accw

                def embed_wgrad():       # (round 5: off the chain -- a 131 072-row weight gradient sat in front of the data
                    gw2 = rt.buf(co, KT)  #  gradient on the tail of the backward pass)
                    if gb is not None and accb:       # one accumulate flag covers dw and dbias: start the scratch dw at 0
                        hip.fill_(gw2, 0.0)
                    hip.conv2d_wgrad(patches.t, KT, g, co, gw2, gb, None, B, OH, OW, KT, OH, OW, co, 1, 1, 1, 0, 1,
                                     accumulate=0 if gb is None else accb, precision=rt.prec_wgrad(KT, co, KT, co))
                    hip.weight_ohwi(gw2, gw, co, ci, kh, kw, 1, accumulate=accw)
                    if rt.on_param_grad:
                        rt.on_param_grad(conv.weight)
                        if gb is not None:
                            rt.on_param_grad(conv.bias)
                rt.aside(embed_wgrad, (patches.t, g))
            if act.need_grad:
                dp = rt.buf(B, OH, OW, KT)
                wd, _, prec = rt.dgrad_operands(conv, w2, w2, co, KT, 1, 1, None, co)
                hip.conv2d(g, co, wd, None, dp, KT, B, OH, OW, KT, OH, OW, co, 1, 1, 1, 0, 1, mode=1, precision=prec)
                buf, acc = rt.grad_target(act)
                hip.patch_scatter(dp, buf, act.C, B, H, W, C, CP, k, accumulate=acc)
        rt.push(bwd)
        return y
    xe, re_ = x, r
    dims = [bb.network[3 * i][0].norm1.weight.shape[0] for i in range(4)]
    # Two-stream mode (default): the image and the radar chain of every stage run as ONE batch of 2B samples in one
    # (2B,H,W,C) buffer, image samples first -- one launch per layer with twice the tiles and per-half parameters
    # instead of two half-filled launches on two streams (vr_coc.py:589-600 calls network[idx] / network_radar[idx]
    # back to back on equal shapes).  model.pair_streams = False keeps the two chains on two forked streams.
    outs, outs_r = [], []

    def chain(act, blocks, prefix):
        for j, blk in enumerate(blocks):
            act = cluster_block(rt, act, blk, f"{prefix}.{j}.token_mixer")
        return act

    if overlapped:
        return _backbone_overlapped(rt, bb, x, r, embed, chain)
    mask = os.environ.get("VRNET_PAIR_MASK")      # diagnostic: which stages (bits 0-3) / reducers (bits 4-6) run two-stream

    def can_pair(h, w, bit=None):    # rows of one stream must be whole 128-row tiles (true for every stage from 256 px at bs 2)
        on = rt.pair_streams if (mask is None or bit is None) else bool(int(mask, 0) >> bit & 1)
        return on and (B * h * w) % 128 == 0
    xr = rt.new_pair(2 * B, H // 4, W // 4, dims[0])             # stage-0 input: both patch embeddings
    xh, rh = xr.halves()
    rt.parallel([lambda: embed(xe, bb.patch_embed, out=xh), lambda: embed(re_, bb.patch_embed_radar, out=rh)], site=1)
    for i in range(4):
        pi, pr = f"backbone.backbone.network.{3 * i}", f"backbone.backbone.network_radar.{3 * i}"
        if can_pair(xr.H, xr.W, i):
            for j, (bi, br) in enumerate(zip(bb.network[3 * i], bb.network_radar[3 * i])):
                xr = cluster_block(rt, xr, (bi, br), (f"{pi}.{j}.token_mixer", f"{pr}.{j}.token_mixer"))
            xs, rs = xr.halves()
        else:                                                    # tiny maps (test sizes): two chains on two forked streams
            xi, ri = xr.halves()
            xs, rs = rt.parallel([lambda: chain(xi, bb.network[3 * i], pi), lambda: chain(ri, bb.network_radar[3 * i], pr)], site=2)
        if i < 3:                                                # fused maps go into the reducer's two-stream input
            fused = rt.new_pair(2 * B, xs.H, xs.W, xs.C)
            fx, fr = fused.halves()
        else:
            fx = fr = None
        x = image_enhance(rt, xs, rs, bb.network[3 * i + 1], out=fx)
        r = radar_enhance(rt, x, rs, bb.network_radar[3 * i + 1], out=fr)
        if i in (0, 3):
            outs.append(x)
            outs_r.append(r)
        if i < 3:
            ci, cr = bb.network[3 * i + 2].proj, bb.network_radar[3 * i + 2].proj
            if can_pair(fused.H // 2, fused.W // 2, 4 + i):
                xr = simple_conv(rt, fused, (ci, cr))
            else:
                xr = rt.new_pair(2 * B, fused.H // 2, fused.W // 2, dims[i + 1])
                ox, orr = xr.halves()
                rt.parallel([lambda: simple_conv(rt, fx, ci, out=ox), lambda: simple_conv(rt, fr, cr, out=orr)], site=3)
            if i < 2:
                xs, rs = xr.halves()
                outs.append(xs)
                outs_r.append(rs)
    return outs, outs_r


def branch_alias(rt, x):
    """A second handle on x's tensor for a consumer that runs on ANOTHER chain of the next parallel section: it gets a
    gradient buffer of its own (two chains must not accumulate into one buffer concurrently), which a closure on the MAIN tape
    -- pushed here, so replayed right after the section's backward -- adds to x's."""
    xa = Act(x.t, need_grad=x.need_grad)

    def bwd():
        g = take_grad(xa)
        if g is not None:
            rt.give_grad(x, g)
    rt.push(bwd)
    return xa


def _backbone_overlapped(rt, bb, x, r, embed, chain):
    """forward_tokens (vr_coc.py:589-675) with the asymmetric fusion un-serialised (round 5).  The reference runs, per level,
    stage blocks (both streams) -> ImageEnhanceByRadar -> RadarEnhanceByImage -> both reducers -> next stage.  Only the radar
    stream needs RadarEnhanceByImage: the image stream's reducer and next stage depend on the ImageEnhance output alone.  So a
    section is  A: image reducer (or patch embedding) -> image blocks   beside   B: RadarEnhanceByImage of the previous level ->
    radar reducer -> radar blocks,  joined in front of the next ImageEnhanceByRadar (which needs both).  The ~16 small launches
    of a RadarEnhanceByImage (ShuffleAttention, concat, ECA, 1x1 conv, two BatchNorms -- one stream's worth of work on a chip that
    is otherwise idle) and their ~25 backward launches then run beside the image chain's GEMMs instead of in front of both
    chains.  Same arithmetic; the gradient of an ImageEnhance output is summed from two buffers (branch_alias) instead of
    accumulated in one."""
    outs, outs_r = [None] * 4, [None] * 4
    re_mod, r_in = bb.radar_enhance_by_image1, r
    for i in range(4):
        pi, pr = f"backbone.backbone.network.{3 * i}", f"backbone.backbone.network_radar.{3 * i}"
        x_prev, xa = x, branch_alias(rt, x)

        def branch_a(i=i, x_prev=x_prev, pi=pi):
            rt.stamp(f"s{i} A start")
            a = embed(x_prev, bb.patch_embed) if i == 0 else simple_conv(rt, x_prev, bb.network[3 * (i - 1) + 2].proj)
            out = a, chain(a, bb.network[3 * i], pi)
            rt.stamp(f"s{i} A end")
            return out

        def branch_b(i=i, xa=xa, r_in=r_in, re_mod=re_mod, pr=pr):
            rt.stamp(f"s{i} B start")
            rr = radar_enhance(rt, xa, r_in, re_mod)
            rt.stamp(f"s{i} B RE done")
            b = embed(rr, bb.patch_embed_radar) if i == 0 else simple_conv(rt, rr, bb.network_radar[3 * (i - 1) + 2].proj)
            out = rr, b, chain(b, bb.network_radar[3 * i], pr)
            rt.stamp(f"s{i} B end")
            return out
        rt.stamp(f"s{i} fork")
        (tap_a, xs), (rr, tap_b, rs) = rt.parallel([branch_a, branch_b], site=2)
        if i == 1:
            outs_r[0] = rr                                       # RadarEnhance output at 1/4 resolution
        if i in (1, 2):
            outs[i], outs_r[i] = tap_a, tap_b                    # the reducers' outputs (inputs of stages 1, 2)
        rt.stamp(f"s{i} join")
        x = image_enhance(rt, xs, rs, bb.network[3 * i + 1])
        if i in (0, 3):
            outs[0 if i == 0 else 3] = x
        re_mod, r_in = bb.network_radar[3 * i + 1], rs
    # the last RadarEnhanceByImage only feeds the detection branch of the neck: it runs at the head of that branch
    # (neck_forward), beside the segmentation branch's ASPP, not in front of both
    outs_r[3] = (branch_alias(rt, x), r_in, re_mod)
    return outs, outs_r


def neck_forward(rt, nk, x, r, seg_out):
    """CoCFpnDual.forward (coc_fpn_dual.py:184-224). seg_out: NCHW tensor for the seg logits."""
    (x2, x3, x4, x5), (r2, r3, r4, r5) = backbone_forward(rt, nk.backbone, x, r)

    def seg_branch():        # image-stream features only (coc_fpn_dual.py:193-209)
        a5 = aspp(rt, x5, nk.aspp)
        t = shuffle_attention(rt, cat2(rt, x4, coc_upsample(rt, a5, nk.upsample5_4), True), nk.sc_attn_seg4)
        t = shuffle_attention(rt, cat2(rt, coc_upsample(rt, t, nk.upsample4_3), x3, True), nk.sc_attn_seg3)
        t = shuffle_attention(rt, cat2(rt, coc_upsample(rt, t, nk.upsample3_2), x2, True), nk.sc_attn_seg2)
        return coc_upsample(rt, t, nk.upsample2_0, nchw_out=seg_out)

    def det_branch():        # radar-stream features only (coc_fpn_dual.py:213-221)
        r5_ = radar_enhance(rt, *r5) if isinstance(r5, tuple) else r5      # (deferred by _backbone_overlapped)
        p5 = coc_conv(rt, r5_, nk.p5_out_det, "backbone.p5_out_det.coc.token_mixer")
        p4 = coc_conv(rt, cat2(rt, r4, coc_upsample(rt, p5, nk.p5_4_det), False), nk.p4_out_det,
                      "backbone.p4_out_det.coc.token_mixer")
        p3 = coc_conv(rt, cat2(rt, r3, coc_upsample(rt, p4, nk.p4_3_det), False), nk.p3_out_det,
                      "backbone.p3_out_det.coc.token_mixer")
        return (p3, p4, p5)
    seg_lo, feats = rt.parallel([seg_branch, det_branch], site=4)
    return feats, seg_lo


def head_forward(rt, hd, feats, det_outs):
    """DecoupleHead.forward (decouplehead.py:42-88): the three prediction convs store straight into the
    channel ranges [reg 0:4 | obj 4:5 | cls 5:] of the NCHW output (the reference's torch.cat).  The three
    pyramid levels are independent chains."""
    ctot = 5 + hd.num_classes

    def level(k, x):
        s = base_conv(rt, x, hd.stems[k])
        c = base_conv(rt, base_conv(rt, s, hd.cls_convs[k][0]), hd.cls_convs[k][1])
        g = base_conv(rt, base_conv(rt, s, hd.reg_convs[k][0]), hd.reg_convs[k][1])
        out = det_outs[k]
        conv_call(rt, g, hd.reg_preds[k], None, nchw=(out, ctot, 0))
        conv_call(rt, g, hd.obj_preds[k], None, nchw=(out, ctot, 4))
        conv_call(rt, c, hd.cls_preds[k], None, nchw=(out, ctot, 5))

        def bwd():
            dg = rt.det_grads[k]
            if dg is None:
                return
            B, _, h, w = dg.shape
            d = rt.buf(B, h, w, ctot)
            hip.nchw_to_nhwc(dg.contiguous(), d, ctot, B, ctot, h * w)
            conv_backward(rt, g, hd.reg_preds[k], d, ctot)
            conv_backward(rt, g, hd.obj_preds[k], d[..., 4:], ctot)
            conv_backward(rt, c, hd.cls_preds[k], d[..., 5:], ctot)
        rt.push(bwd)
        return None
    rt.parallel([(lambda k=k, x=x: level(k, x)) for k, x in enumerate(feats)], site=5)


class WeightPlanes:
    """Pre-split weights for the x6 kernels (hip.conv2d `w_planes`): the six-product scheme spends its VALU time on
    splitting fragments into bf16 planes, and a weight tile is the same for every row tile of a step -- so every 1x1 weight
    whose launch runs on an x6 tile kernel is split ONCE per forward, all of them in ONE launch (hip.conv_planes_pack), into
    the kernel's LDS stage image.  Derived caches like FusedQKV: the first forward that needs a pack builds it on the spot
    and registers it; from then on `refresh()` re-splits the whole table at the start of every forward (the parameters may
    have been updated in place)."""

    def __init__(self, model, device):
        self.owner, self.device = id(model), device
        self.entries = {}          # key -> [source tensor, J, K, sj, sk, kscale tensor or None, planes buffer]
        self.table, self.nblocks, self.dirty = None, 0, False

    def get(self, key, w, J, K, sj, sk, kscale):
        ent = self.entries.get(key)
        ids = (w.data_ptr(), None if kscale is None else kscale.data_ptr())
        if ent is not None and ent[7] == ids:
            return ent[6]
        buf = torch.empty((hip.conv_planes_bytes(J, K),), dtype=torch.uint8, device=self.device)
        ent = [w, J, K, sj, sk, kscale, buf, ids]
        self.entries[key] = ent
        self.dirty = True
        tab, nb = self._table([ent])
        hip.conv_planes_pack(tab, 1, nb)          # first use: split now (later forwards: refresh())
        return buf

    def _table(self, ents):
        rows, first = [], 0
        for w, J, K, sj, sk, kscale, buf, _ in ents:
            rows += [w.data_ptr(), J, K, sj, sk, 0 if kscale is None else kscale.data_ptr(), buf.data_ptr(), first]
            first += (K // 16) * 2 * ((J + 127) // 128)
        return torch.tensor(rows, dtype=torch.int64, device=self.device), first

    def refresh(self):
        # entries whose source storage has been replaced since they were registered (model.to, load_state_dict(assign=True),
        # p.data = ..., a rebuilt FusedQKV) must not be re-split from the old address: drop them, their next use re-registers
        stale = [k for k, e in self.entries.items()
                 if e[7] != (e[0].data_ptr(), None if e[5] is None else e[5].data_ptr())]
        for k in stale:
            del self.entries[k]
            self.dirty = True
        if not self.entries:
            return
        if self.dirty:
            self.ents = list(self.entries.values())
            self.table, self.nblocks = self._table(self.ents)
            self.dirty = False
        hip.conv_planes_pack(self.table, len(self.ents), self.nblocks)


BN_ZMASK = os.environ.get("VRNET_BN_ZMASK", "1") != "0"      # (diagnostic A/B switch)
PG_MAX_ROWS = int(os.environ.get("VRNET_PG_MAX_ROWS", "1000000000"))      # plane GEMMs only for maps of at most this many pixels (diagnostic override)
# which GEMM kinds of the ClusterBlocks run on plane operands by default, per compute_dtype (measured: DESIGN 3.6)
PLANE_GEMMS_DEFAULT = {"f32": False, "bf16": "fwd+wgrad", "off": False}


class PlaneWeights:
    """Weights of the plane GEMMs (hip.gemm_planes): every 1x1 weight a ClusterBlock multiplies with is split into bf16 planes
    ONCE per forward, all of them in one launch (hip.planes_split) -- forward form w[Cout][Cin] as it is, data-gradient form
    the transpose with the layer scale folded in.  Same life cycle as WeightPlanes: the first forward that needs a pack
    builds it on the spot and registers it, `refresh()` re-splits the whole table at the start of every later forward."""

    def __init__(self, model, device, np_):
        self.owner, self.device, self.np = id(model), device, np_
        self.entries = {}          # key -> [source, R, K, sr, sk, kscale or None, Planes, (pointers)]
        self.table, self.nblocks, self.dirty, self.ents = None, 0, False, []

    def get(self, key, w, R, K, sr, sk, kscale=None):
        ent = self.entries.get(key)
        ids = (w.data_ptr(), None if kscale is None else kscale.data_ptr())
        if ent is not None and ent[7] == ids:
            return ent[6]
        ent = [w, R, K, sr, sk, kscale, hip.Planes.empty(self.np, (R, K), self.device), ids]
        self.entries[key] = ent
        self.dirty = True
        tab, nb = self._table([ent])
        hip.planes_split(tab, 1, nb, self.np)          # first use: split now (later forwards: refresh())
        return ent[6]

    def _table(self, ents):
        rows, first = [], 0
        for w, R, K, sr, sk, kscale, pl, _ in ents:
            rows += [w.data_ptr(), R, K, sr, sk, 0 if kscale is None else kscale.data_ptr(), pl.t.data_ptr(), pl.ld, pl.plane, first]
            first += hip.planes_split_blocks(R, K)
        return torch.tensor(rows, dtype=torch.int64, device=self.device), first

    def refresh(self):
        # entries whose source storage has been replaced since they were registered (model.to, load_state_dict(assign=True),
        # p.data = ...) are dropped: their next use re-registers them
        stale = [k for k, e in self.entries.items()
                 if e[7] != (e[0].data_ptr(), None if e[5] is None else e[5].data_ptr())]
        for k in stale:
            del self.entries[k]
            self.dirty = True
        if not self.entries:
            return
        if self.dirty:
            self.ents = list(self.entries.values())
            self.table, self.nblocks = self._table(self.ents)
            self.dirty = False
        hip.planes_split(self.table, len(self.ents), self.nblocks, self.np)

    def fwd(self, w):
        """planes of w[Cout][Cin] (a 1x1 conv weight or a 2-D matrix): the B operand of the forward GEMM."""
        co, ci = w.shape[0], w.shape[1]
        return self.get((id(w), 0), w, co, ci, ci, 1)

    def dgrad(self, w, kscale=None):
        """planes of w^T [Cin][Cout] with kscale[Cout] folded in: the B operand of the data-gradient GEMM."""
        co, ci = w.shape[0], w.shape[1]
        return self.get((id(w), 1), w, ci, co, 1, ci, kscale)


class FusedQKV:
    """Concatenated [fc1 ; fc_v] weights and biases of every Cluster module (vr_coc.py:145-147): both 1x1 convs read
    the same normalised input, so each block runs them as ONE GEMM with 2*E*D output channels (twice the tiles of the
    small-M layers, the input read once) and their data / weight gradients as one GEMM each.  The copies are derived
    caches (the state_dict keeps fc1 / fc_v): persistent buffers refreshed by one multi-tensor copy per forward."""

    CHUNK = 4096

    def __init__(self, model, device):
        # bound to THIS module tree: copy.deepcopy (ModelEMA) and nn.DataParallel replicas carry a stale copy of the
        # object along (views lose their aliasing under deepcopy), so the owner id is checked on every forward
        self.owner = id(model)
        dst, src = [], []
        for mod in model.modules():
            tm = getattr(mod, "token_mixer", None)
            if tm is None or not hasattr(tm, "fc_v"):
                continue
            ed, c = tm.fc1.weight.shape[0], tm.fc1.weight.shape[1]
            w = torch.zeros((2 * ed, c), device=device)
            b = torch.zeros((2 * ed,), device=device)
            tm._fused_qkv = (w, b)
            dst += [w[:ed], w[ed:], b[:ed], b[ed:]]
            src += [tm.fc1.weight, tm.fc_v.weight, tm.fc1.bias, tm.fc_v.bias]
        self.dst, self.src, self.key = dst, src, None

    def refresh(self):
        key = tuple(t.data_ptr() for t in self.src)
        if key != self.key:
            dev = self.dst[0].device
            sizes = [t.numel() for t in self.dst]
            ct, ci = [], []
            for i, n in enumerate(sizes):
                k = (n + self.CHUNK - 1) // self.CHUNK
                ct += [i] * k
                ci += list(range(k))
            self.addrs = torch.tensor([t.data_ptr() for t in self.dst] + list(key), dtype=torch.int64, device=dev)
            self.sizes = torch.tensor(sizes, dtype=torch.int64, device=dev)
            self.ct = torch.tensor(ct, dtype=torch.int32, device=dev)
            self.ci = torch.tensor(ci, dtype=torch.int32, device=dev)
            self.n, self.nc, self.key = len(sizes), len(ct), key
        hip.mt_copy(self.addrs, self.sizes, self.ct, self.ci, self.n, self.nc, self.CHUNK)


def forward_pass(model, x, x_radar, record, need_dx=False, need_dr=False):
    """Runs the forward program.  Returns (rt, (xa, ra), dets, seg); with record=True rt.tape holds the backward."""
    if not (x.is_cuda and x_radar.is_cuda):
        raise RuntimeError("EfficientVRNet (HIP hot path) needs inputs on a HIP device; there is no CPU fallback")
    if x.device != x_radar.device:
        raise RuntimeError(f"image on {x.device} but radar on {x_radar.device}")
    if x.dim() != 4 or x_radar.dim() != 4 or x.shape[1] != 3 or x_radar.shape[1] != 4 or \
            x.shape[0] != x_radar.shape[0] or x.shape[2:] != x_radar.shape[2:]:
        raise RuntimeError(f"expected image (B,3,H,W) and radar (B,4,H,W), got {tuple(x.shape)} and {tuple(x_radar.shape)}")
    if x.dtype != torch.float32 or x_radar.dtype != torch.float32:
        x, x_radar = x.float(), x_radar.float()
    B, _, H, W = x.shape
    if H % 64 or W % 64:
        raise RuntimeError(f"input size {H}x{W} must be a multiple of 64 (fold-2 Cluster on the H/32 map)")
    first = next(model.parameters())
    if first.device != x.device:
        raise RuntimeError(f"model on {first.device} but inputs on {x.device}")
    with torch.cuda.device(x.device):          # kernels launch on the inputs' device, whatever the caller's current one
        rt = RT(x.device, model.training, record)
        rt.concurrent = bool(getattr(model, "concurrent", True))
        # measured (A/B inside one gpurun call, phi=l bs 8 512 px): two chains on two streams 30.8 ms/step, one
        # two-stream chain 31.6 ms/step (and 1 480 instead of 2 070 launches): the default is the faster one
        rt.pair_streams = bool(getattr(model, "pair_streams", False))
        if getattr(model, "debug_stamps", False):
            if getattr(model, "_stamp_buf", None) is None:
                model._stamp_buf = torch.zeros(512, dtype=torch.int64, device=x.device)
            model._stamp_names = []
            rt.stamps = (model._stamp_buf, model._stamp_names)
        rt.early_wgrads = int(getattr(model, "early_wgrads", 2))
        rt.fused_fusion = bool(getattr(model, "fused_fusion", True))
        rt.fused_upsample = bool(getattr(model, "fused_upsample", True))
        rt.mlp_recompute = getattr(model, "mlp_recompute", "auto")
        rt.overlap_fusion = bool(getattr(model, "overlap_fusion", True))     # RadarEnhanceByImage beside the image chain (round 5)
        rt.fused_mlp = bool(getattr(model, "fused_mlp", True))
        rt.bn_colstats = bool(getattr(model, "bn_colstats", True))
        rt.forced_idx = getattr(model, "forced_idx_maps", None)
        rt.sync_bn = getattr(model, "_sync_bn", None)
        if rt.sync_bn is not None:
            rt.bn_colstats = False          # the statistics come from the per-sample moments pass that feeds the collective
            if model.training:
                rt.sync_bn.begin_forward(B, x.device)
                rt.sync_batch_total = rt.sync_bn.batch_total      # this pass's count, also for ITS backward (bn_backward)
        cd = str(os.environ.get("VRNET_COMPUTE_DTYPE") or getattr(model, "compute_dtype", "f32")).lower()   # env: diagnostics
        if cd not in ("f32", "fp32", "float32", "torch.float32", "f32-mfma", "bf16", "bfloat16", "torch.bfloat16"):
            raise RuntimeError(f"compute_dtype {cd!r}: expected 'f32', 'f32-mfma' or 'bf16'")
        rt.bf16 = cd in ("bf16", "bfloat16", "torch.bfloat16")
        rt.fp32_precision = 0 if cd == "f32-mfma" else 2
        # torch.autocast around the call (the reference trains under torch.cuda.amp.autocast, utils/utils_fit.py:86-88):
        # the reduced-precision policy of this path is bf16-rounded conv operands with fp32 accumulation, fp32 tensors,
        # norms, clustering and outputs -- for autocast(float16) too (the fp32-range outputs need no GradScaler; one is
        # harmless)
        if torch.is_autocast_enabled():
            rt.bf16 = True
        rt.bucketer = getattr(model, "_grad_bucketer", None)
        rt.via_autograd = bool(getattr(model, "_via_autograd", False))   # stock DDP around the module (run_forward)
        rt.ready = [] if rt.bucketer is not None else None
        rt.on_param_grad = rt.ready.append if rt.bucketer is not None else None
        if getattr(model, "record_relu_masks", False):
            rt.relu_masks = {}
        fq = getattr(model, "_fused_qkv", None)
        if fq is None or fq.owner != id(model) or fq.dst[0].device != x.device:
            fq = model._fused_qkv = FusedQKV(model, x.device)
        refreshes = [fq.refresh]
        if getattr(model, "weight_planes", True) and not rt.bf16 and rt.fp32_precision == 2:
            wp = getattr(model, "_weight_planes", None)
            if wp is None or wp.owner != id(model) or wp.device != x.device:
                wp = model._weight_planes = WeightPlanes(model, x.device)
            refreshes.append(wp.refresh)       # after fq.refresh(): the concatenated fc1 | fc_v weights are sources too
            rt.wplanes = wp
        # plane GEMMs in the ClusterBlocks (csrc/pgemm.hip): model.plane_gemms = None (default: see below), False, True or a
        # string of GEMM kinds "fwd", "wgrad", "fwd+wgrad"
        pg = getattr(model, "plane_gemms", None)
        if pg is None:
            pg = PLANE_GEMMS_DEFAULT["bf16" if rt.bf16 else ("f32" if rt.fp32_precision == 2 else "off")]
        if pg:
            kinds = pg if isinstance(pg, str) else "fwd+wgrad"
            rt.pg_fwd, rt.pg_wgrad = "fwd" in kinds, "wgrad" in kinds
            rt.pnp = 1 if rt.bf16 else 3
            pwts = getattr(model, "_plane_weights", None)
            if pwts is None or pwts.owner != id(model) or pwts.device != x.device or pwts.np != rt.pnp:
                pwts = model._plane_weights = PlaneWeights(model, x.device, rt.pnp)
            refreshes.append(pwts.refresh)
            rt.pweights = pwts

        def refresh_all():
            for fn in refreshes:
                fn()
        xa = Act(torch.empty((B, H, W, 3), device=x.device), need_grad=need_dx)
        ra = Act(torch.empty((B, H, W, 4), device=x.device), need_grad=need_dr)
        rt.stamp("step start")
        hip.nchw_to_nhwc(x.contiguous(), xa.t, 3, B, 3, H * W)
        hip.nchw_to_nhwc(x_radar.contiguous(), ra.t, 4, B, 4, H * W)
        # (round 5 issued the per-step weight preparation -- [fc1 ; fc_v] copies, plane splits, Mlp and k x k packs -- on a side
        # stream beside the input fusion; same-call A/B in rounds 5 and 6, fp32 / bf16 bs 16 / nano: 25.00-25.03 vs 24.95-25.09,
        # 29.31-29.50 vs 29.42-29.47, 11.43-11.52 vs 11.40-11.51 ms -- neutral everywhere, removed: profiles/r06_ab_switches.txt)
        refresh_all()
        nc, ns = model.num_classes, model.num_seg_classes
        seg = torch.empty((B, ns, H, W), device=x.device)
        dets = [torch.empty((B, 5 + nc, H // s, W // s), device=x.device) for s in (8, 16, 32)]
        feats, seg_lo = neck_forward(rt, model.backbone, xa, ra, seg)
        rt.stamp("neck done")
        head_forward(rt, model.head, feats, dets)
        rt.stamp("forward done")
    model._last_idx_maps = rt.idx_maps
    if rt.relu_masks is not None:        # {BatchNorm state_dict prefix: (B,H,W,C) bool}: which ReLU inputs were > 0
        names = {mod: k for k, mod in model.named_modules()}
        model._last_relu_masks = {names[bn]: (a.t > 0) for bn, a in rt.relu_masks.items()}
    return rt, (xa, ra), dets, seg


def _mark_ready(rt, params, pos):
    """Hands parameters whose gradient kernels are ordered before the current stream to the bucketer (which may start
    a bucket's all-reduce behind them)."""
    if rt.bucketer is not None:
        for prm in params:
            rt.bucketer.mark_ready(prm, pos)


def _take_ready(rt):
    if not rt.ready:
        return []
    done = list(rt.ready)
    rt.ready.clear()                    # in place: rt.on_param_grad is this list's bound append
    return done


def backward_begin(rt, gdets, gseg):
    if rt.tape is None:
        raise RuntimeError("EfficientVRNet backward called twice (activations are freed after the first pass)")
    rt.det_grads, rt.seg_grad = tuple(gdets), gseg
    rt.tape_pos = len(rt.tape)
    rt.stamp("backward begin")


def backward_range(rt, lo, hi, flush_each=False):
    """Replays tape closures hi-1 ... lo (the backward of forward sections lo ... hi-1).  flush_each: every weight
    gradient is issued and joined after each closure (the bucketer's recording pass: a parameter is then attributed
    to the closure a cut could be placed behind)."""
    with torch.cuda.device(rt.device):
        for i in range(hi - 1, lo - 1, -1):
            rt.tape_pos = i
            rt.tape[i]()
            rt.flush_cluster_ab()
            if rt.stamps is not None:
                rt.stamp(f"bwd {i} {'P ' if getattr(rt.tape[i], 'is_parallel', False) else ''}{getattr(rt.tape[i], '__qualname__', '?').split('.')[0]}")
            if not flush_each and rt.early_wgrads and i > lo and not getattr(rt.tape[i - 1], "is_parallel", False):
                # the next closure is main-chain work: the section's weight gradients run beside it.  Mode 2 (default): only
                # behind the LAST section of chains (stage 0) -- its weight gradients (the largest maps) otherwise run after
                # everything else, 0.8 ms during which two kernels at a time own the chip -- and on four streams
                left = sum(1 for f in rt.tape[:i] if getattr(f, "is_parallel", False))
                if rt.early_wgrads == 1 or left <= 1:
                    rt.start_deferred_wgrads(4 if rt.early_wgrads == 2 else 0)
            done = rt.join_aside(0 if flush_each else ASIDE_LAG)
            if flush_each:
                # (recording pass: a parameter is attributed to the closure behind which its gradient IS complete in the real
                # schedule -- weight gradients a section deferred run, and are joined, in the NEXT section's backward; they are not
                # flushed here any more (round 5): a cut between the two sections then leaves them to the next segment)
                done += _take_ready(rt)
            _mark_ready(rt, done, i)


def backward_cut(rt, final=True):
    """Every gradient kernel issued so far is ordered before whatever the current stream runs next (joins the
    side streams): the point where a captured segment ends / a bucket's collective may start.  final=False (a cut between two
    segments, round 5): weight gradients a section DEFERRED stay deferred -- they run beside the next section's chains in the next
    segment, where the recording pass attributed them -- instead of running alone in front of the cut (0.7 ms per step)."""
    with torch.cuda.device(rt.device):
        rt.flush_cluster_ab()
        done = rt.join_aside(0)
        if final:
            rt.flush_deferred_wgrads()
        else:
            rt.join_started_wgrads()
        done += _take_ready(rt)
        _mark_ready(rt, done, rt.tape_pos)


def backward_end(rt, model, inputs, needs, params=None, needs_params=None):
    """Input gradients (NCHW) and publication of the parameter gradients; frees the tape."""
    with torch.cuda.device(rt.device):
        backward_cut(rt)
        rt.stamp("backward joined")
        rt.tape = None
        outs = []
        for act, need in zip(inputs, needs):
            if need and act.grad is not None:
                g = torch.empty((act.B, act.C, act.H, act.W), device=act.t.device)
                hip.nhwc_to_nchw(act.grad, act.C, g, act.B, act.C, act.H * act.W)
                outs.append(g)
            else:
                outs.append(None)
        pouts = None
        params = list(model.parameters()) if params is None else params
        if rt.bucketer is not None:          # data parallel: buckets own the gradients (all-reduced, then .grad = view)
            stray = [p for p in rt.pgrads if rt.bucketer.view(p) is None]
            if stray:
                raise RuntimeError(f"{len(stray)} parameter gradient(s) were produced outside the data-parallel buckets")
            rt.bucketer.finish()
        elif (getattr(model, "autograd_param_grads", False) or rt.via_autograd) and needs_params is not None:
            # through torch.autograd: every parameter's AccumulateGrad node runs, so the hooks a stock
            # DistributedDataParallel reducer hangs on it fire (train.py:367-368).  Under such a wrapper a parameter that
            # asks for a gradient and gets none (the six zero-size ShuffleAttention(channel=3) entries) receives zeros:
            # the reducer counts one hook call per parameter it found in the graph
            pouts = [rt.pgrads.get(p) if need else None for p, need in zip(params, needs_params)]
            if rt.via_autograd:
                pouts = [torch.zeros_like(p) if (g is None and need) else g for p, g, need in zip(params, pouts, needs_params)]
        else:
            # Default: publish parameter gradients directly (.grad = buffer, or += into an existing .grad, as
            # loss.backward() would) -- saves one copy of every gradient per step.  Set
            # model.autograd_param_grads = True to receive them through torch.autograd.grad instead.
            for i, p in enumerate(params):
                g = rt.pgrads.get(p) if (needs_params is None or needs_params[i]) else None
                if g is not None:
                    if p.grad is None:
                        p.grad = g
                    else:
                        hip.add_(p.grad, g)
        rt.pgrads.clear()
        rt.det_grads = rt.seg_grad = None
        rt.release()
    return outs, pouts


class _VRNetFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, x, x_radar, *params):
        record = any(ctx.needs_input_grad)
        rt, inputs, dets, seg = forward_pass(model, x, x_radar, record, ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        # The tape belongs to the autograd node, not to the runtime: its closures capture `rt`, and rt -> tape -> closure ->
        # rt would be a reference cycle that keeps every saved activation (GBs) alive until Python's cycle collector runs
        # when an output is dropped without a backward pass.  Without the back edge the node's death frees them at once.
        ctx.tape, rt.tape = rt.tape, None
        ctx.rt, ctx.inputs = rt, inputs
        ctx.params = params
        ctx.model = model
        ctx.set_materialize_grads(False)
        return (*dets, seg)

    @staticmethod
    def backward(ctx, g0, g1, g2, gseg):
        rt, model = ctx.rt, ctx.model
        if ctx.tape is None:
            raise RuntimeError("EfficientVRNet: backward through the same forward twice (its saved activations were freed)")
        rt.tape, ctx.tape = ctx.tape, None
        backward_begin(rt, (g0, g1, g2), gseg)
        rec = rt.bucketer is not None and rt.bucketer.recording
        backward_range(rt, 0, len(rt.tape), flush_each=rec)
        outs, pouts = backward_end(rt, model, ctx.inputs, ctx.needs_input_grad[1:3], list(ctx.params),
                                   ctx.needs_input_grad[3:])
        ctx.rt = ctx.inputs = None
        return (None, *outs, *(pouts if pouts is not None else [None] * len(ctx.params)))


_warned_unwrapped = [False]


def _ddp_wrapper_of(model):
    """The torch DistributedDataParallel instance whose forward() is calling `model` right now (None: there is none).
    DDP's reducer hangs its hooks on the parameters' AccumulateGrad nodes from C++ -- nothing on the module or its
    parameters shows them -- so the wrapper is looked for where it must be: up the call stack
    (DistributedDataParallel.forward -> _run_ddp_forward -> module(...) -> Module._call_impl -> forward -> run_forward;
    the WHOLE stack is walked -- it stops at the first wrapper -- so any depth of nesting modules and hooks is covered).
    A wrapper that calls the module from somewhere the stack does not show (a helper thread) can say so explicitly:
    `model.autograd_param_grads = True` returns the parameter gradients through autograd without any detection."""
    import sys
    try:
        from torch.nn.parallel import DistributedDataParallel as DDP
    except Exception:      # a torch build without distributed
        return None
    f = sys._getframe(1)
    while f is not None:
        s = f.f_locals.get("self")
        if isinstance(s, DDP) and any(m is model for m in s.module.modules()):
            return s
        f = f.f_back
    return None


def _warn_if_unreduced(model):
    """Several ranks, gradients wanted, and neither parallel.DataParallelVRNet nor a DistributedDataParallel wrapper nor
    model.autograd_param_grads: the ranks would diverge silently -- say so once."""
    if _warned_unwrapped[0]:
        return
    try:
        import torch.distributed as dist
        many = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    except Exception:
        many = False
    if many:
        import warnings
        _warned_unwrapped[0] = True
        warnings.warn("EfficientVRNet runs on several ranks without parallel.DataParallelVRNet, without a torch "
                      "DistributedDataParallel wrapper on its call stack and without model.autograd_param_grads = True: "
                      "parameter gradients are published by assignment (.grad = buffer), no reducer hook will see them and "
                      "the ranks' gradients are NOT averaged", RuntimeWarning, stacklevel=3)


def run_forward(model, x, x_radar):
    params = tuple(model.parameters())
    # Stock DistributedDataParallel around the module (the reference's train.py:367-368, unchanged): its reducer only sees
    # gradients that arrive through autograd's AccumulateGrad nodes, so under it the parameter gradients are returned
    # through torch.autograd instead of being published by assignment (one extra copy of every gradient per step;
    # parallel.DataParallelVRNet is the fast path -- INTEGRATION.md).
    ddp = _ddp_wrapper_of(model) if torch.is_grad_enabled() else None
    if ddp is not None and getattr(model, "_grad_bucketer", None) is not None:
        raise RuntimeError("EfficientVRNet is wrapped by torch DistributedDataParallel AND by parallel.DataParallelVRNet: "
                           "its gradients would be all-reduced twice -- use one of the two")
    model._via_autograd = ddp is not None
    if (ddp is None and not getattr(model, "autograd_param_grads", False) and getattr(model, "_grad_bucketer", None) is None
            and torch.is_grad_enabled() and model.training):
        _warn_if_unreduced(model)
    if not torch.is_grad_enabled() or not (x.requires_grad or x_radar.requires_grad or any(p.requires_grad for p in params)):
        # nothing to differentiate (torch.no_grad(), frozen model): no tape, no saved activations, no autograd node
        rt, _, dets, seg = forward_pass(model, x, x_radar, record=False)
        rt.release()
        return list(dets), seg
    out = _VRNetFunction.apply(model, x, x_radar, *params)
    return [out[0], out[1], out[2]], out[3]
