"""Whole-path parity on the GPU: EfficientVRNet.forward/backward through the C-ABI kernels against
(a) the CPU oracle with teacher-forced assignments (tests/parity.py) and (b) the golden vectors the
reference itself produced (tests/golden/net_*.npz)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def A():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import asy_vrnet_amd
    return asy_vrnet_amd


def build(A, phi, size, pseed, training, pair=False):
    """pair: the image and the radar chain of every backbone stage as one two-stream batch (model.pair_streams)."""
    m = A.EfficientVRNet(4, 9, phi, img_size=size).cuda()
    A.randomize_state_dict(m.state_dict(), seed=pseed)
    m.pair_streams = pair
    return m.train(training)


@pytest.mark.parametrize("phi,size,batch,training,seeds", [
    ("nano", 64, 2, True, (21, 31)), ("nano", 64, 2, False, (21, 31)), ("nano", 128, 2, True, (21, 31)),
    ("tiny", 128, 3, True, (21, 31)), ("nano", 256, 2, True, (21, 31)), ("l", 128, 2, True, (21, 31)),
    ("s", 128, 2, True, (21, 31)), ("m", 128, 2, True, (21, 31)), ("m", 128, 2, True, (3, 9)),
    ("nano", 128, 2, True, (11, 5)), ("nano", 128, 2, True, (11, 6)),
    ("nano", 256, 2, True, (21, 31, "pair")), ("l", 128, 2, True, (21, 31, "pair")), ("nano", 128, 4, False, (21, 31, "pair")),
    ("s", 256, 2, True, (3, 9, "pair"))])
def test_against_oracle(A, phi, size, batch, training, seeds):
    """One seed pair for every width, nothing hand-picked: with (21, 31) phi=m has ONE BatchNorm+ReLU pre-activation (of
    98 304 in that layer) within fp32 rounding of zero whose mask bit differs between this path and the fp64 oracle;
    (11, 5) and (11, 6) are two more such cases found by a seed scan.  The comparison is ReLU-mask-aware (tests/parity.py,
    2b): the oracle takes the path's masks and every differing element must be within rounding of zero."""
    from tests.parity import compare_with_oracle
    m = build(A, phi, size, seeds[0], training, pair="pair" in seeds)      # "pair": one two-stream chain per stage
    rep = compare_with_oracle(m, batch, size, iseed=seeds[1], check_grads=training, oracle_dtype=torch.float64)
    print(rep)
    assert rep["ok"], rep


def test_1024_forward_against_oracle(A):
    """BASELINE config 5's input size: 32x32-point regions in every backbone stage, 64x64 (4096 points) in neck p3
    (streaming Cluster kernel); needs EfficientVRNet(..., img_size=1024) as the reference's fea_pos is 512-only."""
    from tests.parity import compare_with_oracle
    m = build(A, "nano", 1024, 13, False)
    rep = compare_with_oracle(m, 1, 1024, iseed=17, check_grads=False, oracle_dtype=torch.float32)
    print(rep)
    assert rep["ok"], rep
    with pytest.raises(RuntimeError, match="fea_pos"):
        build(A, "nano", 512, 13, False)(torch.zeros(1, 3, 1024, 1024, device="cuda"), torch.zeros(1, 4, 1024, 1024, device="cuda"))


@pytest.mark.parametrize("name", ["net_nano_128_train", "net_nano_512_train"])
def test_against_reference_golden_with_its_decisions(A, name, golden_dir):
    """The robust form of test_against_reference_golden: the fixtures also hold the reference's own Cluster
    assignments and ReLU masks.  (1) The HIP path, left to itself, must take the same decisions except for rare points;
    (2) teacher-forced to the reference's assignments (model.forced_idx_maps -> vrnet_cluster_fwd_forced_f32) a tied
    arg-max can no longer move anything: det / seg within 1e-3 of the reference EVERYWHERE, no "fraction beyond" escape,
    and -- when no ReLU bit differs either -- the input gradients within 5e-3 everywhere."""
    from tests.parity import load_reference_decisions, rel_err
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    meta = json.load(open(os.path.join(golden_dir, name + ".json")))
    ref_idx, ref_masks = load_reference_decisions(golden_dir, name)
    m = build(A, meta["phi"], meta["size"], meta["pseed"], True)
    x, r = A.synthetic_inputs(meta["batch"], meta["size"], meta["iseed"])
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    with torch.no_grad():
        m(x.cuda(), r.cuda())
    own = {k: v.permute(0, 3, 1, 2).cpu() for k, v in m._last_idx_maps.items()}
    assert set(own) == set(ref_idx)
    points = sum(v.numel() for v in own.values())
    flips = sum(int((own[k] != ref_idx[k]).sum()) for k in own)
    print(name, "assignments", points, "decided differently from the reference", flips)
    assert flips <= max(3, points // 1000)
    m.load_state_dict(sd0)
    m.forced_idx_maps = {k: v.permute(0, 2, 3, 1).contiguous().cuda() for k, v in ref_idx.items()}
    m.record_relu_masks = True
    try:
        xg, rg = x.cuda().requires_grad_(True), r.cuda().requires_grad_(True)
        det, seg = m(xg, rg)
    finally:
        m.forced_idx_maps = None
        m.record_relu_masks = False
    for k, v in m._last_idx_maps.items():
        assert torch.equal(v.permute(0, 3, 1, 2).cpu(), ref_idx[k]), k
    mine = {k: v.permute(0, 3, 1, 2).cpu() for k, v in m._last_relu_masks.items()}
    assert set(mine) == set(ref_masks)
    rflips = sum(int((mine[k] != ref_masks[k]).sum()) for k in mine)
    relem = sum(v.numel() for v in mine.values())
    print("ReLU elements", relem, "decided differently from the reference", rflips)
    assert rflips <= max(3, relem // 10000)
    st = meta.get("seg_stride", 1)
    for i in range(3):
        assert rel_err(det[i], torch.from_numpy(z[f"det{i}"])) < 1e-3, (i, rel_err(det[i], torch.from_numpy(z[f"det{i}"])))
    assert rel_err(seg[:, :, ::st, ::st], torch.from_numpy(z["seg"])) < 1e-3
    sd = m.state_dict()
    for k in z.files:
        if k.startswith("s:"):
            assert rel_err(sd[k[2:]], torch.from_numpy(z[k])) < 1e-3, k
    sum((d * d).mean() for d in det).add((seg * seg).mean()).backward()
    errs = [rel_err(xg.grad[:, :, ::st, ::st], torch.from_numpy(z["dx"])), rel_err(rg.grad[:, :, ::st, ::st], torch.from_numpy(z["dr"]))]
    print("input gradients vs the reference", errs)
    if rflips == 0:
        assert max(errs) < 5e-3, errs
    else:      # a ReLU bit within rounding of zero decided the other way moves the gradient behind it by a few per cent
        assert max(errs) < 0.1, errs


def test_512_bs2_against_oracle(A):
    from tests.parity import compare_with_oracle
    m = build(A, "nano", 512, 3, True)
    rep = compare_with_oracle(m, 2, 512, iseed=9, check_grads=True, oracle_dtype=torch.float64)
    print(rep)
    assert rep["ok"], rep


@pytest.mark.parametrize("name", ["net_nano_64_train", "net_nano_64_eval", "net_nano_128_train", "net_tiny_128_eval",
                                  "net_nano_512_train", "net_nano_128x192_train"])
def test_against_reference_golden(A, name, golden_dir):
    """Direct comparison with the reference's own outputs.  These cases have no numerical near-ties between the
    reference and this implementation (checked: zero flips), so plain 1e-3 holds on the outputs -- also at the
    benchmark resolution (net_nano_512_train, the reference in fp32: measured 2e-6; the reference in fp64 assigns
    0.07 % of the points differently than ANY fp32 evaluation, which is why that fixture pins the oracle, not this).
    Gradients at 512 px: the reference's own fp32 backward is 1-25 % off the exact gradient on single parameters
    (tests/test_oracle_golden.py), so the per-parameter bar there is 3e-2 (measured worst 9e-3)."""
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    meta = json.load(open(os.path.join(golden_dir, name + ".json")))
    m = build(A, meta["phi"], meta["size"], meta["pseed"], meta["training"])
    x, r = A.synthetic_inputs(meta["batch"], meta["size"], meta["iseed"])
    grads = "dx" in z.files
    xg, rg = x.cuda().requires_grad_(grads), r.cuda().requires_grad_(grads)
    det, seg = m(xg, rg)
    from tests.parity import rel_err
    st = meta.get("seg_stride", 1)
    small = np.max(meta["size"]) <= 192        # (size: a side, or [H, W] for the rectangular case)
    gtol = 5e-3 if small else 3e-2
    def same(mine, ref, what):
        # (512 px, 0.9 M hard assignments: this fixture is tie-free for the kernel set as it stands.  It is not a robust
        # property: in round 3 a direct kernel for nano's 16-channel patch embedding -- another summation order, 1e-7
        # apart -- turned ONE tied decision the other way, and through data_normal's batch-wide max (vr_coc.py:59-67) 87 % of
        # det0 moved by more than 1e-3 (up to 9.5 %) while the teacher-forced comparison of the same configuration,
        # test_512_bs2_against_oracle, stayed at 2e-6.  That test is the gate; if this one ever fails with a large
        # `fraction`, look there first.)
        e = (mine.detach().double().cpu() - torch.from_numpy(ref).double()).abs() / float(np.abs(ref).max())
        assert float(e.max()) < 1e-3, (what, "max", float(e.max()), "fraction beyond 1e-3", float((e > 1e-3).double().mean()))
    for i in range(3):
        same(det[i], z[f"det{i}"], f"det{i}")
    same(seg[:, :, ::st, ::st], z["seg"], "seg")
    if meta["training"]:
        sd = m.state_dict()
        for k in z.files:
            if k.startswith("s:"):
                assert rel_err(sd[k[2:]], torch.from_numpy(z[k])) < 1e-3, k
    if grads:
        sum((d * d).mean() for d in det).add((seg * seg).mean()).backward()
        for mine, ref in ((xg.grad[:, :, ::st, ::st], z["dx"]), (rg.grad[:, :, ::st, ::st], z["dr"])):
            if small:
                assert rel_err(mine, torch.from_numpy(ref)) < 5e-3
            else:
                # 512 px: 0.9 M hard assignments and 3 M ReLU masks; ONE of them decided differently by two fp32
                # evaluations (a numerical tie) moves the input gradient around that point by a few per cent while
                # everything else agrees to 1e-4 (the teacher-forced comparison with the oracle covers this case
                # exactly: test_512_bs2_against_oracle, same seeds).  Here: all but 1 % of the elements within 5e-3
                # (measured: 0.35 % off, i.e. the receptive field of a single decision), none off by more than 10 %.
                e = (mine.detach().double().cpu() - torch.from_numpy(ref).double()).abs() / float(np.abs(ref).max())
                assert float((e > 5e-3).double().mean()) < 1e-2 and float(e.max()) < 0.1, (float(e.max()), float((e > 5e-3).double().mean()))
        pd = dict(m.named_parameters())
        for k in z.files:
            if k.startswith("g:"):
                ref = torch.from_numpy(z[k])
                if ref.abs().max() < 1e-6:
                    assert pd[k[2:]].grad.abs().max() < 1e-4, k
                else:
                    # (a scalar such as sim_alpha is one sum over every region of the map: at 512 px a single
                    # differently-decided point moves it further than it moves a weight matrix)
                    lim = gtol if (small or ref.numel() >= 16) else 0.2
                    assert rel_err(pd[k[2:]].grad, ref) < lim, (k, rel_err(pd[k[2:]].grad, ref))


@pytest.mark.parametrize("pair", [False, True])
@pytest.mark.parametrize("phi,size,batch", [("nano", 64, 2), ("nano", 256, 2), ("s", 128, 4), ("nano", (128, 192), 2)])
def test_captured_step_equals_eager_step_bit_for_bit(A, phi, size, batch, pair):
    """graph.GraphedStep (what bench.py times) replays the same kernels as the eager step, with the chains really
    concurrent on their side streams: every parameter gradient, the loss and the BatchNorm statistics must equal the
    eager results exactly, replay after replay.  (This is the test that catches a missing stream dependency: the eager
    step hides such races behind its Python launch gaps, a replayed graph does not.)"""
    from asy_vrnet_amd.graph import GraphedStep

    def loss_of(det, seg):
        return sum((d * d).mean() for d in det) + (seg * seg).mean()
    x, r = A.synthetic_inputs(batch, size, 3, "cuda")
    ref = build(A, phi, size, 7, True, pair)
    sd0 = {k: v.clone() for k, v in ref.state_dict().items()}
    loss_ref = loss_of(*ref(x, r))
    loss_ref.backward()
    sd1 = {k: v.clone() for k, v in ref.state_dict().items()}
    m = build(A, phi, size, 7, True, pair)
    gs = GraphedStep(m, loss_of, batch, size, x.device, warmup=2)
    for rep in range(3):
        m.load_state_dict(sd0)
        loss = gs(x, r)
        torch.cuda.synchronize()
        assert torch.equal(loss, loss_ref.detach())
        for (k, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
            if p.numel():
                assert torch.equal(p.grad, q.grad), (rep, k)
        for k, v in m.state_dict().items():
            assert torch.equal(v, sd1[k]), (rep, k)


def test_runs_under_autocast_like_the_reference_training_loop(A):
    """utils/utils_fit.py:86-88 calls the model under torch.cuda.amp.autocast: here that selects the bf16-operand conv
    path (fp32 tensors and outputs), exactly what compute_dtype = "bf16" does; backward works through a GradScaler."""
    m = build(A, "nano", 128, 5, True)
    x, r = A.synthetic_inputs(2, 128, 1, "cuda")
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    for dt in (torch.bfloat16, torch.float16):
        m.load_state_dict(sd0)
        with torch.autocast("cuda", dtype=dt):
            det, seg = m(x, r)
        assert seg.dtype == torch.float32 and all(d.dtype == torch.float32 for d in det)
        m.load_state_dict(sd0)
        m.compute_dtype = "bf16"
        det2, seg2 = m(x, r)
        m.compute_dtype = "f32"
        assert torch.equal(seg, seg2) and all(torch.equal(a, b) for a, b in zip(det, det2))
    m.load_state_dict(sd0)
    det3, seg3 = m(x, r)                                  # fp32 path: close to, not equal to, the autocast result
    assert not torch.equal(seg3, seg) and ((seg3 - seg).abs().max() / seg3.abs().max()) < 0.1
    scaler = torch.amp.GradScaler("cuda")
    m.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.float16):
        det, seg = m(x, r)
        loss = sum((d * d).mean() for d in det) + (seg * seg).mean()
    scaler.scale(loss).backward()
    g = m.head.stems[0].conv.weight.grad
    assert g is not None and torch.isfinite(g).all() and g.abs().max() > 0


def test_other_device_index_and_data_parallel_replicas(A):
    """`.cuda(i)` with i != 0 (train.py:285-287 uses cuda:1) and nn.DataParallel replicas (yolo.py:103-104): kernels
    launch on the inputs' device and stream whatever the caller's current device is.  Needs >= 2 visible GPUs."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs (the round's GPU box has one)")
    m0 = build(A, "nano", 64, 5, False)
    x, r = A.synthetic_inputs(2, 64, 1, "cuda:0")
    with torch.no_grad():
        d0, s0 = m0(x, r)
        m1 = build(A, "nano", 64, 5, False).to("cuda:1")
        d1, s1 = m1(x.to("cuda:1"), r.to("cuda:1"))               # current device stays cuda:0
        assert s1.device.index == 1 and torch.equal(s1.cpu(), s0.cpu())
        dp = torch.nn.DataParallel(m0, device_ids=[0, 1])
        dd, sd = dp(torch.cat([x, x]), torch.cat([r, r]))
        assert sd.shape[0] == 4 and torch.isfinite(sd).all()
    with pytest.raises(RuntimeError, match="inputs on"):
        m0(x.to("cuda:1"), r.to("cuda:1"))


def test_module_surface_behaviour(A):
    """deepcopy (ModelEMA, yolo_training.py:457), no_grad eval, repeated backward error, seg-only loss."""
    import copy
    m = build(A, "nano", 64, 5, True)
    x, r = A.synthetic_inputs(2, 64, 1)
    x, r = x.cuda(), r.cuda()
    ema = copy.deepcopy(m).eval()
    with torch.no_grad():
        d1, s1 = ema(x, r)
        d2, s2 = ema(x, r)
    assert torch.equal(s1, s2) and all(torch.equal(a, b) for a, b in zip(d1, d2))   # deterministic
    det, seg = m(x, r)
    seg.mean().backward()                                                           # det grads absent
    assert m.head.cls_preds[0].weight.grad is None or m.head.cls_preds[0].weight.grad.abs().max() == 0
    assert m.backbone.upsample2_0.upsample[0].conv.weight.grad.abs().max() > 0
    with pytest.raises(RuntimeError):
        m(x.cpu(), r.cpu())


def test_data_parallel_wrapper_single_rank(A):
    """parallel.DataParallelVRNet on one rank (no process group): gradients land in the flat buckets, equal the
    plain run bit for bit, and the hipGraph-captured step (deferred all-reduce) reproduces them."""
    from asy_vrnet_amd.parallel import DataParallelVRNet
    from asy_vrnet_amd.graph import GraphedStep

    def loss_of(det, seg):
        return sum((d * d).mean() for d in det) + (seg * seg).mean()
    x, r = A.synthetic_inputs(2, 64, 3)
    x, r = x.cuda(), r.cuda()
    ref = build(A, "nano", 64, 7, True)
    loss_of(*ref(x, r)).backward()
    sd_after = {k: v.clone() for k, v in ref.state_dict().items()}
    m = build(A, "nano", 64, 7, True)
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    dp = DataParallelVRNet(m, bucket_bytes=1 << 20)
    assert len(dp.bucketer.buckets) > 3 and dp.bucketer.recording
    for step in range(2):       # pass 0 records the execution order, pass 1 runs on the rebuilt arena (3 segments)
        m.load_state_dict(sd0)
        loss_of(*dp(x, r)).backward()
        for (k, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
            if p.numel():
                assert p.grad is not None and torch.equal(p.grad, q.grad), (step, k)
                assert p.grad.data_ptr() == dp.bucketer.view(p).data_ptr()
        for k, v in m.state_dict().items():
            assert torch.equal(v, sd_after[k]), k
    bk = dp.bucketer
    assert not bk.recording and len(bk.cuts) == 2 and len(bk.segment_slices) == 3
    names = {p: k for k, p in m.named_parameters()}
    assert names[bk.params[0]].startswith("head.") and "network" in names[bk.params[len(bk.params) // 2]]
    # the two chains of a stage interleave in the arena (execution order), unlike the registration order
    st2 = [names[p] for p in bk.params if ".network.6." in names[p] or ".network_radar.6." in names[p]]
    flips = sum(1 for a, b in zip(st2, st2[1:]) if ("radar" in a) != ("radar" in b))
    assert flips >= 2, st2[:8]
    # captured step on a fresh replica
    m2 = build(A, "nano", 64, 7, True, pair=True)
    dp2 = DataParallelVRNet(m2, bucket_bytes=1 << 20)
    gs = GraphedStep(dp2, loss_of, 2, 64, x.device, warmup=2)
    assert len(gs.graphs) == 3                                          # backward cut into 3 captured segments
    m2.load_state_dict(build(A, "nano", 64, 7, True).state_dict())      # undo the warm-up's BN statistics
    loss = gs(x, r)
    torch.cuda.synchronize()
    # (the eager reference of a two-stream replica is a two-stream model: since round 3 the single-stream chain runs
    #  GroupNorm through the one-launch kernels, whose rounding differs from the two-stream chain's in the last bit)
    ref2 = build(A, "nano", 64, 7, True, pair=True)
    loss2 = loss_of(*ref2(x, r))
    loss2.backward()
    assert torch.equal(loss, loss2.detach())
    for (k, p), (_, q) in zip(m2.named_parameters(), ref2.named_parameters()):
        if p.numel():
            assert torch.equal(p.grad, q.grad), k
    assert abs(float(loss) - float(loss_of(*ref(x, r)).detach())) < 1e-5 * abs(float(loss))


@pytest.mark.parametrize("phi,size,batch", [("nano", 128, 2), ("s", 256, 2)])
def test_round5_schedule_and_fused_passes_agree_with_the_round4_forms(A, phi, size, batch):
    """Round 5 changed HOW the program runs, not what it computes: RadarEnhanceByImage beside the image chain of the next stage
    (model.overlap_fusion), the fused passes of the fusion blocks (model.fused_fusion: csrc/fusion.hip, sa_cat_sums), early
    weight gradients of the last section (model.early_wgrads); the side-stream weight preparation of that round measured neutral
    and was removed in round 6.
    With all of them off the program is round 4's; outputs, BatchNorm statistics and every gradient must agree to rounding
    (the fused passes reassociate a few sums: column statistics per workgroup instead of per chunk), and the default program
    must repeat itself bit for bit."""
    def loss_of(det, seg):
        return sum((d * d).mean() for d in det) + (seg * seg).mean()
    x, r = A.synthetic_inputs(batch, size, 9)
    x, r = x.cuda(), r.cuda()

    def run(**flags):
        m = build(A, phi, size, 13, True)
        for k, v in flags.items():
            setattr(m, k, v)
        det, seg = m(x, r)
        loss_of(det, seg).backward()
        torch.cuda.synchronize()
        return ([d.detach() for d in det] + [seg.detach()], {k: p.grad for k, p in m.named_parameters() if p.grad is not None},
                {k: v.clone() for k, v in m.state_dict().items() if "running" in k})
    new = run()
    again = run()
    old = run(overlap_fusion=False, fused_fusion=False, early_wgrads=0)
    for a, b in zip(new[0], again[0]):
        assert torch.equal(a, b)
    assert all(torch.equal(new[1][k], again[1][k]) for k in new[1])
    for a, b in zip(new[0], old[0]):
        assert ((a - b).abs().max() / b.abs().max()).item() < 2e-5
    for k in old[2]:
        assert torch.allclose(new[2][k], old[2][k], rtol=1e-5, atol=1e-6), k
    assert new[1].keys() == old[1].keys()
    worst = max(((new[1][k] - old[1][k]).norm() / old[1][k].norm().clamp_min(1e-20)).item() for k in old[1])
    assert worst < 2e-3, worst      # (a tied arg-max or ReLU bit decided the other way would show up here as percents)


def test_recorded_order_fixture_is_current(A, golden_dir):
    """tests/golden/dp_ready_pos.json -- the backward execution order the CPU test of the 8-GPU segment / bucket plan is built
    on (tests/test_data_parallel_gloo.py::test_n8_plan_at_l) -- equals what a recording pass of THIS program stamps (the
    order is a function of the program's structure only: identical for every width, batch and image size)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from make_golden_dp_plan import recorded_positions
    want = json.load(open(os.path.join(golden_dir, "dp_ready_pos.json")))
    got = recorded_positions("nano", size=64, batch=2)
    # (the section index is what the plan depends on; the order INSIDE a section only permutes a bucket's members)
    assert {k: v[0] for k, v in got.items()} == {k: v[0] for k, v in want.items()}, \
        "the backward program changed its section order: rerun tools/make_golden_dp_plan.py on the GPU"


def test_stock_distributed_data_parallel_single_rank(A):
    """The reference's own wrapper, unchanged (train.py:367-368): torch DistributedDataParallel(model,
    find_unused_parameters=True) on a 1-rank RCCL process group.  Under it the parameter gradients go back through
    torch.autograd (program.run_forward finds the wrapper on the call stack), so the reducer's AccumulateGrad hooks fire:
    two steps, gradients equal to the plain run's, and the six zero-size ShuffleAttention(channel=3) parameters -- which
    never receive a real gradient -- do not leave the reducer waiting ("Expected to have finished reduction")."""
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    from asy_vrnet_amd.parallel import DataParallelVRNet

    def loss_of(det, seg):
        return sum((d * d).mean() for d in det) + (seg * seg).mean()
    own_group = not dist.is_initialized()
    if own_group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29591")
        dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        x, r = A.synthetic_inputs(2, 64, 3)
        x, r = x.cuda(), r.cuda()
        ref = build(A, "nano", 64, 7, True)
        m = build(A, "nano", 64, 7, True)
        ddp = DDP(m, device_ids=[torch.cuda.current_device()], find_unused_parameters=True)
        for step in range(2):
            for mod in (ref, m):
                mod.zero_grad(set_to_none=True)
            loss_of(*ref(x, r)).backward()
            loss_of(*ddp(x, r)).backward()          # a second step raises inside DDP if a hook of step 0 never fired
            assert m._via_autograd and not ref._via_autograd
            n_zero = 0
            for (k, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
                if p.numel() == 0:
                    n_zero += 1
                    continue
                assert p.grad is not None and q.grad is not None, k
                assert torch.equal(p.grad, q.grad), (step, k)
            assert n_zero == 6
        # BatchNorm statistics moved the same way (the wrapper broadcasts buffers from rank 0: a no-op on one rank)
        for (k, v), (_, w) in zip(m.state_dict().items(), ref.state_dict().items()):
            assert torch.equal(v, w), k
        # an eval forward through the wrapper under no_grad: no autograd node, nothing for the reducer to wait for
        with torch.no_grad():
            ddp.eval()(x, r)
        ddp.train()
        # both wrappers at once would reduce twice: refused with a message naming them
        DataParallelVRNet(m, bucket_bytes=1 << 20)
        with pytest.raises(RuntimeError, match="DataParallelVRNet"):
            ddp(x, r)
    finally:
        if own_group:
            dist.destroy_process_group()


@pytest.mark.gpu
def test_deepcopy_and_updated_weights_use_fresh_fused_weights(A):
    """The concatenated fc1|fc_v weights are derived caches: a deep copy (ModelEMA, yolo_training.py:457) and an
    in-place parameter update (optimizer step) must both be reflected by the next forward."""
    import copy
    model = A.EfficientVRNet(4, 9, "nano", img_size=(64, 64)).cuda().eval()
    A.randomize_state_dict(model.state_dict(), seed=3)
    x, r = A.synthetic_inputs(2, 64, 7, "cuda")
    with torch.no_grad():
        det0, seg0 = model(x, r)
        clone = copy.deepcopy(model)
        det1, seg1 = clone(x, r)
        assert torch.equal(seg0, seg1) and all(torch.equal(a, b) for a, b in zip(det0, det1))
        # change a fused weight of the copy only: the copy's output moves, the original's does not
        w = clone.backbone.backbone.network[0][0].token_mixer.fc_v.weight
        w.mul_(1.5)
        det2, seg2 = clone(x, r)
        assert not torch.equal(seg2, seg1)
        det3, seg3 = model(x, r)
        assert torch.equal(seg3, seg0)
        w.div_(1.5)
        w2 = model.backbone.backbone.network[0][0].token_mixer.fc1.weight
        w2.add_(0.01)
        det4, seg4 = model(x, r)
        assert not torch.equal(seg4, seg0)


@pytest.mark.parametrize("phi,size,batch", [("nano", 128, 2), ("l", 128, 2), ("nano", 256, 2)])
def test_bf16_operand_mode_against_oracle(A, phi, size, batch):
    """model.compute_dtype = "bf16" (BASELINE configs "bf16 with MFMA conv path"): dense convs multiply bf16-rounded
    operands with fp32 accumulation (the kernels' rounding itself is pinned to 2e-5 in test_hip_ops.py).  The oracle
    applies the same rounding to the same layers (O.OPERAND_ROUND) and is teacher-forced with the kernels' Cluster
    assignments.  Rounding is a discontinuity too: fp32 and fp64 values that straddle a bf16 boundary round apart
    (2^-8 relative, once per ~4000 operands), and ~100 layers compound that to ~1e-2 -- the bound of this test; the
    gradients are compared in aggregate (relative L2 over all parameters, cosine)."""
    from tests.parity_bf16 import bf16_report
    m = build(A, phi, size, 21, True)
    rep = bf16_report(A, m, phi, batch, size, iseed=31)
    print(rep)
    assert rep["flips"] <= max(30, rep["points"] // 1000), rep
    assert rep["det_err"] < 4e-2 and rep["seg_err"] < 4e-2, rep
    assert rep["grad_cos"] > 0.97 and rep["grad_rel_l2"] < 0.25, rep


@pytest.mark.gpu
def test_forward_without_backward_frees_its_activations_at_once(A):
    """Inference / validation loops: a forward whose outputs are dropped holds no memory afterwards -- under
    torch.no_grad() nothing is recorded at all, and with autograd on the saved activations die with the autograd node
    (no reference cycle that only Python's cycle collector would break; it is switched off here)."""
    import gc
    net = A.EfficientVRNet(4, 9, "nano", img_size=256).cuda().train()
    x, r = A.synthetic_inputs(4, 256, 3, "cuda")
    det, seg = net(x, r)                            # workspaces, caches
    (seg.mean() + sum(d.mean() for d in det)).backward()
    del det, seg
    net.zero_grad(set_to_none=True)
    gc.collect()
    torch.cuda.synchronize()
    gc.disable()
    try:
        base = torch.cuda.memory_allocated()
        for mode in ("no_grad", "recorded, no backward", "backward"):
            for _ in range(3):
                if mode == "no_grad":
                    with torch.no_grad():
                        det, seg = net(x, r)
                    assert not seg.requires_grad
                else:
                    det, seg = net(x, r)
                    assert seg.requires_grad
                    if mode == "backward":
                        net.zero_grad(set_to_none=True)
                        (seg.mean() + sum(d.mean() for d in det)).backward()
                peak = torch.cuda.memory_allocated() - base
                del det, seg
                net.zero_grad(set_to_none=True)
                torch.cuda.synchronize()
                held = torch.cuda.memory_allocated() - base
                assert held <= max(1 << 20, peak // 50), (mode, held, peak)
    finally:
        gc.enable()


def test_section_stamps_follow_the_program_order(A):
    """model.debug_stamps (tools/debug/section_stamps.py): device-clock stamps at the forks, chain ends and joins of the program --
    the untraced timeline of the step.  On one stream they must be monotone; a chain starts after its fork and the join comes
    after both chain ends; outputs are unchanged by the stamps."""
    m = build(A, "nano", 128, 5, True)
    g = torch.Generator().manual_seed(3)
    x, r = torch.randn(2, 3, 128, 128, generator=g).cuda(), torch.randn(2, 4, 128, 128, generator=g).cuda()
    det0, seg0 = m(x, r)
    m.debug_stamps = True
    det, seg = m(x, r)
    (sum(d.square().mean() for d in det) + seg.square().mean()).backward()
    torch.cuda.synchronize()
    assert torch.equal(seg, seg0) and all(torch.equal(a, b) for a, b in zip(det, det0))
    names = m._stamp_names
    ticks = m._stamp_buf[:len(names)].cpu().tolist()
    t = dict(zip(names, ticks))         # (the chain stamps of rt.parallel repeat per section: the named ones below are unique)
    assert names[0] == "step start" and names[-1] == "backward joined"
    main = [v for n, v in zip(names, ticks) if not n.startswith(" ") and " A " not in n and " B " not in n]
    assert all(a <= b for a, b in zip(main, main[1:])), list(zip(names, ticks))
    for i in range(4):
        assert t[f"s{i} fork"] <= min(t[f"s{i} A start"], t[f"s{i} B start"])
        assert max(t[f"s{i} A end"], t[f"s{i} B end"]) <= t[f"s{i} join"]
    assert t["backward joined"] - t["step start"] > 0


@pytest.mark.parametrize("phi,hw,batch,training", [("nano", (128, 192), 2, True), ("nano", (192, 64), 3, False), ("s", (64, 128), 2, True)])
def test_rectangular_input_against_oracle(A, phi, hw, batch, training):
    """img_size = (H, W) with H != W (the reference builds fea_pos, the folds and the proposals per axis: vr_coc.py:160-189,
    400-412): every kernel that takes H and W separately -- region folds, adaptive pooling, bilinear gathers, patch embedding,
    the NCHW boundary -- against the fp64 oracle."""
    from tests.parity import compare_with_oracle
    m = A.EfficientVRNet(4, 9, phi, img_size=hw).cuda()
    A.randomize_state_dict(m.state_dict(), seed=7)
    m.train(training)
    rep = compare_with_oracle(m, batch, hw, iseed=13, check_grads=training, oracle_dtype=torch.float64)
    print(rep)
    assert rep["ok"], rep


def test_batch_one_training_raises_like_the_reference(A):
    """A training-mode forward of ONE sample: ASPP's global-pool branch puts a (1, C, 1, 1) tensor through a train-mode
    BatchNorm2d (coc_fpn_dual.py:97-101), which PyTorch rejects ("Expected more than 1 value per channel when training");
    the reference therefore cannot train at batch 1 and neither can this path -- same error text, no silent statistics.
    Evaluation at batch 1 works (net_tiny_128_eval)."""
    m = build(A, "nano", 64, 1, True)
    x, r = A.synthetic_inputs(1, 64, 2)
    with pytest.raises(RuntimeError, match="Expected more than 1 value per channel when training"):
        m(x.cuda(), r.cuda())
    m.eval()
    with torch.no_grad():
        det, seg = m(x.cuda(), r.cuda())
    assert seg.shape == (1, 9, 64, 64) and all(torch.isfinite(d).all() for d in det)


@pytest.mark.parametrize("phi,size", [("nano", 128), ("s", 128)])
def test_fused_upsample_has_the_bits_of_the_three_launch_form(A, phi, size):
    """CoCUpsample with BatchNorm + ReLU applied on the taps of the bilinear gather (model.fused_upsample, K11:
    vrnet_bn_relu_upsample_bilinear_f32) against conv -> BN apply -> upsample: the tap expression is the apply kernel's, so
    outputs, statistics and every gradient are EQUAL, not close."""
    x, r = A.synthetic_inputs(2, size, 4)
    x, r = x.cuda(), r.cuda()

    def run(fused):
        m = build(A, phi, size, 17, True)
        m.fused_upsample = fused
        det, seg = m(x, r)
        (sum((d * d).mean() for d in det) + (seg * seg).mean()).backward()
        torch.cuda.synchronize()
        return [d.detach() for d in det] + [seg.detach()], {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    (o1, g1), (o0, g0) = run(True), run(False)
    assert all(torch.equal(a, b) for a, b in zip(o1, o0))
    assert g1.keys() == g0.keys() and all(torch.equal(g1[k], g0[k]) for k in g0), [k for k in g0 if not torch.equal(g1[k], g0[k])][:5]


@pytest.mark.parametrize("phi,size,batch", [("nano", 128, 2), ("s", 128, 2)])
def test_segment_k_of_the_captured_backward_is_complete_when_graph_k_ends(A, phi, size, batch):
    """What N > 1 correctness rests on: the captured step of a data-parallel model is three hipGraphs, and the
    all-reduce of arena slice k is issued right behind graph k.  So when graph k has finished, EVERY gradient of slice k must
    have its final value -- a weight gradient that a later graph still writes (deferred past the cut, started early and joined
    late) would be averaged half-finished on every rank, silently.  One rank is enough to see it: replay the graphs one at a
    time without the collectives, snapshot slice k after graph k, compare with the arena after the whole step; and the arena of
    the cut step must agree with the gradients of the plain, uncut step."""
    import torch.distributed as dist
    from asy_vrnet_amd.graph import GraphedStep
    from asy_vrnet_amd.parallel import DataParallelVRNet

    def loss_of(det, seg):
        return sum((d * d).mean() for d in det) + (seg * seg).mean()
    own_group = not dist.is_initialized()
    if own_group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29593")
        dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        x, r = A.synthetic_inputs(batch, size, 3, "cuda")
        ref = build(A, phi, size, 7, True)
        sd0 = {k: v.clone() for k, v in ref.state_dict().items()}
        loss_of(*ref(x, r)).backward()
        m = build(A, phi, size, 7, True)
        net = DataParallelVRNet(m, force_collective=True)
        gs = GraphedStep(net, loss_of, batch, size, x.device)
        bk = m._grad_bucketer
        assert len(gs.graphs) == 3 and sorted(bk.segment_slices) == [0, 1, 2]
        for rep in range(2):
            m.load_state_dict(sd0)
            gs.x.copy_(x)
            gs.r.copy_(r)
            snaps = {}
            for k, g in enumerate(gs.graphs):
                g.replay()
                torch.cuda.synchronize()
                lo, hi = bk.segment_slices[k]
                assert hi > lo
                snaps[k] = bk.arena[lo:hi].clone()
            for k, snap in snaps.items():
                lo, hi = bk.segment_slices[k]
                assert torch.equal(bk.arena[lo:hi], snap), f"replay {rep}: a later graph still wrote gradients of segment {k}"
            for (k, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
                if p.numel() and q.grad is not None:
                    v = bk.view(p)      # (rounding-level: in the arena a gradient is accumulated into a zeroed view, and the
                    assert v is not None, k      #  flush at a cut changes which slab-reduce kernel finishes a layer-scale gradient)
                    assert ((v - q.grad).norm() / q.grad.norm().clamp_min(1e-20)).item() < 1e-5, k
    finally:
        if own_group:
            dist.destroy_process_group()
