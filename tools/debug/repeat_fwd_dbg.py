"""Debug: forward-only repeatability (train mode, no backward), many runs.  python tools/debug/repeat_fwd_dbg.py dtype runs [grad 0/1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import asy_vrnet_amd as A
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 100
record = int(sys.argv[3]) if len(sys.argv) > 3 else 0
net = A.EfficientVRNet(4, 9, "l", img_size=512).cuda().train()
A.randomize_state_dict(net.state_dict(), seed=2)
net.compute_dtype = dtype
x, r = A.synthetic_inputs(16, 512, 11, "cuda")
state = {k: v.clone() for k, v in net.state_dict().items()}
ref = None
nbad = 0
for it in range(runs):
    net.load_state_dict(state)
    if record:
        det, seg = net(x, r)
    else:
        with torch.no_grad():
            det, seg = net(x, r)
    torch.cuda.synchronize()
    cur = [seg.detach().clone()] + [d.detach().clone() for d in det]
    del det, seg
    if ref is None:
        ref = cur
        continue
    bad = [i for i in range(4) if not torch.equal(cur[i], ref[i])]
    if bad:
        nbad += 1
        print(f"run {it}: outputs {bad} differ, max rel {[float((cur[i] - ref[i]).abs().max() / ref[i].abs().max()) for i in bad]}")
print(f"{nbad} of {runs - 1} runs differ (record={record})")
