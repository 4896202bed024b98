"""Diagnostic (needs libvrnet_stamp2.so = igemm.hip built with -DVR_IGEMM_STAMP2): per-workgroup timeline of one
LDS-DMA igemm launch -- prologue / main loop / epilogue durations, workgroups per CU over time, effective clock.

    python tools/stamp_wg.py B H W Cin Cout [mode]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import numpy as np
from asy_vrnet_amd import hip
B, H, W, Ci, Co = [int(v) for v in sys.argv[1:6]]
mode = int(sys.argv[6]) if len(sys.argv) > 6 else 0
precision = int(sys.argv[7]) if len(sys.argv) > 7 else 2
x = torch.randn(B, H, W, Ci if mode == 0 else Co, device="cuda")
w = torch.randn(Co, Ci, 1, 1, device="cuda") * 0.05
y = torch.empty(B, H, W, Co if mode == 0 else Ci, device="cuda")
M = B * H * W
nwg = 8 * ((M // 64 + 7) // 8) * ((y.shape[-1] + 63) // 64)
st = torch.zeros(8 * nwg + 64, dtype=torch.float64, device="cuda")
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(3):
    st.zero_()
    ev0.record()
    hip.conv2d(x, x.shape[-1], w, None, y, y.shape[-1], B, H, W, Ci, H, W, Co, 1, 1, 1, 0, 1, mode=mode, stats=st, precision=precision)
    ev1.record()
torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1)
t = st.view(torch.int64)[:8 * nwg].view(nwg, 8).cpu().numpy()
t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
start, first, loop_end, end = [(t[:, i] - t0) for i in range(4)]
span = end.max()
flops = 2.0 * M * Ci * Co
# s_memtime ticks at 100 MHz (constant); compare with the event time
print(f"launch {ms*1e3:.1f} us by events; span {span} memtime ticks -> {span/100.0:.1f} us at 100 MHz; {flops/ms/1e9:.1f} TF/s; {len(t)} workgroups")
print(f"prologue (start -> first stage landed): mean {np.mean(first-start)/100:.2f} us  p90 {np.percentile(first-start,90)/100:.2f}")
print(f"main loop: mean {np.mean(loop_end-first)/100:.2f} us  min {np.min(loop_end-first)/100:.2f}  max {np.max(loop_end-first)/100:.2f}")
print(f"epilogue: mean {np.mean(end-loop_end)/100:.2f} us  p90 {np.percentile(end-loop_end,90)/100:.2f}")
hw = t[:, 4]
cu = (hw >> 8) & 0xF
se = (hw >> 13) & 0x7        # gfx9 HW_ID: wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]
sh = (hw >> 12) & 0x1
key = se * 32 + sh * 16 + cu
print("distinct (se,sh,cu) ids seen:", len(set(key.tolist())), "(per XCD; 8 XCDs alias)")
# residency over time: number of workgroups alive
ev = sorted([(s, 1) for s in start] + [(e, -1) for e in end])
live, last, area = 0, 0, 0.0
hist = {}
for tt, d in ev:
    hist[live] = hist.get(live, 0) + (tt - last)
    area += live * (tt - last)
    last = tt
    live += d
print(f"mean workgroups resident: {area/span:.1f} (of 768 slots)")
order = np.argsort(start)
print("start times of workgroups at rank 0, 25%, 50%, 75%, 100% (ticks):", np.round(start[order][[0, len(t)//4, len(t)//2, 3*len(t)//4, len(t)-1]], 1))
stage = (loop_end - first) / (Ci / 32)
print(f"per-stage time inside the loop: mean {np.mean(stage)/100*1e3:.0f} ns  (16 MFMAs = {1024/2.4:.0f} ns at 2.4 GHz)")
