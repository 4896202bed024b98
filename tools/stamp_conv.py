"""Diagnostic (needs a -DVR_IGEMM_STAMP build of igemm.hip): per-stage s_memtime stamps of one wave of the DMA kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asy_vrnet_amd import hip
B, H, W, Ci, Co = [int(v) for v in sys.argv[1:6]]
precision = int(sys.argv[6]) if len(sys.argv) > 6 else 2
BK = int(sys.argv[7]) if len(sys.argv) > 7 else 16
x = torch.randn(B, H, W, Ci, device="cuda"); w = torch.randn(Co, Ci, 1, 1, device="cuda") * 0.05
y = torch.empty(B, H, W, Co, device="cuda")
st = torch.zeros(max(4096, (B * H * W // 32) * ((Co + 31) // 32) * 2), dtype=torch.float64, device="cuda")
for _ in range(3):
    hip.conv2d(x, Ci, w, None, y, Co, B, H, W, Ci, H, W, Co, 1, 1, 1, 0, 1, mode=0, stats=st, precision=precision)
torch.cuda.synchronize()
t = st.view(torch.int64)[:256].view(64, 4).cpu()
n = min(64, Ci // BK)
top, after_bar, frags, end = t[:n, 3], t[:n, 0], t[:n, 1], t[:n, 2]
print("stage: wait+barrier | frag reads | mfma+dma issue | total   (s_memtime ticks)")
for s in range(2, min(n, 18)):
    print(f"{s:3d}: {int(after_bar[s]-top[s]):6d} {int(frags[s]-after_bar[s]):6d} {int(end[s]-frags[s]):6d} {int(top[s+1]-top[s]) if s+1<n else 0:6d}")
print("mean stage time", float((top[n-1]-top[2]))/(n-3))
