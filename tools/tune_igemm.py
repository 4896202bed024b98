"""Tile-config sweep for the igemm (tuning aid): runs tools/bench_kernels.py under VRNET_IGEMM_CFG=0,1,2
(128x128, 128x64, 64x64 tiles) in child processes and prints, per conv shape, the time of each config."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
res = {}
WG = "--wgrad" in args
BKS = "--bk" in args           # cfg0 = BK 16, cfg1 = BK 32
for f in ("--wgrad", "--bk"):
    if f in args:
        args.remove(f)
for cfg in ((0, 1) if (WG or BKS) else (0, 1, 2)):
    if BKS:
        env = dict(os.environ, VRNET_IGEMM_BK=str(16 if cfg == 0 else 32))
    else:
        env = dict(os.environ, **({"VRNET_WGRAD_CFG": str(cfg)} if WG else {"VRNET_IGEMM_CFG": str(cfg)}))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_kernels.py")] + args, env=env,
                         capture_output=True, text=True).stdout
    for line in out.splitlines():
        if line.startswith("#") or not line.strip() or "amdgpu" in line:
            continue
        f = line.split()
        key = " ".join(f[5:])
        if (f[5] == "2") != WG:
            continue
        res.setdefault(key, {})[cfg] = (float(f[2]), int(f[1]), float(f[3]))
rows = []
for key, d in res.items():
    if len(d) < (2 if (WG or BKS) else 3):
        continue
    d.setdefault(2, d[0])
    best = min(d, key=lambda c: d[c][0])
    rows.append((d[best][0] * d[best][1], key, d, best))
rows.sort(reverse=True)
tot = {c: sum(d[c][0] * d[c][1] for _, _, d, _ in rows) for c in (0, 1, 2)}
print("# totals us/step per forced config:", tot, "best-of:", sum(r[0] for r in rows))
print("# key = mode B H W Cin OH OW Cout k s d act res ypre aux nchw | us cfg0 cfg1 cfg2 | best | M N K")
for t, key, d, best in rows:
    f = key.split()
    mode, B, H, W, Ci, OH, OW, Co = map(int, f[:8])
    M = B * (OH * OW if mode == 0 else H * W)
    N, K = (Co, Ci) if mode == 0 else (Ci, Co)
    print(f"{key:60s} | {d[0][0]:8.1f} {d[1][0]:8.1f} {d[2][0]:8.1f} | {best} | M={M} N={N} K={K} x{d[0][1]}")
