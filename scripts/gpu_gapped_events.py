import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import founderblockgraphs_amd as F
eng = F.Engine(0)
m, n = 256, 400000
d = torch.empty(m * n, dtype=torch.uint8, device="cuda")
eng.msa_synthetic(d.data_ptr(), m, n, gap_fraction=0.05, gap_run=16)
msa = d.cpu().numpy().reshape(m, n)
v = eng.gapped_v(msa)
s, prev, b = eng.gapped_dp(v)
ev = np.flatnonzero(prev[1:] != prev[:-1]) + 1
print("columns", n, "events", len(ev), "per 64-tile", len(ev) / (n / 64), "blocks", None if b is None else len(b))
tiles = np.bincount(ev // 64, minlength=n // 64 + 1)
print("tile events: mean", tiles.mean(), "max", tiles.max(), "hist", np.bincount(np.minimum(tiles, 40))[:41].tolist())
L = np.arange(n) - v.astype(np.int64) + 1
print("min block len: mean", L[L > 0].mean(), "score", int(s[-1]))
