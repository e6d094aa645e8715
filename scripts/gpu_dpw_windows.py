"""The wide-window sweep on made-up f[] (plateaus, as tests/test_gpu_parity.py builds them) for several largest extensions, once
with the matrices of rounds 2-3 (dpw_matrix = 1) and once without (0): wall time of fbg_minmax_dp, which window ran, equal
boundaries.  Under scripts/gpu_trace_order.sh the kernel times follow in launch order.  usage: gpu_dpw_windows.py [n]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import founderblockgraphs_amd as F


def plateau_f(rng, n, max_ext):
    x = np.arange(n, dtype=np.int64)
    ends = np.sort(rng.choice(np.arange(1, n), size=max(1, n // max(2, max_ext)), replace=False))
    nxt = ends[np.minimum(np.searchsorted(ends, x, side="left"), len(ends) - 1)]
    f = np.minimum(x + np.clip(nxt - x, 0, max_ext), n - 1)
    f[0] = 0
    return f


n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
with F.Engine(0) as eng:
    for max_ext in (900, 1800, 3500, 6000, 12000):
        f = plateau_f(np.random.default_rng(max_ext), n, max_ext)
        d_f = torch.from_numpy(f).cuda()
        res = {}
        for val in (1, 0, 1, 0):
            eng.set_option("dpw_matrix", val)
            d_b = torch.empty(n + 1, dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            cnt = eng.minmax_dp_device(d_f.data_ptr(), n, d_b.data_ptr())
            eng.sync()
            ms = 1e3 * (time.perf_counter() - t0)
            res[val] = (cnt, d_b[:cnt].clone())
            print("max_ext", max_ext, "dpw_matrix", val, "ms", round(ms, 2), "dp_kind", eng.get_option("dp_kind"), "blocks", cnt,
                  "stage", round(eng.stage_ms().get("dp", (0,))[0], 2), flush=True)
        assert res[0][0] == res[1][0] and torch.equal(res[0][1], res[1][1])
