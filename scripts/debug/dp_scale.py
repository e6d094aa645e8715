import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import founderblockgraphs_amd as F
eng = F.Engine(0)
for n in (1_000_000, 4_000_000, 8_000_000):
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.arange(n, device="cuda")
    f = torch.minimum(x + torch.randint(14, 26, (n,), device="cuda", generator=g), torch.tensor(n - 1, device="cuda"))
    f[0] = 0
    d_b = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    for it in range(2):
        torch.cuda.synchronize()
        t = time.perf_counter()
        cnt = eng.minmax_dp_device(f.data_ptr(), n, d_b.data_ptr())
        eng.sync()
        dt = time.perf_counter() - t
    print(n, cnt, round(dt * 1e3, 2), "ms", {k: round(v[0], 2) for k, v in eng.stage_ms().items() if k == "dp"}, flush=True)
