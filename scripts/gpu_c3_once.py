"""BASELINE config 3 (1000 x 1,000,000) index build + scan, a few times, with debug options from the command line
(KEY=VALUE ...): the thing to put under rocprofv3.  usage: python scripts/gpu_c3_once.py [reps] [key=value ...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import founderblockgraphs_amd as F

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
m, n = 1000, 1_000_000
d = torch.empty(m * n, dtype=torch.uint8, device="cuda")
with F.Engine(0) as eng:
    for kv in sys.argv[2:]:
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    eng.msa_synthetic(d.data_ptr(), m, n, 0x5EED0001)
    d_f = torch.zeros(n, dtype=torch.int64, device="cuda")
    eng.msa_set_device(d.data_ptr(), m, n)
    for r in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.index_build()
        eng.scan_f(0, n, d_f.data_ptr())
        eng.sync()
        print(json.dumps({"ms": round(1e3 * (time.perf_counter() - t0), 2), "index_kind": eng.get_option("index_kind"), "msd_decline": eng.get_option("msd_decline"),
                          "stages": {k: round(v[0], 2) for k, v in eng.stage_ms().items()}, "f_sum": int(d_f.sum())}), flush=True)
