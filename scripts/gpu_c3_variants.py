"""BASELINE config 3 index build + scan under several settings of ONE debug option, one after the other in one process
(for a kernel trace read in launch order, or plain wall times).  usage: python scripts/gpu_c3_variants.py KEY v1,v2,... [reps] [key=value ...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import founderblockgraphs_amd as F

key = sys.argv[1]
values = [int(v) for v in sys.argv[2].split(",")]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
m, n = 1000, 1_000_000
d = torch.empty(m * n, dtype=torch.uint8, device="cuda")
with F.Engine(0) as eng:
    for kv in sys.argv[4:]:
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    eng.msa_synthetic(d.data_ptr(), m, n, 0x5EED0001)
    d_f = torch.zeros(n, dtype=torch.int64, device="cuda")
    eng.msa_set_device(d.data_ptr(), m, n)
    for val in values:
        eng.set_option(key, val)
        for r in range(reps):
            d_f.zero_()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.index_build()
            eng.scan_f(0, n, d_f.data_ptr())
            eng.sync()
            print(json.dumps({key: val, "ms": round(1e3 * (time.perf_counter() - t0), 2), "index_kind": eng.get_option("index_kind"),
                              "stages": {k: round(v[0], 2) for k, v in eng.stage_ms().items()}, "f_sum": int(d_f.sum())}), flush=True)
