"""Per-rank cost of the key-range partitioned index at weak-scaled sizes, on ONE GPU: the P partitions of a
P-GPU job are played one after the other by P contexts (as tests/test_gpu_parity.py::_partitioned does), so each
context's stage timers show what its rank would spend on a GPU of its own.  Usage: gpu_part_sim.py P [cols_per_gpu]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import founderblockgraphs_amd as F
from founderblockgraphs_amd._lib import PART_HALO_BYTES

P = int(sys.argv[1])
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
m, n = 1000, cols * P
d = torch.empty(m * n, dtype=torch.uint8, device="cuda")
engines = [F.Engine() for _ in range(P)]
engines[0].msa_synthetic(d.data_ptr(), m, n)
blobs = torch.zeros(P * PART_HALO_BYTES, dtype=torch.uint8, device="cuda")
gm = [torch.zeros(n + 1, dtype=torch.int32, device="cuda") for _ in range(P)]
torch.cuda.synchronize()
for it in range(2):
    wall = [0.0] * P
    for r, e in enumerate(engines):
        e.msa_set_device(d.data_ptr(), m, n)
        t = time.perf_counter()
        ok = e.part_index_build(r, P, blobs.data_ptr() + r * PART_HALO_BYTES)
        e.sync()
        wall[r] += time.perf_counter() - t
        assert ok
    for r, e in enumerate(engines):
        t = time.perf_counter()
        ok = e.part_scan(blobs.data_ptr(), gm[r].data_ptr())
        e.sync()
        wall[r] += time.perf_counter() - t
        assert ok
    red = gm[0]
    for g in gm[1:]:
        red = torch.maximum(red, g)
    torch.cuda.synchronize()
    for r, e in enumerate(engines):
        assert e.part_finish(red.data_ptr())
    d_f = torch.zeros(n, dtype=torch.int64, device="cuda")
    d_b = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    t = time.perf_counter()
    engines[0].scan_f(0, n, d_f.data_ptr())
    blocks = engines[0].minmax_dp_device(d_f.data_ptr(), n, d_b.data_ptr())
    engines[0].sync()
    tail = time.perf_counter() - t
for r, e in enumerate(engines):
    print(json.dumps({"part": r, "wall_ms": 1e3 * wall[r], "stages": {k: round(v[0], 2) for k, v in e.stage_ms().items()},
                      "device_GB": e.device_bytes() / 1e9}))
print(json.dumps({"P": P, "n": n, "blocks": blocks, "rank0_tail_ms": 1e3 * tail,
                  "est_step_ms": 1e3 * (max(wall) + tail), "est_cols_per_s": n / (max(wall) + tail)}))
