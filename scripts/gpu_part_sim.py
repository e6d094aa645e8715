"""Per-rank cost of the key-range partitioned index at weak-scaled sizes, on ONE GPU.  The P partitions of a P-GPU job
are played one after the other; each partition's stage timers show what its rank would spend on a GPU of its own.
Default: P contexts (as tests/test_gpu_parity.py::_partitioned does).  --sequential: one context only, each
partition built twice (once for its edge slots, once more before its scan) -- for sizes where P contexts do not
fit one GPU's memory.  --c5: BASELINE config 5 instead (256 x 2,000,000 with gaps and N, --ignore-chars=N; the same MSA
whatever P: strong scaling).  Usage: gpu_part_sim.py P [cols_per_gpu] [--sequential] [--c5]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import founderblockgraphs_amd as F
from founderblockgraphs_amd._lib import PART_HALO_BYTES

args = [a for a in sys.argv[1:] if not a.startswith("--")]
sequential = "--sequential" in sys.argv
P = int(args[0])
cols = int(args[1]) if len(args) > 1 else 1_000_000
c5 = "--c5" in sys.argv
m, n = (256, 2_000_000) if c5 else (1000, cols * P)
ign = "N" if c5 else ""
d = torch.empty(m * n, dtype=torch.uint8, device="cuda")
engines = [F.Engine() for _ in range(1 if sequential else P)]
if c5:
    engines[0].msa_synthetic(d.data_ptr(), m, n, gap_fraction=0.05, gap_run=16, n_fraction=0.001)
else:
    engines[0].msa_synthetic(d.data_ptr(), m, n)
blobs = torch.zeros(P * PART_HALO_BYTES, dtype=torch.uint8, device="cuda")
red = torch.zeros(n + 1, dtype=torch.int32, device="cuda")
gm = torch.zeros(n + 1, dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
rows = []
for it in range(1 if sequential else 2):
    wall = [0.0] * P
    stages = [None] * P
    red.zero_()
    if sequential:
        e = engines[0]
        e.msa_set_device(d.data_ptr(), m, n)
        for r in range(P):
            assert e.part_index_build(r, P, blobs.data_ptr() + r * PART_HALO_BYTES, ignorechars=ign)
            e.sync()
        for r in range(P):
            t = time.perf_counter()
            assert e.part_index_build(r, P, blobs.data_ptr() + r * PART_HALO_BYTES, ignorechars=ign)
            assert e.part_scan(blobs.data_ptr(), gm.data_ptr())
            e.sync()
            wall[r] = time.perf_counter() - t
            stages[r] = {k: round(v[0], 2) for k, v in e.stage_ms().items()}
            torch.maximum(red, gm, out=red)
            torch.cuda.synchronize()
    else:
        for r, e in enumerate(engines):
            e.msa_set_device(d.data_ptr(), m, n)
            t = time.perf_counter()
            assert e.part_index_build(r, P, blobs.data_ptr() + r * PART_HALO_BYTES, ignorechars=ign)
            e.sync()
            wall[r] += time.perf_counter() - t
        for r, e in enumerate(engines):
            t = time.perf_counter()
            assert e.part_scan(blobs.data_ptr(), gm.data_ptr())
            e.sync()
            wall[r] += time.perf_counter() - t
            stages[r] = {k: round(v[0], 2) for k, v in e.stage_ms().items()}
            torch.maximum(red, gm, out=red)
            torch.cuda.synchronize()
    for e in engines:
        assert e.part_finish(red.data_ptr()) == 1, "a threshold was not cleared: this script does not model the re-scan"
    d_f = torch.zeros(n, dtype=torch.int64, device="cuda")
    d_b = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    for _ in range(2):                                 # the second time: buffers exist
        d_f.zero_()
        torch.cuda.synchronize()
        t = time.perf_counter()
        engines[0].scan_f(0, n, d_f.data_ptr())
        blocks = engines[0].minmax_dp_device(d_f.data_ptr(), n, d_b.data_ptr())
        engines[0].sync()
        tail = time.perf_counter() - t
for r in range(P):
    print(json.dumps({"part": r, "wall_ms": 1e3 * wall[r], "stages": stages[r]}))
print(json.dumps({"config": "C5 256 x 2,000,000 gaps + N" if c5 else "1000 rows, 10^6 columns per partition", "P": P, "n": n, "text": m * (n + 1) + 1, "sequential": sequential, "blocks": blocks, "rank0_tail_ms": 1e3 * tail,
                  "device_GB": engines[0].device_bytes() / 1e9,
                  "est_step_ms": 1e3 * (max(wall) + tail), "est_cols_per_s": n / (max(wall) + tail),
                  # bench.py runs the sweep of a step on a second stream beside the next step's index build
                  "est_step_ms_sweep_overlapped": 1e3 * max(wall), "est_cols_per_s_sweep_overlapped": n / max(wall)}))
