"""BASELINE config 4 (1000 x 8,000,000, 8.0e9 symbols) on ONE GPU through the group API: 8 key-range partitions worked off in
turn (two passes).  Prints wall time and the stage times of the last partition build; under rocprofv3 --kernel-trace --stats
the per-kernel times are what one of 8 GPUs would spend on its partition (scripts/gpu_prof_c4.sh)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
import founderblockgraphs_amd as F  # noqa: E402

m, n = 1000, int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 0
with F.Group([0]) as grp:
    if parts:
        grp.set_option("partitions", parts)
        grp.set_option("plan", 1)
    grp.msa_synthetic(m, n)
    for rep in range(2):
        t0 = time.perf_counter()
        grp.scan_f()
        dt = time.perf_counter() - t0
        eng = grp.member(0)
        print(json.dumps({"rows": m, "cols": n, "plan": grp.plan_used(), "wall_ms": round(1e3 * dt, 1),
                          "last_partition_stages_ms": {k: round(v[0], 2) for k, v in eng.stage_ms().items()},
                          "device_GB": round(eng.device_bytes() / 1e9, 1)}), flush=True)
