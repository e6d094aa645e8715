"""What a process pays before its first segmentation: loading the library, creating a context, the first call (code objects, workspaces)
and the second, on a small MSA and on the C3 shape.  usage: python scripts/gpu_first_call.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
t0 = time.perf_counter()
import numpy as np
import founderblockgraphs_amd as F
t1 = time.perf_counter()
eng = F.Engine(0)
t2 = time.perf_counter()
rng = np.random.default_rng(1)
small = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=(50, 2000))
ms = []
for _ in range(3):
    t = time.perf_counter()
    f = eng.elastic_f(small)
    eng.minmax_dp(f)
    ms.append(round(1e3 * (time.perf_counter() - t), 2))
print("import", round(1e3 * (t1 - t0)), "ms; context", round(1e3 * (t2 - t1)), "ms; 50 x 2000 MSA, calls 1-3:", ms, flush=True)
big = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=(1000, 1_000_000))
ms = []
for _ in range(3):
    t = time.perf_counter()
    f = eng.elastic_f(big)
    eng.minmax_dp(f)
    ms.append(round(1e3 * (time.perf_counter() - t), 2))
print("1000 x 1 000 000 MSA from pageable memory, calls 1-3:", ms, flush=True)
