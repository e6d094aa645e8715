"""Star phylogeny 1000 x 200000 with 2 % gap cells in runs of 8 (bench.py's other_workloads[1]) alone: step time, stages,
which scan ran.  usage: python scripts/gpu_stargaps.py [reps] [span_scan option] [key=value ...]
FBG_STAR_ROWS / FBG_STAR_COLS in the environment: another shape (e.g. more than 1024 rows: the big-group kernels)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import founderblockgraphs_amd as F

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
opt = int(sys.argv[2]) if len(sys.argv) > 2 else 0
m, n = int(os.environ.get("FBG_STAR_ROWS", 1000)), int(os.environ.get("FBG_STAR_COLS", 200_000))
g = torch.Generator(device="cuda").manual_seed(7)
anc = torch.randint(0, 4, (n,), device="cuda", generator=g, dtype=torch.uint8)
d = torch.empty((m, n), dtype=torch.uint8, device="cuda")
lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")
for i0 in range(0, m, 50):
    i1 = min(m, i0 + 50)
    mut = torch.rand((i1 - i0, n), device="cuda", generator=g) < 0.01
    sub = torch.randint(0, 4, (i1 - i0, n), device="cuda", generator=g, dtype=torch.uint8)
    d[i0:i1] = lut[torch.where(mut, sub, anc.expand(i1 - i0, n)).long()]
for i0 in range(0, m, 50):
    i1 = min(m, i0 + 50)
    start = (torch.rand((i1 - i0, n), device="cuda", generator=g) < 0.02 / 8).float().unsqueeze(1)
    gap = torch.nn.functional.max_pool1d(torch.nn.functional.pad(start, (7, 0)), 8, 1).squeeze(1) > 0
    d[i0:i1][gap] = ord("-")
# FBG_STAR_SHARED=d: a part of the gap runs is shared -- at one column in 2000 (FBG_STAR_SHARED_RATE) a run of 8 starts in a fraction d of the rows at once
# (deletions that hundreds of rows have in common: the groups whose odd members are compared pair by pair, k_sp_odd_slow)
shared = float(os.environ.get("FBG_STAR_SHARED", 0))
if shared > 0:
    ev = (torch.rand((n,), device="cuda", generator=g) < float(os.environ.get("FBG_STAR_SHARED_RATE", 1 / 2000))).unsqueeze(0)
    for i0 in range(0, m, 50):
        i1 = min(m, i0 + 50)
        start = ((torch.rand((i1 - i0, n), device="cuda", generator=g) < shared) & ev).float().unsqueeze(1)
        gap = torch.nn.functional.max_pool1d(torch.nn.functional.pad(start, (7, 0)), 8, 1).squeeze(1) > 0
        d[i0:i1][gap] = ord("-")
d = d.reshape(-1)
with F.Engine(0) as eng:
    eng.set_option("span_scan", opt)
    for kv in sys.argv[3:]:                                   # further debug options: key=value
        k, val = kv.split("=")
        eng.set_option(k, int(val))
    d_f = torch.zeros(n, dtype=torch.int64, device="cuda")
    d_b = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    eng.msa_set_device(d.data_ptr(), m, n)
    for r in range(reps):
        d_f.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.index_build()
        eng.scan_f(0, n, d_f.data_ptr())
        blocks = eng.minmax_dp_device(d_f.data_ptr(), n, d_b.data_ptr())
        eng.sync()
        dt = time.perf_counter() - t0
        print(json.dumps({"ms": round(1e3 * dt, 2), "blocks": blocks, "index_kind": eng.get_option("index_kind"),
                          "span": eng.get_option("span_scan_used"), "work": eng.get_option("span_scan_work"), "G": eng.get_option("span_groups"), "odd": eng.get_option("span_odd_groups"), "irr": eng.get_option("span_irregular"), "chain": eng.get_option("span_chain"), "slow": eng.get_option("span_slow_groups"), "decline": eng.get_option("span_decline"), "dp_kind": eng.get_option("dp_kind"),
                          "stages": {k: round(v[0], 2) for k, v in eng.stage_ms().items()},
                          "f_sum": int(d_f.sum())}), flush=True)
        if os.environ.get("FBG_PHASES"):                      # a library built with -DSP_PHASE_TIMERS: cycles of k_sp_odd_pairs<28,64> per phase
            print("phases", [eng.get_option(f"span_dbg{k}") for k in range(7)], flush=True)
