"""Star phylogeny with gaps beyond 2^30 cells (the group-level scan with the slots' flags in the key word, span_scan.hip):
step time at FBG_STAR_ROWS x FBG_STAR_COLS (default 1000 x 2 000 000), and -- argument `check` -- the same f from the
record path (option span_scan = -1).  usage: python scripts/gpu_span_big.py [reps] [check]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import founderblockgraphs_amd as F

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
check = len(sys.argv) > 2 and sys.argv[2] == "check"
m, n = int(os.environ.get("FBG_STAR_ROWS", 1000)), int(os.environ.get("FBG_STAR_COLS", 2_000_000))
g = torch.Generator(device="cuda").manual_seed(7)
anc = torch.randint(0, 4, (n,), device="cuda", generator=g, dtype=torch.uint8)
d = torch.empty((m, n), dtype=torch.uint8, device="cuda")
lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")
step = max(1, 10_000_000 // n)
for i0 in range(0, m, step):
    i1 = min(m, i0 + step)
    mut = torch.rand((i1 - i0, n), device="cuda", generator=g) < 0.01
    sub = torch.randint(0, 4, (i1 - i0, n), device="cuda", generator=g, dtype=torch.uint8)
    d[i0:i1] = lut[torch.where(mut, sub, anc.expand(i1 - i0, n)).long()]
    start = (torch.rand((i1 - i0, n), device="cuda", generator=g) < 0.02 / 8).float().unsqueeze(1)
    gap = torch.nn.functional.max_pool1d(torch.nn.functional.pad(start, (7, 0)), 8, 1).squeeze(1) > 0
    d[i0:i1][gap] = ord("-")
    del mut, sub, start, gap
d = d.reshape(-1)
torch.cuda.empty_cache()
print(json.dumps({"rows": m, "columns": n, "cells": (m + 1) * (n + 1), "beyond_2_30": (m + 1) * (n + 1) >= 1 << 30}), flush=True)
with F.Engine(0) as eng:
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
        print(json.dumps({"ms": round(1e3 * dt, 2), "columns_per_s": round(n / dt), "blocks": blocks, "span": eng.get_option("span_scan_used"),
                          "key_flags": eng.get_option("span_key_flags_used"), "decline": eng.get_option("span_decline"), "alloc_calls": eng.get_option("alloc_calls"), "alloc_ms": eng.get_option("alloc_us") // 1000, "G": eng.get_option("span_groups"), "odd": eng.get_option("span_odd_groups"),
                          "slow": eng.get_option("span_slow_groups"), "dp_kind": eng.get_option("dp_kind"),
                          "stages": {k: round(v[0], 2) for k, v in eng.stage_ms().items() if v[0] > 0.05}, "f_sum": int(d_f.sum())}), flush=True)
    if check:
        f_span = d_f.clone()
        eng.set_option("span_scan", -1)
        d_f.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.index_build()
        eng.scan_f(0, n, d_f.data_ptr())
        eng.sync()
        dt = time.perf_counter() - t0
        bad = int((f_span != d_f).sum())
        print(json.dumps({"record_path_ms": round(1e3 * dt, 2), "span": eng.get_option("span_scan_used"), "index_kind": eng.get_option("index_kind"),
                          "columns_that_differ": bad}), flush=True)
        if bad:
            sys.exit(1)
