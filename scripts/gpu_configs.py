#!/usr/bin/env python3
"""Runs the BASELINE configs other than the headline (and the harder 'star phylogeny' distribution of
SURVEY.md 8d) through the staged device API and prints per-stage times.  Not part of bench.py's contract;
used to fill the tables of DESIGN.md / BASELINE.md."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import founderblockgraphs_amd as F  # noqa: E402


def star_msa(m, n, p=0.01, seed=7):
    g = torch.Generator(device="cuda").manual_seed(seed)
    anc = torch.randint(0, 4, (n,), device="cuda", generator=g, dtype=torch.uint8)
    out = torch.empty((m, n), dtype=torch.uint8, device="cuda")
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")
    for i0 in range(0, m, 50):
        i1 = min(m, i0 + 50)
        mut = torch.rand((i1 - i0, n), device="cuda", generator=g) < p
        sub = torch.randint(0, 4, (i1 - i0, n), device="cuda", generator=g, dtype=torch.uint8)
        out[i0:i1] = lut[torch.where(mut, sub, anc.expand(i1 - i0, n)).long()]
    return out.reshape(-1)


def run(name, m, n, elastic=True, ignore="", gap_fraction=0.0, gap_run=0, n_fraction=0.0, star=False, reps=2, gapped=False, star_gaps=0.0):
    eng = F.Engine(0)
    for kv in filter(None, os.environ.get("FBG_OPTS", "").split(",")):      # debug options for every run: FBG_OPTS=key=value,key=value
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    st = torch.cuda.Stream()
    torch.cuda.set_stream(st)
    eng.set_stream(st.cuda_stream)
    if star:
        d = star_msa(m, n)
        if star_gaps > 0:        # gap runs of 8 on top: every cell starts one with probability star_gaps / 8 (rows keep resembling each other)
            g = torch.Generator(device="cuda").manual_seed(11)
            dd = d.view(m, n)
            for i0 in range(0, m, 50):
                i1 = min(m, i0 + 50)
                start = (torch.rand((i1 - i0, n), device="cuda", generator=g) < star_gaps / 8).to(torch.int32)
                run_ = torch.nn.functional.max_pool1d(torch.nn.functional.pad(start.float().unsqueeze(1), (7, 0)), 8, 1).squeeze(1) > 0
                dd[i0:i1][run_] = ord("-")
    else:
        d = torch.empty(m * n, dtype=torch.uint8, device="cuda")
        eng.msa_synthetic(d.data_ptr(), m, n, gap_fraction=gap_fraction, gap_run=gap_run, n_fraction=n_fraction)
    d_f = torch.zeros(n, dtype=torch.int64, device="cuda")
    d_b = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    best = None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.msa_set_device(d.data_ptr(), m, n)
        eng.index_build(reversed=not (elastic or gapped), ignorechars=ignore)
        d_f.zero_()
        if gapped:       # segment2elasticValid: non-elastic mode, rows with gaps
            eng.scan_gapped_v(d_f.data_ptr())
            blocks = eng.gapped_dp_device(d_f.data_ptr(), n, d_b.data_ptr())
        elif elastic:
            eng.scan_f(0, n, d_f.data_ptr())
            blocks = eng.minmax_dp_device(d_f.data_ptr(), n, d_b.data_ptr())
        else:
            eng.scan_v(0, n, d_f.data_ptr())
            blocks = eng.repeatfree_dp_device(d_f.data_ptr(), n, d_b.data_ptr())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        stages = {k: round(v[0], 3) for k, v in eng.stage_ms().items()}
        if best is None or dt < best[0]:
            best = (dt, stages, blocks)
    dt, stages, blocks = best
    ext = int((d_f - torch.arange(n, device="cuda")).max()) if elastic else None
    print(json.dumps({"config": name, "index_kind": eng.get_option("index_kind"), "rows": m, "cols": n, "ms": round(dt * 1e3, 2), "columns_per_s": round(n / dt),
                      "blocks": blocks, "max_extension": ext, "stages_ms": stages,
                      "device_GB": round(eng.device_bytes() / 1e9, 1)}), flush=True)
    eng.close()


if __name__ == "__main__":
    which = sys.argv[1:] or ["c2", "c5", "star"]
    if "c2" in which:
        run("C2 64x100k non-elastic", 64, 100_000, elastic=False)
    if "c5" in which:
        run("C5 256x2M gaps+N elastic ignore=N", 256, 2_000_000, ignore="N", gap_fraction=0.05, gap_run=16, n_fraction=0.001)
    if "c5gapped" in which:
        run("C5-shaped 256x2M gaps, non-elastic (segment2elasticValid)", 256, 2_000_000, elastic=False, gapped=True,
            gap_fraction=0.05, gap_run=16)
    if "c3gapped" in which:
        run("C3 1000x1M, non-elastic through segment2elasticValid", 1000, 1_000_000, elastic=False, gapped=True)
    if "star" in which:
        run("star phylogeny 1000x200k p=0.01", 1000, 200_000, star=True)
    if "stargaps" in which:
        run("star phylogeny 1000x200k p=0.01 with 2% gap cells (runs of 8)", 1000, 200_000, star=True, star_gaps=0.02)
    if "star1m" in which:
        run("star phylogeny 1000x1M p=0.01", 1000, 1_000_000, star=True, reps=1)
    if "c3" in which:
        run("C3 1000x1M elastic", 1000, 1_000_000)
