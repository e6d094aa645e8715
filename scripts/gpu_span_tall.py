"""The 2000 x 12000 star phylogeny with gaps of test_span_scan_through_a_group_and_at_scale (c), many times: how often, and
where, the group-level scan differs from the record path with the slow groups' odd members on 1 / 32 workgroups."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import founderblockgraphs_amd as F
from test_gpu_parity import star_msa
rng = np.random.default_rng(515)
star_msa(rng, 150, 9000, gap_cells=0.02, gap_run=8)
star_msa(rng, 400, 50000, gap_cells=0.02, gap_run=8)
tall = star_msa(rng, 2000, 12000, gap_cells=0.02, gap_run=8)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 15
with F.Engine(0) as e:
    ref = {}
    with e.options(span_scan=-1):
        for t in (False, True):
            ref[t] = e.elastic_f(tall, disable_efg_tricks=t)
    for split in (1, 32):
        e.set_option("span_slow_split", split)
        fails = 0
        for r in range(reps):
            for t in (False, True):
                a = e.elastic_f(tall, disable_efg_tricks=t)
                bad = np.flatnonzero(a != ref[t])
                if bad.size:
                    fails += 1
                    print("split", split, "rep", r, "tricks_off", t, "differ", bad.size, bad[:12].tolist(), a[bad[:12]].tolist(), ref[t][bad[:12]].tolist(), flush=True)
        print("split", split, "failures", fails, "of", 2 * reps, flush=True)
