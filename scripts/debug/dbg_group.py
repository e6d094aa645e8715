import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
from conftest import random_msa
from oracle import pyoracle as O
import founderblockgraphs_amd as F
rng = np.random.default_rng(701)
with F.Group([0]) as grp:
    for (m, n, kw, ign) in [(24, 500, {}, ""), (50, 301, dict(alphabet="AC"), ""), (16, 400, dict(gap_p=0.02, gap_run=7), ""),
                            (30, 300, dict(gap_p=0.05, gap_run=3, n_p=0.02), "N"), (40, 600, dict(similar=0.97), ""),
                            (9, 1, {}, ""), (1, 40, {}, ""), (65, 257, dict(alphabet="ACGTN"), "N")]:
        msa = random_msa(rng, m, n, **kw)
        for tricks_off in (False, True):
            f = O.compute_f(msa, ignore=ign, disable_tricks=tricks_off)
            try:
                got = grp.elastic_f(msa, ignorechars=ign, disable_efg_tricks=tricks_off)
                print(m, n, kw, ign, tricks_off, "ok" if np.array_equal(got, f) else "MISMATCH", grp.plan_used(), flush=True)
            except Exception as e:
                print(m, n, kw, ign, tricks_off, "EXC", e, flush=True)
