"""Rank-order scan of gapped MSAs (gapped_rank.hip) against the oracle on a spread of small cases; prints which index
form each case got and the first mismatching columns."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import random_msa
from oracle import pyoracle as O
import founderblockgraphs_amd as F

eng = F.Engine(0)
bad = 0
kinds = {}
stats = {}
t0 = time.time()
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for seed in range(seeds):
    rng = np.random.default_rng(seed)
    m = int(rng.choice([1, 2, 3, 5, 17, 64, 130]))
    n = int(rng.choice([2, 7, 33, 100, 257, 1000, 3000]))
    alphabet = str(rng.choice(["AC", "ACGT", "ACGTN", "ACGTRYKM"]))
    kw = {}
    if rng.random() < 0.4:
        kw["similar"] = float(rng.choice([0.5, 0.9, 0.99]))
    if rng.random() < 0.85:
        kw["gap_p"] = float(rng.choice([0.005, 0.05, 0.3])); kw["gap_run"] = int(rng.choice([1, 4, 20]))
    if rng.random() < 0.5:
        kw["n_p"] = 0.02
    msa = random_msa(rng, m, n, alphabet=alphabet, **kw)
    if rng.random() < 0.4 and n >= 33:               # long gap runs: rows that start late, end early, or skip a stretch
        for i in rng.choice(m, size=max(1, m // 3), replace=False):
            kind = rng.integers(3)
            w = int(rng.integers(1, max(2, n // 2)))
            if kind == 0:
                msa[i, :w] = ord("-")
            elif kind == 1:
                msa[i, n - w:] = ord("-")
            else:
                a0 = int(rng.integers(0, n - w))
                msa[i, a0:a0 + w] = ord("-")
    if (msa != ord("-")).sum(axis=1).min() == 0:
        continue
    ign = "N" if rng.random() < 0.6 else ""
    if "gap_p" not in kw and not ign:
        ign = "N"
    eng.set_option("gapped_rank", int(rng.choice([0, 2, 3, 4, 4])))
    for off in (False, True):
        f = O.compute_f(msa, ignore=ign, disable_tricks=off)
        try:
            g = eng.elastic_f(msa, ignorechars=ign, disable_efg_tricks=off)
        except F.NoSegmentation:
            g = None
        kind = eng.get_option("index_kind")
        kinds[kind] = kinds.get(kind, 0) + 1
        if kind == 2:
            thr = eng.get_option("grs_threshold")
            stats["thresholded"] = stats.get("thresholded", 0) + (thr > 1)
            stats["redone_cols"] = stats.get("redone_cols", 0) + eng.get_option("grs_redone")
        if g is None:
            okay = f[0] == n
        else:
            okay = np.array_equal(g, f)
        if not okay:
            bad += 1
            d = np.flatnonzero(g != f)[:6] if g is not None else []
            print(f"MISMATCH seed={seed} m={m} n={n} alphabet={alphabet} kw={kw} ign={ign!r} tricks_off={off} kind={kind} opt={eng.get_option('gapped_rank')} thr={eng.get_option('grs_threshold')} cols={list(d)} "
                  f"got={[int(g[x]) for x in d]} want={[int(f[x]) for x in d]}", flush=True)
            if bad > 12:
                sys.exit(1)
print(f"stats={stats} kinds={kinds} bad={bad} in {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
