"""Times single cases of scripts/gpu_fuzz_span.py (same generator) on the GPU and checks them against the oracle.
usage: python scripts/gpu_case_time.py SEED [SEED ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import pyoracle as O
import founderblockgraphs_amd as F


def case(seed):
    rng = np.random.default_rng(seed)
    m = int(rng.choice([1, 2, 3, 7, 20, 64, 65, 130, 300, 900, 1030, 1300]))
    n = int(rng.choice([1, 2, 9, 40, 150, 600, 2500])) if m > 300 else int(rng.choice([1, 5, 33, 200, 1000, 4000, 12000]))
    alphabet = str(rng.choice(["A", "AC", "ACGT", "ACGT", "ACGTN", "ACGTRYKM"]))
    alpha = np.frombuffer(alphabet.encode(), dtype=np.uint8)
    anc = alpha[rng.integers(0, len(alpha), n)]
    if rng.random() < 0.3 and n >= 40:
        u = int(rng.integers(2, 12)); a0 = int(rng.integers(0, n - 30)); anc[a0:a0 + 30] = np.resize(anc[a0:a0 + u], 30)
    msa = np.tile(anc, (m, 1))
    sub = float(rng.choice([0.0, 0.001, 0.01, 0.05, 0.3]))
    mut = rng.random((m, n)) < sub
    msa[mut] = alpha[rng.integers(0, len(alpha), int(mut.sum()))]
    gap_run = int(rng.choice([1, 3, 8, 30])); gap_cells = float(rng.choice([0.0, 0.005, 0.02, 0.1])); shared = float(rng.choice([0.0, 0.0, 0.3, 0.9]))
    if gap_cells > 0:
        for i, j in np.argwhere(rng.random((m, n)) < gap_cells / gap_run):
            msa[i, j:j + gap_run] = ord("-")
            if shared and rng.random() < shared:
                msa[rng.random(m) < float(rng.choice([0.05, 0.4, 0.9])), j:j + gap_run] = ord("-")
    if n >= 9 and rng.random() < 0.4:
        for i in range(m):
            if rng.random() < 0.3:
                msa[i, :int(rng.integers(1, n // 2 + 1))] = ord("-")
            if rng.random() < 0.3:
                msa[i, n - int(rng.integers(1, n // 2 + 1)):] = ord("-")
    ign = "N" if (alphabet == "ACGTN" and rng.random() < 0.6) else ""
    return msa, ign, (1 if rng.random() < 0.3 else -1)


with F.Engine(0) as eng:
    for seed in map(int, sys.argv[1:]):
        msa, ign, msd_min = case(seed)
        eng.set_option("span_scan", 1)
        eng.set_option("msd_min", msd_min)
        for tricks_off in (False, True):
            f = O.compute_f(msa, ignore=ign, disable_tricks=tricks_off)
            t = time.time()
            try:
                g = eng.elastic_f(msa, ignorechars=ign, disable_efg_tricks=tricks_off)
                ok = bool((g == f).all())
            except F.NoSegmentation:
                ok = bool(tricks_off and f[0] == msa.shape[1])
            print(seed, msa.shape, "tricks_off", tricks_off, "ok", ok, "s", round(time.time() - t, 2), "span", eng.get_option("span_scan_used"),
                  "odd", eng.get_option("span_odd_groups"), "slow", eng.get_option("span_slow_groups"), "chain", eng.get_option("span_chain"),
                  {k: round(v[0], 1) for k, v in eng.stage_ms().items() if v[0] > 1}, flush=True)
