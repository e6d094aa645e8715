"""Randomised parity campaign for the group-level scan on column spans (span_scan.hip; option span_scan = 1 takes it whatever
the input): star phylogenies and noisy copies with gap runs, deletions shared by many rows, rows that start late / end
early, ignore characters, small alphabets (repeats: stray suffixes in the groups), 1 .. 1300 rows (FBG_FUZZ_TALL=1: a
quarter of the cases with 2100 .. 6000 rows); f with the elastic
tricks on and off against the oracle.  Usage: gpu_fuzz_span.py SECONDS [SEED]; exits non-zero on the first mismatch."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import pyoracle as O
import founderblockgraphs_amd as F

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 31000
eng = F.Engine(0)
t0 = time.time()
last = t0
it = used = 0
while time.time() - t0 < budget:
    seed = seed0 + it
    it += 1
    if time.time() - last > 30:
        last = time.time()
        print(f"... {it} cases ({used} through the span scan), {time.time() - t0:.0f} s", flush=True)
    rng = np.random.default_rng(seed)
    m = int(rng.choice([1, 2, 3, 7, 20, 64, 65, 130, 300, 900, 1030, 1300]))
    n = int(rng.choice([1, 2, 9, 40, 150, 600, 2500])) if m > 300 else int(rng.choice([1, 5, 33, 200, 1000, 4000, 12000]))
    if os.environ.get("FBG_FUZZ_TALL") and rng.random() < 0.25:      # thousands of rows: the large instances of k_sp_odd_pairs
        m = int(rng.choice([2100, 3000, 4200, 6000]))
        n = int(rng.choice([1, 9, 40, 150, 400]))
    alphabet = str(rng.choice(["A", "AC", "ACGT", "ACGT", "ACGTN", "ACGTRYKM"]))
    if m > 2000 and alphabet == "A":                          # (thousands of rows of one letter, forced through the all-pairs kernel: a minute per case)
        alphabet = "AC"
    alpha = np.frombuffer(alphabet.encode(), dtype=np.uint8)
    anc = alpha[rng.integers(0, len(alpha), n)]
    if rng.random() < 0.3 and n >= 40:                        # a tandem repeat in the ancestor: strays in the groups
        u = int(rng.integers(2, 12)); a0 = int(rng.integers(0, n - 30)); anc[a0:a0 + 30] = np.resize(anc[a0:a0 + u], 30)
    msa = np.tile(anc, (m, 1))
    sub = float(rng.choice([0.0, 0.001, 0.01, 0.05, 0.3]))
    mut = rng.random((m, n)) < sub
    msa[mut] = alpha[rng.integers(0, len(alpha), int(mut.sum()))]
    gap_run = int(rng.choice([1, 3, 8, 30]))
    gap_cells = float(rng.choice([0.0, 0.005, 0.02, 0.1]))
    shared = float(rng.choice([0.0, 0.0, 0.3, 0.9]))
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
    eng.set_option("span_scan", 1)
    eng.set_option("msd_min", 1 if rng.random() < 0.3 else -1)
    tag = f"seed={seed} m={m} n={n} alphabet={alphabet} sub={sub} gaps={gap_cells}/{gap_run} shared={shared} ignore={ign!r}"
    if os.environ.get("FBG_FUZZ_TRACE"):
        print("case", tag, "msd_min", eng.get_option("msd_min"), flush=True)
    try:
        for tricks_off in (False, True):
            f = O.compute_f(msa, ignore=ign, disable_tricks=tricks_off)
            try:
                g = eng.elastic_f(msa, ignorechars=ign, disable_efg_tricks=tricks_off)
                bad = np.flatnonzero(g != f)
                assert bad.size == 0, f"f differs (tricks_off={tricks_off}) at {bad[:6].tolist()}: {g[bad[:6]].tolist()} vs {f[bad[:6]].tolist()}"
            except F.NoSegmentation:
                assert tricks_off and f[0] == n, "no segmentation"
            used += eng.get_option("span_scan_used")
    except Exception as ex:      # noqa: BLE001
        print("MISMATCH", tag, repr(ex), flush=True)
        sys.exit(1)
print(f"span fuzz ok: {it} cases ({used} index builds through the span scan) in {time.time() - t0:.0f} s (seeds {seed0}..{seed0 + it - 1})")
