"""Randomised parity campaign for the sweep alone (fbg_minmax_dp against the oracle's minmax_dp): random f[] of every
shape the scans can produce -- extensions from 0 to beyond the widest window, plateaus (long repeats), rare long
extensions, f[0] > 0 -- so that every form of the sweep (byte matrices, 16-bit matrices over 1024 / 2048 / 4096 columns
with their source pruning, wave, literal) is met thousands of times.  Usage: gpu_fuzz_dp.py SECONDS [SEED]."""
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
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 4242
eng = F.Engine(0)
t0 = time.time()
last = t0
it = 0
kinds = {}
while time.time() - t0 < budget:
    seed = seed0 + it
    it += 1
    if time.time() - last > 30:
        last = time.time()
        print(f"... {it} cases, {time.time() - t0:.0f} s, sweeps used {kinds}", flush=True)
    rng = np.random.default_rng(seed)
    n = int(rng.choice([1, 2, 100, 255, 256, 257, 1000, 5000, 20000, 60000]))
    max_ext = int(rng.choice([0, 3, 40, 120, 250, 254, 255, 256, 400, 900, 1020, 1023, 1500, 2044, 2047, 3000, 4092, 4095, 6000]))
    x = np.arange(n, dtype=np.int64)
    style = int(rng.integers(5))
    if style == 0 or n < 4:
        ext = rng.integers(0, max_ext + 1, n)
    elif style == 1:                  # plateaus: many columns share one right end
        ends = np.sort(rng.choice(np.arange(1, n), size=max(1, n // max(2, max_ext)), replace=False))
        ext = np.clip(ends[np.minimum(np.searchsorted(ends, x, side="left"), len(ends) - 1)] - x, 0, max_ext)
    elif style == 2:                  # mostly tiny, rarely long
        ext = np.where(rng.random(n) < 0.03, rng.integers(0, max_ext + 1, n), rng.integers(0, 3, n))
    elif style == 3:                  # stretches of long extensions between stretches of short ones
        seg = (x // max(1, int(rng.integers(50, 3000)))) % 2 == 0
        ext = np.where(seg, rng.integers(max_ext // 2, max_ext + 1, n), rng.integers(0, 8, n))
    else:                             # every extension near the maximum: block lengths near 2 * max_ext
        ext = rng.integers(max(0, max_ext - 3), max_ext + 1, n)
    f = np.minimum(x + ext, n - 1)
    f[0] = 0 if rng.random() < 0.8 else min(n - 1, int(rng.integers(0, max_ext + 2)))
    f = f.astype(np.uint64)
    tag = f"seed={seed} n={n} max_ext={max_ext} style={style} f0={int(f[0])}"
    try:
        mml, bt, b = O.minmax_dp(f)
        gb, gmml, gbt = eng.minmax_dp(f, full=True)
        assert np.array_equal(gmml, mml), "minmaxlength"
        assert np.array_equal(gbt, bt), "backtrack"
        assert np.array_equal(gb, b), "boundaries"
        k = eng.get_option("dp_kind")
        kinds[k] = kinds.get(k, 0) + 1
    except AssertionError as e:
        print(f"MISMATCH ({e}) {tag} dp_kind={eng.get_option('dp_kind')}", flush=True)
        sys.exit(1)
print(f"fuzz ok: {it} cases in {time.time() - t0:.0f} s (seeds {seed0}..{seed0 + it - 1}), sweeps used {kinds}")
