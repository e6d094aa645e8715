"""Randomised parity campaign (not part of the test suite): random MSAs, shapes, alphabets, code paths against the
oracle.  Usage: gpu_fuzz.py SECONDS [SEED].  Exits non-zero on the first mismatch, printing how to reproduce it."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from conftest import random_msa
from oracle import pyoracle as O
import founderblockgraphs_amd as F
from founderblockgraphs_amd._lib import PART_HALO_BYTES

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 12345
ENVS = [{}, {"FBG_NO_PACKED": "1"}, {"FBG_FORCE_WIDE": "1"}, {"FBG_MSD_MIN": "1"}, {"FBG_RANK_NO_THRESHOLD": "1"},
        {"FBG_NO_RANKED": "1"}, {"FBG_FULL_KEYS": "1"}, {"FBG_MSD_MIN": "1", "FBG_FORCE_WIDE": "1", "FBG_FULL_KEYS": "1"},
        {"FBG_MSD_MIN": "1", "FBG_FULL_KEYS": "1"}, {"FBG_BP_MIN": "1"}, {"FBG_BP_MIN": "1", "FBG_NO_RANKED": "1"},
        {"FBG_MSD_MIN": "1", "FBG_NO_RANKED": "1"}, {"FBG_MSD_MIN": "1", "FBG_BP_MIN": "1", "FBG_NO_RANKED": "1"},
        {"FBG_PURE_SCAN": "1"}, {"FBG_PURE_SCAN": "1", "FBG_NO_PACKED": "1"}, {"FBG_PURE_SCAN": "1", "FBG_MSD_MIN": "1"},
        {"FBG_GAPPED_RANK": "-1"}, {"FBG_GAPPED_RANK": "-1", "FBG_BP_MIN": "1"}, {"FBG_GAPPED_RANK": "-1", "FBG_MSD_MIN": "1"},
        {"FBG_GAPPED_RANK": "4"}, {"FBG_GAPPED_RANK": "4", "FBG_MSD_MIN": "1"}, {"FBG_GAPPED_RANK": "2"}, {"FBG_GAPPED_RANK": "3"},
        {"FBG_MSD_MIN": "1", "FBG_MSD_SAMPLE_BINS": "1"}, {"FBG_MSD_MIN": "1", "FBG_GAPPED_RANK": "-1", "FBG_MSD_SAMPLE_BINS": "1"},
        {"FBG_MSD_MIN": "1", "FBG_GAPPED_RANK": "-1"}, {"FBG_MSD_MIN": "1"},
        {"FBG_SPAN_SCAN": "1"}, {"FBG_SPAN_SCAN": "1"}, {"FBG_SPAN_SCAN": "1", "FBG_MSD_MIN": "1"}, {"FBG_SPAN_SCAN": "-1"}]
ALL_KEYS = sorted({k for e in ENVS for k in e})
eng = F.Engine(0)
parts = [F.Engine(0) for _ in range(3)]
t0 = time.time()
last_note = t0
it = 0
while time.time() - t0 < budget:
    seed = seed0 + it
    it += 1
    if time.time() - last_note > 30 or os.environ.get("FBG_FUZZ_BIG"):
        last_note = time.time()
        print(f"... {it} cases, {time.time() - t0:.0f} s", flush=True)
    rng = np.random.default_rng(seed)
    if os.environ.get("FBG_FUZZ_BIG"):
        # texts of 1.7e7 .. 2.4e7 symbols: above the sizes from which the three-pass sorts and the by-position passes
        # run by themselves (2^24); a case takes the oracle 10-20 s
        m = int(rng.choice([40, 200, 1000]))
        n = int(rng.integers(17_000_000, 24_000_000)) // m
    else:
        m = int(rng.choice([1, 2, 3, 5, 17, 64, 130, 400]))
        n = int(rng.choice([1, 2, 7, 33, 100, 257, 1000, 3000]))
    alphabet = str(rng.choice(["A", "AC", "ACGT", "ACGTN", "ACGTRYKM"]))
    kw = {}
    if rng.random() < 0.5:
        kw["similar"] = float(rng.choice([0.5, 0.9, 0.99]))
    gaps = rng.random() < 0.45
    if gaps:
        kw["gap_p"] = float(rng.choice([0.01, 0.05, 0.3])); kw["gap_run"] = int(rng.choice([1, 4, 20]))
    msa = random_msa(rng, m, n, alphabet=alphabet, **kw)
    if gaps and n >= 33 and rng.random() < 0.4:          # rows that start late, end early or skip a long stretch
        for i in rng.choice(m, size=max(1, m // 3), replace=False):
            kind = int(rng.integers(3))
            w = int(rng.integers(1, max(2, n // 2)))
            a0 = 0 if kind == 0 else n - w if kind == 1 else int(rng.integers(0, n - w))
            msa[i, a0:a0 + w] = ord("-")
    env = ENVS[int(rng.integers(len(ENVS)))]
    for e in [eng] + parts:                               # fbg_set_option: the library does not read the environment
        for k in ALL_KEYS:
            e.set_option(k[4:].lower(), int(env.get(k, -1 if k in ("FBG_MSD_MIN", "FBG_BP_MIN") else 0)))
    tag = f"seed={seed} m={m} n={n} alphabet={alphabet} kw={kw} env={env}"
    try:
        ign = "N" if (alphabet == "ACGTN" and rng.random() < 0.5) else ""
        tricks_off = bool(rng.random() < 0.3)
        f = O.compute_f(msa, ignore=ign, disable_tricks=tricks_off, threads=8 if os.environ.get("FBG_FUZZ_BIG") else 1)
        try:
            g = eng.elastic_f(msa, ignorechars=ign, disable_efg_tricks=tricks_off)
            assert np.array_equal(g, f), "f"
            mml, bt, b = O.minmax_dp(f)
            gb, gmml, gbt = eng.minmax_dp(g, full=True)
            assert np.array_equal(gmml, mml) and np.array_equal(gbt, bt) and np.array_equal(gb, b), "dp"
            if not all(all(c == ord("-") for c in msa[i]) for i in range(m)):
                node_of, first, rep_row, ecount, edges = eng.block_graph(b)
                assert int(first[-1]) > 0, "graph"
        except F.NoSegmentation:
            assert tricks_off and f[0] == n, "no segmentation"
        # non-elastic mode with gaps (segment2elasticValid): v[] and its chain DP
        gv = eng.gapped_v(msa)
        assert np.array_equal(gv, O.gapped_v(msa, literal=m * n <= 3000)), "gapped v"
        s2, p2, b2 = O.segment2_dp(gv)
        gs2, gp2, gb2 = eng.gapped_dp(gv)
        assert np.array_equal(gs2, s2) and np.array_equal(gp2, p2), "gapped dp"
        assert (b2 is None) == (gb2 is None) and (b2 is None or np.array_equal(gb2, b2)), "gapped boundaries"
        if not gaps:
            v = O.segment_v(msa)
            assert np.array_equal(eng.repeatfree_v(msa), v), "v"
        if not all(all(c == ord("-") for c in msa[i]) for i in range(m)):
            # key-range partitions played by separate contexts (gaps / ignore characters: fbg_part_index_build_ignore)
            P = int(rng.integers(1, 4))
            es = parts[:P]
            pign = ign if gaps or ign else ""
            poff = tricks_off if (gaps or ign) else False
            blobs = torch.zeros(P * PART_HALO_BYTES, dtype=torch.uint8, device="cuda")
            gm = [torch.zeros(n + 1, dtype=torch.int32, device="cuda") for _ in range(P)]
            torch.cuda.synchronize()
            for e in es:
                e.msa_load_host(msa)
            for r, e in enumerate(es):
                e.part_index_build(r, P, blobs.data_ptr() + r * PART_HALO_BYTES, False, pign, poff)
                e.sync()
            for r, e in enumerate(es):
                e.part_scan(blobs.data_ptr(), gm[r].data_ptr())
                e.sync()
            red = gm[0]
            for x in gm[1:]:
                red = torch.maximum(red, x)
            torch.cuda.synchronize()
            vd = [e.part_finish(red.data_ptr()) for e in es]
            assert len(set(vd)) == 1, f"verdicts {vd}"
            if vd[0] == 2:
                for r, e in enumerate(es):
                    e.part_rescan(gm[r].data_ptr()); e.sync()
                red = gm[0]
                for x in gm[1:]:
                    red = torch.maximum(red, x)
                torch.cuda.synchronize()
                vd = [e.part_finish(red.data_ptr()) for e in es]
                assert len(set(vd)) == 1, f"verdicts after the re-scan {vd}"
            if vd[0] == 1:
                d_f = torch.zeros(n, dtype=torch.int64, device="cuda")
                torch.cuda.synchronize()
                es[0].scan_f(0, n, d_f.data_ptr(), disable_efg_tricks=poff); es[0].sync()
                want = O.compute_f(msa, ignore=pign, disable_tricks=poff)
                assert np.array_equal(d_f.cpu().numpy().astype(np.uint64), want), f"partitioned f (P={P})"
    except Exception as ex:      # noqa: BLE001
        print("MISMATCH", tag, repr(ex), flush=True)
        sys.exit(1)
print(f"fuzz ok: {it} cases in {time.time() - t0:.0f} s (seeds {seed0}..{seed0 + it - 1})")
