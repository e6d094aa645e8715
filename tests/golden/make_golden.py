#!/usr/bin/env python3
"""Regenerates oracle_vectors.json: seeded inputs -> outputs of the CPU oracle (oracle/fbg_oracle.c).

The reference binary cannot be built in this pipeline (sdsl-lite submodule empty), so these vectors
pin the ORACLE, not the reference: they let the GPU tests run against committed data and make any
later drift of the oracle itself visible.  Inputs are stored as generator parameters (seed, shape,
kwargs of tests/conftest.random_msa); outputs as full arrays for small cases and sha256 digests of the
little-endian uint64 arrays / xGFA bytes for the larger ones.
"""
import hashlib
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from conftest import random_msa  # noqa: E402
from oracle import pyoracle as O  # noqa: E402

CASES = [
    dict(seed=1, m=4, n=40, kw={}),
    dict(seed=2, m=9, n=300, kw=dict(gap_p=0.03, gap_run=5)),
    dict(seed=3, m=64, n=700, kw=dict(similar=0.97)),
    dict(seed=4, m=33, n=513, kw=dict(similar=0.98, gap_p=0.01, gap_run=9, n_p=0.01), ignore="N"),
    dict(seed=5, m=257, n=1500, kw=dict(similar=0.95)),
    dict(seed=6, m=1000, n=2000, kw={}),
    dict(seed=7, m=12, n=4000, kw=dict(alphabet="AC", similar=0.995)),
    dict(seed=8, m=64, n=3000, kw={}, nonelastic=True),
    dict(seed=9, m=10, n=900, kw=dict(similar=0.9), nonelastic=True),
    dict(seed=10, m=16, n=2500, kw=dict(gap_p=0.02, gap_run=4), nonelastic=True, gapped=True),
    dict(seed=11, m=40, n=1200, kw=dict(similar=0.9, gap_p=0.01, gap_run=7), nonelastic=True, gapped=True),
]


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype="<u8").tobytes()).hexdigest()


def run_case(c):
    msa = random_msa(np.random.default_rng(c["seed"]), c["m"], c["n"], **c["kw"])
    out = dict(c)
    out["msa_sha256"] = hashlib.sha256(msa.tobytes()).hexdigest()
    small = c["m"] * c["n"] <= 4000
    if c.get("nonelastic"):
        if c.get("gapped"):      # segment2elasticValid (fbg.cpp:738-866)
            if c["seed"] == 10:
                msa[:, 0] = np.arange(c["m"]) % 4 + ord("E")      # a first block exists: the heuristic finds a segmentation
            v = O.gapped_v(msa)
            s, prev, b = O.segment2_dp(v)
            out["msa_sha256"] = hashlib.sha256(msa.tobytes()).hexdigest()
        else:
            v = O.segment_v(msa)
            s, prev, b = O.segment_dp(v)
        out.update(v=digest(v), s=digest(s), prev=digest(prev), boundaries=None if b is None else b.tolist())
        if b is not None:
            out["stats"] = O.segment_stats(msa, b)
    else:
        f = O.compute_f(msa, ignore=c.get("ignore", ""))
        mml, bt, b = O.minmax_dp(f)
        with tempfile.TemporaryDirectory() as td:
            x = O.write_xgfa(msa, b, os.path.join(td, "o.xgfa"), ids=[f"r{i}" for i in range(c["m"])])
        out.update(f=f.tolist() if small else digest(f), mml=digest(mml), bt=digest(bt), boundaries=digest(b),
                   n_blocks=len(b), optimal=int(mml[-1]), xgfa_sha256=hashlib.sha256(x).hexdigest(), xgfa_bytes=len(x))
    return out


if __name__ == "__main__":
    res = [run_case(c) for c in CASES]
    with open(os.path.join(HERE, "oracle_vectors.json"), "w") as fh:
        json.dump(res, fh, indent=1)
    print(f"wrote {len(res)} cases")
