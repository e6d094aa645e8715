"""CPU tests (no GPU): the oracle against the golden vectors, against its own second formulation,
and against brute-force definition checkers.  The oracle is test infrastructure (oracle/)."""
import json
import os
import random

import numpy as np
import pytest

from conftest import random_msa
from fasta_util import read_fasta
from oracle import pyoracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")
with open(os.path.join(GOLD, "appendix_b.json")) as fh:
    APPX = json.load(fh)


def parse_xgfa(text):
    out = {"S": [], "L": [], "P": {}}
    for line in text.decode().splitlines():
        t = line.split("\t")
        if t[0] == "S":
            assert int(t[1]) == len(out["S"])
            out["S"].append(t[2])
        elif t[0] == "L":
            assert t[2] == "+" and t[4] == "+" and t[5] == "0M"
            out["L"].append([int(t[1]), int(t[3])])
        elif t[0] == "P":
            out["P"][t[1]] = [int(v.rstrip("+")) for v in t[2].split(",")]
            assert t[3] == "*"
        else:
            out[t[0]] = line
    return out


@pytest.mark.parametrize("name", ["msa.fasta", "test.fasta", "test2.fasta", "test3.fasta"])
def test_golden_elastic(name, tmp_path):
    msa, ids = read_fasta(os.path.join(GOLD, name))
    exp = APPX[name]["elastic"]
    for literal in (False, True):
        f = O.compute_f(msa, literal=literal)
        assert f.tolist() == exp["f"]
    mml, bt, b = O.minmax_dp(f)
    assert b.tolist() == exp["boundaries"]
    if "mml" in exp:
        assert mml.tolist() == exp["mml"] and bt.tolist() == exp["bt"]
    x = parse_xgfa(O.write_xgfa(msa, b, str(tmp_path / "o.xgfa"), ids=[i.decode() for i in ids]))
    assert x["M"] == f"M\t{msa.shape[0]}\t{msa.shape[1]}"
    for key in ("X", "B"):
        if key in exp:
            assert x[key] == exp[key]
    if "S" in exp:
        assert x["S"] == exp["S"] and x["L"] == exp["L"] and x["P"] == exp["P"]


def test_golden_disable_tricks(tmp_path):
    msa, _ = read_fasta(os.path.join(GOLD, "msa.fasta"))
    exp = APPX["msa.fasta"]["elastic_disable_tricks"]
    for literal in (False, True):
        assert O.compute_f(msa, disable_tricks=True, literal=literal).tolist() == exp["f"]
    b = O.minmax_dp(np.array(exp["f"], dtype=np.uint64))[2]
    assert b.tolist() == exp["boundaries"]
    assert parse_xgfa(O.write_xgfa(msa, b, str(tmp_path / "o.xgfa")))["X"] == exp["X"]


def test_golden_nonelastic():
    msa, _ = read_fasta(os.path.join(GOLD, "test.fasta"), elastic=False, gap_limit=1)
    exp = APPX["test.fasta"]["nonelastic_gap_limit_1"]
    for literal in (False, True):
        assert O.segment_v(msa, literal=literal).tolist() == exp["v"]
    s, prev, b = O.segment_dp(np.array(exp["v"], dtype=np.uint64))
    assert s.tolist() == exp["s"] and prev.tolist() == exp["prev"] and b.tolist() == exp["boundaries"]
    assert int(s[-1]) == exp["score"]
    st = O.segment_stats(msa, b)
    assert st == {k: exp[k] for k in ("nodes", "total_label_length", "founders", "edges")}
    # config 1 of BASELINE.json: every row of msa.fasta has a gap -> all rows filtered (fbg.cpp:3351-3355)
    msa2, _ = read_fasta(os.path.join(GOLD, "msa.fasta"), elastic=False, gap_limit=1)
    assert msa2 is None


def test_suffix_array_vs_naive():
    rng = np.random.default_rng(3)
    for n, sigma in [(1, 1), (2, 1), (50, 1), (200, 2), (500, 4), (1000, 3), (3000, 2)]:
        body = bytes(rng.integers(1, sigma + 1, n).astype(np.uint8))
        text = body + b"\0"
        sa = O.suffix_array(text)
        naive = sorted(range(len(text)), key=lambda i: text[i:])
        assert sa.tolist() == naive


def test_lcp_and_isa_vs_naive():
    rng = np.random.default_rng(4)
    msa = random_msa(rng, 5, 60, alphabet="AC", gap_p=0.05, gap_run=3)
    T, SA, ISA, LCP = O.msa_index(msa)
    t = bytes(T)
    assert t.count(b"#") == 5 and t[-1] == 0
    assert sorted(SA.tolist()) == list(range(len(t)))
    assert all(ISA[SA[r]] == r for r in range(len(t)))
    for r in range(1, len(t)):
        a, b = t[SA[r - 1]:], t[SA[r]:]
        assert a < b
        h = 0
        while h < min(len(a), len(b)) and a[h] == b[h]:
            h += 1
        assert LCP[r] == h


def brute_f(msa, ignore=b"", disable_tricks=False):
    """Appendix A.1 straight from the definition: g = 1 + max lcp with any text position that is not
    the pointer of an active row; O(N) per cell, tiny inputs only."""
    m, n = msa.shape
    rows = [bytes(r).replace(b"-", b"") for r in msa]
    text = b"#".join(rows) + b"#\0"
    pos, k = [], 0
    for r in rows:
        pos.append(k)
        k += len(r) + 1
    N = len(text)

    def lcp(a, b):
        h = 0
        while a + h < N and b + h < N and text[a + h] == text[b + h]:
            h += 1
        return h
    nz = [0] * m
    f = []
    for x in range(n):
        active = [i for i in range(m) if disable_tricks or nz[i] > 0]
        ptr = {pos[i] + nz[i] for i in active}
        fimax = x
        for i in active:
            p = pos[i] + nz[i]
            g = 1 + max(lcp(p, q) for q in range(N) if q not in ptr)
            cols = [c for c in range(n) if msa[i, c] != 0x2D]
            gg = nz[i] + g
            if gg > len(cols):
                fi = n if disable_tricks else cols[-1]
            else:
                fi = cols[gg - 1]
            ign = [c for c in range(x, n) if bytes([msa[i, c]]) in [bytes([b]) for b in ignore]]
            if ign:
                fi = min(fi, ign[0])
            fimax = max(fimax, fi)
        f.append(fimax)
        for i in range(m):
            nz[i] += msa[i, x] != 0x2D
    return f


@pytest.mark.parametrize("seed", range(6))
def test_f_vs_bruteforce_definition(seed):
    rng = np.random.default_rng(seed)
    m, n = int(rng.integers(1, 6)), int(rng.integers(1, 30))
    msa = random_msa(rng, m, n, alphabet="ACN" if seed % 2 else "AC", gap_p=0.08 if seed % 3 else 0.0, gap_run=2,
                     similar=0.8 if seed > 2 else 0.0)
    for tricks_off in (False, True):
        for ign in (b"", b"N"):
            exp = brute_f(msa, ign, tricks_off)
            assert O.compute_f(msa, ignore=ign, disable_tricks=tricks_off).tolist() == exp
            assert O.compute_f(msa, ignore=ign, disable_tricks=tricks_off, literal=True).tolist() == exp


@pytest.mark.parametrize("seed", range(12))
def test_two_formulations_agree(seed):
    rng = np.random.default_rng(50 + seed)
    m, n = int(rng.integers(1, 40)), int(rng.integers(1, 400))
    kw = [dict(), dict(gap_p=0.03, gap_run=4), dict(similar=0.95), dict(similar=0.98, gap_p=0.02, gap_run=8),
          dict(alphabet="AC", similar=0.99), dict(alphabet="ACGTN", n_p=0.03)][seed % 6]
    msa = random_msa(rng, m, n, **kw)
    for tricks_off in (False, True):
        a = O.compute_f(msa, ignore="N", disable_tricks=tricks_off)
        b = O.compute_f(msa, ignore="N", disable_tricks=tricks_off, literal=True)
        assert np.array_equal(a, b)
    if b"-" not in msa.tobytes():
        assert np.array_equal(O.segment_v(msa), O.segment_v(msa, literal=True))


def test_threaded_partition_equals_single():
    """--threads partition of fbg.cpp:2278-2289 gives the same f (tricks enabled)."""
    rng = np.random.default_rng(9)
    msa = random_msa(rng, 20, 997, similar=0.95, gap_p=0.02, gap_run=5)
    f1 = O.compute_f(msa, threads=1)
    for t in (2, 3, 8):
        assert np.array_equal(O.compute_f(msa, threads=t), f1)


def brute_minmax(f):
    """min over valid segmentations of the longest block: the quantity minmaxlength[n] claims to be."""
    n = len(f)
    INF = 10 ** 9
    best = [INF] * (n + 1)
    best[0] = 0
    for j in range(1, n + 1):
        for x in range(j):
            if f[x] + 1 <= j and best[x] < INF:
                best[j] = min(best[j], max(best[x], j - x))
    return best


@pytest.mark.parametrize("seed", range(8))
def test_dp_is_optimal_and_boundaries_valid(seed):
    rng = np.random.default_rng(70 + seed)
    n = int(rng.integers(1, 120))
    x = np.arange(n)
    f = np.minimum(x + rng.integers(0, 9, n), n - 1)
    f[0] = 0
    mml, bt, b = O.minmax_dp(f.astype(np.uint64))
    assert mml.tolist() == brute_minmax(f.tolist())
    starts = [0] + [int(e) + 1 for e in b[:-1]]
    ends = [int(e) for e in b[:-1]] + [n - 1]
    assert b[-1] == n
    for s0, e0 in zip(starts, ends):
        assert f[s0] <= e0                      # every block is at least its minimal valid width
    assert max(e0 - s0 + 1 for s0, e0 in zip(starts, ends)) == mml[n]


def test_nonelastic_dp_no_segmentation():
    v = np.array([1, 2, 3], dtype=np.uint64)    # no valid block ends anywhere
    s, prev, b = O.segment_dp(v)
    assert b is None and s[-1] == 4


def test_xgfa_edge_cases(tmp_path):
    # all-gap block labels are skipped; substr clamps at the row end (fbg.cpp:1214-1216,1234-1236)
    msa = O.msa_array(["AC--GT", "AC--GA", "ACTTGT"])
    out = O.write_xgfa(msa, np.array([1, 3, 6], dtype=np.uint64), str(tmp_path / "e.xgfa"), ids=["a", "b", "c"])
    x = parse_xgfa(out)
    assert x["X"] == "X\t1\t3\t5" and x["B"] == "B\t1\t1\t2"
    assert x["S"] == ["AC", "TT", "GT", "GA"]
    assert x["L"] == [[0, 1], [1, 2]]           # rows a, b have no node in block 2 -> no edge into block 3
    assert x["P"] == {"a": [0, 2], "b": [0, 3], "c": [0, 1, 2]}


# ---- non-elastic mode with gaps: segment2elasticValid (fbg.cpp:738-866) -------------------------------------

def brute_gapped_v(rows):
    """fbg.cpp:763-822 by definition: the union of the occurrence sets (by start position in T) of the rows'
    gap-stripped strings has exactly m members; an empty string occurs at all N = |T|+1 suffixes."""
    m, n = len(rows), len(rows[0])
    T = "".join(r.replace("-", "") + "#" for r in rows)
    N = len(T) + 1

    def occ(p):
        if p == "":
            return set(range(N))
        out, k = set(), T.find(p)
        while k >= 0:
            out.add(k)
            k = T.find(p, k + 1)
        return out
    v, jp = [0] * n, n
    for j in range(n - 1, -1, -1):
        v[j] = j + 1
        while True:
            u = set()
            for r in rows:
                u |= occ(r[jp:j + 1].replace("-", ""))
            if len(u) == m:
                v[j] = jp
                break
            if jp == 0:
                break
            jp -= 1
    return v


def brute_segment2_dp(v):
    """fbg.cpp:827-846 in explicit arithmetic modulo 2^64."""
    n, M = len(v), 1 << 64
    s, prev = [n + 1] * n, [n + 1] * n
    for j in range(1, n):
        jp = v[j]
        if jp > j:
            continue
        if jp == 0:
            s[j], prev[j] = j + 1, 0
            continue
        a = max(s[jp - 1], (j - jp + 1) % M)
        b = max(s[j - 1], (j - prev[j - 1] + 1) % M)
        if a < b:
            s[j], prev[j] = a, jp
        else:
            s[j], prev[j] = b, prev[j - 1]
    return s, prev


@pytest.mark.parametrize("name", ["msa.fasta", "test.fasta", "test2.fasta", "test3.fasta"])
def test_golden_gapped_nonelastic(name):
    """--gap-limit=0 keeps every row (fbg.cpp:105-106) and runs segment2elasticValid (3438-3439)."""
    msa, _ = read_fasta(os.path.join(GOLD, name), elastic=False, gap_limit=0)
    exp = APPX[name]["nonelastic_gap_limit_0"]
    for literal in (False, True):
        assert O.gapped_v(msa, literal=literal).tolist() == exp["v"]
    s, prev, b = O.segment2_dp(np.array(exp["v"], dtype=np.uint64))
    assert s.tolist() == exp["s"] and prev.tolist() == exp["prev"]
    assert (None if b is None else b.tolist()) == exp["boundaries"]


@pytest.mark.parametrize("seed", range(6))
def test_gapped_v_and_dp_vs_bruteforce(seed):
    rng = random.Random(900 + seed)
    solved = 0
    for it in range(150):
        m, n = rng.randint(1, 6), rng.randint(1, 14)
        sig = rng.choice(["AC", "ACGT", "A"])
        pg = rng.choice([0, 0.1, 0.3, 0.6])
        rows = ["".join("-" if rng.random() < pg else rng.choice(sig) for _ in range(n)) for _ in range(m)]
        if it % 3 == 0:
            rows = ["".join(c if rng.random() > 0.2 else rng.choice(sig + "-") for c in rows[0]) for _ in range(m)]
        msa = O.msa_array(rows)
        exp = brute_gapped_v(rows)
        assert O.gapped_v(msa, literal=True).tolist() == exp, rows
        assert O.gapped_v(msa).tolist() == exp, rows
        s, prev, b = O.segment2_dp(np.array(exp, dtype=np.uint64))
        es, ep = brute_segment2_dp(exp)
        assert s.tolist() == es and prev.tolist() == ep, rows
        assert (b is None) == (es[-1] == n + 1)
        if b is not None:
            solved += 1
            # every block of the answer passes the reference's own test: it starts at or before v[end]
            lo = 0
            for e in b.tolist():
                assert lo <= exp[e] <= e
                lo = e + 1
            assert b[-1] == n - 1
    assert solved > 10


def test_literal_scan_text_length_multiple_of_64():
    """N = 64: the occurrence table of the literal scan needs its closing block."""
    rows = ['AACACCAAAACC', 'AACACCAAAACC', 'CACACCAAAACC', 'AACACCAAAAC-', 'AACCCC-AAACC']
    msa = O.msa_array(rows)
    assert O.gapped_v(msa, literal=True).tolist() == brute_gapped_v(rows) == O.gapped_v(msa).tolist()
