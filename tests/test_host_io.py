"""The host program's FASTA reader and xGFA writer without a GPU: founderblockgraphs_amd/fbg_host_selftest (host/fasta.cpp +
host/xgfa.cpp behind a command line) against the oracle's writer (fbg.cpp:1185-1301) and the test-side reader with the
rules of read_input (fbg.cpp:151-201).  The GPU tests check the same bytes through the whole program (tests/test_cli.py)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import random_msa
from fasta_util import read_fasta, write_fasta
from oracle import pyoracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "founderblockgraphs_amd", "fbg_host_selftest")
GOLD = os.path.join(ROOT, "tests", "golden")


def selftest(fasta, gap_limit, elastic, paths, out, boundaries=(), graph=None):
    extra = [f"GRAPH={graph}"] if graph else []
    p = subprocess.run([EXE, fasta, str(gap_limit), str(int(elastic)), str(int(paths)), out, *extra, *map(str, boundaries)],
                       capture_output=True)
    assert p.returncode == 0, p.stderr
    lines = p.stdout.decode().split("\n")
    m, n = map(int, lines[0].split())
    stats = list(map(int, lines[1].split())) if len(lines) > 1 and lines[1] else None
    return m, n, stats, p.stderr


CASES = [
    dict(m=1, n=1), dict(m=2, n=7), dict(m=5, n=64, similar=0.9), dict(m=17, n=300, similar=0.97),
    dict(m=9, n=200, similar=0.95, gap_p=0.05, gap_run=3), dict(m=33, n=120, gap_p=0.2, gap_run=1),
    dict(m=4, n=90, alphabet="ACGTN", similar=0.9, n_p=0.02), dict(m=64, n=1000, similar=0.99, gap_p=0.01, gap_run=8),
]


@pytest.mark.parametrize("case", range(len(CASES)))
@pytest.mark.parametrize("paths", [False, True])
def test_xgfa_bytes_and_statistics_match_the_oracle(case, paths, tmp_path):
    kw = dict(CASES[case])
    m, n = kw.pop("m"), kw.pop("n")
    msa = random_msa(np.random.default_rng(100 + case), m, n, **kw)
    ids = [f"row {i} of case {case}" for i in range(m)]
    fasta = str(tmp_path / "in.fasta")
    write_fasta(fasta, msa, ids, width=None if case % 2 else 60)
    f = O.compute_f(msa)
    b = O.minmax_dp(f)[2]
    expected = O.write_xgfa(msa, b, str(tmp_path / "oracle.gfa"), ids=ids if paths else None)
    got_m, got_n, stats, _ = selftest(fasta, 1, True, paths, str(tmp_path / "host.gfa"), b.tolist())
    assert (got_m, got_n) == (m, n)
    assert open(tmp_path / "host.gfa", "rb").read() == expected
    st = O.segment_stats(msa, b)
    assert stats == [st["nodes"], st["total_label_length"], st["founders"], st["edges"]]
    # the same bytes from node / edge arrays (what fbg_block_graph hands the program): output_efg's numbering restated
    # with dicts stands in for the GPU here
    nb = len(b)
    node_of = np.full((nb, m), 0xFFFFFFFF, dtype=np.uint32)
    rep_row = np.zeros((nb, m), dtype=np.uint32)
    first = np.zeros(nb + 1, dtype=np.uint64)
    ecount = np.zeros(nb, dtype=np.uint64)
    edges = np.zeros((nb, m), dtype=np.uint64)
    nodecount, prev = 0, 0
    for j, end in enumerate(b):
        cur, e = {}, set()
        for i in range(m):
            lab = bytes(c for c in msa[i, prev:min(int(end) + 1, n)] if c != ord("-"))
            if not lab:
                continue
            if lab not in cur:
                rep_row[j, len(cur)] = i
                cur[lab] = nodecount
                nodecount += 1
            node_of[j, i] = cur[lab]
            if j > 0 and node_of[j - 1, i] != 0xFFFFFFFF:
                e.add((int(node_of[j - 1, i]) << 32) | cur[lab])
        first[j + 1] = nodecount
        ecount[j] = len(e)
        edges[j, :len(e)] = sorted(e)
        prev = int(end) + 1
    with open(tmp_path / "graph.bin", "wb") as fh:
        fh.write(np.array([nb, m], dtype=np.uint64).tobytes())
        for arr in (node_of, rep_row, first, ecount, edges):
            fh.write(arr.tobytes())
    selftest(fasta, 1, True, paths, str(tmp_path / "host_graph.gfa"), b.tolist(), graph=str(tmp_path / "graph.bin"))
    assert open(tmp_path / "host_graph.gfa", "rb").read() == expected


@pytest.mark.parametrize("name", sorted(x for x in os.listdir(GOLD) if x.endswith(".fasta")))
@pytest.mark.parametrize("elastic,gap_limit", [(True, 1), (False, 1), (False, 3), (False, 0)])
def test_reader_applies_the_reference_row_filters(name, elastic, gap_limit, tmp_path):
    path = os.path.join(GOLD, name)
    msa, _ = read_fasta(path, elastic=elastic, gap_limit=gap_limit)
    m, n, _, _ = selftest(path, gap_limit, elastic, False, str(tmp_path / "unused.gfa"))
    if msa is None:
        assert m == 0
    else:
        assert (m, n) == msa.shape


def test_rows_of_another_length_are_dropped_with_a_warning(tmp_path):
    fasta = tmp_path / "ragged.fasta"
    fasta.write_bytes(b">a\nACGTACGT\n>b\nACGT\n>c\nACGAACGT\n")
    m, n, _, err = selftest(str(fasta), 1, True, False, str(tmp_path / "unused.gfa"))
    assert (m, n) == (2, 8) and b"b" in err
