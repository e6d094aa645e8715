"""The C++ host program (founderblockgraphs_amd/founderblockgraph) against the reference's command
line contract.  CPU part: option parser vs the reference's OWN generated parser (compiled from
/root/reference by `make -C oracle ref` into oracle/_ref/), messages, exit codes, BASELINE config 1.
GPU part: emitted xGFA byte-identical to the oracle's writer."""
import os
import subprocess

import numpy as np
import pytest

from conftest import random_msa
from oracle import pyoracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "founderblockgraphs_amd", "founderblockgraph")
DUMP = os.path.join(ROOT, "founderblockgraphs_amd", "fbg_options_dump")
REF = os.path.join(ROOT, "oracle", "_ref", "refcmdline")
GOLD = os.path.join(ROOT, "tests", "golden")


def run(exe, *args, env=None):
    p = subprocess.run([exe, *args], capture_output=True, env=None if env is None else {**os.environ, **env})
    return p.returncode, p.stdout, p.stderr.replace(exe.encode(), b"PROG")


def ref_available():
    if not os.path.exists(REF) and os.path.isdir("/root/reference"):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True, capture_output=True)
    return os.path.exists(REF)


needs_ref = pytest.mark.skipif(not ref_available(), reason="oracle/_ref/refcmdline not built (reference tree absent)")

ARGVS = [
    ["--help"], ["-h"], ["--full-help"], ["--version"], ["-V"], [],
    ["--input", "a"], ["--output", "b"], ["--input=a", "--output=b"],
    ["--input", "a", "--output", "b", "-e", "--gfa", "-p", "-t", "8", "--gap-limit=5", "--ignore-chars=N-"],
    ["--input", "a", "--output", "b", "--elastic", "--elastic"], ["--input", "a", "--input", "b", "--output", "c"],
    ["--input", "a", "--output", "b", "--gap-limit", "zz"], ["--input", "a", "--output", "b", "-t", "3x"],
    ["--input", "a", "--output", "b", "--gap-limit", "0x10", "--threads=-1"],
    ["--input", "a", "--output", "b", "--heuristic-subset", "7", "--disable-elastic-tricks"],
    ["--input", "a", "--output", "b", "--graphviz-output", "g.dot", "--memory-chart-output=m.html"],
    ["--input", "a", "--output", "b", "--bogus"], ["--input"], ["--inp", "a", "--outp", "b", "--gf"],
    ["--input", "a", "--output", "b", "-ep"], ["--input", "a", "--output", "b", "-x"],
    ["--input", "a", "--output", "b", "stray", "-e"],
]


@needs_ref
@pytest.mark.parametrize("argv", ARGVS, ids=[" ".join(a) or "<none>" for a in ARGVS])
def test_option_parser_matches_reference_parser(argv):
    assert run(DUMP, *argv) == run(REF, *argv)


@needs_ref
def test_help_and_version_text_of_the_real_binary():
    for flag in ("--help", "--full-help", "--version"):
        assert run(BIN, flag)[:2] == run(REF, flag)[:2]


def test_flag_validation_messages():
    def err(*a):
        rc, _, e = run(BIN, "--input", os.path.join(GOLD, "test.fasta"), "--output", "/tmp/fbg_unused", *a)
        return rc, e.decode()
    assert err("--gap-limit=-1") == (1, "Gap limit needs to be non-negative.\n")
    assert err("-p") == (1, "Output of original sequences as paths without option --elastic is not implemented!\n")
    assert err("--gfa") == (1, "--elastic and --gfa options are currently only supported when both are used!\n")
    assert err("--elastic") == (1, "--elastic and --gfa options are currently only supported when both are used!\n")
    assert err("--heuristic-subset=0") == (1, "wrong value for --heuristic-subset!\n")
    assert err("--heuristic-subset=-5") == (1, "wrong value for --heuristic-subset!\n")
    rc, e = err("--elastic", "--gfa", "--threads=0")
    assert rc == 1 and e.endswith("Invalid number of threads.\n")


def test_baseline_config1_all_rows_filtered():
    """BASELINE.json configs[0]: test/msa.fasta, non-elastic, --gap-limit=1 -> every row has a gap, all four
    are dropped with a NOTICE, 'Unable to read sequences', EXIT_FAILURE, nothing written."""
    out = "/tmp/fbg_cfg1.index"
    if os.path.exists(out):
        os.remove(out)
    rc, so, se = run(BIN, "--input", os.path.join(GOLD, "msa.fasta"), "--output", out, "--gap-limit=1")
    assert rc == 1 and so == b""
    lines = se.decode().split("\n")
    assert [l for l in lines if l.startswith("NOTICE")] == [
        "NOTICE: Sequence “file1” contained a gap run with 1 characters.",
        "NOTICE: Sequence “file2” contained a gap run with 2 characters.",
        "NOTICE: Sequence “file3” contained a gap run with 1 characters.",
        "NOTICE: Sequence “file4” contained a gap run with 2 characters."]
    assert se.decode().endswith("Unable to read sequences from the input\n.")
    assert not os.path.exists(out)


def test_length_mismatch_warning(tmp_path):
    p = tmp_path / "bad.fasta"
    p.write_bytes(b">a\nACGT\n>b\nACG\n>c\nAC\nGT\n")
    rc, _, se = run(BIN, "--input", str(p), "--output", str(tmp_path / "o"), "--elastic", "--gfa")
    assert "WARNING: length of the sequence “b” does not match that of the first sequence; skipping. (4 vs. 3)" \
        in se.decode()
    assert "Input MSA[1..2,1..4]" in se.decode()


# ---- GPU: end-to-end bytes -----------------------------------------------------------------------

from fasta_util import write_fasta  # noqa: E402


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["msa.fasta", "test.fasta", "test2.fasta", "test3.fasta"])
def test_cli_fixtures_xgfa_bytes(name, tmp_path):
    from fasta_util import read_fasta
    msa, ids = read_fasta(os.path.join(GOLD, name))
    for paths in (False, True):
        out = tmp_path / f"{name}.{paths}.xgfa"
        args = ["--input", os.path.join(GOLD, name), "--output", str(out), "--elastic", "--gfa"] + (["-p"] if paths else [])
        rc, _, se = run(BIN, *args)
        assert rc == 0, se.decode()
        b = O.minmax_dp(O.compute_f(msa))[2]
        exp = O.write_xgfa(msa, b, str(tmp_path / "exp.xgfa"), ids=[i.decode() for i in ids] if paths else None)
        assert out.read_bytes() == exp
        assert "Time taken: " in se.decode() and "Writing the xGFA to disk" in se.decode()


@pytest.mark.gpu
@pytest.mark.parametrize("case", range(7))
def test_cli_random_xgfa_bytes(case, tmp_path):
    rng = np.random.default_rng(300 + case)
    m, n, kw, ign, notricks = [(6, 200, {}, "", False), (40, 900, dict(similar=0.97, gap_p=0.02, gap_run=6), "", False),
                               (25, 500, dict(similar=0.95, n_p=0.02), "N", False),
                               (8, 300, dict(gap_p=0.05, gap_run=3), "", True), (300, 400, dict(similar=0.99), "", False),
                               # similar rows WITH gaps: the group-level scan on column spans (span_scan.hip), by the
                               # library's own choice and forced
                               (200, 3000, dict(similar=0.99, gap_p=0.003, gap_run=8), "", False),
                               (120, 2500, dict(similar=0.98, gap_p=0.004, gap_run=5, n_p=0.001), "N", True)][case]
    msa = random_msa(rng, m, n, **kw)
    # every row needs a non-gap character for -p (reference: undefined otherwise, fbg.cpp:1295)
    msa[:, 0] = np.where(msa[:, 0] == ord("-"), ord("A"), msa[:, 0])
    ids = [f"r{i} sample" for i in range(m)]
    src = tmp_path / "in.fasta"
    write_fasta(src, msa, ids, width=70 if case % 2 else None)
    out = tmp_path / "out.xgfa"
    args = ["--input", str(src), "--output", str(out), "-e", "--gfa", "-p"]
    if ign:
        args.append(f"--ignore-chars={ign}")
    if notricks:
        args.append("--disable-elastic-tricks")
    rc, _, se = run(BIN, *args, env={"FBG_DEBUG_ENV": "1", "FBG_SPAN_SCAN": "1"} if case == 6 else None)
    f = O.compute_f(msa, ignore=ign, disable_tricks=notricks)
    if notricks and f[0] == n:
        assert rc == 1 and "No valid segmentation found!" in se.decode()
        return
    assert rc == 0, se.decode()
    b = O.minmax_dp(f)[2]
    assert out.read_bytes() == O.write_xgfa(msa, b, str(tmp_path / "exp.xgfa"), ids=ids)


@pytest.mark.gpu
def test_cli_nonelastic_stats(tmp_path):
    """BASELINE config 2 shape (non-elastic, gap-free): the observable result is the stderr statistics."""
    rc, _, se = run(BIN, "--input", os.path.join(GOLD, "test.fasta"), "--output", str(tmp_path / "o.index"))
    text = se.decode()
    for line in ("Optimal score: 8", "Number of segments: 2", "#nodes=3", "total length of node labels=23",
                 "#founders=2", "#edges=2"):
        assert line in text
    assert rc == 1     # no defined .index output at this commit of the reference (see main.cpp)


@pytest.mark.gpu
@pytest.mark.parametrize("case", range(4))
def test_cli_gapped_nonelastic_stats(case, tmp_path):
    """Non-elastic mode with --gap-limit != 1 -> segment2elasticValid (fbg.cpp:3438-3439): score, segment count and
    graph statistics on stderr (738-929), then the same undefined make_efg call as the gap-free mode."""
    rng = np.random.default_rng(500 + case)
    m, n, kw, limit = [(6, 120, dict(gap_p=0.03, gap_run=2), 0), (12, 400, dict(gap_p=0.02, gap_run=3), 5),
                       (30, 900, dict(similar=0.9, gap_p=0.01, gap_run=4), 6), (3, 40, dict(gap_p=0.3), 0)][case]
    msa = random_msa(rng, m, n, **kw)
    msa[:, 0] = np.arange(m) % 4 + ord("E")          # rows start differently: a first block exists quickly
    src = tmp_path / "in.fasta"
    write_fasta(src, msa, [f"r{i}" for i in range(m)])
    from fasta_util import read_fasta
    kept, _ = read_fasta(str(src), elastic=False, gap_limit=limit)
    rc, _, se = run(BIN, "--input", str(src), "--output", str(tmp_path / "o.index"), f"--gap-limit={limit}")
    text = se.decode()
    v = O.gapped_v(kept)
    s, prev, b = O.segment2_dp(v)
    assert f"Input MSA[1..{kept.shape[0]},1..{n}]" in text
    assert f"Optimal score: {int(s[-1])}" in text
    if b is None:
        assert rc == 1 and "No valid segmentation found!" in text
        return
    st = O.segment_stats(kept, b)
    for line in (f"Number of segments: {len(b)}", f"#nodes={st['nodes']}", f"total length of node labels={st['total_label_length']}",
                 f"#founders={st['founders']}", f"#edges={st['edges']}"):
        assert line in text
    assert rc == 1     # no defined .index output at this commit of the reference (see main.cpp)
