"""Several GPUs as one engine (fbg_group_*, include/fbg_hip.h): the multi-device code of the C ABI, exercised on ONE
GPU with several contexts on device 0 (an id may repeat in dev_ids), against the oracle; and the two multi-GPU
configurations of BASELINE.json at full size with property checks (no CPU oracle finishes them):

  C4  synthetic 1000 x 8,000,000, --elastic: 8e9 symbols, positions beyond 32 bits, 8 key-range partitions worked off
      by one device in turn (the same partitions 8 GPUs hold one each);
  C5  synthetic 256 x 2,000,000, 5 % gap runs + N, --ignore-chars=N, --elastic: column shards of a replicated index.
"""
import os
import subprocess

import numpy as np
import pytest

from conftest import random_msa
from fasta_util import write_fasta
from oracle import pyoracle as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "founderblockgraphs_amd", "founderblockgraph")


@pytest.mark.parametrize("members", [1, 2, 3])
def test_group_elastic_f_matches_oracle(members):
    """f through fbg_group_elastic_f for gap-free (partitioned index), gapped / ignore-character and similar-row inputs
    (column shards of a replicated index); f is max-merged into like fbg.cpp:1681."""
    import founderblockgraphs_amd as F
    rng = np.random.default_rng(700 + members)
    with F.Group([0] * members) as grp:
        assert grp.size() == members
        for (m, n, kw, ign) in [(24, 500, {}, ""), (50, 301, dict(alphabet="AC"), ""), (16, 400, dict(gap_p=0.02, gap_run=7), ""),
                                (30, 300, dict(gap_p=0.05, gap_run=3, n_p=0.02), "N"), (40, 600, dict(similar=0.97), ""),
                                (9, 1, {}, ""), (1, 40, {}, ""), (65, 257, dict(alphabet="ACGTN"), "N")]:
            msa = random_msa(rng, m, n, **kw)
            for tricks_off in (False, True):
                f = O.compute_f(msa, ignore=ign, disable_tricks=tricks_off)
                if tricks_off and f[0] == n:
                    with pytest.raises(F.NoSegmentation):
                        grp.elastic_f(msa, ignorechars=ign, disable_efg_tricks=True)
                    continue
                got = grp.elastic_f(msa, ignorechars=ign, disable_efg_tricks=tricks_off)
                assert np.array_equal(got, f), (members, m, n, kw, ign, tricks_off, grp.plan_used())
            plan, parts = grp.plan_used()
            if members > 1 and (m, n) == (24, 500):
                assert plan == "partitioned" and parts == members, (plan, parts, m, n)
            # max-merge: a caller's f that is larger somewhere stays
            base = np.zeros(n, dtype=np.uint64)
            base[n // 2] = n - 1
            want = np.maximum(O.compute_f(msa, ignore=ign), base)
            assert np.array_equal(grp.elastic_f(msa, ignorechars=ign, f=base), want)
            # the sweep and the graph run on member 0
            eng = grp.member(0)
            mml, bt, b = O.minmax_dp(O.compute_f(msa, ignore=ign))
            gb, gmml, gbt = eng.minmax_dp(O.compute_f(msa, ignore=ign), full=True)
            assert np.array_equal(gb, b) and np.array_equal(gmml, mml) and np.array_equal(gbt, bt)


def test_group_more_partitions_than_members():
    """4 and 6 partitions on 2 members (each works its partitions off in turn, two passes), 3 on 1: the same f."""
    import founderblockgraphs_amd as F
    rng = np.random.default_rng(81)
    cases = [random_msa(rng, 40, 700), random_msa(rng, 120, 300, alphabet="ACGTN"), random_msa(rng, 64, 400, similar=0.5)]
    for members, parts in [(2, 4), (2, 6), (1, 3)]:
        with F.Group([0] * members) as grp:
            grp.set_option("partitions", parts)
            grp.set_option("plan", 1)
            for msa in cases:
                assert np.array_equal(grp.elastic_f(msa), O.compute_f(msa)), (members, parts)
                assert grp.plan_used() == ("partitioned", parts)
                assert np.array_equal(grp.repeatfree_v(msa), O.segment_v(msa)), (members, parts)
            # MSAs with gaps / ignore characters: the partitions are scanned in suffix order (gapped_rank.hip), with and
            # without the elastic tricks
            for msa, ign in [(random_msa(rng, 60, 900, gap_p=0.02, gap_run=5, n_p=0.01), "N"), (random_msa(rng, 90, 500, gap_p=0.05, gap_run=2), "")]:
                for off in (False, True):
                    want = O.compute_f(msa, ignore=ign, disable_tricks=off)
                    if off and want[0] == msa.shape[1]:
                        continue
                    assert np.array_equal(grp.elastic_f(msa, ignorechars=ign, disable_efg_tricks=off), want), (members, parts, off)
                    assert grp.plan_used() == ("partitioned", parts)


def test_group_row_pairs_and_non_elastic():
    """The row-group-pair plan (texts beyond 2^32 symbols that the partitioned index declines), forced on small inputs
    with 7 rows per pair text; v[] of segment() and of segment2elasticValid through the group."""
    import founderblockgraphs_amd as F
    rng = np.random.default_rng(5)
    with F.Group([0, 0]) as grp:
        grp.set_option("plan", 3)
        grp.set_option("pair_rows", 14)
        for (m, n, kw, ign) in [(30, 400, {}, ""), (25, 300, dict(similar=0.9), ""), (21, 350, dict(gap_p=0.03, gap_run=4, n_p=0.02), "N")]:
            msa = random_msa(rng, m, n, **kw)
            assert np.array_equal(grp.elastic_f(msa, ignorechars=ign), O.compute_f(msa, ignore=ign)), (m, n, kw)
            assert grp.plan_used()[0] == "row_pairs"
    with F.Group([0, 0, 0]) as grp:
        for (m, n, kw) in [(20, 500, {}), (33, 257, dict(similar=0.95)), (12, 900, dict(alphabet="AC"))]:
            msa = random_msa(rng, m, n, **kw)
            assert np.array_equal(grp.repeatfree_v(msa), O.segment_v(msa))
        for (m, n, kw) in [(20, 500, dict(gap_p=0.02, gap_run=3)), (16, 300, {})]:
            msa = random_msa(rng, m, n, **kw)
            assert np.array_equal(grp.gapped_v(msa), O.gapped_v(msa))


def test_group_rccl_exchange_on_one_rank():
    """option exchange=2 routes the two exchanges through librccl (dlopen, ncclCommInitAll, ncclAllGather,
    ncclAllReduce on the member's stream).  One GPU means one rank -- two ranks on one device are refused by RCCL --
    so this pins the plumbing, not the transport."""
    import founderblockgraphs_amd as F
    rng = np.random.default_rng(9)
    with F.Group([0]) as grp:
        grp.set_option("exchange", 2)
        grp.set_option("plan", 1)
        grp.set_option("partitions", 2)
        msa = random_msa(rng, 40, 600)
        assert np.array_equal(grp.elastic_f(msa), O.compute_f(msa))
        assert grp.plan_used() == ("partitioned", 2)
        grp.set_option("plan", 2)
        assert np.array_equal(grp.elastic_f(msa), O.compute_f(msa))
    with F.Group([0, 0]) as grp:
        with pytest.raises(F.FbgError):
            grp.set_option("exchange", 2)            # two members on one device: not an RCCL communicator


def _run_cli(args, devices):
    env = dict(os.environ)
    env["FBG_DEVICES"] = devices
    return subprocess.run([HOST] + args, capture_output=True, env=env, timeout=600)


@pytest.mark.parametrize("fixture", ["msa", "test", "test2", "test3", "random"])
def test_cli_two_contexts_byte_identical(tmp_path, fixture):
    """The host program on a 2-context group (FBG_DEVICES=0,0) writes the xGFA of the 1-context run and of the
    oracle's writer, byte for byte."""
    if fixture == "random":
        rng = np.random.default_rng(12)
        msa = random_msa(rng, 24, 800)
        path = str(tmp_path / "in.fasta")
        write_fasta(path, msa, [f"r{i}" for i in range(24)])
    else:
        path = os.path.join(ROOT, "tests", "golden", f"{fixture}.fasta")
        msa = None
    outs = []
    for devices in ("0", "0,0"):
        out = str(tmp_path / f"out{len(outs)}.xgfa")
        r = _run_cli(["--input", path, "--output", out, "--elastic", "--gfa", "-p"], devices)
        assert r.returncode == 0, r.stderr.decode()
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1]
    if msa is not None:
        f = O.compute_f(msa)
        b = O.minmax_dp(f)[2]
        assert outs[0] == O.write_xgfa(msa, b, str(tmp_path / "exp.xgfa"), ids=[f"r{i}" for i in range(24)])


def _blocks_ok(d_f, d_b, cnt, n):
    import torch
    b = d_b[:cnt]
    assert int(b[-1]) == n and bool((b[1:] > b[:-1]).all())
    starts = torch.cat([torch.zeros(1, dtype=torch.int64, device="cuda"), b[:-1] + 1])
    ends = torch.cat([b[:-1], torch.tensor([n - 1], device="cuda")])
    assert bool((d_f[starts] <= ends).all())             # every block is semi-repeat-free: f[start] <= end
    return int((ends - starts + 1).max())


def test_c4_full_size_eight_partitions_on_one_gpu():
    """BASELINE config 4: 1000 x 8,000,000 (text of 8.0e9 symbols), the key-range partitions that eight GPUs would hold
    one each, worked off by one device.  Checks: f in range, f[0] = 0, 8 partitions == 16 partitions, every block of the
    sweep valid against f, a 40-row sub-MSA's f never exceeds the whole MSA's, and columns [0, 40000) equal the
    row-group-pair plan's (six 4e9-symbol indexes with 32-bit positions: a different index, sort and scan)."""
    import torch
    import founderblockgraphs_amd as F
    from founderblockgraphs_amd.api import device_view
    m, n = 1000, 8_000_000
    ar = torch.arange(n, device="cuda")
    with F.Group([0]) as grp:
        grp.msa_synthetic(m, n)
        p = grp.scan_f()
        plan, parts = grp.plan_used()
        assert plan == "partitioned" and parts == 8, (plan, parts)
        eng = grp.member(0)

        def view(ptr):
            # member 0's f lives in the engine's memory and is overwritten by the next scan: keep a copy
            torch.cuda.synchronize()
            return device_view(ptr, n).clone()

        f8 = view(p)
        assert int(f8[0]) == 0 and bool((f8 >= ar).all()) and bool((f8 < n).all())
        d_b = torch.empty(n + 1, dtype=torch.int64, device="cuda")
        d_mml = torch.empty(n + 1, dtype=torch.int64, device="cuda")
        cnt = eng.minmax_dp_device(f8.data_ptr(), n, d_b.data_ptr(), d_mml.data_ptr())
        assert _blocks_ok(f8, d_b, cnt, n) == int(d_mml[n])
        grp.set_option("partitions", 16)
        f16 = view(grp.scan_f())
        assert grp.plan_used() == ("partitioned", 16)
        assert torch.equal(f8, f16)
        del f16
        # first 40,000 columns against the row-pair plan (4 groups of 250 rows, 6 pair texts of 4.0e9 symbols)
        grp.set_option("partitions", 0)
        grp.set_option("plan", 3)
        fp = view(grp.scan_f())
        assert grp.plan_used()[0] == "row_pairs"
        assert torch.equal(f8[:40000], fp[:40000]) and torch.equal(f8, fp)
        del fp
        # rows 0..39 alone: fewer suffixes to match, never a longer extension
        grp.set_option("plan", 0)
        grp.msa_synthetic(40, n)            # the generator's cell (i, j) depends on i * n + j: the first 40 rows of the same MSA
        fs = view(grp.scan_f())
        assert bool((fs <= f8).all()) and bool((fs >= ar).all())


def test_c5_full_size_column_shards():
    """BASELINE config 5: 256 x 2,000,000, 5 % of the cells in gap runs of 16, 0.1 % N, --ignore-chars=N.  The scan in 4
    column shards (compute_f_range's partition) equals the whole scan, and so does a 2-member group's; every block valid; the literal sweep
    (fbg.cpp:1968-2014 statement by statement) equals the matrix-chain sweep."""
    import torch
    import founderblockgraphs_amd as F
    m, n = 256, 2_000_000
    with F.Engine(0) as eng:
        d = torch.empty(m * n, dtype=torch.uint8, device="cuda")
        eng.msa_synthetic(d.data_ptr(), m, n, gap_fraction=0.05, gap_run=16, n_fraction=0.001)
        eng.msa_set_device(d.data_ptr(), m, n)
        eng.index_build(ignorechars="N")
        whole = torch.zeros(n, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        eng.scan_f(0, n, whole.data_ptr())
        shards = torch.zeros(n, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        for r in range(4):
            eng.scan_f(n * r // 4, n * (r + 1) // 4, shards.data_ptr())
        eng.sync()
        assert torch.equal(whole, shards)
        ar = torch.arange(n, device="cuda")
        assert int(whole[0]) == 0 and bool((whole >= ar).all()) and bool((whole < n).all())
        d_b = torch.empty(n + 1, dtype=torch.int64, device="cuda")
        d_mml = torch.empty(n + 1, dtype=torch.int64, device="cuda")
        d_bt = torch.empty(n + 1, dtype=torch.int64, device="cuda")
        cnt = eng.minmax_dp_device(whole.data_ptr(), n, d_b.data_ptr(), d_mml.data_ptr(), d_bt.data_ptr())
        assert _blocks_ok(whole, d_b, cnt, n) == int(d_mml[n])
        with eng.options(dp_literal=1):
            d_b2 = torch.empty(n + 1, dtype=torch.int64, device="cuda")
            d_mml2 = torch.empty(n + 1, dtype=torch.int64, device="cuda")
            d_bt2 = torch.empty(n + 1, dtype=torch.int64, device="cuda")
            cnt2 = eng.minmax_dp_device(whole.data_ptr(), n, d_b2.data_ptr(), d_mml2.data_ptr(), d_bt2.data_ptr())
        assert cnt2 == cnt and torch.equal(d_mml, d_mml2) and torch.equal(d_bt, d_bt2) and torch.equal(d_b[:cnt], d_b2[:cnt])
        del d_b2, d_mml2, d_bt2, shards
    # the same through the group: 2 members on this one device, each sorting and scanning half of the suffixes
    from founderblockgraphs_amd.api import device_view
    with F.Group([0, 0]) as grp:
        grp.msa_synthetic(m, n, gap_fraction=0.05, gap_run=16, n_fraction=0.001)
        p = grp.scan_f(ignorechars="N")
        assert grp.plan_used() == ("partitioned", 2)         # key-range partitions scanned in suffix order (gapped_rank.hip)
        torch.cuda.synchronize()
        assert torch.equal(device_view(p, n), whole)
