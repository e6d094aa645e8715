"""CPU tests of host-side logic that needs no device: the wide-position row arithmetic of the rank-order scan
(rank_common.h, rs_rem<FBG_SLOTS_WIDE>), restated with Python integers at the sizes a GPU test cannot reach, and the
group API's behaviour on a box without a GPU."""
import ctypes as C

import pytest


def column_of(p, d):
    """rs_rem<FBG_SLOTS_WIDE>: p mod d by one multiply with floor((2^64 - 1) / d) and one conditional correction."""
    magic = ((1 << 64) - 1) // d
    c = p - ((p * magic) >> 64) * d
    if c >= d:
        c -= d
    return c


@pytest.mark.parametrize("m,n", [(4096, 200_000_000), (64, 1_500_000_000), (1000, 8_000_000), (3, 2), (1, (1 << 31) - 2),
                                 (7, 1 << 20), (255, (1 << 32) // 255)])
def test_wide_row_arithmetic_is_exact(m, n):
    """ADVICE (round 1): the single-multiply form with ceil(2^64 / d) was wrong at p = k*d - 1 near 2^40 (m = 4096,
    n = 2e8).  The floor form with a correction is exact for every 64-bit p and every d >= 2."""
    d = n + 1
    N = m * d + 1
    probes = {0, 1, d - 1, d, d + 1, N - 1, N - 2, (1 << 40) - 1, (1 << 64) - 1}
    for k in (1, 2, m // 2, m - 1, m, (1 << 40) // d, (1 << 63) // d, ((1 << 64) - 1) // d):
        for delta in (-1, 0, 1):
            probes.add(k * d + delta)
    for p in probes:
        if 0 <= p < (1 << 64):
            assert column_of(p, d) == p % d, (m, n, p)


def test_group_needs_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import founderblockgraphs_amd as F
    with pytest.raises(F.FbgError) as ei:
        F.Group([0, 0])
    assert ei.value.code == 6 and "no CPU fallback" in str(ei.value)
    from founderblockgraphs_amd import _lib
    L = _lib.lib()
    assert L.fbg_group_size(None) == 0 and L.fbg_group_member(None, 0) is None
    assert L.fbg_host_alloc(1 << 20) is None          # pinned memory comes from the HIP runtime: none without a device


def test_partition_plan_helpers():
    """distributed.py: shard ranges cover [0, n) without overlap; the row-pair plan covers every pair of groups."""
    from founderblockgraphs_amd import distributed as D
    for n, w in [(10, 3), (1_000_000, 8), (7, 8), (1, 1)]:
        cuts = [D.shard_range(n, r, w) for r in range(w)]
        assert cuts[0][0] == 0 and cuts[-1][1] == n and all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
    G, groups, plan = D.plan_row_pairs(1000, 8_000_000, 8)
    assert G == 4 and sorted(p for r in plan for p in r) == D.group_pairs(4)
    assert all((b1 - a1) + (b2 - a2) == 500 for (a1, b1) in groups for (a2, b2) in groups if (a1, b1) != (a2, b2))
