"""The wide-window sweep (dp.hip, k_dpw_*) restated in numpy at toy sizes and checked against the oracle on the CPU:
rectangular (min,max) matrices per block of B steps over the WS values before it, a chain over the blocks, and the rule
by which the chain leaves out sources that cannot win (minmaxlength grows by at most 1 per column; a source k costs at
least WS - k).  The GPU tests compare the kernels themselves with the oracle (tests/test_gpu_parity.py,
test_dp_sweep_wide_windows*); this one pins the formulation and its exactness argument."""
import numpy as np
import pytest

from oracle import pyoracle as O

INF = 0xFFFF


def _random_f(rng, n, max_ext, style):
    x = np.arange(n, dtype=np.int64)
    if style == "uniform":
        ext = rng.integers(0, max_ext + 1, n)
    elif style == "plateau":
        ends = np.sort(rng.choice(np.arange(1, n), size=max(1, n // max(2, max_ext)), replace=False))
        nxt = ends[np.minimum(np.searchsorted(ends, x, side="left"), len(ends) - 1)]
        ext = np.clip(nxt - x, 0, max_ext)
    else:
        ext = np.where(rng.random(n) < 0.03, rng.integers(0, max_ext + 1, n), rng.integers(0, 3, n))
    f = np.minimum(x + ext, n - 1)
    f[0] = 0
    return f.astype(np.uint64)


def block_matrices(f, WS, B):
    """k_dpw_blockM: M[b][t][k] = smallest longest block over the cuttings of [source k, target t) whose inner cut
    points lie in block b (source k = prefix length B*b - WS + 1 + k, target t = prefix length B*b + 1 + t)."""
    n = len(f)
    ext = np.full(n + 1, INF, dtype=np.int64)
    ext[:n] = np.minimum(f.astype(np.int64) + 1 - np.arange(n), INF)
    nblocks = (n + B - 1) // B
    M = np.full((nblocks, B, WS), INF, dtype=np.int64)
    for b in range(nblocks):
        jb = B * b
        s_ext = np.array([ext[jb + 1 + t] if jb + 1 + t < n else INF for t in range(B)])
        for k in range(WS):
            xs = jb - (WS - 1) + k
            ext_src = ext[xs] if xs >= 0 else INF
            row = np.full(B, INF, dtype=np.int64)
            for t in range(B):
                age = t + WS - k
                w = age if ext_src <= age <= WS else INF
                for tp in range(t):
                    if t - tp >= max(1, s_ext[tp]):
                        w = min(w, max(row[tp], t - tp))
                row[t] = w
            M[b, :, k] = row
    return M


def chain(f, M, WS, B, CH, G):
    """k_dpw_chain with its pruning: chunks of CH sources from the youngest down, lanes of G sources; the youngest
    chunk of a block is asked for one block early (bound L + 2B), the others with the block's own bound L + B."""
    n = len(f)
    first_valid = min(int(f[0]) + 1, n + 1)
    nblocks = M.shape[0]
    ring = np.full(WS, INF, dtype=np.int64)
    ring[WS - 1] = 0
    mml = np.zeros(n + 1, dtype=np.int64)
    flagged, read = False, 0
    L, NQ, early = 0, WS // CH, 0
    for b in range(nblocks):
        jb = B * b
        kmin_here = WS - min(L + B, WS)
        ns = NQ - min(kmin_here // CH, NQ - 1)
        acc = np.full(B, INF, dtype=np.int64)
        for sg in range(ns):
            c = NQ - 1 - sg
            kmin = early if sg == 0 else kmin_here
            for k in range(CH * c, CH * c + CH):
                if (k // G) * G + G - 1 < kmin:
                    continue
                read += 1
                acc = np.minimum(acc, np.maximum(ring[(jb + k) & (WS - 1)], M[b, :, k]))
        early = WS - min(L + 2 * B, WS)
        for t in range(B):
            j = jb + 1 + t
            ring[(j - 1) & (WS - 1)] = acc[t]
            if j <= n:
                if j < first_valid:
                    mml[j] = n + j
                else:
                    mml[j] = acc[t]
                    flagged = flagged or acc[t] >= WS
        L = acc[B - 1]
    return mml, flagged, read / (nblocks * WS)


@pytest.mark.parametrize("WS,B,CH,G", [(64, 16, 16, 4), (128, 16, 64, 8)])
def test_rectangular_chain_with_pruning_matches_the_oracle(WS, B, CH, G):
    rng = np.random.default_rng(WS)
    exact, fractions = 0, []
    for it in range(36):
        n = int(rng.integers(40, 260))
        max_ext = int(rng.integers(1, WS - 8))
        f = _random_f(rng, n, max_ext, ["uniform", "plateau", "spiky"][it % 3])
        if it % 5 == 0:               # f[0] > 0: only without the elastic tricks
            f[0] = min(n - 1, int(rng.integers(1, max_ext + 1)))
        ref = O.minmax_dp(f)[0].astype(np.int64)
        mml, flagged, frac = chain(f, block_matrices(f, WS, B), WS, B, CH, G)
        if flagged:                   # a value reached the window: the engine tries the next size / the literal sweep
            assert int(ref[1:].max()) >= WS or int(f[0]) > 0
            continue
        assert np.array_equal(mml, ref), (it, n, max_ext, np.flatnonzero(mml != ref)[:5])
        exact += 1
        fractions.append(frac)
    assert exact >= 24
    assert min(fractions) < 0.6       # the pruning does leave sources out
