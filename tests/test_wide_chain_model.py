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


# ---- round 4: matrices for the block before only, the older sources in closed form (k_dpw_blockY / k_dpw_chain2) ----
#
# A source that lies more than B columns before a block can only enter it by ONE block that is longer than everything
# that follows inside (inner blocks are at most B - 1 long), so for such a source the cost of reaching target t is
# max(value, c - x) for the first inner cut point c from which t can be reached at all: what the block has to know about
# its inside is a B x B bit matrix ("t is reachable from c"), and G[c] = min over the old sources that may end a block at
# c of max(value, c - x) is a lower envelope that needs no matrix: a source gives its value while that
# dominates its age and the age from then on: three B-entry tables (values of sources valid from the block's first column on, by
# the last step they dominate; ages, by the first step they do; values of sources that become valid inside the block and
# dominate to its end, by that step) read as running minima, and a list for the values that dominate on a part of the block only.

def young_and_reach(f, B):
    n = len(f)
    ext = np.full(n + 1, INF, dtype=np.int64)
    ext[:n] = np.minimum(f.astype(np.int64) + 1 - np.arange(n), INF)
    nblocks = (n + B - 1) // B
    My = np.full((nblocks, B, B), INF, dtype=np.int64)
    reach = np.zeros((nblocks, B, B), dtype=bool)       # reach[b][t][c]: target t from inner cut point c < t
    for b in range(nblocks):
        jb = B * b
        s_ext = np.array([ext[jb + 1 + t] if jb + 1 + t < n else INF for t in range(B)])
        for k in range(2 * B):                           # B sources of the block before, then the block's own columns
            xs = jb - (B - 1) + k
            ext_src = ext[xs] if 0 <= xs <= n else INF
            row = np.full(B, INF, dtype=np.int64)
            for t in range(B):
                age = t + B - k
                w = age if (age >= 1 and ext_src <= age) else INF
                for tp in range(t):
                    if t - tp >= max(1, s_ext[tp]):
                        w = min(w, max(row[tp], t - tp))
                row[t] = w
            if k < B:
                My[b, :, k] = row
            else:
                reach[b, :, k - B] = row < INF
    return My, reach, ext


def chain2(f, My, reach, ext, WS, B):
    n = len(f)
    first_valid = min(int(f[0]) + 1, n + 1)
    nblocks = My.shape[0]
    val = {0: 0}                                         # prefix length -> minmaxlength (the LDS rings of the kernel)
    mml = np.zeros(n + 1, dtype=np.int64)
    flagged, looked = False, 0
    L = 0
    for b in range(nblocks):
        jb = B * b
        acc = np.full(B, INF, dtype=np.int64)
        for k in range(B):                               # the block before, through its matrix
            x = jb - (B - 1) + k
            if x in val:
                acc = np.minimum(acc, np.maximum(val[x], My[b, :, k]))
        pfx = np.full(B, INF, dtype=np.int64)            # by the last step at which the value dominates (valid from step 0)
        slope = np.full(B, INF, dtype=np.int64)          # by the first step at which the age dominates: age at step 0
        csfx = np.full(B, INF, dtype=np.int64)           # by the first valid step, for values that dominate to the block's end
        part = []                                        # (lo, rr, v): valid from lo > 0, the value dominates on [lo, rr], rr < B - 1
        for a0 in range(B + 1, min(WS, L + B) + 1):     # older sources, youngest first; beyond L + B none can win
            x = jb + 1 - a0
            if x < 0 or val.get(x, INF) >= INF:
                continue
            looked += 1
            v, e = val[x], ext[x]
            lo = max(0, e - a0)                          # first step of the block at which [x, c) is long enough
            if lo >= B:
                continue
            r = v - a0                                   # the value dominates up to step r, the age from r + 1 on
            if lo == 0:
                if r >= 0:
                    pfx[min(r, B - 1)] = min(pfx[min(r, B - 1)], v)
            elif r >= B - 1:
                csfx[lo] = min(csfx[lo], v)
            elif r >= lo:
                part.append((lo, r, v))
            if r + 1 <= B - 1:
                s0 = max(lo, r + 1)
                slope[s0] = min(slope[s0], a0)
        G = np.full(B, INF, dtype=np.int64)
        for t in range(B):
            g = min(pfx[t:].min(), t + slope[:t + 1].min(), csfx[:t + 1].min())
            for lo, r, v in part:
                if lo <= t <= r:
                    g = min(g, v)
            G[t] = min(g, INF)
        for t in range(B):
            best = G[t]
            for c in range(t):
                if reach[b, t, c]:
                    best = min(best, G[c])
            acc[t] = min(acc[t], best)
        for t in range(B):
            j = jb + 1 + t
            if acc[t] < INF:
                val[j] = int(acc[t])
            if j <= n:
                if j < first_valid:
                    mml[j] = n + j
                else:
                    mml[j] = acc[t]
                    flagged = flagged or acc[t] >= WS
        L = int(acc[B - 1])
    return mml, flagged, looked / max(1, nblocks * (WS - B))


@pytest.mark.parametrize("WS,B", [(64, 16), (128, 16), (256, 32)])
def test_closed_form_for_old_sources_matches_the_oracle(WS, B):
    rng = np.random.default_rng(WS + B)
    exact, fractions = 0, []
    for it in range(48):
        n = int(rng.integers(40, 400))
        max_ext = int(rng.integers(1, WS - 8))
        f = _random_f(rng, n, max_ext, ["uniform", "plateau", "spiky"][it % 3])
        if it % 5 == 0:
            f[0] = min(n - 1, int(rng.integers(1, max_ext + 1)))
        ref = O.minmax_dp(f)[0].astype(np.int64)
        My, reach, ext = young_and_reach(f, B)
        mml, flagged, frac = chain2(f, My, reach, ext, WS, B)
        if flagged:
            assert int(ref[1:].max()) >= WS or int(f[0]) > 0
            continue
        assert np.array_equal(mml, ref), (it, n, max_ext, np.flatnonzero(mml != ref)[:5])
        exact += 1
        fractions.append(frac)
    assert exact >= 30
    assert min(fractions) < 0.6
