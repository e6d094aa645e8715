import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def random_msa(rng, m, n, alphabet="ACGT", gap_p=0.0, gap_run=1, similar=0.0, n_p=0.0):
    """Random MSA as (m, n) uint8.  similar>0: rows are noisy copies of one ancestor
    (long runs of consecutive suffix ranks); gap_p: fraction of cells that start a gap run."""
    alpha = np.frombuffer(alphabet.encode(), dtype=np.uint8)
    if similar > 0:
        anc = alpha[rng.integers(0, len(alpha), n)]
        a = np.tile(anc, (m, 1))
        mut = rng.random((m, n)) > similar
        a[mut] = alpha[rng.integers(0, len(alpha), int(mut.sum()))]
    else:
        a = alpha[rng.integers(0, len(alpha), (m, n))]
    a = a.copy()
    if n_p > 0:
        a[rng.random((m, n)) < n_p] = ord("N")
    if gap_p > 0:
        starts = np.argwhere(rng.random((m, n)) < gap_p)
        for i, j in starts:
            a[i, j:j + gap_run] = ord("-")
    return a


@pytest.fixture(scope="session")
def engine():
    import founderblockgraphs_amd as F
    eng = F.Engine(0)
    yield eng
    eng.close()
