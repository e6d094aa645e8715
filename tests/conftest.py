import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The library, the host programs and the oracle are build products (git-ignored): a checkout that has not seen
    __graft_entry__.build() yet gets them built here (hipcc cross-compiles without a GPU)."""
    import subprocess
    pkg = os.path.join(ROOT, "founderblockgraphs_amd")
    need = [os.path.join(pkg, "libfbg_hip.so"), os.path.join(pkg, "founderblockgraph"), os.path.join(pkg, "fbg_options_dump")]
    if not all(os.path.exists(f) for f in need):
        subprocess.run(["make", "-C", os.path.join(pkg, "csrc"), "-j8"], check=True, stdout=subprocess.DEVNULL)
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, stdout=subprocess.DEVNULL)


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
