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
    # make is a no-op when everything is current, and the only thing that notices an edited header
    subprocess.run(["make", "-C", os.path.join(pkg, "csrc"), "-j8"], check=True, stdout=subprocess.DEVNULL)
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


from contextlib import contextmanager


@contextmanager
def fbg_options(engines, switches):
    """{"FBG_NO_RANKED": "1", ...} -> fbg_set_option("no_ranked", 1) on the given engine(s); the previous values come
    back on exit.  (The library does not read the environment: include/fbg_hip.h, fbg_set_option.)"""
    engines = engines if isinstance(engines, (list, tuple)) else [engines]
    kv = {k[4:].lower() if k.startswith("FBG_") else k: int(v) for k, v in switches.items()}
    old = [{k: e.get_option(k) for k in kv} for e in engines]
    try:
        for e in engines:
            for k, v in kv.items():
                e.set_option(k, v)
        yield
    finally:
        for e, o in zip(engines, old):
            for k, v in o.items():
                e.set_option(k, v)


@pytest.fixture(scope="session")
def engine():
    import founderblockgraphs_amd as F
    eng = F.Engine(0)
    yield eng
    eng.close()
