"""a small gap-free MSA through index_build + elastic f against the oracle (first thing to run after an engine change)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import founderblockgraphs_amd as F
from oracle import pyoracle as O
rng = np.random.default_rng(1)
msa = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=(4, 40))
with F.Engine(0) as eng:
    print("engine up", flush=True)
    eng.msa_load_host(msa)
    print("loaded", flush=True)
    eng.index_build()
    print("index built", flush=True)
    f = eng.elastic_f(msa)
    print("f equal:", np.array_equal(f, O.compute_f(msa)), flush=True)
