import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
from conftest import random_msa
from oracle import pyoracle as O
import founderblockgraphs_amd as F

rng = np.random.default_rng(4242)
cases = [random_msa(rng, 20, 900, similar=0.96), random_msa(rng, 64, 400), random_msa(rng, 7, 1500, alphabet="AC", similar=0.99)]
msa = cases[2]
eng = F.Engine(0)
for env in ({}, {"FBG_FULL_KEYS": "1"}, {"FBG_FULL_KEYS": "1", "FBG_NO_PACKED": "1"}):
    for k in ("FBG_FULL_KEYS", "FBG_NO_PACKED"):
        os.environ.pop(k, None)
    os.environ.update(env)
    f = O.compute_f(msa)
    g = eng.elastic_f(msa)
    bad = np.nonzero(f != g)[0]
    print(env, "f mismatches:", len(bad), bad[:10], f[bad[:10]], g[bad[:10]])
    T, SA, ISA, LCP = O.msa_index(msa)
    eng.msa_load_host(msa); eng.index_build()
    gT, gSA, gISA, gPL, gPR = eng.index_download()
    b2 = np.nonzero(gSA.astype(np.int64) != SA.astype(np.int64))[0]
    print("   SA mismatches:", len(b2), b2[:10])
    lcp_ext = np.concatenate([LCP, [0]]).astype(np.int64)
    b3 = np.nonzero(gPL.astype(np.int64) != lcp_ext[ISA])[0]
    print("   PL mismatches:", len(b3), b3[:10], gPL[b3[:5]], lcp_ext[ISA][b3[:5]])
    print("   stages", {k: v[1] for k, v in eng.stage_ms().items()})
