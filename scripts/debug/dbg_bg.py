import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import pyoracle as O
import founderblockgraphs_amd as F
msa = O.msa_array(["AGCGA-CTAGATAC", "AGC--ACTAGTT--", "AGCGA-CTCGTTAC", "AGC--ACT-GTTAC"])
eng = F.Engine(0)
f = eng.elastic_f(msa)
b = eng.minmax_dp(f)
print("f", f, "b", b)
node_of, first, rep_row, ecount, edges = eng.block_graph(b)
print("first", first); print("node_of", node_of); print("rep", rep_row); print("ecount", ecount); print("edges", [[hex(int(x)) for x in r] for r in edges])
