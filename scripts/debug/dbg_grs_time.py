"""k_grs_scan variants on C5 (debug switches through the gapped_rank option): stage times"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import founderblockgraphs_amd as F
m, n = 256, 2_000_000
eng = F.Engine(0)
d = torch.empty(m * n, dtype=torch.uint8, device="cuda")
eng.msa_synthetic(d.data_ptr(), m, n, gap_fraction=0.05, gap_run=16, n_fraction=0.001)
for v in [int(a) for a in sys.argv[1:]] or [0]:
    eng.set_option("gapped_rank", v)
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.msa_set_device(d.data_ptr(), m, n)
        eng.index_build(ignorechars="N")
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print("thr", eng.get_option("grs_threshold"), "redone", eng.get_option("grs_redone"), "kind", eng.get_option("index_kind"))
    print(v, round(dt * 1e3, 2), {k: round(x[0], 2) for k, x in eng.stage_ms().items()}, flush=True)
