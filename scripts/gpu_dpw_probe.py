"""The sweep of the star phylogeny with gaps alone, once per value of the option dpw_matrix given on the command line (1: the
wide matrices of rounds 2-3; even values above: timing probes of k_dpw_chain2 that skip parts of it -- wrong results); read
the kernel times from a trace in launch order (scripts/gpu_trace_order.sh).  usage: gpu_dpw_probe.py VALUE [VALUE ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import founderblockgraphs_amd as F

m, n = int(os.environ.get("FBG_STAR_ROWS", 1000)), int(os.environ.get("FBG_STAR_COLS", 200_000))
g = torch.Generator(device="cuda").manual_seed(7)
anc = torch.randint(0, 4, (n,), device="cuda", generator=g, dtype=torch.uint8)
d = torch.empty((m, n), dtype=torch.uint8, device="cuda")
lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")
for i0 in range(0, m, 50):
    i1 = min(m, i0 + 50)
    mut = torch.rand((i1 - i0, n), device="cuda", generator=g) < 0.01
    sub = torch.randint(0, 4, (i1 - i0, n), device="cuda", generator=g, dtype=torch.uint8)
    d[i0:i1] = lut[torch.where(mut, sub, anc.expand(i1 - i0, n)).long()]
for i0 in range(0, m, 50):
    i1 = min(m, i0 + 50)
    start = (torch.rand((i1 - i0, n), device="cuda", generator=g) < 0.02 / 8).float().unsqueeze(1)
    gap = torch.nn.functional.max_pool1d(torch.nn.functional.pad(start, (7, 0)), 8, 1).squeeze(1) > 0
    d[i0:i1][gap] = ord("-")
d = d.reshape(-1)
with F.Engine(0) as eng:
    d_f = torch.zeros(n, dtype=torch.int64, device="cuda")
    d_b = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    d_mml = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    eng.msa_set_device(d.data_ptr(), m, n)
    eng.index_build()
    eng.scan_f(0, n, d_f.data_ptr())
    eng.sync()
    ext = (d_f + 1 - torch.arange(n, device="cuda")).float()
    print("extensions: mean", float(ext.mean()), "max", float(ext.max()), "share above 128", float((ext > 128).float().mean()), flush=True)
    for val in map(int, sys.argv[1:]):
        eng.set_option("dpw_matrix", val)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        try:
            blocks = eng.minmax_dp_device(d_f.data_ptr(), n, d_b.data_ptr(), d_mml.data_ptr())
        except F.FbgError as e:
            blocks = str(e)
        eng.sync()
        mm = d_mml[1:].float()
        print("dpw_matrix", val, "ms", round(1e3 * (time.perf_counter() - t0), 2), "blocks", blocks, "dp_kind", eng.get_option("dp_kind"),
              "minmaxlength mean", float(mm.mean()), "max", float(mm.max()), flush=True)
