#!/usr/bin/env python3
"""bench.py -- MSA columns segmented per second on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one synthetic MSA already resident in HBM:
    index build: text, key sort, extension scan in rank order      -> load_cst + compute_f   fbg.cpp:361-436, 1579-1695
    f[] from the per-column maxima                                  -> fbg.cpp:1656-1681
    [N>1: an all-gather of 1.5 KB per rank and one all-reduce(max) of n+1 words, see below]
    min-max-length sweep + backtrack (rank 0)                       -> fbg.cpp:1940-2039
Workload at N=1: BASELINE config C3, synthetic 1000 rows x 1,000,000 columns, iid ACGT
(SURVEY.md 8d generator), --elastic.  N>1: weak scaling, 1,000,000 columns per GPU: every rank holds
the whole MSA and text but sorts and scans only its own key range of the suffixes (key-range
partitioned index, founderblockgraphs_amd/distributed.py; texts of 2^32 symbols and more -- N >= 5 --
included).  If a rank finds its input unsuitable all ranks fall back together: to the replicated
index with column shards (--replicated-index forces it) while the text fits 32-bit
positions, else to the exact row-group-pair plan (one all-reduce(max) of f).

The sweep of step i runs on a context and stream of its own on rank 0, beside the index build of
step i+1 (it is a chain of small launches that would leave the GPU -- at N>1 all GPUs -- idle); all
of it is joined before the closing barrier, inside the timed region.  --serial-sweep switches that off.

Prints ONE JSON line on rank 0.  `roofline` prices the kernel of the step that takes longest among those
with HIP-event brackets of their own: the extension scan (k_rank_scan_lean / k_rank_scan; k_scan_stream when
the MSA has gaps) with (13*m + 8) bytes per column, the three passes of the MSD sort with
the bytes a pass must move (9 / 16 / 16 per suffix); `roofline.kernels` lists all of them.  `cpu_baseline` times the CPU restatement (oracle/, kind "port") on a bounded column prefix
of the same MSA.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED, SEED2, SEED3 = 0x5EED0001, 0x5EED0002, 0x5EED0003
HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec


def splitmix64(x):
    x = x + np.uint64(0x9E3779B97F4A7C15)
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def synth_msa_host(m, n_total, cols, seed=SEED):
    """Columns [0, cols) of the synthetic m x n_total MSA, on the host."""
    i = np.arange(m, dtype=np.uint64)[:, None]
    j = np.arange(cols, dtype=np.uint64)[None, :]
    with np.errstate(over="ignore"):
        h = splitmix64(np.uint64(seed) + i * np.uint64(n_total) + j)
    return np.frombuffer(b"ACGT", dtype=np.uint8)[(h >> np.uint64(62)).astype(np.int64)]


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def physical_cores():
    """(logical CPUs, physical cores) of the box, from /proc/cpuinfo -- stated next to the cores the baseline may use."""
    logical, phys = os.cpu_count() or 1, set()
    try:
        pid = cid = None
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("physical id"):
                    pid = line.split(":", 1)[1].strip()
                elif line.startswith("core id"):
                    cid = line.split(":", 1)[1].strip()
                elif not line.strip():
                    if pid is not None and cid is not None:
                        phys.add((pid, cid))
                    pid = cid = None
    except OSError:
        pass
    return logical, (len(phys) or logical)


def host_cores():
    """Cores this process may really use: the affinity mask, cut down to the cgroup's CPU quota if there is one."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()
        if quota != "max":
            cores = max(1, min(cores, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return cores


def cpu_baseline(m, n_total, sample_cols):
    """Oracle (CPU restatement) on a column prefix, both legs of SURVEY.md 8(d): 1 thread = the reference's default
    --threads=-1 (fbg.cpp:3392-3393), and T threads = all host cores with the reference's partition (ranges of
    floor(n/T)+1 columns, fbg.cpp:2278-2284; index build and sweep stay single-threaded there as here)."""
    from oracle import pyoracle as O
    lib = None
    try:   # the reference builds with -march=native (Makefile:3); rebuild the port that way on this host
        out = os.path.join("/tmp", f"liboracle_native_{os.getpid()}.so")
        O.build(march="native", out=out)
        lib = O.lib(out)
    except Exception:
        lib = O.lib()
    msa = synth_msa_host(m, n_total, sample_cols)
    legs = {}
    for threads in (1, host_cores()):
        t = {}
        t0 = time.perf_counter()
        f = O.compute_f(msa, threads=threads, timings=t, library=lib)
        O.minmax_dp(f, library=lib)
        dt = time.perf_counter() - t0
        legs[threads] = (dt, t)
        if host_cores() == 1:
            break
    dt1, t1 = legs[1]
    T = max(legs)
    dtT, tT = legs[T]
    logical, phys = physical_cores()
    return {
        "value": sample_cols / dt1, "unit": "columns/s", "cores": 1, "kind": "port", "cpu_model": cpu_model(),
        "host": {"logical_cpus": logical, "physical_cores": phys, "usable_by_this_process": host_cores(),
                 "note": "usable = affinity mask cut to the cgroup quota; the threaded leg runs on that many threads"},
        "sample": f"first {sample_cols} columns of the same {m}-row synthetic MSA, elastic, "
                  f"index {t1.get('index_s', 0):.2f}s + scan {t1.get('scan_s', 0):.2f}s + DP, {dt1:.2f}s total",
        "threaded": {"value": sample_cols / dtT, "unit": "columns/s", "cores": T,
                     "sample": f"same sample, scan over {T} threads (column ranges of floor(n/T)+1): index {tT.get('index_s', 0):.2f}s "
                               f"+ scan {tT.get('scan_s', 0):.2f}s + DP, {dtT:.2f}s total"},
    }


def boundary_and_latency(F, torch, dev, m, n, d_msa):
    """What the JSON `value` does not show: (a) the host-buffer boundary -- fbg_elastic_f + fbg_minmax_dp,
    what INTEGRATION.md binds: MSA and f cross PCIe -- first call of a fresh context (workspaces allocated) and later
    calls, from pageable memory and from memory of fbg_host_alloc; (b) the latency of ONE job with the sweep not
    overlapped with a next job's index build."""
    import ctypes as C
    from founderblockgraphs_amd import _lib
    out = {}
    host = d_msa.view(m, n).cpu().numpy()                  # pageable
    with F.Engine(dev) as eng:
        ms = []
        for _ in range(3):
            t0 = time.perf_counter()
            f = eng.elastic_f(host)
            eng.minmax_dp(f)
            ms.append(1e3 * (time.perf_counter() - t0))
        out["boundary_ms"] = {"first_call": ms[0], "warm": min(ms[1:]), "host_memory": "pageable (library stages it through pinned buffers)"}
        # SURVEY.md 8(d)'s T_seg: MSA resident in host RAM -> boundaries on the host (index build + scan + sweep, the copies
        # over PCIe included), warm context; columns / s
        out["t_seg"] = {"definition": "n / T_seg, T_seg = MSA in host RAM -> boundaries in host RAM through the C ABI "
                                      "(fbg_elastic_f + fbg_minmax_dp), warm context",
                        "unit": "columns/s", "pageable": n / (1e-3 * min(ms[1:]))}
        L = _lib.lib()
        p = L.fbg_host_alloc(m * n)
        if p:
            pinned = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(m, n))
            np.copyto(pinned, host)
            ms = []
            for _ in range(2):
                t0 = time.perf_counter()
                f = eng.elastic_f(pinned)
                eng.minmax_dp(f)
                ms.append(1e3 * (time.perf_counter() - t0))
            out["boundary_ms"]["warm_pinned"] = min(ms)
            out["t_seg"]["pinned"] = n / (1e-3 * min(ms))
            del pinned
            L.fbg_host_free(C.c_void_p(p))
        del host
        # one job, start to boundaries on the device, nothing overlapped
        d_f = torch.zeros(n, dtype=torch.int64, device="cuda")
        d_b = torch.empty(n + 1, dtype=torch.int64, device="cuda")
        eng.msa_set_device(d_msa.data_ptr(), m, n)
        ms = []
        for _ in range(3):
            d_f.zero_()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.index_build()
            eng.scan_f(0, n, d_f.data_ptr())
            eng.minmax_dp_device(d_f.data_ptr(), n, d_b.data_ptr())
            eng.sync()
            ms.append(1e3 * (time.perf_counter() - t0))
        out["latency_ms"] = min(ms[1:])
    return out


def other_workloads(F, torch, dev):
    """The inputs that leave the headline's fast path, so that the JSON line shows them next to it (not part of
    `value`): similar rows (star phylogeny, one iid ancestor, every cell substituted with p = 0.01), the same with gaps
    (the slow path that is left) and BASELINE config 5 (gap runs + N, --ignore-chars=N)."""
    res = []

    def run(name, m, n, d, ignore=""):
        with F.Engine(dev) as eng:
            d_f = torch.zeros(n, dtype=torch.int64, device="cuda")
            d_b = torch.empty(n + 1, dtype=torch.int64, device="cuda")
            eng.msa_set_device(d.data_ptr(), m, n)
            best, stages, blocks = None, None, None
            for _ in range(3):
                d_f.zero_()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                eng.index_build(ignorechars=ignore)
                eng.scan_f(0, n, d_f.data_ptr())
                blocks = eng.minmax_dp_device(d_f.data_ptr(), n, d_b.data_ptr())
                eng.sync()
                dt = time.perf_counter() - t0
                if best is None or dt < best:
                    best, stages = dt, {k: round(v[0], 3) for k, v in eng.stage_ms().items()}
            res.append({"workload": name, "rows": m, "cols": n, "ms_per_step": 1e3 * best, "columns_per_s": n / best,
                        "blocks": blocks, "stages_ms": stages, "index_kind": eng.get_option("index_kind"),
                        "span_scan_used": eng.get_option("span_scan_used"), "span_key_flags_used": eng.get_option("span_key_flags_used"),
                        "dp_kind": eng.get_option("dp_kind")})

    m, n = 1000, 200_000
    g = torch.Generator(device="cuda").manual_seed(7)
    anc = torch.randint(0, 4, (n,), device="cuda", generator=g, dtype=torch.uint8)
    d = torch.empty((m, n), dtype=torch.uint8, device="cuda")
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")
    for i0 in range(0, m, 50):
        i1 = min(m, i0 + 50)
        mut = torch.rand((i1 - i0, n), device="cuda", generator=g) < 0.01
        sub = torch.randint(0, 4, (i1 - i0, n), device="cuda", generator=g, dtype=torch.uint8)
        d[i0:i1] = lut[torch.where(mut, sub, anc.expand(i1 - i0, n)).long()]
    run("star phylogeny 1000 x 200000, p = 0.01 (similar rows), --elastic", m, n, d.reshape(-1))
    # the same rows with 2 % of the cells in gap runs of 8 -- what a pangenome MSA looks like: the group-level scan on column
    # spans (span_scan.hip; round 2: per-position records with six doubling rounds, 158 ms); the extensions (a row's string
    # behind a deletion occurs in the other rows, eight columns on) reach hundreds of columns: the 16-bit wide-window sweep
    for i0 in range(0, m, 50):
        i1 = min(m, i0 + 50)
        start = (torch.rand((i1 - i0, n), device="cuda", generator=g) < 0.02 / 8).float().unsqueeze(1)
        gap = torch.nn.functional.max_pool1d(torch.nn.functional.pad(start, (7, 0)), 8, 1).squeeze(1) > 0
        d[i0:i1][gap] = ord("-")
    run("star phylogeny 1000 x 200000, p = 0.01, 2 % gap cells in runs of 8 (similar rows WITH gaps), --elastic", m, n, d.reshape(-1))
    del d
    m, n = 256, 2_000_000
    d = torch.empty(m * n, dtype=torch.uint8, device="cuda")
    with F.Engine(dev) as eng:
        eng.msa_synthetic(d.data_ptr(), m, n, gap_fraction=0.05, gap_run=16, n_fraction=0.001)
        eng.sync()
    run("BASELINE config 5: synthetic 256 x 2000000, 5% gap runs of 16 + 0.1% N, --elastic --ignore-chars=N", m, n, d, ignore="N")
    del d
    # similar rows with gaps beyond 2^30 cells (2 * 10^9): the group-level scan with the slots' flags in the key word
    # (span_key_flags_used; rounds 1-3: the record path, seconds)
    m, n = 1000, 2_000_000
    g = torch.Generator(device="cuda").manual_seed(7)
    anc = torch.randint(0, 4, (n,), device="cuda", generator=g, dtype=torch.uint8)
    d = torch.empty((m, n), dtype=torch.uint8, device="cuda")
    for i0 in range(0, m, 5):
        i1 = min(m, i0 + 5)
        mut = torch.rand((i1 - i0, n), device="cuda", generator=g) < 0.01
        sub = torch.randint(0, 4, (i1 - i0, n), device="cuda", generator=g, dtype=torch.uint8)
        d[i0:i1] = lut[torch.where(mut, sub, anc.expand(i1 - i0, n)).long()]
        start = (torch.rand((i1 - i0, n), device="cuda", generator=g) < 0.02 / 8).float().unsqueeze(1)
        gap = torch.nn.functional.max_pool1d(torch.nn.functional.pad(start, (7, 0)), 8, 1).squeeze(1) > 0
        d[i0:i1][gap] = ord("-")
        del mut, sub, start, gap
    torch.cuda.empty_cache()
    run("star phylogeny 1000 x 2000000, p = 0.01, 2 % gap cells in runs of 8 (similar rows WITH gaps, 2 * 10^9 cells), --elastic", m, n, d.reshape(-1))
    return res


def measured_traffic(m, n, kernel):
    """HBM bytes per launch of the scan kernel from the committed rocprofv3 PMC passes (profiles/, collected as
    MI355X_MICROARCH.md prescribes: separate --pmc passes, KiB units, FETCH_SIZE doubled on gfx950); None when no pass
    exists for this workload / kernel."""
    for name in ("r04_pmc_kernels.json", "r03_pmc_kernels.json", "r02_pmc_kernels.json", "r01_pmc_scan_kernels.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as fh:
                d = json.load(fh)
            if d["workload"] == {"rows": m, "cols": n}:
                for k, v in d["kernels"].items():
                    if (k == kernel or k.startswith(kernel + "<")) and "hbm_bytes_per_launch" in v:
                        return v["hbm_bytes_per_launch"], f"profiles/{name}"
        except Exception:
            continue
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows", type=int, default=1000)
    ap.add_argument("--cols-per-gpu", type=int, default=1_000_000)
    ap.add_argument("--cpu-sample-cols", type=int, default=100_000, help="column prefix the CPU port is timed on (SURVEY.md 8d: n' >= 10^5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip boundary_ms / latency_ms / other_workloads (N=1 extras outside the timed region)")
    ap.add_argument("--force-row-pairs", type=int, default=0, help="debug: use the row-group-pair plan with this many rows per pair text")
    ap.add_argument("--replicated-index", action="store_true", help="N>1: every rank builds the whole index (column shards) instead of one key range of it")
    ap.add_argument("--backend", default="nccl", help="nccl (RCCL, default) | gloo (debug: several ranks on one GPU)")
    ap.add_argument("--serial-sweep", action="store_true", help="debug: run the sweep of a step before the next step starts (no second stream)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import founderblockgraphs_amd as F

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    ndev = torch.cuda.device_count()
    dev = local_rank % max(1, ndev) if args.backend == "gloo" else local_rank
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        from datetime import timedelta
        # a rank that dies leaves the others in a collective: bounded, so that the job ends with an error instead of hanging
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev), timeout=timedelta(seconds=300))
        else:
            dist.init_process_group("gloo", timeout=timedelta(seconds=300))

    from founderblockgraphs_amd import distributed as D
    m, n = args.rows, args.cols_per_gpu * world
    eng = F.Engine(dev)
    # one stream for everything: the engine's kernels, torch's fills and the RCCL exchange
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    eng.set_stream(stream.cuda_stream)

    d_fs = [torch.zeros(n, dtype=torch.int64, device="cuda") for _ in range(2)]
    d_b = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    state = {"step": 0}

    # The sweep of a step (sort of the extensions, min-max DP, backtrack: fbg.cpp:1940-2039) is a chain of small,
    # latency-bound launches on rank 0.  It runs on a context and stream of its own, beside the index build of the
    # NEXT step (at N > 1 the other ranks would otherwise idle through it); f is double-buffered, and everything is
    # joined before the closing fence, so all of its work lies inside the timed region.
    class Sweeper:
        def __init__(self):
            from concurrent.futures import ThreadPoolExecutor
            self.eng = F.Engine(dev)
            self.stream = torch.cuda.Stream()
            self.eng.set_stream(self.stream.cuda_stream)
            self.pool = ThreadPoolExecutor(max_workers=1)
            self.pending = None

        def submit(self, d_f):
            self.join()
            ev = torch.cuda.Event()
            ev.record(stream)
            self.stream.wait_event(ev)
            # ctypes drops the GIL for the duration of the call: the main thread goes on to the next step
            self.pending = self.pool.submit(self.eng.minmax_dp_device, d_f.data_ptr(), n, d_b.data_ptr())

        def join(self):
            if self.pending is not None:
                state["blocks"] = self.pending.result()
                self.pending = None

    sweeper = Sweeper() if rank == 0 and not args.serial_sweep else None

    def sweep(d_f):
        if rank != 0:
            return
        if sweeper is not None:
            sweeper.submit(d_f)
        else:
            state["blocks"] = eng.minmax_dp_device(d_f.data_ptr(), n, d_b.data_ptr())
    fits32 = m * (n + 1) + 1 < (1 << 32) - 1
    want_partition = world > 1 and not args.replicated_index and not args.force_row_pairs
    x0, x1 = D.shard_range(n, rank, world)
    bufs = {}

    def whole_msa():
        if "whole" not in bufs:
            bufs["whole"] = torch.empty(m * n, dtype=torch.uint8, device="cuda")
            eng.msa_synthetic(bufs["whole"].data_ptr(), m, n, SEED)
        return bufs["whole"]

    def step_partitioned():
        """Each rank sorts and scans one key range of the suffixes (any text length); False = declined by some rank."""
        d_msa = whole_msa()
        eng.msa_set_device(d_msa.data_ptr(), m, n)
        if not D.partitioned_index(eng, n, rank, world):
            return False
        if rank == 0:
            d_f = d_fs[state["step"] & 1]
            d_f.zero_()
            eng.scan_f(0, n, d_f.data_ptr())
            sweep(d_f)
        return True

    def step_replicated():
        """Column shards of one (replicated) index: SURVEY.md 8e, compute_f_range's partition."""
        d_msa = whole_msa()
        eng.msa_set_device(d_msa.data_ptr(), m, n)
        eng.index_build()
        d_f = d_fs[state["step"] & 1]
        d_f.zero_()
        eng.scan_f(x0, x1, d_f.data_ptr())
        f_full = D.all_gather_columns(d_f[x0:x1], n, rank, world)   # the one exchange of the path
        if f_full.data_ptr() != d_f.data_ptr():
            d_f.copy_(f_full)                                       # the sweep reads the step's own buffer
        sweep(d_f)

    def step_row_pairs():
        """Text too long for 32-bit positions on one GPU: exact row-group-pair plan, one all-reduce(max) of f."""
        if "plan" not in bufs:
            bufs.pop("whole", None)
            G, groups, plan = (D.plan_row_pairs(m, n, world, limit=args.force_row_pairs * (n + 1) + 1)
                               if args.force_row_pairs else D.plan_row_pairs(m, n, world))
            rows_pair = max((groups[a][1] - groups[a][0]) + (groups[b][1] - groups[b][0]) for a, b in D.group_pairs(G))
            bufs["plan"] = (G, groups, plan[rank], rows_pair, torch.empty(rows_pair * n, dtype=torch.uint8, device="cuda"))
        G, groups, mine, rows_pair, d_msa = bufs["plan"]
        d_f = d_fs[state["step"] & 1]
        d_f.zero_()
        for a, b in mine:
            off = 0
            for g in (a, b):
                r0, r1 = groups[g]
                eng.msa_synthetic(d_msa.data_ptr() + off * n, r1 - r0, n, SEED + r0 * n)
                off += r1 - r0
            eng.msa_set_device(d_msa.data_ptr(), off, n)
            eng.index_build()
            eng.scan_f(0, n, d_f.data_ptr())
        D.all_reduce_max(d_f)
        sweep(d_f)

    state["path"] = "partitioned" if want_partition else ("replicated" if fits32 and not args.force_row_pairs else "row_pairs")

    def step():
        state["step"] += 1
        if state["path"] == "partitioned":
            if step_partitioned():
                return
            state["path"] = "replicated" if fits32 else "row_pairs"     # every rank sees the same verdict
        if state["path"] == "replicated":
            step_replicated()
        else:
            step_row_pairs()

    def fence():
        if sweeper is not None:
            sweeper.join()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    stage_acc = {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        stream.synchronize()                       # this step's own stream only: the sweep may still be running
        for k, (ms, ln) in eng.stage_ms().items():
            a = stage_acc.setdefault(k, [0.0, 0])
            a[0] += ms
            a[1] += ln
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        ms_per_step = 1e3 * dt / args.steps
        # the extension scan is k_rank_scan (rank order, inside the index build) for gap-free MSAs,
        # k_scan_stream (text order) otherwise; one launch per step either way
        ranked = stage_acc.get("rank_kernel", [0.0, 0])[1] > 0
        scan_total_ms, scan_launches = stage_acc["rank_kernel"] if ranked else stage_acc["scan"]
        scan_launches = max(1, scan_launches if ranked else args.steps)      # a large rank-order scan goes out in pieces (rank_scan.hip)
        scan_ms = scan_total_ms / scan_launches                             # average duration of ONE launch of the kernel
        launches_per_step = scan_launches / max(1, args.steps)
        if state["path"] == "partitioned":
            scan_cols, scan_rows, mode_used = x1 - x0, m, "key-range partitioned index, all-reduce(max) of the column maxima"
        elif state["path"] == "replicated":
            scan_cols, scan_rows, mode_used = x1 - x0, m, "column shards, replicated index"
        else:
            G, _, _, rows_pair, _ = bufs["plan"]
            scan_cols, scan_rows = n, rows_pair
            mode_used = f"row-group pairs (G={G}, {len(D.group_pairs(G))} pairs), all-reduce(max)"
        scan_bytes = (13 * scan_rows + 8) * scan_cols / launches_per_step    # algorithmic bytes one launch covers
        scan_kernel = "k_scan_stream" if not ranked else ("k_rank_scan_lean" if (world == 1 and eng.get_option("rank_no_lean") == 0) else "k_rank_scan")
        traffic, traffic_src = measured_traffic(scan_rows, scan_cols, scan_kernel)
        achieved = scan_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
        # The kernels with HIP-event brackets of their own: the extension scan, priced with SURVEY.md 8(d)'s
        # (13 m + 8) bytes per column, and the three passes of the MSD sort, priced with the bytes a sort pass cannot avoid
        # (pass 1: 1 text byte read + 8 written per suffix; passes 2 and 3: 8 read + 8 written).  `roofline` is the one
        # that takes longest; `roofline.kernels` has them all.
        slots = scan_rows * (scan_cols + 1) + 1
        priced = {scan_kernel: {"avg_launch_ms": scan_ms, "algorithmic_bytes_per_launch": scan_bytes,
                                "bytes_per_unit": f"(13 * {scan_rows} + 8) B per column (SURVEY.md 8d)"}}
        for stage, kname, per_slot in (("sort_pass1", "k_msd_pack_split", 9), ("sort_pass2", "k_msd_split", 16), ("sort_pass3", "k_msd_finish", 16)):
            tot, ln = stage_acc.get(stage, [0.0, 0])
            if ln > 0:
                priced[kname] = {"avg_launch_ms": tot / ln, "algorithmic_bytes_per_launch": per_slot * slots,
                                 "bytes_per_unit": f"{per_slot} B per suffix (DESIGN.md 5)"}
        for v in priced.values():
            v["achieved"] = v["algorithmic_bytes_per_launch"] / (v["avg_launch_ms"] * 1e-3) / 1e9 if v["avg_launch_ms"] > 0 else 0.0
            v["frac"] = v["achieved"] / HBM_PEAK_GBPS
        dominant = max(priced, key=lambda k: priced[k]["avg_launch_ms"])
        dom = priced[dominant]
        dom_traffic, dom_src = (traffic, traffic_src) if dominant == scan_kernel else measured_traffic(scan_rows, scan_cols, dominant)
        out = {
            "metric": "MSA columns segmented/sec", "value": n * args.steps / dt, "unit": "columns/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": f"synthetic {m} rows x {n} cols iid ACGT (seed 0x5EED0001), --elastic"
                                   f"{'' if world == 1 else f', {args.cols_per_gpu} columns per GPU, {mode_used}'}",
                       "value_is": "columns of the whole job / wall time of the timed steps, MSA resident in HBM, boundaries left on the "
                                   "device, the sweep of a step running beside the next step's index build; T_seg of SURVEY.md 8(d) "
                                   "(host RAM to host RAM) is `t_seg`, one job alone is `latency_ms`",
                       "rows": m, "cols": n, "text_length": m * (n + 1) + 1, "blocks": state.get("blocks")},
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": dom["achieved"], "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": dom["frac"],
                         "traffic": dom_traffic if world == 1 else None,
                         "traffic_source": (f"{dom_src} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE doubled for "
                                            "the wide coalesced streams, as the guide prescribes for gfx950)") if dom_src else None,
                         "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_launch"], "avg_launch_ms": dom["avg_launch_ms"],
                         "bytes_per_unit": dom["bytes_per_unit"],
                         "note": "the kernel of the step that takes longest; the extension scan and every sort pass are in `kernels`",
                         "kernels": priced, "scan_kernel": scan_kernel, "scan_launches_per_step": launches_per_step},
            "stages_ms_per_step": {k: v[0] / max(1, args.steps) for k, v in stage_acc.items()},
            "sweep": ("serial" if sweeper is None else
                      {"overlapped_with": "the next step's index build (own context and stream on rank 0)",
                       "dp_ms_last_step": sweeper.eng.stage_ms().get("dp", (0.0, 0))[0]}),
            "device_bytes": eng.device_bytes(),
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(m, n, min(args.cpu_sample_cols, n))
        else:
            out["cpu_baseline"] = None
        if world == 1 and not args.no_extras:
            if sweeper is not None:
                sweeper.eng.release_scratch()
            extras = boundary_and_latency(F, torch, dev, m, n, whole_msa())
            if "t_seg" in extras:
                out["t_seg_columns_per_s"] = extras["t_seg"].get("pinned", extras["t_seg"]["pageable"])
            out.update(extras)
            bufs.clear()
            eng.release_scratch()
            out["other_workloads"] = other_workloads(F, torch, dev)
        print(json.dumps(out), flush=True)
    if sweeper is not None:
        sweeper.pool.shutdown()
        sweeper.eng.close()
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
