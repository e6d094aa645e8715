"""world_size-2 gloo test of the column-shard plumbing (shard ranges, uneven all-gather, rank-0
sweep).  The per-shard compute is the oracle's compute_f_range partition here; on the GPU the same
plumbing runs on libfbg_hip.so (tests/test_gpu_parity.py::test_sharded_pipeline_single_gpu)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import random_msa


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, seed, out_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from founderblockgraphs_amd import distributed as D
    from oracle import pyoracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    msa = random_msa(np.random.default_rng(seed), 12, n, similar=0.95, gap_p=0.02, gap_run=4)
    f_ref = O.compute_f(msa)

    def scan(x0, x1):
        # stand-in for the HIP scan of one shard: the oracle restricted to the rank's columns
        return torch.from_numpy(f_ref[x0:x1].astype(np.int64))

    def sweep(f_full):
        return torch.from_numpy(O.minmax_dp(f_full.numpy().astype(np.uint64))[2].astype(np.int64))

    f_full, b = D.segment_columns_sharded(n, scan, sweep)
    assert np.array_equal(f_full.numpy().astype(np.uint64), f_ref)
    if rank == 0:
        np.save(os.path.join(out_dir, "b.npy"), b.numpy())
        np.save(os.path.join(out_dir, "ref.npy"), O.minmax_dp(f_ref)[2].astype(np.int64))
    else:
        assert b is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [1001, 512, 3])
def test_two_rank_column_shards(n, tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), n, 42, str(tmp_path)), nprocs=2, join=True)
    assert np.array_equal(np.load(tmp_path / "b.npy"), np.load(tmp_path / "ref.npy"))


def test_shard_ranges_partition_the_columns():
    from founderblockgraphs_amd.distributed import shard_range
    for n in (1, 7, 1000, 1_000_003):
        for w in (1, 2, 3, 8):
            edges = [shard_range(n, r, w) for r in range(w)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[r][1] == edges[r + 1][0] for r in range(w - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1


def _worker_pairs(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from founderblockgraphs_amd import distributed as D
    from oracle import pyoracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m, n = 13, 300
    msa = random_msa(np.random.default_rng(5), m, n, similar=0.96, gap_p=0.02, gap_run=5, n_p=0.01)
    # pretend ranks hold at most 8 rows x (n+1) symbols: forces the row-group-pair plan
    G, groups, plan = D.plan_row_pairs(m, n, world, limit=8 * (n + 1) + 1)
    f = torch.zeros(n, dtype=torch.int64)
    for a, b in plan[rank]:
        rows = list(range(*groups[a])) + list(range(*groups[b]))
        part = O.compute_f(msa[rows], ignore="N")       # stand-in for index_build + scan_f on one pair
        f = torch.maximum(f, torch.from_numpy(part.astype(np.int64)))
    D.all_reduce_max(f)
    assert np.array_equal(f.numpy().astype(np.uint64), O.compute_f(msa, ignore="N"))
    assert sorted(p for r in plan for p in r) == D.group_pairs(G) and G >= 3
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_row_group_pairs(tmp_path):
    """The capacity plan used when m*(n+1)+1 >= 2^32: pairs of row groups, element-wise max of f."""
    mp.spawn(_worker_pairs, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)


class _OraclePartEngine:
    """CPU stand-in for the fbg_part_* calls of libfbg_hip.so, built on the oracle's index arrays: partition r
    owns the suffix-array slots [N*r/W, N*(r+1)/W) and contributes the extensions g of exactly those suffixes
    (fbg.cpp:1633-1657 for gap-free rows).  Same calling convention: raw addresses of caller-owned buffers."""

    def __init__(self, msa, fail=False, lazy=False):
        from oracle import pyoracle as O
        self.msa, self.fail = msa, fail
        self.lazy = lazy               # first scan with a threshold (drops small extensions), like the engine's g_min
        self.m, self.n = msa.shape
        _, self.SA, self.ISA, lcp = O.msa_index(msa)
        self.LCP = np.concatenate([lcp.astype(np.int64), [0]])
        self.gmax = None

    @staticmethod
    def _view(ptr, nbytes, dtype):
        import ctypes
        return np.frombuffer((ctypes.c_uint8 * nbytes).from_address(ptr), dtype=dtype)

    def sync(self):
        pass

    def part_index_build(self, part, nparts, blob_ptr, reversed=False, ignorechars="", disable_efg_tricks=False):
        from founderblockgraphs_amd._lib import PART_HALO, PART_HALO_BYTES
        assert not reversed
        self.part, self.nparts = part, nparts
        blob = self._view(blob_ptr, PART_HALO_BYTES, np.uint8)
        blob[:] = 0
        tail = self._view(blob_ptr + 2 * PART_HALO * 12, 16, np.uint64)
        tail[0] = 0 if self.fail else 1
        return not self.fail

    def part_scan(self, blobs_ptr, gmax_ptr):
        from founderblockgraphs_amd._lib import PART_HALO, PART_HALO_BYTES
        oks = [int(self._view(blobs_ptr + p * PART_HALO_BYTES + 2 * PART_HALO * 12, 16, np.uint64)[0])
               for p in range(self.nparts)]
        g = self._view(gmax_ptr, 4 * (self.n + 1), np.int32)
        g[:] = 0
        if not all(oks):
            g[self.n] = 1
            return False
        N = len(self.SA)
        lo, hi = N * self.part // self.nparts, N * (self.part + 1) // self.nparts
        for x in range(self.n):
            r = np.sort(self.ISA[np.arange(self.m) * (self.n + 1) + x].astype(np.int64))
            k = 0
            while k < self.m:
                e = k
                while e + 1 < self.m and r[e + 1] == r[e] + 1:
                    e += 1
                lb, rb = r[k], r[e]
                for q in range(lb, rb + 1):
                    if lo <= q < hi:
                        ext = 1 + max(self.LCP[lb:q + 1].min(), self.LCP[q + 1:rb + 2].min())
                        if self.lazy and ext < 4:
                            continue
                        g[x] = max(g[x], ext)
                k = e + 1
        return True

    def part_finish(self, gmax_ptr):
        g = self._view(gmax_ptr, 4 * (self.n + 1), np.int32)
        if g[self.n] != 0:
            return 0
        if self.lazy and g[:self.n].min() < 4:
            return 2                   # a column did not clear the threshold: the caller must re-scan
        self.gmax = g[:self.n].astype(np.int64)
        return 1

    def part_rescan(self, gmax_ptr):
        self.lazy = False
        blobs = np.zeros(1, dtype=np.uint8)
        from founderblockgraphs_amd._lib import PART_HALO, PART_HALO_BYTES
        ok = np.zeros(self.nparts * PART_HALO_BYTES, dtype=np.uint8)
        for p in range(self.nparts):
            ok[p * PART_HALO_BYTES + 2 * PART_HALO * 12] = 1
        self.part_scan(ok.ctypes.data, gmax_ptr)
        del blobs

    def f(self):
        """k_rank_finish: fbg.cpp:1618-1666 with rank_i(x) = x, tot_i = n (gap-free rows)."""
        n, x = self.n, np.arange(self.n)
        fi = np.where(x + self.gmax > n, n - 1, x + self.gmax - 1)
        f = np.maximum(x, fi)
        f[0] = 0
        return f


def _worker_part(rank, world, port, fail_rank, out_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from founderblockgraphs_amd import distributed as D
    from oracle import pyoracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    msa = random_msa(np.random.default_rng(11), 9, 160, similar=0.8)
    eng = _OraclePartEngine(msa, fail=(rank == fail_rank), lazy=(fail_rank == -2))
    ok = D.partitioned_index(eng, msa.shape[1], device="cpu")
    assert ok == (fail_rank < 0)                     # one failing partition: every rank declines
    assert not eng.lazy                              # fail_rank -2: the re-scan branch was taken (on every rank)
    if ok:
        assert np.array_equal(eng.f().astype(np.uint64), O.compute_f(msa))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("fail_rank", [-1, 0, 1, -2])
def test_two_rank_partitioned_index(fail_rank, tmp_path):
    """Key-range partitioned index: halo all-gather + all-reduce(max) of the column maxima, verdict included."""
    mp.spawn(_worker_part, args=(2, _free_port(), fail_rank, str(tmp_path)), nprocs=2, join=True)


def _worker_raises(rank, world, port, where, out_dir):
    """One rank's engine call raises: EVERY rank must raise (the failing one its own error, the others RankFailed),
    promptly -- not after the process group's timeout."""
    import sys
    import time
    from datetime import timedelta
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from founderblockgraphs_amd import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=timedelta(seconds=120))
    assert D.collective_device().type == "cpu"       # gloo; under nccl it is the rank's GPU, never taken from a result
    t0 = time.time()
    msa = random_msa(np.random.default_rng(3), 6, 90, similar=0.8)
    n = msa.shape[1]
    raised = None
    try:
        if where == "scan_shard":
            def scan(x0, x1):
                if rank == 1:
                    raise RuntimeError("engine failure on rank 1")
                return torch.zeros(x1 - x0, dtype=torch.int64)
            D.segment_columns_sharded(n, scan, lambda f: f)
        else:
            eng = _OraclePartEngine(msa)
            if rank == 1:
                def boom(*a, **k):
                    raise RuntimeError("engine failure on rank 1")
                setattr(eng, where, boom)
            D.partitioned_index(eng, n, device="cpu")
    except D.RankFailed as e:
        raised = "RankFailed"
        assert rank == 0, e
    except RuntimeError as e:
        raised = "own"
        assert rank == 1 and "engine failure" in str(e)
    assert raised == ("own" if rank == 1 else "RankFailed")
    assert time.time() - t0 < 60
    dist.barrier()                                   # both ranks are still in step: the next collective works
    dist.destroy_process_group()


@pytest.mark.parametrize("where", ["scan_shard", "part_index_build", "part_scan"])
def test_two_rank_failure_is_raised_on_every_rank(where, tmp_path):
    mp.spawn(_worker_raises, args=(2, _free_port(), where, str(tmp_path)), nprocs=2, join=True)
