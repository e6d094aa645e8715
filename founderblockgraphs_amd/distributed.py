"""Column-sharded segmentation across the GPUs of one node (one process per GPU).

The scan shards by column range exactly like the reference's compute_f_range threads
(founderblockgraph.cpp:1475-1577, partition 2278-2284): rank r scans columns [n*r/W, n*(r+1)/W) of
every row against the (replicated) index and owns that slice of f.  The path has ONE exchange step:
an all-gather of the per-column minimal extensions (n * 8 bytes in total) ahead of the sequential
sweep on rank 0.  The collective goes through torch.distributed -- backend "nccl" is
RCCL over xGMI on ROCm, "gloo" is used by the CPU tests -- and is the only place torch is needed.

`scan_shard` / `sweep` are callables so the same plumbing runs against libfbg_hip.so (product) and
against stand-ins in the world_size-2 gloo tests.
"""
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Columns owned by `rank`: contiguous, differing by at most one column between ranks."""
    return n * rank // world, n * (rank + 1) // world


def all_gather_columns(f_local, n, rank, world, group=None):
    """Concatenate the ranks' slices (uneven allowed) into the full f[0..n) on every rank."""
    if world == 1:
        return f_local
    sizes = [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]
    if len(set(sizes)) == 1:
        full = torch.empty(n, dtype=f_local.dtype, device=f_local.device)
        dist.all_gather_into_tensor(full, f_local.contiguous(), group=group)
        return full
    pad = max(sizes)
    buf = torch.zeros(pad, dtype=f_local.dtype, device=f_local.device)
    buf[:f_local.numel()] = f_local
    parts = [torch.empty(pad, dtype=f_local.dtype, device=f_local.device) for _ in range(world)]
    dist.all_gather(parts, buf, group=group)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)])


class RankFailed(RuntimeError):
    """Some rank's engine call failed: raised on EVERY rank, after the collective that carried the flag."""


def collective_device(group=None):
    """The device a collective's tensors must live on for this process group: the rank's current GPU under the
    "nccl" (= RCCL) backend, the CPU under "gloo".  Never derived from a result tensor: a rank whose engine call
    failed has none, and a CPU tensor handed to an RCCL group raises on that rank alone."""
    backend = str(dist.get_backend(group)).lower() if dist.is_initialized() else "gloo"
    if "nccl" in backend:
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def agree(error, world, group=None, device=None):
    """All ranks learn whether any of them caught an exception during the phase that just ended (one all-reduce of a
    single word) and raise together -- a rank that raised alone would leave the others waiting in the next collective
    until the process group's timeout."""
    if world > 1:
        if device is None:
            device = collective_device(group)
        flag = torch.tensor([1 if error is not None else 0], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
        failed = bool(flag.item())
    else:
        failed = error is not None
    if error is not None:
        raise error
    if failed:
        raise RankFailed("another rank failed in this phase; see its log")


def segment_columns_sharded(n, scan_shard, sweep, rank=None, world=None, group=None, device=None):
    """scan_shard(x0, x1) -> 1-D integer tensor with f[x0..x1); sweep(f_full) -> boundaries (rank 0 only).
    Returns (f_full, boundaries or None).  `device`: where the agreement word lives (default: what the process
    group's backend needs, collective_device)."""
    world = dist.get_world_size(group) if world is None else world
    rank = dist.get_rank(group) if rank is None else rank
    x0, x1 = shard_range(n, rank, world)
    error, f_local = None, None
    try:
        f_local = scan_shard(x0, x1)
        if f_local.numel() != x1 - x0:
            raise ValueError(f"scan_shard returned {f_local.numel()} values for columns [{x0}, {x1})")
    except Exception as e:          # noqa: BLE001 -- carried to every rank below
        error = e
    agree(error, world, group, device=device)
    f_full = all_gather_columns(f_local, n, rank, world, group)
    boundaries = sweep(f_full) if rank == 0 else None
    return f_full, boundaries


def engine_scan_shard(engine, n, disable_efg_tricks=False):
    """scan_shard callable running on libfbg_hip.so; the engine must already hold the index."""
    def scan(x0, x1):
        d_f = torch.zeros(n, dtype=torch.int64, device="cuda")
        # the engine may run on its own stream (bench.py hands it torch's): order the fill, the scan
        # and whatever torch does with the result explicitly
        torch.cuda.current_stream().synchronize()
        engine.scan_f(x0, x1, d_f.data_ptr(), disable_efg_tricks)
        engine.sync()
        return d_f[x0:x1]
    return scan


def engine_sweep(engine):
    def sweep(f_full):
        n = f_full.numel()
        d_b = torch.empty(n + 1, dtype=torch.int64, device=f_full.device)
        torch.cuda.current_stream().synchronize()
        cnt = engine.minmax_dp_device(f_full.data_ptr(), n, d_b.data_ptr())
        return d_b[:cnt]
    return sweep


# ---- when the replicated index does not fit: exact row-group decomposition ----------------------
#
# g_i(x) of fbg.cpp:1656 is 1 + the longest match of row i's suffix with ANY text position that is not
# an aligned pointer of an active row.  Split the rows into G groups: the match is the maximum over
# the groups the other position lies in, so indexing every unordered PAIR of groups separately and
# taking the element-wise maximum of the partial f arrays gives the same f (matches running across a
# '#' into a different neighbouring row do not matter: any g reaching the row end is clamped the same
# way, fbg.cpp:1659-1664).  The reference itself merges partial f arrays by max (fbg.cpp:1681).
# One all-reduce(max) of n values is the only exchange.  Costs (G+1)/2 x the sorting work: it buys
# capacity (text of 2/G of the rows per GPU), not speed -- used only when m*(n+1)+1 >= 2^32.

def row_groups(m, G):
    return [(m * g // G, m * (g + 1) // G) for g in range(G)]


def group_pairs(G):
    return [(a, b) for a in range(G) for b in range(a + 1, G)] if G > 1 else [(0, 0)]


def plan_row_pairs(m, n, world, limit=(1 << 32) - 2):
    """Smallest G whose pair texts fit 32-bit ranks; pairs are dealt to ranks round-robin."""
    for G in range(2, 65):
        groups = row_groups(m, G)
        biggest = max((groups[a][1] - groups[a][0]) + (groups[b][1] - groups[b][0]) for a, b in group_pairs(G))
        if biggest * (n + 1) + 1 <= limit:
            pairs = group_pairs(G)
            return G, groups, [pairs[k::world] for k in range(world)]
    raise ValueError("MSA too large for the row-pair plan")


def all_reduce_max(f_partial, group=None):
    dist.all_reduce(f_partial, op=dist.ReduceOp.MAX, group=group)
    return f_partial


# ---- partitioned index: sort state and sort time divided by the number of GPUs ----------------------
#
# Every rank holds the whole MSA but sorts and scans only the suffixes whose leading symbols fall into its key
# range (include/fbg_hip.h, fbg_part_*).  Two collectives: an all-gather of the partitions' edge slots (1.5 KB
# per rank) and an all-reduce(MAX) of the per-column maxima ((n + 1) * 4 bytes).  Every rank learns from the
# collectives whether all partitions succeeded, so the fall-back to the replicated index is taken by all ranks or none.

def partitioned_index(engine, n, rank=None, world=None, group=None, reversed=False, device="cuda", ignorechars="", disable_efg_tricks=False):
    """Build the index of the engine's current MSA partitioned over the ranks.  True: the engine is ready for
    scan_f / scan_v (any column range, normally all of them on rank 0).  False: nothing usable was built --
    call engine.index_build() on every rank instead.  MSAs with gaps / ignore characters (ignorechars) are scanned for
    the setting of the elastic tricks given here; scan_f must ask for the same."""
    from ._lib import PART_HALO_BYTES
    world = dist.get_world_size(group) if world is None else world
    rank = dist.get_rank(group) if rank is None else rank
    blob = torch.empty(PART_HALO_BYTES, dtype=torch.uint8, device=device)
    blobs = torch.empty(world * PART_HALO_BYTES, dtype=torch.uint8, device=device)
    gmax = torch.empty(n + 1, dtype=torch.int32, device=device)

    def fence():       # torch's stream and the engine's may differ: order their work explicitly
        if torch.device(device).type == "cuda":
            torch.cuda.current_stream().synchronize()
        engine.sync()

    # Failure agreement is folded into the two collectives of the path: a rank whose engine call raised sends a blob
    # whose verdict word has its top bit set / a column-maxima array whose verdict word is 2; every rank checks the
    # gathered / reduced words and raises, instead of one rank raising alone and the others hanging in the collective.
    ERR_BIT = 1 << 63
    tail_word = (PART_HALO_BYTES - 16) // 8                 # the blob's verdict word (u64), see k_halo_export

    def tails():
        return blobs.view(torch.int64).view(world, PART_HALO_BYTES // 8)[:, tail_word].cpu()

    fence()
    error = None
    try:
        engine.part_index_build(rank, world, blob.data_ptr(), reversed, ignorechars, disable_efg_tricks)   # verdict travels inside the blob
    except Exception as e:          # noqa: BLE001
        error = e
        blob.zero_()
        blob.view(torch.int64)[tail_word] = -(1 << 63)      # ERR_BIT as a signed word
    fence()
    if world > 1:
        dist.all_gather_into_tensor(blobs, blob, group=group)
    else:
        blobs.copy_(blob)
    fence()
    if error is not None:
        raise error
    if bool((tails() < 0).any()):
        raise RankFailed("fbg_part_index_build failed on another rank; see its log")
    try:
        engine.part_scan(blobs.data_ptr(), gmax.data_ptr())           # verdict travels in gmax[n]
    except Exception as e:          # noqa: BLE001
        error = e
        gmax.zero_()
        gmax[n] = 2
    fence()
    if world > 1:
        dist.all_reduce(gmax, op=dist.ReduceOp.MAX, group=group)
    fence()
    if error is not None:
        raise error
    if int(gmax[n].item()) >= 2:
        raise RankFailed("fbg_part_scan failed on another rank; see its log")
    verdict = engine.part_finish(gmax.data_ptr())
    if verdict == 2:
        # some column's maximum did not clear the scan's threshold (every rank sees the same reduced maxima and takes
        # this branch together): exact re-scan without threshold, reduce once more
        try:
            engine.part_rescan(gmax.data_ptr())
        except Exception as e:      # noqa: BLE001
            error = e
            gmax.zero_()
            gmax[n] = 2
        fence()
        if world > 1:
            dist.all_reduce(gmax, op=dist.ReduceOp.MAX, group=group)
        fence()
        if error is not None:
            raise error
        if int(gmax[n].item()) >= 2:
            raise RankFailed("fbg_part_rescan failed on another rank; see its log")
        verdict = engine.part_finish(gmax.data_ptr())
    return verdict == 1
