// group.hip -- several contexts driven as ONE engine: the multi-GPU form of the hot path, inside the C ABI.
//
// The reference fans compute_f_range out over std::threads inside the binary (fbg.cpp:2278-2289) and joins them
// before the sequential sweep; here the fan-out is over GPUs.  A group holds one context per entry of dev_ids
// (fbg_ctx_create(ndev, dev_ids)); an id may repeat -- several contexts on one device -- which is how
// the multi-device code is exercised on a one-GPU box and how a text too long for one index is split into
// partitions that one device works off one after the other.  One host thread per member drives its context
// (the entry points block on their stream), the members meet at the exchange steps:
//
//   plan "partitioned"  every member sorts and scans key ranges of the suffixes (fbg_part_*): an all-gather of the
//                       partitions' edge slots (FBG_PART_HALO_BYTES each) and an all-reduce(MAX) of the n + 1
//                       per-column maxima.  More partitions than members: each member works its partitions off one
//                       after the other (two passes: edge slots first, then the scans), so index memory is that of
//                       ONE partition -- a 1000 x 8,000,000 MSA (8e9 symbols) runs on a single MI355X.
//   plan "columns"      replicated index, member r scans columns [r * chunk, (r + 1) * chunk) -- compute_f_range's
//                       partition -- and one all-gather brings f to member 0.  The fall-back for MSAs
//                       the partitioned index declines (gaps, ignore characters, similar rows).
//   plan "row pairs"    texts of 2^32 symbols and more that the partitioned index declines: rows in G groups, every
//                       unordered pair of groups indexed by some member, one all-reduce(MAX) of f (exact: g is a
//                       maximum over text positions; the reference merges partial f by max itself, fbg.cpp:1681).
//
// Exchange: RCCL (ncclAllGather / ncclAllReduce on the members' streams, one communicator per member from
// ncclCommInitAll) when every member has a device of its own; device-to-device / peer copies and a max kernel when
// members share a device (RCCL refuses two ranks on one GPU) or when option "exchange" = 1 asks for it.  librccl is
// opened with dlopen the first time it is needed, so a single-GPU process never loads it.
//
// The sweep (fbg_minmax_dp and friends) and fbg_block_graph stay single-context calls: run them on
// fbg_group_member(g, 0), which holds f and the MSA.
#include "fbg_internal.h"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <stdarg.h>
#include <algorithm>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool load(std::string &why)
    {
        if (handle) return true;
        // a process that already holds an RCCL (PyTorch-ROCm bundles one under the same soname) gets that one
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (handle) break;
        }
        if (!handle) { why = std::string("dlopen(librccl.so.1): ") + dlerror(); return false; }
        auto sym = [&](const char *n) { return dlsym(handle, n); };
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        AllGather = reinterpret_cast<decltype(AllGather)>(sym("ncclAllGather"));
        AllReduce = reinterpret_cast<decltype(AllReduce)>(sym("ncclAllReduce"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        if (!CommInitAll || !CommDestroy || !AllGather || !AllReduce || !GroupStart || !GroupEnd || !GetErrorString) {
            why = "librccl lacks a collective entry point";
            dlclose(handle); handle = nullptr;
            return false;
        }
        return true;
    }
};

struct Member {
    fbg_ctx *ctx = nullptr;
    int dev = 0;
    int msa_owner = -1;                // member whose buffer holds this device's copy of the MSA (itself, or an earlier one)
    DevBuf msa;                        // the MSA on this device (owners only)
    DevBuf blobs, gmax, acc, f, pair;  // edge slots of all partitions, column maxima (scan / accumulated), f, a pair's rows
    ncclComm_t comm = nullptr;
};

enum { GRP_EXCHANGE_AUTO = 0, GRP_EXCHANGE_COPIES = 1, GRP_EXCHANGE_RCCL = 2 };

} // namespace

struct fbg_group {
    std::vector<Member> mem;
    bool distinct = true;              // every member has a device of its own
    bool use_rccl = false;
    RcclApi rccl;
    std::string err;
    std::mutex err_mutex;              // members fail on their own threads
    uint64_t m = 0, n = 0;
    bool have_msa = false;
    int64_t opt_partitions = 0, opt_exchange = GRP_EXCHANGE_AUTO, opt_plan = FBG_PLAN_AUTO, opt_pair_rows = 0;
    int plan_used = FBG_PLAN_AUTO;
    int parts_used = 0;
};

namespace {

int grp_fail(fbg_group *g, int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    std::lock_guard<std::mutex> lock(g->err_mutex);
    g->err = buf;
    return code;
}

void set_err(fbg_group *g, const char *what)
{
    std::lock_guard<std::mutex> lock(g->err_mutex);
    g->err = what ? what : "";
}

#define GRP_HIP_TRY(g, expr)                                                                                     \
    do {                                                                                                         \
        hipError_t e_ = (expr);                                                                                  \
        if (e_ != hipSuccess)                                                                                    \
            return grp_fail((g), e_ == hipErrorOutOfMemory ? FBG_ERR_OOM : FBG_ERR_HIP, "%s failed: %s (%s:%d)", \
                            #expr, hipGetErrorString(e_), __FILE__, __LINE__);                                   \
    } while (0)

// fn(member index) on every member, one host thread each (member 0 on the calling thread).  All members run to the
// end of the phase whatever the others return: the first failure becomes the group's, and every member has left its
// context in a defined state by the time the caller sees it -- the in-process form of "all ranks raise together".
int parallel(fbg_group *g, const std::function<int(int)> &fn)
{
    const int nm = (int)g->mem.size();
    std::vector<int> rc(nm, FBG_OK);
    std::vector<std::thread> th;
    for (int i = 1; i < nm; i++) th.emplace_back([&, i]() { rc[i] = fn(i); });
    rc[0] = fn(0);
    for (auto &t : th) t.join();
    for (int i = 0; i < nm; i++)
        if (rc[i] != FBG_OK) {
            const char *why = fbg_last_error(g->mem[i].ctx);
            if (g->err.empty() || (why && *why)) set_err(g, (std::string("member ") + std::to_string(i) + ": " + (why ? why : "")).c_str());
            return rc[i];
        }
    return FBG_OK;
}

int mem_reserve(fbg_group *g, Member &mb, DevBuf &b, size_t bytes)
{
    GRP_HIP_TRY(g, hipSetDevice(mb.dev));
    const int rc = fbg_reserve(mb.ctx, b, bytes);
    if (rc != FBG_OK) set_err(g, fbg_last_error(mb.ctx));
    return rc;
}

__global__ void k_grp_max32(uint32_t *__restrict__ acc, const uint32_t *__restrict__ in, uint64_t cnt)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cnt) acc[i] = max(acc[i], in[i]);
}
__global__ void k_grp_max64(unsigned long long *__restrict__ acc, const unsigned long long *__restrict__ in, uint64_t cnt)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cnt) acc[i] = max(acc[i], in[i]);
}

int sync_all(fbg_group *g)
{
    for (Member &mb : g->mem) {
        GRP_HIP_TRY(g, hipSetDevice(mb.dev));
        GRP_HIP_TRY(g, hipStreamSynchronize(mb.ctx->stream));
    }
    return FBG_OK;
}

int copy_between(fbg_group *g, Member &dst, void *d, Member &src, const void *s, size_t bytes)
{
    GRP_HIP_TRY(g, hipSetDevice(dst.dev));
    if (dst.dev == src.dev) GRP_HIP_TRY(g, hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, dst.ctx->stream));
    else GRP_HIP_TRY(g, hipMemcpyPeerAsync(d, dst.dev, s, src.dev, bytes, dst.ctx->stream));
    return FBG_OK;
}

#define GRP_NCCL_TRY(g, expr)                                                                                       \
    do {                                                                                                            \
        ncclResult_t r_ = (expr);                                                                                   \
        if (r_ != ncclSuccess)                                                                                      \
            return grp_fail((g), FBG_ERR_HIP, "%s failed: %s (%s:%d)", #expr, (g)->rccl.GetErrorString(r_), __FILE__, \
                            __LINE__);                                                                              \
    } while (0)

// Every member's buffer `at(i)` holds `per` bytes of its own at offset i * per; afterwards all hold all of them.
// The callers have synchronised the members' streams (the producing phase ended with a join).
int all_gather(fbg_group *g, const std::function<uint8_t *(int)> &at, size_t per)
{
    const int nm = (int)g->mem.size();
    if (nm == 1 && !g->use_rccl) return FBG_OK;           // (a forced RCCL exchange runs on a single rank as well)
    if (g->use_rccl) {
        GRP_NCCL_TRY(g, g->rccl.GroupStart());
        for (int i = 0; i < nm; i++) {
            Member &mb = g->mem[i];
            GRP_NCCL_TRY(g, g->rccl.AllGather(at(i) + (size_t)i * per, at(i), per, ncclUint8, mb.comm, mb.ctx->stream));
        }
        GRP_NCCL_TRY(g, g->rccl.GroupEnd());
    } else {
        for (int dst = 0; dst < nm; dst++)
            for (int src = 0; src < nm; src++)
                if (src != dst)
                    FBG_TRY(copy_between(g, g->mem[dst], at(dst) + (size_t)src * per, g->mem[src], at(src) + (size_t)src * per, per));
    }
    return sync_all(g);
}

// element-wise maximum of the members' arrays (cnt words of `width` bytes, 4 or 8), result in every member's array
int all_reduce_max(fbg_group *g, const std::function<void *(int)> &at, uint64_t cnt, int width)
{
    const int nm = (int)g->mem.size();
    if (nm == 1 && !g->use_rccl) return FBG_OK;
    if (g->use_rccl) {
        GRP_NCCL_TRY(g, g->rccl.GroupStart());
        for (int i = 0; i < nm; i++) {
            Member &mb = g->mem[i];
            GRP_NCCL_TRY(g, g->rccl.AllReduce(at(i), at(i), cnt, width == 4 ? ncclUint32 : ncclUint64, ncclMax, mb.comm, mb.ctx->stream));
        }
        GRP_NCCL_TRY(g, g->rccl.GroupEnd());
        return sync_all(g);
    }
    // copies: everything to member 0 (through its scratch), reduced there, sent back
    Member &m0 = g->mem[0];
    FBG_TRY(mem_reserve(g, m0, m0.pair, cnt * width));     // `pair` is free during the exchanges
    for (int src = 1; src < nm; src++) {
        FBG_TRY(copy_between(g, m0, m0.pair.p, g->mem[src], at(src), cnt * width));
        if (width == 4) hipLaunchKernelGGL(k_grp_max32, dim3(fbg_blocks(cnt, 256)), dim3(256), 0, m0.ctx->stream, (uint32_t *)at(0), m0.pair.as<uint32_t>(), cnt);
        else hipLaunchKernelGGL(k_grp_max64, dim3(fbg_blocks(cnt, 256)), dim3(256), 0, m0.ctx->stream, (unsigned long long *)at(0), m0.pair.as<unsigned long long>(), cnt);
    }
    GRP_HIP_TRY(g, hipGetLastError());
    GRP_HIP_TRY(g, hipStreamSynchronize(m0.ctx->stream));
    for (int dst = 1; dst < nm; dst++) FBG_TRY(copy_between(g, g->mem[dst], at(dst), m0, at(0), cnt * width));
    return sync_all(g);
}

const uint8_t *member_msa(fbg_group *g, int i) { return g->mem[g->mem[i].msa_owner].msa.as<uint8_t>(); }

// partitions: the option, else as many as there are members -- more (a multiple) when the sort state of N / P
// suffixes would not fit the free memory of the smallest device (12-byte slots in three buffers of the MSD sort,
// about 40 bytes per suffix, next to the MSA and the text)
int pick_partitions(fbg_group *g, int *parts)
{
    const int nm = (int)g->mem.size();
    if (g->opt_partitions > 0) {
        if (g->opt_partitions % nm != 0)
            return grp_fail(g, FBG_ERR_INVALID, "option partitions=%lld must be a multiple of the %d members", (long long)g->opt_partitions, nm);
        *parts = (int)g->opt_partitions;
        return FBG_OK;
    }
    const double N = (double)g->m * (double)(g->n + 1) + 1;
    size_t min_free = ~(size_t)0;
    for (Member &mb : g->mem) {
        size_t fr = 0, tot = 0;
        GRP_HIP_TRY(g, hipSetDevice(mb.dev));
        GRP_HIP_TRY(g, hipMemGetInfo(&fr, &tot));
        int sharing = 0;
        size_t held = 0;
        for (Member &o : g->mem)
            if (o.dev == mb.dev) { sharing++; held += o.ctx->held_bytes; }
        // workspaces the contexts already own are reused (grow-only), the MSA among them
        min_free = std::min(min_free, (fr + held) / (size_t)sharing);
    }
    // a partition of more than ~1e9 suffixes leaves the three-pass MSD sort for the slower library sort
    // (msd_sort_pairs.hip: 512^2 sub-buckets of 4608 slots); per context: MSA + text + ~40 bytes per suffix + slack
    int k = 1;
    while (k < 64 && (N / ((double)nm * k) > 1.05e9 || 2.0 * N + 40.0 * N / ((double)nm * k) + 4e9 > (double)min_free * 0.9)) k++;
    *parts = nm * k;
    return FBG_OK;
}

// ---- plan: key-range partitioned index --------------------------------------------------------------------------
// *done = 1: member 0 (every member when each holds one partition) is ready for fbg_scan_f / fbg_scan_v.
// *done = 0: some partition declined; nothing usable.
int plan_partitioned(fbg_group *g, int reversed, const uint8_t *ignore, uint64_t ignore_len, int disable_tricks, int *done,
                     bool is_rescan = false)
{
    *done = 0;
    for (Member &mb : g->mem) mb.ctx->opt.part_tricks_off = disable_tricks ? 1 : 0;     // (MSAs with gaps / ignore characters: one setting per scan)
    const int nm = (int)g->mem.size();
    int P = 0;
    FBG_TRY(pick_partitions(g, &P));
    const int k = P / nm;                                  // partitions per member: [i * k, (i + 1) * k)
    const uint64_t n = g->n;
    const size_t HB = FBG_PART_HALO_BYTES;
    g->parts_used = P;
    for (Member &mb : g->mem) {
        FBG_TRY(mem_reserve(g, mb, mb.blobs, (size_t)P * HB));
        FBG_TRY(mem_reserve(g, mb, mb.gmax, (n + 1) * 4));
        if (k > 1) FBG_TRY(mem_reserve(g, mb, mb.acc, (n + 1) * 4));
    }
    std::vector<int> good(nm, 1);
    // pass 1: every partition's sort and classification; its edge slots land in the owner's blob table
    FBG_TRY(parallel(g, [&](int i) -> int {
        Member &mb = g->mem[i];
        for (int q = 0; q < k; q++) {
            const int p = i * k + q;
            int ok = 0;
            FBG_TRY(fbg_part_index_build_ignore(mb.ctx, reversed, p, P, ignore, ignore_len, mb.blobs.as<uint8_t>() + (size_t)p * HB, &ok));
            if (!ok) good[i] = 0;                          // the verdict travels in the blob as well
        }
        return fbg_sync(mb.ctx);
    }));
    FBG_TRY(all_gather(g, [&](int i) { return g->mem[i].blobs.as<uint8_t>(); }, (size_t)k * HB));
    if (std::find(good.begin(), good.end(), 0) != good.end()) return FBG_OK;
    // pass 2: halos in, runs walked, column maxima out.  With several partitions per member the sort state of a
    // partition was overwritten by the next one: it is built once more
    FBG_TRY(parallel(g, [&](int i) -> int {
        Member &mb = g->mem[i];
        FBG_HIP_TRY(mb.ctx, hipSetDevice(mb.dev));
        for (int q = 0; q < k; q++) {
            const int p = i * k + q;
            int ok = 0;
            if (k > 1) {
                FBG_TRY(mem_reserve(g, mb, mb.pair, HB));
                FBG_TRY(fbg_part_index_build_ignore(mb.ctx, reversed, p, P, ignore, ignore_len, mb.pair.p, &ok));
                if (!ok) { good[i] = 0; break; }
            }
            FBG_TRY(fbg_part_scan(mb.ctx, mb.blobs.p, mb.gmax.as<uint32_t>(), &ok));
            if (!ok) good[i] = 0;
            if (k > 1) {
                if (q == 0) FBG_HIP_TRY(mb.ctx, hipMemcpyAsync(mb.acc.p, mb.gmax.p, (n + 1) * 4, hipMemcpyDeviceToDevice, mb.ctx->stream));
                else hipLaunchKernelGGL(k_grp_max32, dim3(fbg_blocks(n + 1, 256)), dim3(256), 0, mb.ctx->stream, mb.acc.as<uint32_t>(), mb.gmax.as<uint32_t>(), n + 1);
            }
        }
        return fbg_sync(mb.ctx);
    }));
    // a partition that declined in this pass (its rebuild, or its scan) leaves nothing to reduce: with k > 1 the
    // member's accumulator may not even have been written.  All members live in this process and see good[].
    if (std::find(good.begin(), good.end(), 0) != good.end()) return FBG_OK;
    auto red = [&](int i) -> void * { return k > 1 ? g->mem[i].acc.p : g->mem[i].gmax.p; };
    FBG_TRY(all_reduce_max(g, red, n + 1, 4));             // word n carries every partition's verdict
    std::vector<int> verdict(nm, 0);
    FBG_TRY(parallel(g, [&](int i) -> int { return fbg_part_finish(g->mem[i].ctx, (const uint32_t *)red(i), &verdict[i]); }));
    if (verdict[0] == 2) {
        // some column did not clear the largest threshold a partition scanned with (all members see the same reduced
        // maxima): exact maxima without thresholds, reduced once more
        if (k == 1) {
            FBG_TRY(parallel(g, [&](int i) -> int { return fbg_part_rescan(g->mem[i].ctx, g->mem[i].gmax.as<uint32_t>()); }));
            FBG_TRY(all_reduce_max(g, red, n + 1, 4));
            FBG_TRY(parallel(g, [&](int i) -> int { return fbg_part_finish(g->mem[i].ctx, (const uint32_t *)red(i), &verdict[i]); }));
        } else {
            // once: the re-scan runs without thresholds, a second verdict 2 would be a defect, not a reason to go on
            if (is_rescan) return grp_fail(g, FBG_ERR_HIP, "partitioned index: the exact re-scan asked for another re-scan");
            std::vector<int64_t> old(nm), old_g(nm);
            for (int i = 0; i < nm; i++) {
                old[i] = g->mem[i].ctx->opt.rank_no_threshold; g->mem[i].ctx->opt.rank_no_threshold = 1;
                old_g[i] = g->mem[i].ctx->opt.gapped_rank; if (old_g[i] == 0 || old_g[i] == 4) g->mem[i].ctx->opt.gapped_rank = 3;
            }
            const int rc = plan_partitioned(g, reversed, ignore, ignore_len, disable_tricks, done, true);
            for (int i = 0; i < nm; i++) { g->mem[i].ctx->opt.rank_no_threshold = old[i]; g->mem[i].ctx->opt.gapped_rank = old_g[i]; }
            return rc;
        }
    }
    *done = verdict[0] == 1;
    return FBG_OK;
}

// ---- plan: replicated index, column shards ----------------------------------------------------------------------
// mode FBG_SCAN_F / FBG_SCAN_V; every member's f buffer holds nm * chunk words, member 0's the whole array afterwards
int plan_columns(fbg_group *g, int reversed, const uint8_t *ignore, uint64_t ignore_len, int disable_tricks)
{
    const int nm = (int)g->mem.size();
    const uint64_t n = g->n, chunk = (n + nm - 1) / nm;
    FBG_TRY(parallel(g, [&](int i) -> int {
        Member &mb = g->mem[i];
        FBG_TRY(fbg_index_build(mb.ctx, reversed, ignore, ignore_len));
        const uint64_t x0 = std::min(n, (uint64_t)i * chunk), x1 = std::min(n, x0 + chunk);
        if (reversed) FBG_TRY(fbg_scan_v(mb.ctx, x0, x1, mb.f.as<uint64_t>()));
        else FBG_TRY(fbg_scan_f(mb.ctx, x0, x1, disable_tricks, mb.f.as<uint64_t>()));
        return fbg_sync(mb.ctx);
    }));
    // the one exchange of the path: the per-column minimal extensions, ahead of the sweep on member 0
    return all_gather(g, [&](int i) { return g->mem[i].f.as<uint8_t>(); }, (size_t)chunk * 8);
}

// ---- plan: row-group pairs (texts of 2^32 symbols and more; elastic f only) --------------------------------------
int plan_row_pairs(fbg_group *g, const uint8_t *ignore, uint64_t ignore_len, int disable_tricks)
{
    const int nm = (int)g->mem.size();
    const uint64_t m = g->m, n = g->n;
    // option pair_rows (tests): rows a pair text may hold, instead of what fits 32-bit positions
    const uint64_t limit = g->opt_pair_rows > 0 ? (uint64_t)g->opt_pair_rows * (n + 1) + 1 : (1ull << 32) - 2;
    uint64_t G = 0, rows_pair = 0;
    for (uint64_t cand = 2; cand <= 64 && cand <= m; cand++) {
        // two largest groups: sizes are floor / ceil of m / G
        const uint64_t big = (m + cand - 1) / cand;
        if (2 * big * (n + 1) + 1 <= limit) { G = cand; rows_pair = 2 * big; break; }
    }
    if (!G) return grp_fail(g, FBG_ERR_TOO_LARGE, "MSA of %llu x %llu cells is too large for the row-pair plan", (unsigned long long)m, (unsigned long long)n);
    std::vector<std::pair<uint64_t, uint64_t>> pairs;
    for (uint64_t a = 0; a < G; a++)
        for (uint64_t b = a + 1; b < G; b++) pairs.emplace_back(a, b);
    auto lo = [&](uint64_t q) { return m * q / G; };
    int rc = parallel(g, [&](int i) -> int {
        Member &mb = g->mem[i];
        FBG_TRY(mem_reserve(g, mb, mb.pair, rows_pair * n));
        const uint8_t *whole = member_msa(g, i);
        for (size_t e = (size_t)i; e < pairs.size(); e += (size_t)nm) {
            uint64_t off = 0;
            for (uint64_t q : {pairs[e].first, pairs[e].second}) {
                const uint64_t r0 = lo(q), r1 = lo(q + 1);
                FBG_HIP_TRY(mb.ctx, hipMemcpyAsync(mb.pair.as<uint8_t>() + off * n, whole + r0 * n, (r1 - r0) * n, hipMemcpyDeviceToDevice, mb.ctx->stream));
                off += r1 - r0;
            }
            FBG_TRY(fbg_msa_set_device(mb.ctx, mb.pair.as<uint8_t>(), off, n));
            FBG_TRY(fbg_index_build(mb.ctx, 0, ignore, ignore_len));
            FBG_TRY(fbg_scan_f(mb.ctx, 0, n, disable_tricks, mb.f.as<uint64_t>()));   // max-merged: fbg.cpp:1681
        }
        FBG_TRY(fbg_sync(mb.ctx));
        return fbg_msa_set_device(mb.ctx, whole, m, n);   // the whole MSA again (fbg_block_graph reads it)
    });
    if (rc != FBG_OK) {
        for (int i = 0; i < nm; i++) (void)fbg_msa_set_device(g->mem[i].ctx, member_msa(g, i), m, n);
        return rc;
    }
    return all_reduce_max(g, [&](int i) -> void * { return g->mem[i].f.p; }, n, 8);
}

// f (or v: reversed) of the current MSA into member 0's f buffer.  f buffers are prepared by the caller (zeros, or
// the values to max-merge into).
int group_scan(fbg_group *g, int reversed, const uint8_t *ignore, uint64_t ignore_len, int disable_tricks)
{
    if (!g->have_msa) return grp_fail(g, FBG_ERR_INVALID, "no MSA loaded into the group");
    const int nm = (int)g->mem.size();
    const uint64_t m = g->m, n = g->n;
    const bool fits32 = (double)m * (double)(n + 1) + 1 < 4294967295.0;
    g->plan_used = FBG_PLAN_AUTO;
    g->parts_used = 0;
    int plan = (int)g->opt_plan;
    if (plan == FBG_PLAN_AUTO) {
        int P = 0;
        FBG_TRY(pick_partitions(g, &P));
        // one context whose index fits: the plain path.  Otherwise partitions first
        plan = (nm == 1 && P == 1 && fits32) ? FBG_PLAN_COLUMNS : FBG_PLAN_PARTITIONED;
    }
    if (plan == FBG_PLAN_PARTITIONED) {
        int done = 0;
        FBG_TRY(plan_partitioned(g, reversed, ignore, ignore_len, disable_tricks, &done));
        if (done) {
            Member &m0 = g->mem[0];
            if (reversed) FBG_TRY(fbg_scan_v(m0.ctx, 0, n, m0.f.as<uint64_t>()));
            else FBG_TRY(fbg_scan_f(m0.ctx, 0, n, disable_tricks, m0.f.as<uint64_t>()));
            g->plan_used = FBG_PLAN_PARTITIONED;
            return fbg_sync(m0.ctx);
        }
        if (g->opt_plan == FBG_PLAN_PARTITIONED) return grp_fail(g, FBG_ERR_INVALID, "the partitioned index declined this MSA (gaps, similar rows)");
    }
    if (plan != FBG_PLAN_ROW_PAIRS) {                      // a text beyond 2^32 symbols is turned away by the text build
        const int rc = plan_columns(g, reversed, ignore, ignore_len, disable_tricks);
        if (rc != FBG_ERR_TOO_LARGE) { if (rc == FBG_OK) g->plan_used = FBG_PLAN_COLUMNS; return rc; }
    }
    if (reversed) return grp_fail(g, FBG_ERR_TOO_LARGE, "non-elastic scan: text beyond 2^32 symbols that the partitioned index declined");
    FBG_TRY(plan_row_pairs(g, ignore, ignore_len, disable_tricks));
    g->plan_used = FBG_PLAN_ROW_PAIRS;
    return FBG_OK;
}

// f buffers: n (padded to a whole number of column chunks) words on every member, filled from host values or zero
int prepare_f(fbg_group *g, const uint64_t *host_init)
{
    const int nm = (int)g->mem.size();
    const uint64_t chunk = (g->n + nm - 1) / nm, words = chunk * nm;
    for (Member &mb : g->mem) {
        FBG_TRY(mem_reserve(g, mb, mb.f, words * 8));
        GRP_HIP_TRY(g, hipMemsetAsync(mb.f.p, 0, words * 8, mb.ctx->stream));
        if (host_init) GRP_HIP_TRY(g, hipMemcpyAsync(mb.f.p, host_init, g->n * 8, hipMemcpyHostToDevice, mb.ctx->stream));
    }
    return sync_all(g);
}

int share_msa(fbg_group *g, uint64_t m, uint64_t n, const std::function<int(Member &)> &fill)
{
    const int nm = (int)g->mem.size();
    g->have_msa = false;
    for (int i = 0; i < nm; i++) {
        g->mem[i].msa_owner = i;
        for (int j = 0; j < i; j++)
            if (g->mem[j].dev == g->mem[i].dev) { g->mem[i].msa_owner = g->mem[j].msa_owner; break; }
    }
    for (int i = 0; i < nm; i++)
        if (g->mem[i].msa_owner == i) FBG_TRY(mem_reserve(g, g->mem[i], g->mem[i].msa, m * n));
    FBG_TRY(parallel(g, [&](int i) -> int { return g->mem[i].msa_owner == i ? fill(g->mem[i]) : FBG_OK; }));
    for (int i = 0; i < nm; i++) {
        const int rc = fbg_msa_set_device(g->mem[i].ctx, g->mem[g->mem[i].msa_owner].msa.as<uint8_t>(), m, n);
        if (rc != FBG_OK) { set_err(g, fbg_last_error(g->mem[i].ctx)); return rc; }
    }
    g->m = m; g->n = n; g->have_msa = true;
    return FBG_OK;
}

std::string g_group_err;

} // namespace

extern "C" {

int fbg_group_create(int ndev, const int *dev_ids, fbg_group **out)
{
    if (!out) return FBG_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        g_group_err = "no HIP device visible; libfbg_hip has no CPU fallback";
        return FBG_ERR_NO_DEVICE;
    }
    std::vector<int> ids;
    if (ndev <= 0 || !dev_ids) for (int d = 0; d < (ndev > 0 ? std::min(ndev, count) : count); d++) ids.push_back(d);
    else ids.assign(dev_ids, dev_ids + ndev);
    fbg_group *g = new fbg_group();
    for (int d : ids) {
        Member mb;
        mb.dev = d;
        const int rc = fbg_ctx_create(d, &mb.ctx);
        if (rc != FBG_OK) {
            g_group_err = fbg_last_error(nullptr);
            for (Member &o : g->mem) fbg_ctx_destroy(o.ctx);
            delete g;
            return rc;
        }
        g->mem.push_back(mb);
    }
    for (size_t i = 0; i < g->mem.size(); i++)
        for (size_t j = 0; j < i; j++)
            if (g->mem[i].dev == g->mem[j].dev) g->distinct = false;
    *out = g;
    return FBG_OK;
}

void fbg_group_destroy(fbg_group *g)
{
    if (!g) return;
    for (Member &mb : g->mem) {
        (void)hipSetDevice(mb.dev);
        (void)hipStreamSynchronize(mb.ctx->stream);
        if (mb.comm) (void)g->rccl.CommDestroy(mb.comm);
        for (DevBuf *b : {&mb.msa, &mb.blobs, &mb.gmax, &mb.acc, &mb.f, &mb.pair}) fbg_release(mb.ctx, *b);
        fbg_ctx_destroy(mb.ctx);
    }
    delete g;
}

const char *fbg_group_last_error(const fbg_group *g) { return g ? g->err.c_str() : g_group_err.c_str(); }
int fbg_group_size(const fbg_group *g) { return g ? (int)g->mem.size() : 0; }
fbg_ctx *fbg_group_member(fbg_group *g, int i) { return (g && i >= 0 && i < (int)g->mem.size()) ? g->mem[i].ctx : nullptr; }
int fbg_group_plan_used(const fbg_group *g, int *partitions)
{
    if (!g) return FBG_PLAN_AUTO;
    if (partitions) *partitions = g->parts_used;
    return g->plan_used;
}

int fbg_group_set_option(fbg_group *g, const char *key, int64_t value)
{
    if (!g || !key) return FBG_ERR_INVALID;
    if (!strcmp(key, "partitions")) { g->opt_partitions = value; return FBG_OK; }
    if (!strcmp(key, "plan")) { g->opt_plan = value; return FBG_OK; }
    if (!strcmp(key, "pair_rows")) { g->opt_pair_rows = value; return FBG_OK; }
    if (!strcmp(key, "exchange")) {
        if (value == GRP_EXCHANGE_RCCL && !g->distinct && g->mem.size() > 1)
            return grp_fail(g, FBG_ERR_INVALID, "exchange=2 (RCCL) needs a device of its own for every member");
        g->opt_exchange = value;
        return FBG_OK;
    }
    for (Member &mb : g->mem) {
        const int rc = fbg_set_option(mb.ctx, key, value);
        if (rc != FBG_OK) { set_err(g, fbg_last_error(mb.ctx)); return rc; }
    }
    return FBG_OK;
}

} // extern "C"

namespace {

// decide how the members exchange, and open RCCL if that is the way
int prepare_exchange(fbg_group *g)
{
    const bool want = g->opt_exchange == GRP_EXCHANGE_RCCL ||
                      (g->opt_exchange == GRP_EXCHANGE_AUTO && g->distinct && g->mem.size() > 1);
    g->use_rccl = false;
    if (!want) return FBG_OK;
    if (!g->mem[0].comm) {
        std::string why;
        if (!g->rccl.load(why)) {
            if (g->opt_exchange == GRP_EXCHANGE_RCCL) return grp_fail(g, FBG_ERR_HIP, "%s", why.c_str());
            return FBG_OK;                                 // auto: peer copies do the same job
        }
        std::vector<int> devs;
        for (Member &mb : g->mem) devs.push_back(mb.dev);
        std::vector<ncclComm_t> comms(devs.size());
        GRP_NCCL_TRY(g, g->rccl.CommInitAll(comms.data(), (int)devs.size(), devs.data()));
        for (size_t i = 0; i < comms.size(); i++) g->mem[i].comm = comms[i];
    }
    g->use_rccl = true;
    return FBG_OK;
}

} // namespace

extern "C" {

int fbg_group_msa_load_host(fbg_group *g, const uint8_t *msa, uint64_t m, uint64_t n)
{
    if (!g || !msa || m == 0 || n == 0) return FBG_ERR_INVALID;
    g->err.clear();
    return share_msa(g, m, n, [&](Member &mb) -> int {
        FBG_HIP_TRY(mb.ctx, hipSetDevice(mb.dev));
        return fbg_upload(mb.ctx, mb.msa.p, msa, m * n);
    });
}

int fbg_group_msa_synthetic(fbg_group *g, uint64_t m, uint64_t n, uint64_t seed, uint64_t seed2, uint64_t gap_start_threshold,
                            uint32_t gap_run_len, uint64_t seed3, uint64_t n_threshold)
{
    if (!g || m == 0 || n == 0) return FBG_ERR_INVALID;
    g->err.clear();
    return share_msa(g, m, n, [&](Member &mb) -> int {
        FBG_TRY(fbg_msa_synthetic(mb.ctx, mb.msa.as<uint8_t>(), m, n, seed, seed2, gap_start_threshold, gap_run_len, seed3, n_threshold));
        return fbg_sync(mb.ctx);
    });
}

int fbg_group_scan_f(fbg_group *g, const uint8_t *ignore_chars, uint64_t ignore_len, int disable_tricks, uint64_t **d_f)
{
    if (!g || !d_f) return FBG_ERR_INVALID;
    g->err.clear();
    FBG_TRY(prepare_exchange(g));
    FBG_TRY(prepare_f(g, nullptr));
    FBG_TRY(group_scan(g, 0, ignore_chars, ignore_len, disable_tricks));
    *d_f = g->mem[0].f.as<uint64_t>();
    return FBG_OK;
}

int fbg_group_elastic_f(fbg_group *g, const uint8_t *msa, uint64_t m, uint64_t n, const uint8_t *ignore_chars, uint64_t ignore_len,
                        int disable_tricks, uint64_t *f)
{
    if (!g || !f) return FBG_ERR_INVALID;
    FBG_TRY(fbg_group_msa_load_host(g, msa, m, n));
    FBG_TRY(prepare_exchange(g));
    FBG_TRY(prepare_f(g, f));                              // f is max-merged into (fbg.cpp:1681, 3388)
    FBG_TRY(group_scan(g, 0, ignore_chars, ignore_len, disable_tricks));
    Member &m0 = g->mem[0];
    GRP_HIP_TRY(g, hipSetDevice(m0.dev));
    FBG_TRY(fbg_download(m0.ctx, f, m0.f.p, n * 8));
    if (disable_tricks && f[0] == n) return grp_fail(g, FBG_ERR_NO_SEGMENTATION, "No valid segmentation found!");
    return FBG_OK;
}

int fbg_group_repeatfree_v(fbg_group *g, const uint8_t *msa, uint64_t m, uint64_t n, uint64_t *v)
{
    if (!g || !v) return FBG_ERR_INVALID;
    FBG_TRY(fbg_group_msa_load_host(g, msa, m, n));
    FBG_TRY(prepare_exchange(g));
    FBG_TRY(prepare_f(g, nullptr));
    FBG_TRY(group_scan(g, 1, nullptr, 0, 1));
    Member &m0 = g->mem[0];
    GRP_HIP_TRY(g, hipSetDevice(m0.dev));
    return fbg_download(m0.ctx, v, m0.f.p, n * 8);
}

int fbg_group_gapped_v(fbg_group *g, const uint8_t *msa, uint64_t m, uint64_t n, uint64_t *v)
{
    if (!g || !v) return FBG_ERR_INVALID;
    FBG_TRY(fbg_group_msa_load_host(g, msa, m, n));
    FBG_TRY(prepare_exchange(g));
    FBG_TRY(prepare_f(g, nullptr));
    FBG_TRY(group_scan(g, 0, nullptr, 0, 1));              // f without the elastic tricks (dp.hip, fbg_gapped_v_from_f)
    Member &m0 = g->mem[0];
    GRP_HIP_TRY(g, hipSetDevice(m0.dev));
    FBG_TRY(mem_reserve(g, m0, m0.pair, n * 8));
    const int rc = fbg_gapped_v_from_f(m0.ctx, m0.f.as<uint64_t>(), n, m0.pair.as<uint64_t>());
    if (rc != FBG_OK) { set_err(g, fbg_last_error(m0.ctx)); return rc; }
    return fbg_download(m0.ctx, v, m0.pair.p, n * 8);
}

} // extern "C"
