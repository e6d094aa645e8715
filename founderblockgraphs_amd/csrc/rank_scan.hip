// rank_scan.hip -- the extension scan done in suffix-array order (gap-free MSAs, no ignore characters).
//
// Coloured ranks that are consecutive integers (fbg.cpp:1633-1641) are, in suffix-array order, simply
// neighbouring slots whose positions lie in the same MSA column (position mod (n+1)).  So the whole scan of
// fbg.cpp:1610-1694 can be done on the SORTED (key, position) pairs, without the inverse permutation:
//
//   k_rank_scan     per SA slot: its two neighbour LCPs are the numbers of equal leading symbols of its
//                   round-0 key and the neighbours' keys.  A slot whose neighbours are in other columns is a run
//                   of length one: g = 1 + max(LCP[r], LCP[r+1]) goes into the column's maximum (table read
//                   first, atomicMax only when it grows; extensions too small to be a column maximum -- judged
//                   from a sampled histogram and verified afterwards -- skip the table altogether).
//                   Slots with a same-column neighbour, slots tying with a neighbour on the whole key, and their
//                   direct neighbours go to a (short) candidate list instead.
//   k_tie_groups    tie groups are ordered by comparing the text beyond the K key symbols: final SA order.
//   k_runs          candidates, in final order: every maximal run of same-column neighbouring slots gets
//                   g = 1 + max(min LCP towards the run head, min LCP towards the run tail) per member
//                   (fbg.cpp:1644-1678) by one forward and one backward walk.
//   k_rank_finish   f[x] / v[j] from the column maxima (fbg.cpp:1656-1672 with rank_i(x) = x, tot_i = n).
//
// Falls back to the record path (suffix_sort.hip doubling + scan.hip) when more than N/32 slots are
// candidates (similar rows) or a tie group exceeds 64 members.
#include <cstring>
#include <vector>
#include "fbg_internal.h"
#include "text_cmp.h"
#include <rocprim/rocprim.hpp>

// number of equal leading symbols of two keys: (leading equal bits) / b, the division as a multiply-shift
// (exact for numerators <= 64: inv_b = ceil(2^16 / b))
__device__ __forceinline__ uint32_t rs_key_lcp(uint64_t a, uint64_t c, int b, int key_bits)
{
    const uint64_t d = a ^ c;
    const uint32_t bits = (uint32_t)(__clzll((long long)d) - (64 - key_bits));
    const uint32_t inv_b = (65536u + (uint32_t)b - 1) / (uint32_t)b;   // hoisted: b is uniform
    return (bits * inv_b) >> 16;
}

struct RankArgs {
    uint64_t magic;            // floor(2^64 / row_len) + 1: p / row_len == umul64hi(p, magic) for p < 2^32
    const uint64_t *keys;      // sorted round-0 keys
    uint32_t *vals;            // positions in SA order (final once the tie groups are ordered)
    const uint8_t *T;
    uint64_t N, n;             // N: number of SA slots in keys[] / vals[]
    uint64_t Ntext;            // text length (position Ntext-1 is the sentinel)
    uint64_t own_lo, own_hi;   // slots this launch owns; the rest are halo copies of the neighbouring partitions
    int first_part, last_part; // partition holds the globally first / last suffix (no neighbour beyond)
    int part_mode;             // >1 partitions: slots next to a partition edge are re-examined once the halos are in
    uint32_t row_len;          // n + 1
    uint32_t g_min;            // extensions below this cannot be a column maximum (sampled; verified afterwards)
    int b, key_bits, K, reversed;
    uint32_t *gmax;            // per column: max over rows of g (0 = no row pointer seen)
    uint32_t *cand;            // SA slots that need the run treatment (see header)
    uint32_t *blk_count;       // k_rank_scan: candidates found by each workgroup (its private region of cand[])
    uint32_t region;           // capacity of one workgroup's region
    uint32_t *pm;              // scratch parallel to cand: prefix minima of the forward walk
    unsigned long long *counters;   // [0] candidates, [1] fallback flag
};

// column of text position p, or n for '#' / sentinel positions (never a row pointer when there are no gaps)
__device__ __forceinline__ uint32_t rs_col(const RankArgs &a, uint32_t p)
{
    if (p == a.Ntext - 1) return (uint32_t)a.n;
    const uint32_t c = p - (uint32_t)__umul64hi((uint64_t)p, a.magic) * a.row_len;    // p mod (n+1)
    if (c == a.n) return c;
    return a.reversed ? (uint32_t)a.n - 1 - c : c;
}

__device__ __forceinline__ void rs_update(const RankArgs &a, uint32_t col, uint32_t g)
{
    // stale reads are only ever too small (values grow monotonically): never skips a needed update
    if (a.gmax[col] < g) atomicMax(&a.gmax[col], g);
}

__device__ __forceinline__ void rank_scan_slot(const RankArgs &a, const uint64_t k)
{
    const bool in = k < a.own_hi;                      // k >= own_lo by construction
    const int lane = threadIdx.x & 63;
    const uint64_t key = in ? a.keys[k] : 0ull;
    const uint32_t p = in ? a.vals[k] : (uint32_t)(a.Ntext - 1);
    const uint32_t col = rs_col(a, p);
    const bool has_prev = k > a.own_lo, has_next = k + 1 < a.own_hi;     // neighbours inside the owned range
    // keys / columns of the SA neighbours: adjacent lanes, extra loads at the wave's edges
    uint64_t kp = __shfl_up(key, 1, 64), kn = __shfl_down(key, 1, 64);
    uint32_t cp = __shfl_up(col, 1, 64), cn = __shfl_down(col, 1, 64);
    if (lane == 0 && in) { kp = has_prev ? a.keys[k - 1] : ~key; cp = has_prev ? rs_col(a, a.vals[k - 1]) : (uint32_t)a.n; }
    if (lane == 63 || !has_next) {
        kn = has_next ? a.keys[k + 1] : ~key;
        cn = has_next ? rs_col(a, a.vals[k + 1]) : (uint32_t)a.n;
    }
    if (!has_prev) { kp = ~key; cp = (uint32_t)a.n; }
    const bool tie = in && (kp == key || kn == key);
    // is a neighbour slot a tie?  one ballot; the wave's edge lanes look one key further
    const unsigned long long tmask = __ballot(tie);
    bool tie_prev = lane > 0 ? (tmask >> (lane - 1)) & 1ull : false;
    bool tie_next = lane < 63 ? (tmask >> (lane + 1)) & 1ull : false;
    if (lane == 0 && in && has_prev) tie_prev = kp == key || (k > a.own_lo + 1 && a.keys[k - 2] == kp);
    if (lane == 63 && in && has_next) tie_next = kn == key || (k + 2 < a.own_hi && a.keys[k + 2] == kn);
    if (!in) return;
    // candidate: ties (order not final yet), neighbours of ties (their neighbour is not final yet), slots with a
    // same-column neighbour (runs, fbg.cpp:1633-1641), and slots at a partition edge (neighbour not known yet)
    const bool near_tie = tie_prev || tie_next;
    const bool run = col != a.n && (cp == col || cn == col);
    const bool edge = a.part_mode && (k < a.own_lo + 2 || k + 2 >= a.own_hi);
    const bool cand = tie || (col != a.n && (near_tie || run || edge));
    const unsigned long long cmask = __ballot(cand);
    if (cmask) {                                       // one counter update per wave, on the workgroup's own counter
        uint32_t base = 0;
        const int leader = __ffsll((long long)cmask) - 1;
        if (lane == leader) base = atomicAdd(&a.blk_count[blockIdx.x], (uint32_t)__popcll(cmask));
        base = __shfl(base, leader, 64);
        if (cand) {
            const uint32_t slot = base + (uint32_t)__popcll(cmask & ((1ull << lane) - 1));
            if (slot < a.region) a.cand[(size_t)blockIdx.x * a.region + slot] = (uint32_t)k;
            return;
        }
    }
    if (col == a.n) return;
    const uint32_t lp = has_prev ? rs_key_lcp(kp, key, a.b, a.key_bits) : 0u;
    const uint32_t ln = has_next ? rs_key_lcp(key, kn, a.b, a.key_bits) : 0u;
    const uint32_t g = max(lp, ln) + 1;
    // near the end of a row few suffixes compete and extensions stay short: no threshold there
    const uint32_t c_raw = a.reversed ? (uint32_t)a.n - 1 - col : col;
    if (g >= a.g_min || c_raw + 64 >= a.n) rs_update(a, col, g);
}

// a few thousand workgroups, each walking 256-slot chunks of the SA (a launch of N/256 tiny workgroups spends
// more time being dispatched than working)
__global__ __launch_bounds__(256) void k_rank_scan(RankArgs a)
{
    const uint64_t nchunks = (a.own_hi - a.own_lo + 255) / 256;
    for (uint64_t c = blockIdx.x; c < nchunks; c += gridDim.x) rank_scan_slot(a, a.own_lo + c * 256 + threadIdx.x);
}

// cheap regime test + extension histogram over every 1024th block of 256 SA slots:
// counters[2] ties, counters[3] slots looked at, hist[g] = non-tie slots with extension min(g, 63)
__global__ __launch_bounds__(256) void k_tie_sample(const uint64_t *__restrict__ keys, uint64_t N, int b, int key_bits,
                                                    unsigned long long *__restrict__ counters,
                                                    unsigned int *__restrict__ hist)
{
    __shared__ unsigned int sh[64];
    __shared__ unsigned int sties;
    if (threadIdx.x < 64) sh[threadIdx.x] = 0;
    if (threadIdx.x == 0) sties = 0;
    __syncthreads();
    const uint64_t k = (uint64_t)blockIdx.x * 1024 * 256 + threadIdx.x;
    if (k < N) {
        const uint64_t key = keys[k];
        const uint64_t kp = k > 0 ? keys[k - 1] : ~key, kn = k + 1 < N ? keys[k + 1] : ~key;
        if (kp == key || kn == key) {
            atomicAdd(&sties, 1u);
        } else {
            const uint32_t lp = k > 0 ? rs_key_lcp(kp, key, b, key_bits) : 0u;
            const uint32_t ln = k + 1 < N ? rs_key_lcp(key, kn, b, key_bits) : 0u;
            atomicAdd(&sh[min(max(lp, ln) + 1, 63u)], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < 64 && sh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], sh[threadIdx.x]);
    if (threadIdx.x == 0) {
        atomicAdd(&counters[2], (unsigned long long)sties);
        atomicAdd(&counters[3], (unsigned long long)min((uint64_t)256, N - (uint64_t)blockIdx.x * 1024 * 256));
    }
}

// columns that received no value although they have rows: the threshold g_min was too optimistic for them
__global__ void k_count_unfilled(const uint32_t *__restrict__ gmax, uint64_t n, unsigned long long *__restrict__ counters)
{
    const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool miss = x < n && gmax[x] == 0;
    const unsigned long long mask = __ballot(miss);
    if ((threadIdx.x & 63) == 0 && mask) atomicAdd(&counters[1 + 3], (unsigned long long)__popcll(mask));   // counters[4]
}

// candidate regions of the workgroups -> one contiguous list (offsets = exclusive scan of the counts)
__global__ void k_cand_compact(const uint32_t *__restrict__ regions, const uint32_t *__restrict__ counts,
                               const uint32_t *__restrict__ offsets, uint32_t region, uint32_t *__restrict__ out)
{
    const uint32_t c = counts[blockIdx.x] < region ? counts[blockIdx.x] : region, o = offsets[blockIdx.x];
    for (uint32_t i = threadIdx.x; i < c; i += blockDim.x) out[o + i] = regions[(size_t)blockIdx.x * region + i];
}

// second chance for columns the threshold starved: same classification as k_rank_scan, values only
__global__ __launch_bounds__(256) void k_rank_scan_values(RankArgs a)
{
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.N) return;
    const uint64_t key = a.keys[k];
    const uint64_t kp = k > 0 ? a.keys[k - 1] : ~key, kn = k + 1 < a.N ? a.keys[k + 1] : ~key;
    if (kp == key || kn == key) return;
    const uint64_t kpp = k > 1 ? a.keys[k - 2] : ~kp, knn = k + 2 < a.N ? a.keys[k + 2] : ~kn;
    if ((k > 0 && kpp == kp) || (k + 1 < a.N && knn == kn)) return;
    const uint32_t col = rs_col(a, a.vals[k]);
    if (col == a.n) return;
    const uint32_t cp = k > 0 ? rs_col(a, a.vals[k - 1]) : (uint32_t)a.n;
    const uint32_t cn = k + 1 < a.N ? rs_col(a, a.vals[k + 1]) : (uint32_t)a.n;
    if (cp == col || cn == col) return;                // candidates are handled exactly by k_runs
    const uint32_t lp = k > 0 ? rs_key_lcp(kp, key, a.b, a.key_bits) : 0u;
    const uint32_t ln = k + 1 < a.N ? rs_key_lcp(key, kn, a.b, a.key_bits) : 0u;
    rs_update(a, col, max(lp, ln) + 1);
}

// candidates (sorted): a slot whose key equals its successor's but not its predecessor's heads a tie group
__global__ void k_tie_groups(RankArgs a, uint64_t T)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const uint32_t k0 = a.cand[t];
    const uint64_t key = a.keys[k0];
    if (k0 > a.own_lo && a.keys[k0 - 1] == key) return;                    // inside a group
    if ((uint64_t)k0 + 1 >= a.own_hi || a.keys[k0 + 1] != key) return;     // not a tie at all (ties never cross partitions)
    uint32_t s = 1;
    while ((uint64_t)k0 + s < a.own_hi && a.keys[k0 + s] == key) s++;
    if (s > 64) { a.counters[1] = 1; return; }
    // order the s suffixes by the text beyond their K common symbols (insertion sort, s is tiny)
    uint32_t pos[64];
    for (uint32_t i = 0; i < s; i++) pos[i] = a.vals[k0 + i];
    for (uint32_t i = 1; i < s; i++) {
        const uint32_t cur = pos[i];
        uint32_t j = i;
        while (j > 0) {
            const uint32_t o = pos[j - 1];
            const uint32_t h = fbg_extend_match(a.T, (uint64_t)o + a.K, (uint64_t)cur + a.K, 0);
            if (a.T[(uint64_t)o + a.K + h] < a.T[(uint64_t)cur + a.K + h]) break;      // o < cur: in place
            pos[j] = o;
            j--;
        }
        pos[j] = cur;
    }
    for (uint32_t i = 0; i < s; i++) a.vals[k0 + i] = pos[i];
}

// LCP of the suffixes in SA slots k-1 and k (final order)
__device__ __forceinline__ uint32_t rs_slot_lcp(const RankArgs &a, uint64_t k)
{
    if (k == (a.first_part ? a.own_lo : 0) || k >= (a.last_part ? a.own_hi : a.N)) return 0;   // no such neighbour
    const uint64_t x = a.keys[k - 1], y = a.keys[k];
    if (x != y) return rs_key_lcp(x, y, a.b, a.key_bits);
    return fbg_clamp_lcp(fbg_extend_match(a.T, (uint64_t)a.vals[k - 1] + a.K, (uint64_t)a.vals[k] + a.K, 0) + (uint32_t)a.K);
}

// test / debugging aid (fbg_index_download): inverse suffix array and neighbour LCPs by text position
__global__ void k_rank_materialize(RankArgs a, uint32_t *isa, uint32_t *pl, uint32_t *pr)
{
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.N) return;
    const uint32_t p = a.vals[k];
    isa[p] = (uint32_t)k;
    pl[p] = rs_slot_lcp(a, k);
    pr[p] = rs_slot_lcp(a, k + 1);
}

// candidates (sorted, final SA order): one thread per run head walks its run of same-column slots.  With
// partitions a run may begin or end in a halo (copies of the neighbouring partition's edge slots): it is walked
// in full, but only owned members update the column maxima -- the neighbour does the same from its side.
__global__ void k_runs(RankArgs a, uint64_t T)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const uint32_t k0 = a.cand[t];
    const uint32_t col = rs_col(a, a.vals[k0]);
    if (col == a.n) return;                                            // '#' / sentinel: not a row pointer
    if (k0 > a.own_lo && rs_col(a, a.vals[k0 - 1]) == col) return;     // an owned predecessor heads this run
    const uint64_t lo_slot = a.first_part ? a.own_lo : 0, hi_slot = a.last_part ? a.own_hi : a.N;
    uint64_t ks = k0;                                                  // true head: possibly inside the previous halo
    while (ks > lo_slot && rs_col(a, a.vals[ks - 1]) == col) ks--;
    if (ks == 0 && !a.first_part) { a.counters[1] = 1; return; }       // run longer than the halo
    // forward: running minimum of LCP[lb..r]   (owned members of a run are contiguous in cand[])
    uint32_t run = rs_slot_lcp(a, ks);
    uint64_t s = ks;
    for (;;) {
        if (s >= k0 && s < a.own_hi) a.pm[t + (s - k0)] = run;
        if (s + 1 >= hi_slot || rs_col(a, a.vals[s + 1]) != col) break;
        s++;
        run = min(run, rs_slot_lcp(a, s));
    }
    if (s + 1 == a.N && !a.last_part) { a.counters[1] = 1; return; }   // ran through the next halo
    // backward: running minimum of LCP[r+1..rb+1], extension, column maximum   (fbg.cpp:1656)
    uint32_t rmin = 0xffffffffu;
    for (;; s--) {
        rmin = min(rmin, rs_slot_lcp(a, s + 1));
        if (s >= k0 && s < a.own_hi) rs_update(a, col, max(a.pm[t + (s - k0)], rmin) + 1);
        if (s == ks || s <= k0) break;                                 // members before k0 belong to the neighbour
    }
}

struct FinishArgs {
    const uint32_t *gmax;
    uint64_t n, x0, x1;
    int mode, disable_tricks;
    uint64_t *out;
};

__global__ void k_rank_finish(FinishArgs a)
{
    const uint64_t x = a.x0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= a.x1) return;
    const unsigned long long g = a.gmax[x];
    if (a.mode == FBG_SCAN_V) {
        a.out[x] = g <= x + 1 ? x + 1 - g : x + 1;
        return;
    }
    unsigned long long fx = x;                                            // fbg.cpp:1618
    const bool any_active = a.disable_tricks || x != 0;                   // fullrow[], fbg.cpp:1605-1608,1621
    if (any_active && g > 0) {
        const unsigned long long gg = x + g;                              // 1657
        const unsigned long long fi = gg > a.n ? (a.disable_tricks ? a.n : a.n - 1) : gg - 1;   // 1659-1666
        fx = max(fx, fi);
    }
    a.out[x] = max((unsigned long long)a.out[x], fx);                     // 1681
}

template <class F> static int rs_with_tmp(fbg_ctx *ctx, F &&call)
{
    size_t bytes = 0;
    hipError_t e = call(nullptr, bytes);
    if (e != hipSuccess) return fbg_fail(ctx, FBG_ERR_HIP, "rocprim size query: %s", hipGetErrorString(e));
    FBG_TRY(fbg_reserve(ctx, ctx->tmp, bytes));
    size_t have = ctx->tmp.cap;
    e = call(ctx->tmp.p, have);
    if (e != hipSuccess) return fbg_fail(ctx, FBG_ERR_HIP, "rocprim call: %s", hipGetErrorString(e));
    return FBG_OK;
}

static void rs_args_init(fbg_ctx *ctx, RankArgs &a, const uint64_t *keys, uint32_t *vals, uint64_t slots, int b,
                         int key_bits, int K)
{
    a.keys = keys; a.vals = vals; a.T = ctx->text.as<uint8_t>();
    a.N = slots; a.Ntext = ctx->N; a.n = ctx->n; a.row_len = (uint32_t)(ctx->n + 1);
    a.own_lo = 0; a.own_hi = slots; a.first_part = a.last_part = 1; a.part_mode = 0;
    a.magic = ~0ull / (ctx->n + 1) + 1;
    a.b = b; a.key_bits = key_bits; a.K = K; a.reversed = ctx->reversed;
    a.gmax = ctx->gmax.as<uint32_t>();
    a.cand = nullptr; a.pm = nullptr; a.blk_count = nullptr; a.region = 0;
    a.counters = ctx->scalars.as<unsigned long long>() + 32;
    a.g_min = 0;
}

// k_rank_scan over the owned slots, then the candidates: compacted, sorted, tie groups put in final order.
// On return a.cand / a.pm name the sorted list and its scratch; *T = 0xffffffffffffffff when a workgroup's
// candidate region overflowed (similar rows: the caller takes another path).
static int rs_classify(fbg_ctx *ctx, RankArgs &a, uint64_t *T_out, int *launches)
{
    hipStream_t st = ctx->stream;
    const uint64_t own = a.own_hi - a.own_lo;
    const unsigned rs_blocks = fbg_blocks(own, 256, 256 * 32);
    const uint32_t region = (uint32_t)((own / rs_blocks) / 8 + 256);    // a workgroup may find 1/8 of its slots + slack
    FBG_TRY(fbg_reserve(ctx, ctx->list, (size_t)rs_blocks * region * 4));
    FBG_TRY(fbg_reserve(ctx, ctx->dp_c, (size_t)(rs_blocks + 1) * 4));
    FBG_TRY(fbg_reserve(ctx, ctx->dp_d, (size_t)(rs_blocks + 1) * 4));
    FBG_HIP_TRY(ctx, hipMemsetAsync(ctx->dp_c.p, 0, (size_t)(rs_blocks + 1) * 4, st));
    a.cand = ctx->list.as<uint32_t>();
    a.blk_count = ctx->dp_c.as<uint32_t>(); a.region = region;
    FBG_TRY(fbg_stage_begin(ctx, FBG_STAGE_RANK_KERNEL));
    hipLaunchKernelGGL(k_rank_scan, dim3(rs_blocks), dim3(256), 0, st, a);
    FBG_TRY(fbg_stage_end(ctx, FBG_STAGE_RANK_KERNEL, 1));
    (*launches)++;
    // candidate counts per workgroup -> offsets; total and the largest count come back to the host
    uint32_t *d_counts = ctx->dp_c.as<uint32_t>(), *d_offs = ctx->dp_d.as<uint32_t>();
    FBG_TRY(rs_with_tmp(ctx, [&](void *tmp, size_t &bytes) {
        return rocprim::exclusive_scan(tmp, bytes, d_counts, d_offs, 0u, (size_t)(rs_blocks + 1), rocprim::plus<uint32_t>(), st);
    }));
    uint32_t *d_max = reinterpret_cast<uint32_t *>(a.counters + 6);
    FBG_TRY(rs_with_tmp(ctx, [&](void *tmp, size_t &bytes) {
        return rocprim::reduce(tmp, bytes, d_counts, d_max, 0u, (size_t)rs_blocks, rocprim::maximum<uint32_t>(), st);
    }));
    uint32_t tot = 0, mx = 0;
    FBG_HIP_TRY(ctx, hipMemcpyAsync(&tot, d_offs + rs_blocks, 4, hipMemcpyDeviceToHost, st));
    FBG_HIP_TRY(ctx, hipMemcpyAsync(&mx, d_max, 4, hipMemcpyDeviceToHost, st));
    FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
    const uint64_t T = tot;
    *T_out = T;
    if (mx > region) { *T_out = ~0ull; return FBG_OK; }
    if (T > 0) {
        // candidates in SA order; tie groups first (final order), then the runs
        FBG_TRY(fbg_reserve(ctx, ctx->dp_a, T * 4));
        FBG_TRY(fbg_reserve(ctx, ctx->dp_b, T * 4));
        // regions are in SA order already (workgroup b owns chunks b, b+G, ...: not contiguous) -> compact, then sort
        uint32_t *sorted = ctx->dp_a.as<uint32_t>();
        FBG_TRY(fbg_reserve(ctx, ctx->dp_e, T * 4));
        uint32_t *flat = ctx->dp_e.as<uint32_t>();
        hipLaunchKernelGGL(k_cand_compact, dim3(rs_blocks), dim3(256), 0, st, a.cand, d_counts, d_offs, region, flat);
        FBG_TRY(rs_with_tmp(ctx, [&](void *tmp, size_t &bytes) {
            return rocprim::radix_sort_keys(tmp, bytes, flat, sorted, (size_t)T, 0u, 32u, st);
        }));
        a.cand = sorted;
        a.pm = ctx->dp_b.as<uint32_t>();
        hipLaunchKernelGGL(k_tie_groups, dim3(fbg_blocks(T, 64)), dim3(64), 0, st, a, T);
        *launches += 2;
    }
    return FBG_OK;
}

// Called by fbg_suffix_sort right after the round-0 sort.  *done = 1 when the rank-order scan covered the
// whole input (ctx->ranked set, suffix array final in vals); 0 = continue with the record path.
int fbg_rank_scan_try(fbg_ctx *ctx, const uint64_t *keys, uint32_t *vals, int b, int key_bits, int K, int *done)
{
    *done = 0;
    ctx->ranked = false;
    ctx->part_active = false;
    if (!ctx->gapfree || ctx->have_ignore || getenv("FBG_NO_RANKED")) return FBG_OK;
    const uint64_t N = ctx->N, n = ctx->n, m = ctx->m;
    hipStream_t st = ctx->stream;
    FBG_TRY(fbg_stage_begin(ctx, FBG_STAGE_RANKSCAN));
    FBG_TRY(fbg_reserve(ctx, ctx->gmax, (n + 1) * 4));
    unsigned long long *cnt = ctx->scalars.as<unsigned long long>() + 32;
    FBG_HIP_TRY(ctx, hipMemsetAsync(cnt, 0, 8 * sizeof(unsigned long long), st));
    FBG_HIP_TRY(ctx, hipMemsetAsync(ctx->gmax.p, 0, (n + 1) * 4, st));
    RankArgs a;
    rs_args_init(ctx, a, keys, vals, N, b, key_bits, K);
    int launches = 0;
    if (N > (1u << 22)) {
        // sample: (a) similar rows tie almost everywhere -> do not even try the rank-order scan;
        //         (b) extensions so small that >= 32 rows of every column are expected to exceed them cannot be
        //             a column maximum: skipping them removes almost all table reads (verified below)
        unsigned int *d_hist = reinterpret_cast<unsigned int *>(cnt + 8);
        FBG_HIP_TRY(ctx, hipMemsetAsync(d_hist, 0, 64 * sizeof(unsigned int), st));
        hipLaunchKernelGGL(k_tie_sample, dim3(fbg_blocks(N, 1024 * 256)), dim3(256), 0, st, keys, N, b, key_bits, cnt, d_hist);
        launches++;
        unsigned long long hs[4];
        unsigned int hh[64];
        FBG_HIP_TRY(ctx, hipMemcpyAsync(hs, cnt, sizeof(hs), hipMemcpyDeviceToHost, st));
        FBG_HIP_TRY(ctx, hipMemcpyAsync(hh, d_hist, sizeof(hh), hipMemcpyDeviceToHost, st));
        FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
        if (hs[2] * 16 > hs[3]) return fbg_stage_end(ctx, FBG_STAGE_RANKSCAN, launches);
        if (!getenv("FBG_RANK_NO_THRESHOLD")) {
            const double need = 32.0 / (double)m * (double)hs[3];     // sampled slots that must lie at or above g_min
            unsigned long long above = 0;
            for (int g = 63; g >= 1; g--) {
                above += hh[g];
                if ((double)above >= need) { a.g_min = (uint32_t)g; break; }
            }
        }
    }
    uint64_t T = 0;
    FBG_TRY(rs_classify(ctx, a, &T, &launches));
    if (T == ~0ull) return fbg_stage_end(ctx, FBG_STAGE_RANKSCAN, launches);      // a region overflowed: record path
    if (T > 0) {
        hipLaunchKernelGGL(k_runs, dim3(fbg_blocks(T, 64)), dim3(64), 0, st, a, T);
        launches++;
    }
    unsigned long long h[5];
    if (T > 0 || a.g_min > 1) {
        // a large tie group -> record path; a column without a value lost all its rows to the threshold -> redo
        if (a.g_min > 1)
            hipLaunchKernelGGL(k_count_unfilled, dim3(fbg_blocks(n, 256)), dim3(256), 0, st, a.gmax, n, cnt);
        FBG_HIP_TRY(ctx, hipMemcpyAsync(h, cnt, sizeof(h), hipMemcpyDeviceToHost, st));
        FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
        launches++;
        if (h[1] != 0) return fbg_stage_end(ctx, FBG_STAGE_RANKSCAN, launches);
        if (a.g_min > 1 && h[4] != 0) {
            a.g_min = 0;
            hipLaunchKernelGGL(k_rank_scan_values, dim3(fbg_blocks(N, 256)), dim3(256), 0, st, a);
            launches++;
        }
    }
    FBG_HIP_TRY(ctx, hipGetLastError());
    ctx->n_exc = 0;
    ctx->ranked = true;
    ctx->rk_b = b; ctx->rk_key_bits = key_bits; ctx->rk_K = K; ctx->rk_keys = keys;
    *done = 1;
    return fbg_stage_end(ctx, FBG_STAGE_RANKSCAN, launches);
}

// ---- partitioned index (partition.hip): this GPU holds the SA slots of one key range ----------------------
// keys / vals: FBG_PART_HALO + count + FBG_PART_HALO slots; the middle part is sorted, the halos are filled in
// by fbg_rank_part_runs once the neighbouring partitions have published their edge slots.
__global__ void k_halo_export(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ vals, uint64_t own_lo,
                              uint64_t own_hi, uint64_t ok, uint8_t *__restrict__ blob)
{
    uint64_t *bk = reinterpret_cast<uint64_t *>(blob);
    uint32_t *bv = reinterpret_cast<uint32_t *>(blob + 2 * FBG_PART_HALO * 8);
    uint64_t *tail = reinterpret_cast<uint64_t *>(blob + 2 * FBG_PART_HALO * 12);
    const uint32_t t = threadIdx.x;                    // 2 * FBG_PART_HALO threads: head slots, then tail slots
    if (ok) {
        const uint64_t k = t < FBG_PART_HALO ? own_lo + t : own_hi - 2 * FBG_PART_HALO + t;
        bk[t] = keys[k]; bv[t] = vals[k];
    } else { bk[t] = 0; bv[t] = 0; }
    if (t == 0) { tail[0] = ok; tail[1] = own_hi - own_lo; }
}

__global__ void k_halo_import(const uint8_t *__restrict__ blobs, int part, int nparts, uint64_t own_lo, uint64_t own_hi,
                              uint64_t *__restrict__ keys, uint32_t *__restrict__ vals)
{
    const uint32_t t = threadIdx.x;                    // FBG_PART_HALO threads
    if (part > 0) {                                    // tail of the previous partition -> slots [0, H)
        const uint8_t *bl = blobs + (size_t)(part - 1) * FBG_PART_HALO_BYTES;
        keys[t] = reinterpret_cast<const uint64_t *>(bl)[FBG_PART_HALO + t];
        vals[t] = reinterpret_cast<const uint32_t *>(bl + 2 * FBG_PART_HALO * 8)[FBG_PART_HALO + t];
    }
    if (part + 1 < nparts) {                           // head of the next partition -> slots [own_hi, own_hi + H)
        const uint8_t *bl = blobs + (size_t)(part + 1) * FBG_PART_HALO_BYTES;
        keys[own_hi + t] = reinterpret_cast<const uint64_t *>(bl)[t];
        vals[own_hi + t] = reinterpret_cast<const uint32_t *>(bl + 2 * FBG_PART_HALO * 8)[t];
    }
}

static void rs_part_args(fbg_ctx *ctx, RankArgs &a)
{
    rs_args_init(ctx, a, ctx->rk_keys, ctx->sa_ptr, ctx->part_count + 2 * FBG_PART_HALO, ctx->rk_b, ctx->rk_key_bits, ctx->rk_K);
    a.own_lo = FBG_PART_HALO; a.own_hi = FBG_PART_HALO + ctx->part_count;
    a.first_part = ctx->part == 0; a.last_part = ctx->part + 1 == ctx->nparts;
    a.part_mode = ctx->nparts > 1;
}

// Phase 1: classify the owned slots, order the tie groups, publish the edge slots (d_blob, FBG_PART_HALO_BYTES).
// *ok = 0: this partition cannot be handled in rank order (the flag travels in the blob; every rank sees it).
int fbg_rank_part_classify(fbg_ctx *ctx, const uint64_t *keys, uint32_t *vals, uint64_t count, int b, int key_bits, int K,
                           int pre_ok, uint8_t *d_blob, int *ok)
{
    const uint64_t n = ctx->n;
    hipStream_t st = ctx->stream;
    FBG_TRY(fbg_stage_begin(ctx, FBG_STAGE_RANKSCAN));
    int launches = 0;
    ctx->ranked = false;
    ctx->part_count = count;
    ctx->part_T = 0;
    ctx->rk_b = b; ctx->rk_key_bits = key_bits; ctx->rk_K = K; ctx->rk_keys = keys; ctx->sa_ptr = vals;
    FBG_TRY(fbg_reserve(ctx, ctx->gmax, (n + 1) * 4));
    unsigned long long *cnt = ctx->scalars.as<unsigned long long>() + 32;
    FBG_HIP_TRY(ctx, hipMemsetAsync(cnt, 0, 8 * sizeof(unsigned long long), st));
    FBG_HIP_TRY(ctx, hipMemsetAsync(ctx->gmax.p, 0, (n + 1) * 4, st));
    int good = pre_ok && count >= 2 * FBG_PART_HALO;
    RankArgs a;
    rs_part_args(ctx, a);
    if (good && count > (1u << 22)) {                  // similar rows tie almost everywhere: not for this path
        unsigned int *d_hist = reinterpret_cast<unsigned int *>(cnt + 8);
        FBG_HIP_TRY(ctx, hipMemsetAsync(d_hist, 0, 64 * sizeof(unsigned int), st));
        hipLaunchKernelGGL(k_tie_sample, dim3(fbg_blocks(count, 1024 * 256)), dim3(256), 0, st, keys + a.own_lo, count, b,
                           key_bits, cnt, d_hist);
        launches++;
        unsigned long long hs[4];
        FBG_HIP_TRY(ctx, hipMemcpyAsync(hs, cnt, sizeof(hs), hipMemcpyDeviceToHost, st));
        FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
        if (hs[2] * 16 > hs[3]) good = 0;
    }
    if (good) {
        uint64_t T = 0;
        FBG_TRY(rs_classify(ctx, a, &T, &launches));
        if (T == ~0ull) good = 0;
        else {
            ctx->part_T = T;
            if (T > 0) {                               // tie groups longer than 64 raise counters[1]
                unsigned long long h[2];
                FBG_HIP_TRY(ctx, hipMemcpyAsync(h, cnt, sizeof(h), hipMemcpyDeviceToHost, st));
                FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
                if (h[1] != 0) good = 0;
            }
        }
    }
    hipLaunchKernelGGL(k_halo_export, dim3(1), dim3(2 * FBG_PART_HALO), 0, st, keys, vals, a.own_lo, a.own_hi, (uint64_t)good, d_blob);
    launches++;
    FBG_HIP_TRY(ctx, hipGetLastError());
    *ok = good;
    return fbg_stage_end(ctx, FBG_STAGE_RANKSCAN, launches);
}

// Phase 2: halos in, runs walked, column maxima of the owned slots out (d_gmax: n + 1 words; word n = 1 when
// this partition failed, so that the max-reduction over the partitions carries the verdict).
int fbg_rank_part_runs(fbg_ctx *ctx, const uint8_t *d_blobs, uint32_t *d_gmax, int *ok)
{
    const uint64_t n = ctx->n;
    hipStream_t st = ctx->stream;
    FBG_TRY(fbg_stage_begin(ctx, FBG_STAGE_RANKSCAN));
    int launches = 0;
    std::vector<uint8_t> hb((size_t)ctx->nparts * FBG_PART_HALO_BYTES);
    FBG_HIP_TRY(ctx, hipMemcpyAsync(hb.data(), d_blobs, hb.size(), hipMemcpyDeviceToHost, st));
    FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
    int good = 1;
    for (int p = 0; p < ctx->nparts; p++) {
        uint64_t tail[2];
        memcpy(tail, hb.data() + (size_t)p * FBG_PART_HALO_BYTES + 2 * FBG_PART_HALO * 12, sizeof(tail));
        if (!tail[0]) good = 0;
    }
    if (good) {
        RankArgs a;
        rs_part_args(ctx, a);
        a.cand = ctx->dp_a.as<uint32_t>(); a.pm = ctx->dp_b.as<uint32_t>();
        hipLaunchKernelGGL(k_halo_import, dim3(1), dim3(FBG_PART_HALO), 0, st, d_blobs, ctx->part, ctx->nparts, a.own_lo,
                           a.own_hi, const_cast<uint64_t *>(a.keys), a.vals);
        launches++;
        if (ctx->part_T > 0) {
            hipLaunchKernelGGL(k_runs, dim3(fbg_blocks(ctx->part_T, 64)), dim3(64), 0, st, a, ctx->part_T);
            launches++;
            unsigned long long h[2];
            FBG_HIP_TRY(ctx, hipMemcpyAsync(h, a.counters, sizeof(h), hipMemcpyDeviceToHost, st));
            FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
            if (h[1] != 0) good = 0;                   // a run longer than the halo
        }
    }
    FBG_HIP_TRY(ctx, hipMemcpyAsync(d_gmax, ctx->gmax.p, (n + 1) * 4, hipMemcpyDeviceToDevice, st));
    const uint32_t verdict = good ? 0u : 1u;
    FBG_HIP_TRY(ctx, hipMemcpyAsync(d_gmax + n, &verdict, 4, hipMemcpyHostToDevice, st));
    FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
    *ok = good;
    return fbg_stage_end(ctx, FBG_STAGE_RANKSCAN, launches);
}

int fbg_rank_finish(fbg_ctx *ctx, uint64_t x0, uint64_t x1, int mode, int disable_tricks, uint64_t *d_out)
{
    FinishArgs f;
    f.gmax = ctx->gmax.as<uint32_t>();
    f.n = ctx->n; f.x0 = x0; f.x1 = x1; f.mode = mode; f.disable_tricks = disable_tricks; f.out = d_out;
    hipLaunchKernelGGL(k_rank_finish, dim3(fbg_blocks(x1 - x0, 256)), dim3(256), 0, ctx->stream, f);
    FBG_HIP_TRY(ctx, hipGetLastError());
    return FBG_OK;
}

// isa / lcp_prev / lcp_next by text position for fbg_index_download when the index is in rank order
int fbg_rank_materialize(fbg_ctx *ctx, uint32_t *d_isa, uint32_t *d_pl, uint32_t *d_pr)
{
    RankArgs a;
    rs_args_init(ctx, a, ctx->rk_keys, ctx->sa_ptr, ctx->N, ctx->rk_b, ctx->rk_key_bits, ctx->rk_K);
    hipLaunchKernelGGL(k_rank_materialize, dim3(fbg_blocks(ctx->N, 256)), dim3(256), 0, ctx->stream, a, d_isa, d_pl, d_pr);
    FBG_HIP_TRY(ctx, hipGetLastError());
    return FBG_OK;
}
