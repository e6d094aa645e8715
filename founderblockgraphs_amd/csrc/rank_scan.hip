// rank_scan.hip -- the extension scan done in suffix-array order (gap-free MSAs, no ignore characters).
//
// For a column without two consecutive coloured ranks the suffix-tree walk of fbg.cpp:1629-1680 gives every
// active row i the extension g_i = 1 + max(LCP[r_i], LCP[r_i + 1]) and the column only needs max_i g_i
// (f[x] = max(x, min(x + max_i g_i, n) - 1), fbg.cpp:1656-1672 with rank_i(x) = x and tot_i = n).  Both
// LCPs of a suffix are the numbers of equal leading symbols of its round-0 key and its SA neighbours' keys,
// and its column is (position mod (n+1)), so one streaming pass over the SORTED (key, position) pairs
// computes everything -- no inverse permutation, no per-position records:
//
//   k_rank_scan     per SA slot: g from the three keys, column from the position, column maximum kept in a
//                   table that is read first and only updated by atomicMax when it grows (the maximum of m
//                   values changes ~ln m times, so almost every slot costs one L2 read); a slot whose SA
//                   neighbour sits in the same column marks the column as an exception; slots that tie with a
//                   neighbour on the whole key go to a list.
//   k_tie_groups    tie groups (few, small on dissimilar rows) are ordered by comparing the text beyond the K
//                   key symbols, which also yields their LCPs; final SA order is written back.
//   k_exc_collect   for exception columns only: (rank, LCP[rank], LCP[rank+1]) of all m rows into a small
//                   dense table, consumed by k_scan_exceptions (scan.hip: hash set of ranks + pointer jumping).
//   k_rank_finish   f[x] / v[j] from the column maxima.
//
// Falls back to the record path (suffix_sort.hip doubling + scan.hip streaming scan) when more than N/32
// suffixes tie, a tie group exceeds 64 members, or more than n/8 columns are exceptions.
#include "fbg_internal.h"
#include "text_cmp.h"
#include <rocprim/rocprim.hpp>

__device__ __forceinline__ uint32_t rs_key_lcp(uint64_t a, uint64_t c, int b, int key_bits)
{
    const uint64_t d = a ^ c;
    return (uint32_t)((__clzll((long long)d) - (64 - key_bits)) / b);
}

struct RankArgs {
    uint64_t magic;            // floor(2^64 / row_len) + 1: p / row_len == umul64hi(p, magic) for p < 2^32
    const uint32_t *xbits;     // bitmap of the exception columns (k_exc_collect)
    const uint64_t *keys;      // sorted round-0 keys
    uint32_t *vals;            // positions in SA order (final once the tie groups are ordered)
    const uint8_t *T;
    uint64_t N, n;
    uint32_t row_len;          // n + 1
    int b, key_bits, K, reversed;
    uint32_t *gmax;            // per column: max over rows of g (0 = no row pointer seen)
    uint32_t *excol;           // per column: 1 = two SA-adjacent row pointers share this column
    uint32_t *ties;            // SA slots whose key equals a neighbour's
    unsigned long long *counters;   // [0] ties, [1] fallback flag
};

// column of text position p, or n for '#' / sentinel positions (never a row pointer when there are no gaps)
__device__ __forceinline__ uint32_t rs_col(const RankArgs &a, uint32_t p)
{
    if (p == a.N - 1) return (uint32_t)a.n;
    const uint32_t c = p - (uint32_t)__umul64hi((uint64_t)p, a.magic) * a.row_len;    // p mod (n+1)
    if (c == a.n) return c;
    return a.reversed ? (uint32_t)a.n - 1 - c : c;
}

__device__ __forceinline__ void rs_update(const RankArgs &a, uint32_t col, uint32_t g)
{
    // stale reads are only ever too small (values grow monotonically): never skips a needed update
    if (a.gmax[col] < g) atomicMax(&a.gmax[col], g);
}

__global__ __launch_bounds__(256) void k_rank_scan(RankArgs a)
{
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = k < a.N;
    const int lane = threadIdx.x & 63;
    const uint64_t key = in ? a.keys[k] : 0ull;
    const uint32_t p = in ? a.vals[k] : (uint32_t)(a.N - 1);
    const uint32_t col = rs_col(a, p);
    // keys / columns of the SA neighbours: adjacent lanes, one extra load at the wave's edges
    uint64_t kp = __shfl_up(key, 1, 64), kn = __shfl_down(key, 1, 64);
    uint32_t cp = __shfl_up(col, 1, 64), cn = __shfl_down(col, 1, 64);
    if (!in) return;
    if (lane == 0) { kp = k > 0 ? a.keys[k - 1] : ~key; cp = k > 0 ? rs_col(a, a.vals[k - 1]) : (uint32_t)a.n; }
    if (lane == 63 || k + 1 == a.N) {
        kn = k + 1 < a.N ? a.keys[k + 1] : ~key;
        cn = k + 1 < a.N ? rs_col(a, a.vals[k + 1]) : (uint32_t)a.n;
    }
    const bool tie = kp == key || kn == key;           // ties on all K symbols: ordered later
    const unsigned long long tmask = __ballot(tie);
    if (tmask) {                                       // one counter update per wave
        unsigned long long base = 0;
        const int leader = __ffsll((long long)tmask) - 1;
        if (lane == leader) base = atomicAdd(&a.counters[0], (unsigned long long)__popcll(tmask));
        base = __shfl(base, leader, 64);
        if (tie) {
            const unsigned long long slot = base + __popcll(tmask & ((1ull << lane) - 1));
            if (slot < a.N / 32 + 1) a.ties[slot] = (uint32_t)k;
            return;
        }
    }
    if (col == a.n) return;
    const uint32_t lp = k > 0 ? rs_key_lcp(kp, key, a.b, a.key_bits) : 0u;
    const uint32_t ln = k + 1 < a.N ? rs_key_lcp(key, kn, a.b, a.key_bits) : 0u;
    rs_update(a, col, max(lp, ln) + 1);
    // run hint (fbg.cpp:1633-1641): an SA neighbour that is the row pointer of another row in this column
    if (cp == col || cn == col) a.excol[col] = 1;
}

// cheap regime test: ties among every 1024th block of 256 SA slots
__global__ __launch_bounds__(256) void k_tie_sample(const uint64_t *__restrict__ keys, uint64_t N,
                                                    unsigned long long *__restrict__ counters)
{
    const uint64_t k = (uint64_t)blockIdx.x * 1024 * 256 + threadIdx.x;
    bool tie = false;
    if (k < N) {
        const uint64_t key = keys[k];
        tie = (k > 0 && keys[k - 1] == key) || (k + 1 < N && keys[k + 1] == key);
    }
    const unsigned long long mask = __ballot(tie);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&counters[2], (unsigned long long)__popcll(mask));
        atomicAdd(&counters[3], 64ull);
    }
}

// heads[t] = 1 when ties[t] (sorted) starts a group of consecutive SA slots with one key
__global__ void k_tie_groups(RankArgs a, uint64_t T)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const uint32_t k0 = a.ties[t];
    const uint64_t key = a.keys[k0];
    if (t > 0 && a.ties[t - 1] + 1 == k0 && a.keys[k0 - 1] == key) return;     // not the head of its group
    uint32_t s = 1;
    while ((uint64_t)k0 + s < a.N && a.keys[k0 + s] == key) s++;
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
    // LCPs and column maxima of the members (their run hints need the final neighbours: k_tie_hints)
    uint32_t prev_l = k0 > 0 ? rs_key_lcp(a.keys[k0 - 1], key, a.b, a.key_bits) : 0u;
    for (uint32_t i = 0; i < s; i++) {
        const uint32_t p = pos[i];
        const uint32_t col = rs_col(a, p);
        uint32_t next_l;
        if (i + 1 < s)
            next_l = fbg_clamp_lcp(fbg_extend_match(a.T, (uint64_t)p + a.K, (uint64_t)pos[i + 1] + a.K, 0) + (uint32_t)a.K);
        else if ((uint64_t)k0 + s < a.N)
            next_l = rs_key_lcp(key, a.keys[k0 + s], a.b, a.key_bits);
        else
            next_l = 0;
        if (col != a.n) rs_update(a, col, max(prev_l, next_l) + 1);
        prev_l = next_l;
    }
}

// run hints of the tie members, after every group has its final order (their non-tie neighbours are in the
// same column whenever the hint fires, so marking the column covers both sides)
__global__ void k_tie_hints(RankArgs a, uint64_t T)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const uint32_t k = a.ties[t];
    const uint32_t col = rs_col(a, a.vals[k]);
    if (col == a.n) return;
    bool hint = false;
    if (k > 0) hint |= rs_col(a, a.vals[k - 1]) == col;
    if ((uint64_t)k + 1 < a.N) hint |= rs_col(a, a.vals[k + 1]) == col;
    if (hint) a.excol[col] = 1;
}

// test / debugging aid (fbg_index_download): inverse suffix array and neighbour LCPs by text position
__global__ void k_rank_materialize(RankArgs a, uint32_t *__restrict__ isa, uint32_t *__restrict__ pl, uint32_t *__restrict__ pr)
{
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.N) return;
    const uint32_t p = a.vals[k];
    const uint64_t key = a.keys[k];
    uint32_t lp = 0, ln = 0;
    if (k > 0) {
        const uint64_t kq = a.keys[k - 1];
        lp = kq != key ? rs_key_lcp(kq, key, a.b, a.key_bits)
                       : fbg_clamp_lcp(fbg_extend_match(a.T, (uint64_t)a.vals[k - 1] + a.K, (uint64_t)p + a.K, 0) + (uint32_t)a.K);
    }
    if (k + 1 < a.N) {
        const uint64_t kq = a.keys[k + 1];
        ln = kq != key ? rs_key_lcp(key, kq, a.b, a.key_bits)
                       : fbg_clamp_lcp(fbg_extend_match(a.T, (uint64_t)p + a.K, (uint64_t)a.vals[k + 1] + a.K, 0) + (uint32_t)a.K);
    }
    isa[p] = (uint32_t)k; pl[p] = lp; pr[p] = ln;
}

// exception columns: slot[col] = index among the exception columns (exclusive scan of excol)
__global__ __launch_bounds__(256) void k_exc_collect(RankArgs a, const uint32_t *__restrict__ slot, uint64_t m,
                                                     uint4 *__restrict__ exc)
{
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.N) return;
    const uint32_t p = a.vals[k];
    const uint32_t col = rs_col(a, p);
    if (col == a.n || !((a.xbits[col >> 5] >> (col & 31)) & 1u)) return;
    const uint64_t key = a.keys[k];
    uint32_t lp = 0, ln = 0;
    if (k > 0) {
        const uint64_t kq = a.keys[k - 1];
        lp = kq != key ? rs_key_lcp(kq, key, a.b, a.key_bits)
                       : fbg_clamp_lcp(fbg_extend_match(a.T, (uint64_t)a.vals[k - 1] + a.K, (uint64_t)p + a.K, 0) + (uint32_t)a.K);
    }
    if (k + 1 < a.N) {
        const uint64_t kq = a.keys[k + 1];
        ln = kq != key ? rs_key_lcp(key, kq, a.b, a.key_bits)
                       : fbg_clamp_lcp(fbg_extend_match(a.T, (uint64_t)p + a.K, (uint64_t)a.vals[k + 1] + a.K, 0) + (uint32_t)a.K);
    }
    const uint64_t row = __umul64hi((uint64_t)p, a.magic);
    exc[(uint64_t)slot[col] * m + row] = make_uint4((uint32_t)k, lp, ln, 0u);
}

struct FinishArgs {
    const uint32_t *gmax, *excol;
    uint64_t n, x0, x1;
    int mode, disable_tricks;
    uint64_t *out;
};

__global__ void k_rank_finish(FinishArgs a)
{
    const uint64_t x = a.x0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= a.x1 || a.excol[x]) return;
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

__global__ void k_iota_if(const uint32_t *__restrict__ excol, const uint32_t *__restrict__ slot, uint64_t n,
                          uint32_t *__restrict__ xlist, uint32_t *__restrict__ xbits)
{
    const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool e = x < n && excol[x];
    if (e) xlist[slot[x]] = (uint32_t)x;
    const unsigned long long bal = __ballot(e);
    if ((threadIdx.x & 31) == 0 && x < n + 32)       // two 32-bit words per wave
        xbits[x >> 5] = (uint32_t)(bal >> (threadIdx.x & 32));
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

// Called by fbg_suffix_sort right after the round-0 sort.  *done = 1 when the rank-order scan covered the
// whole input (ctx->ranked set, suffix array final in vals); 0 = continue with the record path.
int fbg_rank_scan_try(fbg_ctx *ctx, const uint64_t *keys, uint32_t *vals, int b, int key_bits, int K, int *done)
{
    *done = 0;
    ctx->ranked = false;
    if (!ctx->gapfree || ctx->have_ignore || getenv("FBG_NO_RANKED")) return FBG_OK;
    const uint64_t N = ctx->N, n = ctx->n, m = ctx->m;
    hipStream_t st = ctx->stream;
    FBG_TRY(fbg_stage_begin(ctx, FBG_STAGE_RANKSCAN));
    FBG_TRY(fbg_reserve(ctx, ctx->gmax, (n + 1) * 4));
    FBG_TRY(fbg_reserve(ctx, ctx->excol, (n + 1) * 4));
    FBG_TRY(fbg_reserve(ctx, ctx->xslot, (n + 1) * 4));
    FBG_TRY(fbg_reserve(ctx, ctx->xbits, (n / 32 + 4) * 4));
    FBG_TRY(fbg_reserve(ctx, ctx->xlist, (n + 1) * 4));
    FBG_TRY(fbg_reserve(ctx, ctx->list, (N / 32 + 2) * 4));
    unsigned long long *cnt = ctx->scalars.as<unsigned long long>() + 32;
    FBG_HIP_TRY(ctx, hipMemsetAsync(cnt, 0, 4 * sizeof(unsigned long long), st));
    FBG_HIP_TRY(ctx, hipMemsetAsync(ctx->gmax.p, 0, (n + 1) * 4, st));
    FBG_HIP_TRY(ctx, hipMemsetAsync(ctx->excol.p, 0, (n + 1) * 4, st));
    RankArgs a;
    a.keys = keys; a.vals = vals; a.T = ctx->text.as<uint8_t>();
    a.N = N; a.n = n; a.row_len = (uint32_t)(n + 1);
    a.magic = ~0ull / (n + 1) + 1;
    a.xbits = ctx->xbits.as<uint32_t>();
    a.b = b; a.key_bits = key_bits; a.K = K; a.reversed = ctx->reversed;
    a.gmax = ctx->gmax.as<uint32_t>(); a.excol = ctx->excol.as<uint32_t>();
    a.ties = ctx->list.as<uint32_t>(); a.counters = cnt;
    if (N > (1u << 22)) {   // similar rows tie almost everywhere: do not even try the rank-order scan then
        hipLaunchKernelGGL(k_tie_sample, dim3(fbg_blocks(N, 1024 * 256)), dim3(256), 0, st, keys, N, cnt);
        unsigned long long hs[4];
        FBG_HIP_TRY(ctx, hipMemcpyAsync(hs, cnt, sizeof(hs), hipMemcpyDeviceToHost, st));
        FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
        if (hs[2] * 16 > hs[3]) return fbg_stage_end(ctx, FBG_STAGE_RANKSCAN, 1);
    }
    FBG_TRY(fbg_stage_begin(ctx, FBG_STAGE_RANK_KERNEL));
    hipLaunchKernelGGL(k_rank_scan, dim3(fbg_blocks(N, 256)), dim3(256), 0, st, a);
    FBG_TRY(fbg_stage_end(ctx, FBG_STAGE_RANK_KERNEL, 1));
    int launches = 1;
    unsigned long long h[2];
    FBG_HIP_TRY(ctx, hipMemcpyAsync(h, cnt, sizeof(h), hipMemcpyDeviceToHost, st));
    FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
    const uint64_t T = h[0];
    if (T > N / 32) return fbg_stage_end(ctx, FBG_STAGE_RANKSCAN, launches);       // similar rows: record path
    if (T > 0) {
        // tie slots in SA order, then one thread per group head
        FBG_TRY(fbg_reserve(ctx, ctx->dp_a, T * 4));
        uint32_t *sorted = ctx->dp_a.as<uint32_t>();
        FBG_TRY(rs_with_tmp(ctx, [&](void *tmp, size_t &bytes) {
            return rocprim::radix_sort_keys(tmp, bytes, a.ties, sorted, (size_t)T, 0u, 32u, st);
        }));
        a.ties = sorted;
        hipLaunchKernelGGL(k_tie_groups, dim3(fbg_blocks(T, 64)), dim3(64), 0, st, a, T);
        hipLaunchKernelGGL(k_tie_hints, dim3(fbg_blocks(T, 256)), dim3(256), 0, st, a, T);
        launches += 3;
        FBG_HIP_TRY(ctx, hipMemcpyAsync(h, cnt, sizeof(h), hipMemcpyDeviceToHost, st));
        FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
        if (h[1] != 0) return fbg_stage_end(ctx, FBG_STAGE_RANKSCAN, launches);    // a large tie group: record path
    }
    // exception columns: slots, list, dense (rank, lcp, lcp) table
    unsigned long long *d_ne = cnt + 2;
    FBG_TRY(rs_with_tmp(ctx, [&](void *tmp, size_t &bytes) {
        return rocprim::exclusive_scan(tmp, bytes, a.excol, ctx->xslot.as<uint32_t>(), 0u, (size_t)(n + 1),
                                       rocprim::plus<uint32_t>(), st);
    }));
    uint32_t ne = 0;
    FBG_HIP_TRY(ctx, hipMemcpyAsync(&ne, ctx->xslot.as<uint32_t>() + n, 4, hipMemcpyDeviceToHost, st));
    FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
    (void)d_ne;
    launches += 1;
    if ((uint64_t)ne > n / 8 + 16) return fbg_stage_end(ctx, FBG_STAGE_RANKSCAN, launches);   // many runs: record path
    ctx->n_exc = ne;
    if (ne > 0) {
        FBG_TRY(fbg_reserve(ctx, ctx->exc, (size_t)ne * m * 16));
        hipLaunchKernelGGL(k_iota_if, dim3(fbg_blocks(n + 32, 256)), dim3(256), 0, st, a.excol, ctx->xslot.as<uint32_t>(), n,
                           ctx->xlist.as<uint32_t>(), ctx->xbits.as<uint32_t>());
        hipLaunchKernelGGL(k_exc_collect, dim3(fbg_blocks(N, 256)), dim3(256), 0, st, a, ctx->xslot.as<uint32_t>(), m,
                           ctx->exc.as<uint4>());
        launches += 2;
    }
    FBG_HIP_TRY(ctx, hipGetLastError());
    ctx->ranked = true;
    ctx->rk_b = b; ctx->rk_key_bits = key_bits; ctx->rk_K = K; ctx->rk_keys = keys;
    *done = 1;
    return fbg_stage_end(ctx, FBG_STAGE_RANKSCAN, launches);
}

int fbg_rank_finish(fbg_ctx *ctx, uint64_t x0, uint64_t x1, int mode, int disable_tricks, uint64_t *d_out)
{
    FinishArgs f;
    f.gmax = ctx->gmax.as<uint32_t>(); f.excol = ctx->excol.as<uint32_t>();
    f.n = ctx->n; f.x0 = x0; f.x1 = x1; f.mode = mode; f.disable_tricks = disable_tricks; f.out = d_out;
    hipLaunchKernelGGL(k_rank_finish, dim3(fbg_blocks(x1 - x0, 256)), dim3(256), 0, ctx->stream, f);
    FBG_HIP_TRY(ctx, hipGetLastError());
    return FBG_OK;
}

// isa / lcp_prev / lcp_next by text position for fbg_index_download when the index is in rank order
int fbg_rank_materialize(fbg_ctx *ctx, uint32_t *d_isa, uint32_t *d_pl, uint32_t *d_pr)
{
    RankArgs a;
    a.keys = ctx->rk_keys; a.vals = ctx->sa_ptr; a.T = ctx->text.as<uint8_t>();
    a.N = ctx->N; a.n = ctx->n; a.row_len = (uint32_t)(ctx->n + 1);
    a.magic = ~0ull / (ctx->n + 1) + 1; a.xbits = nullptr;
    a.b = ctx->rk_b; a.key_bits = ctx->rk_key_bits; a.K = ctx->rk_K; a.reversed = ctx->reversed;
    a.gmax = nullptr; a.excol = nullptr; a.ties = nullptr; a.counters = nullptr;
    hipLaunchKernelGGL(k_rank_materialize, dim3(fbg_blocks(ctx->N, 256)), dim3(256), 0, ctx->stream, a, d_isa, d_pl, d_pr);
    FBG_HIP_TRY(ctx, hipGetLastError());
    return FBG_OK;
}
