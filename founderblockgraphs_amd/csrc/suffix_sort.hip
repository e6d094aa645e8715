// suffix_sort.hip -- suffix array and inverse of the indexed text, on the device.
//
// Stands in for the suffix sorting inside sdsl::construct(cst, file, 1) (fbg.cpp:428): suffixes of
// T (bytes, unique smallest sentinel 0 at the end) in unsigned-byte lexicographic order.
//
// Method: prefix doubling with early discard.
//   round 0  every suffix gets a 64-bit key holding its first K symbols (alphabet compacted to b
//            bits, K = 64/b) and all N (key, position) pairs are radix sorted;
//   round k  only suffixes still sharing their key with a neighbour are re-sorted, by
//            (rank of their group, rank of the suffix h symbols further on), h = K, 2K, 4K, ...
// rank[p] is always the SA index of the first member of p's group, so it is final as soon as the
// group is a singleton.  Device-wide sort / scan / select primitives come from rocPRIM; the key
// packing, grouping and doubling kernels are this file's.
//
// Ranks live in a 16-byte record per text position, rec[p] = {rank, lcp_prev, lcp_next, 0}, so that the
// one unavoidable random scatter of the sort (SA order -> text order) also delivers the two neighbour
// LCPs the scan needs: for suffixes whose round-0 keys differ the LCP is the number of equal leading
// symbols of the two keys -- no text access.  Bit 31 of an LCP word is the run hint: the SA neighbour
// sits in the same MSA column, i.e. the two ranks can be coloured together (fbg.cpp:1633-1641).
// Suffixes that tie on their whole key (round-0 groups) are patched after the last doubling round
// (k_fix_dirty: text comparison from symbol K on).  When ties are common (similar rows) or the MSA has
// gaps, lcp.hip recomputes both LCP words for every position instead.
#include "fbg_internal.h"
#include "text_cmp.h"
#include <rocprim/rocprim.hpp>
#include <utility>
#include <cmath>

#define SS_THREADS 256

// onesweep configuration for the N-element (u64 key, u32 position) sort: 9-bit digits, 1024 x 8 items per block;
// measured 57 ms vs 65 ms for rocPRIM's default on 1e9 pairs of 63-bit keys (scripts/micro/sortbench.hip)
using fbg_sort_config = rocprim::radix_sort_config<
    rocprim::default_config, rocprim::default_config,
    rocprim::radix_sort_onesweep_config<rocprim::kernel_config<512, 32>, rocprim::kernel_config<1024, 8>, 9,
                                        rocprim::block_radix_rank_algorithm::match>>;

// key[p] = first K symbol codes of suffix p, most significant first, zero beyond the text.
// Each thread owns PK_ITEMS consecutive positions: the first key is packed symbol by symbol, the following
// ones roll (drop the leading symbol, append one), so the cost per position is one LDS byte read.
//
// COMPACT (rank-order scan only): the code table flags separators ('#', the sentinel) with FBG_SEP; a separator
// and everything after it inside a key count as code 0, which it shares with the smallest symbol.  Such a key is
// the smallest one with its leading symbols, i.e. the suffix lands where it belongs up to ties, and the scan
// knows from the row arithmetic how many symbols of a key are real (rank_scan.hip).  Positions beyond the text
// are separators too.  LAYOUT (FBG_SLOTS_*): pairs, one packed word per suffix, or wide pairs.  FILTER: keep keys in [lo, hi) only,
// compacted in no particular order (partitioned index).
#define PK_ITEMS 8
#define FBG_SEP 0x80
struct PackArgs {
    const uint8_t *T;
    uint64_t N;
    const uint8_t *code;
    int b, K, pb;
    uint64_t *keys;            // PACKED: items
    uint32_t *vals;
    uint64_t lo, hi, cap;      // FILTER
    int nohi;
    unsigned long long *counter;
    const uint64_t *ebits = nullptr;   // PAIRS: bit 31 of the value = an irregular position among the K from p on (gapped_rank.hip)
    const uint32_t *payload = nullptr; // PAIRS: the value of position p is payload[p] (span_scan.hip: cell | flags)
    const uint8_t *key_flags = nullptr; // PAIRS: two bits per position that go below the key (span_scan.hip, 2^30 cells and more)
};

// any irregular position among [p, p + K)?  (K <= 32; the bitmap is padded beyond the text)
__device__ __forceinline__ uint32_t pack_flag(const uint64_t *__restrict__ ebits, uint64_t p, int K)
{
    const uint64_t *e = ebits + (p >> 6);
    const unsigned sh = (unsigned)(p & 63);
    uint64_t bits = e[0] >> sh;
    if (sh) bits |= e[1] << (64 - sh);
    return (bits & ((1ull << K) - 1)) ? 0x80000000u : 0u;
}

template <bool COMPACT, int LAYOUT, bool FILTER>
__global__ __launch_bounds__(SS_THREADS) void k_pack(PackArgs a)
{
    constexpr int TILE = SS_THREADS * PK_ITEMS;
    __shared__ uint8_t tile[TILE + 64];      // symbol codes of the block's positions (+K lookahead)
    __shared__ uint8_t cd[256];
    __shared__ uint64_t skeys[TILE];         // keys leave through LDS so that the global stores are coalesced
    const int b = a.b, K = a.K;
    cd[threadIdx.x] = a.code[threadIdx.x];
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * TILE;
    for (int k = threadIdx.x; k < TILE + 64; k += SS_THREADS) {
        const uint64_t p = base + k;
        tile[k] = p < a.N ? cd[a.T[p]] : (COMPACT ? FBG_SEP : 0);
    }
    __syncthreads();
    const int t0 = threadIdx.x * PK_ITEMS;
    const uint64_t mask = (K * b) >= 64 ? ~0ull : ((1ull << (K * b)) - 1);
    uint64_t key = 0;
    uint32_t seen = 0;
    for (int k = 0; k < K; k++) { const uint32_t c = tile[t0 + k]; seen |= c; key = (key << b) | c; }
#pragma unroll
    for (int i = 0; i < PK_ITEMS; i++) {
        skeys[t0 + i] = key;
        const uint32_t c = tile[t0 + K + i];
        seen |= c;
        key = ((key << b) | c) & mask;
    }
    if (COMPACT && (seen & FBG_SEP)) {       // a separator in reach (K + 8 symbols of a row end): symbol by symbol
        for (int i = 0; i < PK_ITEMS; i++) {
            uint64_t kk = 0;
            bool dead = false;
            for (int k = 0; k < K; k++) {
                const uint32_t c = tile[t0 + i + k];
                dead = dead || (c & FBG_SEP);
                kk = (kk << b) | (dead ? 0u : c);
            }
            skeys[t0 + i] = kk;
        }
    }
    __syncthreads();
    if (!FILTER) {
#pragma unroll
        for (int i = 0; i < PK_ITEMS; i++) {
            const int j = threadIdx.x + i * SS_THREADS;
            const uint64_t p = base + j;
            if (p < a.N) {
                if (LAYOUT == FBG_SLOTS_PACKED) a.keys[p] = (skeys[j] << a.pb) | p;
                else if (LAYOUT == FBG_SLOTS_WIDE) { a.keys[p] = (skeys[j] << a.pb) | (p >> 32); a.vals[p] = (uint32_t)p; }
                else { a.keys[p] = a.key_flags ? (skeys[j] << 2) | a.key_flags[p] : skeys[j]; a.vals[p] = a.payload ? a.payload[p] : (uint32_t)p | (a.ebits ? pack_flag(a.ebits, p, K) : 0u); }
            }
        }
        return;
    }
    __shared__ uint32_t wsum[SS_THREADS / 64];
    __shared__ unsigned long long s_base;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned long long keep[PK_ITEMS];
    uint32_t wtot = 0;
#pragma unroll
    for (int i = 0; i < PK_ITEMS; i++) {
        const int j = threadIdx.x + i * SS_THREADS;
        const uint64_t kj = skeys[j];
        keep[i] = __ballot(base + j < a.N && kj >= a.lo && (a.nohi || kj < a.hi));
        wtot += (uint32_t)__popcll(keep[i]);
    }
    if (lane == 0) wsum[w] = wtot;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int q = 0; q < SS_THREADS / 64; q++) tot += wsum[q];
        s_base = tot ? atomicAdd(a.counter, (unsigned long long)tot) : 0ull;     // one counter update per workgroup
    }
    __syncthreads();
    uint64_t off = s_base;
    for (int q = 0; q < w; q++) off += wsum[q];
#pragma unroll
    for (int i = 0; i < PK_ITEMS; i++) {
        const int j = threadIdx.x + i * SS_THREADS;
        if ((keep[i] >> lane) & 1ull) {
            const uint64_t o = off + (uint64_t)__popcll(keep[i] & ((1ull << lane) - 1));
            if (o < a.cap) {
                const uint64_t p = base + j;
                if (LAYOUT == FBG_SLOTS_PACKED) a.keys[o] = (skeys[j] << a.pb) | p;
                else if (LAYOUT == FBG_SLOTS_WIDE) { a.keys[o] = (skeys[j] << a.pb) | (p >> 32); a.vals[o] = (uint32_t)p; }
                else { a.keys[o] = a.key_flags ? (skeys[j] << 2) | a.key_flags[p] : skeys[j]; a.vals[o] = a.payload ? a.payload[p] : (uint32_t)p | (a.ebits ? pack_flag(a.ebits, p, K) : 0u); }
            }
        }
        off += (uint64_t)__popcll(keep[i]);
    }
}

static void launch_pack(fbg_ctx *ctx, const KeyGeom &g, bool filter, PackArgs &a)
{
    const dim3 grid(fbg_blocks(a.N, SS_THREADS * PK_ITEMS)), block(SS_THREADS);
    hipStream_t st = ctx->stream;
#define FBG_PACK(C, L, F) hipLaunchKernelGGL((k_pack<C, L, F>), grid, block, 0, st, a)
    if (g.compact) {
        if (g.packed) { if (filter) FBG_PACK(true, FBG_SLOTS_PACKED, true); else FBG_PACK(true, FBG_SLOTS_PACKED, false); }
        else if (g.wide) { if (filter) FBG_PACK(true, FBG_SLOTS_WIDE, true); else FBG_PACK(true, FBG_SLOTS_WIDE, false); }
        else { if (filter) FBG_PACK(true, FBG_SLOTS_PAIRS, true); else FBG_PACK(true, FBG_SLOTS_PAIRS, false); }
    } else {
        if (filter) FBG_PACK(false, FBG_SLOTS_PAIRS, true); else FBG_PACK(false, FBG_SLOTS_PAIRS, false);
    }
#undef FBG_PACK
}

// keys of every stride-th position (same definition as k_pack): splitters and the regime pre-test
__global__ void k_sample_keys(const uint8_t *__restrict__ T, uint64_t N, const uint8_t *__restrict__ code, int b, int K,
                              int compact, uint64_t stride, uint64_t S, uint64_t *__restrict__ out)
{
    // the code table in LDS, the text 8 bytes at a time (T is zero padded beyond N): K dependent byte loads through
    // a table in global memory made this small kernel take 0.3 ms
    __shared__ uint8_t cd[256];
    if (threadIdx.x < 256) cd[threadIdx.x] = code[threadIdx.x];
    __syncthreads();
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S) return;
    const uint64_t p = i * stride;
    uint64_t key = 0;
    bool dead = false;
    for (int k0 = 0; k0 < K; k0 += 8) {
        const uint64_t x = p + k0 < N ? fbg_load8(T, p + k0) : 0ull;
        for (int k = k0; k < K && k < k0 + 8; k++) {
            const uint32_t c = p + k < N ? cd[(x >> (8 * (k - k0))) & 255u] : (compact ? FBG_SEP : 0);
            dead = dead || (compact && (c & FBG_SEP));
            key = (key << b) | (dead ? 0u : c);
        }
    }
    out[i] = key;
}

__global__ void k_count_equal_neighbours(const uint64_t *__restrict__ keys, uint64_t S, unsigned long long *__restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool eq = i + 1 < S && keys[i] == keys[i + 1];
    __shared__ uint32_t blk;
    if (threadIdx.x == 0) blk = 0;
    __syncthreads();
    const unsigned long long m = __ballot(eq);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&blk, (uint32_t)__popcll(m));
    __syncthreads();
    if (threadIdx.x == 0 && blk) atomicAdd(out, (unsigned long long)blk);
}

// grp[r] = r if key[r] starts a group, else 0  (max-scan turns it into the group head index)
__global__ void k_mark_heads(const uint64_t *__restrict__ keys, uint64_t cnt, const uint32_t *__restrict__ where,
                             uint32_t *__restrict__ grp)
{
    uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= cnt) return;
    bool head = k == 0 || keys[k] != keys[k - 1];
    uint32_t r = where ? where[k] : (uint32_t)k;
    grp[k] = head ? r : 0u;
}

// number of equal leading symbols of two K-symbol keys (b bits per symbol, right aligned in key_bits)
__device__ __forceinline__ uint32_t key_lcp_inv(uint64_t a, uint64_t c, uint32_t inv_b, int key_bits)
{
    const uint64_t d = a ^ c;
    const uint32_t bits = (uint32_t)(__clzll((long long)d) - (64 - key_bits));
    return (bits * inv_b) >> 16;
}
__device__ __forceinline__ uint32_t key_lcp(uint64_t a, uint64_t c, int b, int key_bits)
{
    return key_lcp_inv(a, c, (65536u + (uint32_t)b - 1) / (uint32_t)b, key_bits);   // exact for numerators <= 64
}

// Run hint of two SA-adjacent positions: can both be the pointer of an active row in one column?
// Row pointer p is current for the columns (column of the previous symbol of its row, column of p]
// (fbg.cpp:1687-1691); gap-free MSAs: exactly one column, p mod (n+1).
struct ColTest {
    const uint32_t *colT;  // MSAs with gaps: column of every text position ('#', sentinel: n); else nullptr
    uint32_t row_len;      // gap-free: n + 1
    uint32_t n;
    uint32_t last;         // N - 1, the sentinel, never a row pointer
    __device__ __forceinline__ uint2 span(uint32_t p) const
    {
        if (p == last) return make_uint2(1u, 0u);                        // empty
        if (!colT) { const uint32_t c = p % row_len; return make_uint2(c, c); }
        const uint32_t c = colT[p];
        uint32_t lo = 0;
        if (p > 0) { const uint32_t cp = colT[p - 1]; lo = cp >= n ? 0u : cp + 1; }
        return make_uint2(lo, c < n ? c : n - 1);
    }
    static __device__ __forceinline__ uint32_t meet(uint2 a, uint2 c)
    {
        return max(a.x, c.x) <= min(a.y, c.y) ? 0x80000000u : 0u;
    }
    __device__ __forceinline__ uint32_t same(uint32_t p, uint32_t q) const { return meet(span(p), span(q)); }
};

// round 0, after the max-scan: one 16-byte record per text position (rank, key-derived neighbour LCPs
// with run hints), unresolved flags.  vals[] doubles as the suffix array.
__global__ void k_apply_groups0(const uint32_t *__restrict__ grp, const uint32_t *__restrict__ vals,
                                const uint64_t *__restrict__ keys, uint64_t N, int b, int key_bits, ColTest ct,
                                uint4 *__restrict__ rec, uint8_t *__restrict__ flags)
{
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = k < N;
    const int lane = threadIdx.x & 63;
    const uint32_t g = in ? grp[k] : 0u, p = in ? vals[k] : ct.last;
    const uint64_t key = in ? keys[k] : 0ull;
    // column span of this suffix's position; the neighbours' spans come from the adjacent lanes
    const uint2 sp = ct.span(p);
    uint2 sprev = make_uint2(__shfl_up(sp.x, 1, 64), __shfl_up(sp.y, 1, 64));
    uint2 snext = make_uint2(__shfl_down(sp.x, 1, 64), __shfl_down(sp.y, 1, 64));
    if (!in) return;
    uint32_t lp = 0, ln = 0;
    bool head = g == (uint32_t)k, next_head = true;
    if (k > 0) {
        const uint64_t kp = keys[k - 1];
        if (kp != key) {
            if (lane == 0) sprev = ct.span(vals[k - 1]);
            lp = key_lcp(kp, key, b, key_bits) | ColTest::meet(sprev, sp);
        }
    }
    if (k + 1 < N) {
        const uint64_t kn = keys[k + 1];
        next_head = kn != key;
        if (next_head) {
            if (lane == 63) snext = ct.span(vals[k + 1]);
            ln = key_lcp(key, kn, b, key_bits) | ColTest::meet(sp, snext);
        }
    }
    rec[p] = make_uint4(g, lp, ln, 0u);
    flags[k] = !(head && next_head);
}

// ---- the same records without a random scatter -------------------------------------------------------------
// rec[p] = ... above writes 16 bytes at a random place per suffix: every one of them a read-modify-write of a memory
// line (36 ms for 5*10^8 suffixes).  The positions are a permutation of 0..N-1, so the way back to text order is a
// sort whose bucket sizes are known exactly beforehand: two splitting passes on the leading bits of p (tiles
// regrouped in LDS and written in runs, as msd_sort.hip does; every bucket owns precisely the stretch of the
// positions it covers, no capacities to bet on) and a last pass that drops each leaf of 4096 positions into place
// inside LDS and writes it out whole.  5 x 16 bytes of streamed traffic per suffix instead: 18 ms for the same input.
#define BP_THREADS 1024                        // 512-thread tiles (two workgroups per CU) were 30 % slower: shorter runs
#define BP_ITEMS 8
#define BP_TILE (BP_THREADS * BP_ITEMS)
#define BP_MAXD 1024
#define BP_LEAF_BITS 12
#define BP_LEAF (1u << BP_LEAF_BITS)

struct BpArgs {
    uint64_t N;
    int shift;                       // bucket of an element = p >> shift; its stretch starts at bucket << shift
    uint32_t nd;                     // buckets a tile can meet (power of two): LDS digit = bucket & (nd - 1)
    const uint4 *in;                 // pass B: stretches of 1 << seg_shift elements
    int seg_shift;
    uint32_t tiles_per_seg;
    uint4 *out;
    unsigned long long *cursor;      // [N >> shift + 1], zeroed: fill of every bucket
    unsigned long long *flag;
    // pass A makes its elements from the sorted keys (k_apply_groups0)
    const uint32_t *grp, *vals;
    const uint64_t *keys;
    int b, key_bits;
    ColTest ct;
    uint8_t *flags;
    // MSAs with gaps: first column and number of gaps of every window of BP_WIN text positions (no row end inside),
    // from which a position's column span is bounded without touching colT
    const uint2 *win;
    const uint32_t *win_lo;          // exact lower end of the span of every window's first position
};

#define BP_WIN_BITS 10
#define BP_WIN (1u << BP_WIN_BITS)
#define BP_WIN_NONE 0xffffffffu

// one wave per window: a window that holds a '#' or the sentinel (column n) bounds nothing
__global__ __launch_bounds__(256) void k_bp_windows(const uint32_t *__restrict__ colT, uint64_t N, uint32_t n,
                                                    uint2 *__restrict__ win, uint32_t *__restrict__ win_lo)
{
    const uint64_t w = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t lo = w << BP_WIN_BITS;
    if (lo >= N) return;
    const uint64_t hi = min(N, lo + BP_WIN);
    bool bad = false;
    for (uint64_t q = lo + lane; q < hi; q += 64) bad = bad || colT[q] >= n;
    if (lane == 0) { const uint32_t cp = lo > 0 ? colT[lo - 1] : n; win_lo[w] = cp >= n ? 0u : cp + 1; }   // as ColTest::span
    if (__ballot(bad)) { if (lane == 0) win[w] = make_uint2(BP_WIN_NONE, 0u); return; }
    if (lane == 0) {
        const uint32_t first = colT[lo], last = colT[hi - 1];
        win[w] = make_uint2(first, last - first - (uint32_t)(hi - 1 - lo));
    }
}

// an interval that contains the column span of text position p (ColTest::span), from the window tables alone
__device__ __forceinline__ uint2 bp_span_bound(const BpArgs &a, uint32_t p)
{
    if (p == a.ct.last) return make_uint2(1u, 0u);
    const uint32_t o = p & (BP_WIN - 1);
    const uint2 w = a.win[p >> BP_WIN_BITS];
    if (w.x == BP_WIN_NONE) return make_uint2(0u, a.ct.n);
    // the window's first position is every 1024th: with a loose lower end half of them would go on to the exact test
    return make_uint2(o ? w.x + o : a.win_lo[p >> BP_WIN_BITS], w.x + o + w.y);
}

// e[r] (valid for r with j = threadIdx.x + r * BP_THREADS < have; e.x = position) -> runs in the buckets' stretches
__device__ __forceinline__ void bp_regroup_write(const BpArgs &a, uint4 (&e)[BP_ITEMS], uint32_t have, uint4 *buf, uint32_t *cnt,
                                                 uint32_t *loff, unsigned long long *gbase, uint32_t *wsum)
{
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t rk[BP_ITEMS];
#pragma unroll
    for (int r = 0; r < BP_ITEMS; r++) {
        const uint32_t j = threadIdx.x + r * BP_THREADS;
        rk[r] = j < have ? atomicAdd(&cnt[(e[r].x >> a.shift) & (a.nd - 1)], 1u) : 0u;
    }
    __syncthreads();
    {   // exclusive offsets of the digit counts; one global atomic per non-empty digit reserves its run
        const uint32_t c = cnt[threadIdx.x];                       // BP_THREADS == BP_MAXD
        uint32_t inc = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d, 64); if ((int)lane >= d) inc += o; }
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        uint32_t pre = 0;
        for (uint32_t q = 0; q < wv; q++) pre += wsum[q];
        loff[threadIdx.x] = pre + inc - c;
        gbase[threadIdx.x] = 0;
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < BP_ITEMS; r++) {
        const uint32_t j = threadIdx.x + r * BP_THREADS;
        if (j < have) {
            const uint32_t d = (e[r].x >> a.shift) & (a.nd - 1);
            buf[loff[d] + rk[r]] = e[r];
            if (rk[r] == 0) {                                      // first of its digit in this tile: reserve the run
                const uint64_t bucket = (uint64_t)e[r].x >> a.shift;
                gbase[d] = (bucket << a.shift) + atomicAdd(&a.cursor[bucket], (unsigned long long)cnt[d]);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < BP_ITEMS; r++) {
        const uint32_t j = threadIdx.x + r * BP_THREADS;
        if (j < have) {
            const uint4 x = buf[j];
            const uint32_t d = (x.x >> a.shift) & (a.nd - 1);
            const uint64_t at = gbase[d] + (j - loff[d]);
            const uint64_t end = min(a.N, (((uint64_t)x.x >> a.shift) + 1) << a.shift);
            if (at < end) a.out[at] = x;
            else *a.flag = 1;                                      // cannot happen for a permutation
        }
    }
}

// first splitting pass: k_apply_groups0's record of every suffix, sent towards its position instead of written there.
// (As two kernels -- a streaming one that writes the records in suffix order, then k_bp_split over the whole array --
// the same work took 15 ms instead of 12.)
__global__ __launch_bounds__(BP_THREADS) void k_bp_groups0(BpArgs a)
{
    __shared__ uint4 buf[BP_TILE];
    __shared__ uint32_t cnt[BP_MAXD], loff[BP_MAXD];
    __shared__ unsigned long long gbase[BP_MAXD];
    __shared__ uint32_t wsum[BP_THREADS / 64];
    const uint64_t first = (uint64_t)blockIdx.x * BP_TILE;
    const uint32_t have = (uint32_t)min((uint64_t)BP_TILE, a.N - first);
    cnt[threadIdx.x] = 0;
    uint4 e[BP_ITEMS];
    // all loads of the tile first, then the window tables, then keys and spans of the whole tile into LDS, where every
    // element finds its two SA neighbours (the tile's first and last element fetch theirs from memory)
    const bool bounded = a.ct.colT != nullptr;
    const uint32_t inv_b = (65536u + (uint32_t)a.b - 1) / (uint32_t)a.b;
    uint32_t gg[BP_ITEMS], pp[BP_ITEMS];
    uint64_t key[BP_ITEMS];
#pragma unroll
    for (int r = 0; r < BP_ITEMS; r++) {
        const uint32_t j = threadIdx.x + r * BP_THREADS;
        const uint64_t k = first + j;
        const bool in = j < have;
        gg[r] = in ? a.grp[k] : 0u;
        pp[r] = in ? a.vals[k] : a.ct.last;
        key[r] = in ? a.keys[k] : 0ull;
    }
    uint2 sp[BP_ITEMS];
    if (bounded) {
        // the run hints need the column spans of SA neighbours.  With gaps a span costs two reads of colT at a random
        // place: bounds from the window tables (small enough to stay in L2) rule nearly every pair out first.  The
        // table reads of the thread's eight slots go out together -- one after the other (the branches of
        // bp_span_bound between them) this kernel spent 88 % of its wave cycles waiting for them
        uint2 wv[BP_ITEMS];
        uint32_t wl[BP_ITEMS];
#pragma unroll
        for (int r = 0; r < BP_ITEMS; r++) wv[r] = a.win[pp[r] >> BP_WIN_BITS];
#pragma unroll
        for (int r = 0; r < BP_ITEMS; r++) wl[r] = (pp[r] & (BP_WIN - 1)) ? 0u : a.win_lo[pp[r] >> BP_WIN_BITS];
#pragma unroll
        for (int r = 0; r < BP_ITEMS; r++) {
            const uint32_t p = pp[r], o = p & (BP_WIN - 1);
            if (p == a.ct.last) sp[r] = make_uint2(1u, 0u);
            else if (wv[r].x == BP_WIN_NONE) sp[r] = make_uint2(0u, a.ct.n);
            else sp[r] = make_uint2(o ? wv[r].x + o : wl[r], wv[r].x + o + wv[r].y);
        }
    } else {
#pragma unroll
        for (int r = 0; r < BP_ITEMS; r++) sp[r] = a.ct.span(pp[r]);
    }
#pragma unroll
    for (int r = 0; r < BP_ITEMS; r++)
        buf[threadIdx.x + r * BP_THREADS] = make_uint4((uint32_t)key[r], (uint32_t)(key[r] >> 32), sp[r].x, sp[r].y);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < BP_ITEMS; r++) {
        const uint32_t j = threadIdx.x + r * BP_THREADS;
        const uint64_t k = first + j;
        const bool in = j < have;
        const uint32_t p = pp[r];
        uint32_t lp = 0, ln = 0;
        const bool head = gg[r] == (uint32_t)k;
        bool next_head = true;
        if (in && k > 0) {
            uint64_t kp;
            uint2 sprev;
            if (j > 0) { const uint4 q = buf[j - 1]; kp = (uint64_t)q.y << 32 | q.x; sprev = make_uint2(q.z, q.w); }
            else { kp = a.keys[k - 1]; const uint32_t pq = a.vals[k - 1]; sprev = bounded ? bp_span_bound(a, pq) : a.ct.span(pq); }
            if (kp != key[r]) {
                uint32_t hint = ColTest::meet(sprev, sp[r]);
                if (hint && bounded) hint = a.ct.same(a.vals[k - 1], p);
                lp = key_lcp_inv(kp, key[r], inv_b, a.key_bits) | hint;
            }
        }
        if (in && k + 1 < a.N) {
            uint64_t kn;
            uint2 snext;
            if (j + 1 < have) { const uint4 q = buf[j + 1]; kn = (uint64_t)q.y << 32 | q.x; snext = make_uint2(q.z, q.w); }
            else { kn = a.keys[k + 1]; const uint32_t pq = a.vals[k + 1]; snext = bounded ? bp_span_bound(a, pq) : a.ct.span(pq); }
            next_head = kn != key[r];
            if (next_head) {
                uint32_t hint = ColTest::meet(sp[r], snext);
                if (hint && bounded) hint = a.ct.same(p, a.vals[k + 1]);
                ln = key_lcp_inv(key[r], kn, inv_b, a.key_bits) | hint;
            }
        }
        e[r] = make_uint4(p, gg[r], lp, ln);
        if (in) a.flags[k] = !(head && next_head);
    }
    bp_regroup_write(a, e, have, buf, cnt, loff, gbase, wsum);
}

// second splitting pass: a stretch of the first, tile by tile, by the next bits of the position
__global__ __launch_bounds__(BP_THREADS) void k_bp_split(BpArgs a)
{
    __shared__ uint4 buf[BP_TILE];
    __shared__ uint32_t cnt[BP_MAXD], loff[BP_MAXD];
    __shared__ unsigned long long gbase[BP_MAXD];
    __shared__ uint32_t wsum[BP_THREADS / 64];
    const uint64_t seg = blockIdx.x / a.tiles_per_seg;
    const uint64_t seg_lo = seg << a.seg_shift, seg_hi = min(a.N, (seg + 1) << a.seg_shift);
    const uint64_t first = seg_lo + (uint64_t)(blockIdx.x % a.tiles_per_seg) * BP_TILE;
    cnt[threadIdx.x] = 0;
    uint4 e[BP_ITEMS];
    const uint32_t have = first < seg_hi ? (uint32_t)min((uint64_t)BP_TILE, seg_hi - first) : 0u;
#pragma unroll
    for (int r = 0; r < BP_ITEMS; r++) {
        const uint32_t j = threadIdx.x + r * BP_THREADS;
        e[r] = j < have ? a.in[first + j] : make_uint4(0u, 0u, 0u, 0u);
    }
    if (have == 0) return;                                         // uniform
    bp_regroup_write(a, e, have, buf, cnt, loff, gbase, wsum);
}

// last pass: the BP_LEAF positions of a leaf, each dropped at its own place in LDS, leave as finished records
__global__ __launch_bounds__(512) void k_bp_leaf(const uint4 *__restrict__ in, uint64_t N, uint4 *__restrict__ rec,
                                                 unsigned long long *__restrict__ flag)
{
    __shared__ uint32_t sg[BP_LEAF], sl[BP_LEAF], sn[BP_LEAF];
    const uint64_t lo = (uint64_t)blockIdx.x << BP_LEAF_BITS;
    const uint32_t have = (uint32_t)min((uint64_t)BP_LEAF, N - lo);
    for (uint32_t j = threadIdx.x; j < have; j += blockDim.x) {
        const uint4 x = in[lo + j];
        const uint32_t i = x.x - (uint32_t)lo;
        if (i < have) { sg[i] = x.y; sl[i] = x.z; sn[i] = x.w; }
        else *flag = 1;
    }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < have; j += blockDim.x) rec[lo + j] = make_uint4(sg[j], sl[j], sn[j], 0u);
}

// doubling rounds: new group heads -> rank word of the records, suffix array slots, unresolved flags
__global__ void k_apply_groups(const uint32_t *__restrict__ grp, const uint32_t *__restrict__ vals, uint64_t cnt,
                               const uint32_t *__restrict__ where, uint4 *__restrict__ rec,
                               uint32_t *__restrict__ sa, uint8_t *__restrict__ flags)
{
    uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= cnt) return;
    const uint32_t r = where[k], g = grp[k], p = vals[k];
    rec[p].x = g;
    sa[r] = p;
    const bool head = g == r;
    const bool next_head = (k + 1 == cnt) || (grp[k + 1] == where[k + 1]);
    flags[k] = !(head && next_head);
}

// After the last round: members of round-0 tie groups get their final neighbour LCPs and hints.
// dirty[] = SA indices of those members.  A neighbour with a different round-0 key keeps its key-derived
// LCP, but its hint depends on who finally sits next to it, so that word is rewritten from here too
// (both sides compute the identical word).
__global__ void k_fix_dirty(const uint32_t *__restrict__ dirty, uint64_t cnt, const uint32_t *__restrict__ sa,
                            const uint64_t *__restrict__ key0, const uint8_t *__restrict__ T, uint64_t N, int b,
                            int key_bits, int K, ColTest ct, uint4 *__restrict__ rec)
{
    uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= cnt) return;
    const uint32_t r = dirty[k], p = sa[r];
    const uint64_t key = key0[r];
    uint32_t lp = 0, ln = 0;
    if (r > 0) {
        const uint32_t q = sa[r - 1];
        const uint64_t kq = key0[r - 1];
        if (kq == key) {
            lp = fbg_clamp_lcp(fbg_extend_match(T, (uint64_t)q + K, (uint64_t)p + K, 0) + (uint32_t)K) | ct.same(q, p);
        } else {
            lp = key_lcp(kq, key, b, key_bits) | ct.same(q, p);
            rec[q].z = lp;
        }
    }
    if ((uint64_t)r + 1 < N) {
        const uint32_t q = sa[r + 1];
        const uint64_t kq = key0[r + 1];
        if (kq == key) {
            ln = fbg_clamp_lcp(fbg_extend_match(T, (uint64_t)p + K, (uint64_t)q + K, 0) + (uint32_t)K) | ct.same(p, q);
        } else {
            ln = key_lcp(key, kq, b, key_bits) | ct.same(p, q);
            rec[q].y = ln;
        }
    }
    rec[p].y = lp;
    rec[p].z = ln;
}

// indices of the set flags (bytes), ascending: per tile of 16 KiB of flags a count, an exclusive scan of the counts
// (small), then every tile writes its indices at its offset.  (rocPRIM's select over a byte array took 5 ms for
// 5 * 10^8 flags: these two passes read 1 byte per element at full width, 0.3 ms.)
#define FC_THREADS 256
#define FC_TILE (FC_THREADS * 64)
template <bool WRITE>
__global__ __launch_bounds__(FC_THREADS) void k_flag_compact(const uint8_t *__restrict__ flags, uint64_t cnt, uint32_t *__restrict__ tile_cnt,
                                                             uint32_t *__restrict__ sel)
{
    __shared__ uint32_t lds[FC_THREADS / 64];
    const uint64_t base = (uint64_t)blockIdx.x * FC_TILE + (uint64_t)threadIdx.x * 64;
    // 64 flags per thread: four 16-byte loads (flags[] is allocated in whole tiles)
    uint32_t bits[2] = {0, 0};
    const uint4 *src = reinterpret_cast<const uint4 *>(flags + base);
#pragma unroll
    for (int q = 0; q < 4; q++) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (base + 16 * q < cnt) v = src[q];
        const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const bool set = ((w4[i >> 2] >> (8 * (i & 3))) & 0xffu) != 0 && base + 16 * q + i < cnt;
            bits[(16 * q + i) >> 5] |= (set ? 1u : 0u) << ((16 * q + i) & 31);
        }
    }
    const uint32_t mine = (uint32_t)__popc(bits[0]) + (uint32_t)__popc(bits[1]);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
    if (lane == 63) lds[w] = inc;
    __syncthreads();
    uint32_t pre = 0, tot = 0;
    for (int k = 0; k < FC_THREADS / 64; k++) { if (k < w) pre += lds[k]; tot += lds[k]; }
    if (!WRITE) {
        if (threadIdx.x == 0) tile_cnt[blockIdx.x] = tot;
        return;
    }
    uint32_t at = tile_cnt[blockIdx.x] + pre + inc - mine;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        uint32_t b = bits[h];
        while (b) {
            const int i = __ffs((int)b) - 1;
            b &= b - 1;
            sel[at++] = (uint32_t)(base + 32 * h + i);
        }
    }
}

struct KeepIdx { uint32_t r, p, g; };

// compact the unresolved members: (SA index, position, group head) triples in SA order
__global__ void k_gather_unresolved(const uint32_t *__restrict__ sel, uint64_t cnt, const uint32_t *__restrict__ where_old,
                                    const uint32_t *__restrict__ vals_old, const uint32_t *__restrict__ grp_old,
                                    uint32_t *__restrict__ where_new, uint32_t *__restrict__ vals_new,
                                    uint32_t *__restrict__ grp_new)
{
    uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= cnt) return;
    uint32_t s = sel[k];
    where_new[k] = where_old ? where_old[s] : s;
    vals_new[k] = vals_old[s];
    grp_new[k] = grp_old[s];
}

// key = (group head, rank of the suffix h further on)
__global__ void k_doubling_keys(const uint32_t *__restrict__ vals, const uint32_t *__restrict__ grp, uint64_t cnt,
                                const uint4 *__restrict__ rec, uint64_t h, uint64_t N, int nb,
                                uint64_t *__restrict__ keys)
{
    uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= cnt) return;
    uint64_t q = (uint64_t)vals[k] + h;
    uint32_t r2 = q < N ? rec[q].x : 0u;   // q < N always holds for unresolved suffixes (unique sentinel)
    keys[k] = ((uint64_t)grp[k] << nb) | r2;                  // both below 2^nb: 2 nb key bits to sort, not 64
}

__global__ void k_iota(uint32_t *__restrict__ a, uint64_t cnt)
{
    uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < cnt) a[k] = (uint32_t)k;
}

static int ensure_tmp(fbg_ctx *ctx, size_t bytes) { return fbg_reserve(ctx, ctx->tmp, bytes); }

template <class F> static int with_tmp(fbg_ctx *ctx, F &&call)
{
    size_t bytes = 0;
    hipError_t e = call(nullptr, bytes);
    if (e != hipSuccess) return fbg_fail(ctx, FBG_ERR_HIP, "rocprim size query: %s", hipGetErrorString(e));
    FBG_TRY(ensure_tmp(ctx, bytes));
    size_t have = ctx->tmp.cap;
    e = call(ctx->tmp.p, have);
    if (e != hipSuccess)
        return fbg_fail(ctx, e == hipErrorOutOfMemory ? FBG_ERR_OOM : FBG_ERR_HIP, "rocprim call: %s", hipGetErrorString(e));
    return FBG_OK;
}

// Alphabet compaction (order preserving) and the key geometry: b bits per symbol, K symbols per key.
// compact = true asks for the separator-free coding of the rank-order scan; g->compact tells whether it applies.
int fbg_key_setup(fbg_ctx *ctx, bool compact, KeyGeom *g, int *launches)
{
    const uint64_t N = ctx->N;
    hipStream_t st = ctx->stream;
    uint8_t *d_code = ctx->small.as<uint8_t>() + 2048;    // small: [0,256) ignore table, [2048,2304) code table, [4096,6144) histogram
    const uint64_t *hist = ctx->byte_hist;                // made by fbg_build_text
    if (hist[0] != 1)
        return fbg_fail(ctx, FBG_ERR_INVALID, "the MSA contains a NUL byte; the text needs a unique 0 sentinel");
    // the compact coding needs '#' to be smaller than every symbol of the rows, and room for the separator flag
    int below = 0, real = 0;
    for (int c = 1; c < 256; c++)
        if (hist[c] && c != '#') { real++; if (c < '#') below++; }
    compact = compact && below == 0 && real >= 1 && real < FBG_SEP;
    uint8_t code[256];
    int sigma = 0;
    if (compact) {
        for (int c = 0; c < 256; c++) {
            if (c == 0 || c == '#') { code[c] = FBG_SEP; continue; }
            code[c] = (uint8_t)sigma;
            if (hist[c]) sigma++;
        }
    } else {
        for (int c = 0; c < 256; c++) { code[c] = (uint8_t)sigma; if (hist[c]) sigma++; }
    }
    int b = 1;
    while ((1 << b) < sigma) b++;
    // symbols per key: enough that only a few percent of the suffixes tie on the whole key (those are ordered by
    // text comparison afterwards), rounded up to whole 9-bit radix passes, at most 64 bits.  The symbol entropy H
    // of the text tells how many symbols that takes: ties ~ N * 2^(-H*K)  =>  K >= (log2 N + 5) / H.
    double H = 0;
    for (int c = 1; c < 256; c++)
        if (hist[c] && !(compact && c == '#')) { const double q = (double)hist[c] / (double)N; H -= q * log2(q); }
    if (H < 0.05) H = 0.05;
    int K = (int)ceil((log2((double)N) + 5.0) / H);
    if (ctx->opt.full_keys || K > 64 / b) K = 64 / b;
    {
        const int passes = (K * b + 8) / 9;                  // 9-bit digits
        const int Kfill = (9 * passes) / b;                  // symbols that fit the same number of passes
        K = Kfill < 64 / b ? Kfill : 64 / b;
    }
    g->compact = compact; g->packed = false; g->wide = false; g->pb = 0;
    if (N >= (1ull << 32) || (compact && ctx->opt.force_wide)) {
        // positions beyond 32 bits: their high bits ride in the low bits of the key word
        if (!compact) return fbg_fail(ctx, FBG_ERR_TOO_LARGE, "text length %llu needs >32-bit positions", (unsigned long long)N);
        int pb = 1;
        while ((1ull << (32 + pb)) < N) pb++;
        g->wide = true; g->pb = pb;
        if (K > (64 - pb) / b) K = (64 - pb) / b;
    } else if (compact && !ctx->opt.no_packed) {
        // one 64-bit word per suffix if the position leaves room for enough symbols (ties up to ~10% are fine:
        // small tie groups are settled inside the scan)
        int pb = 1;
        while ((1ull << pb) < N) pb++;
        const int Kroom = (64 - pb) / b;
        const int Kmin = (int)ceil((log2((double)N) + 3.2) / H);
        if (Kroom >= Kmin) { g->packed = true; g->pb = pb; if (K > Kroom) K = Kroom; }
    }
    g->b = b; g->K = K; g->key_bits = K * b; g->d_code = d_code;
    FBG_HIP_TRY(ctx, hipMemcpyAsync(d_code, code, 256, hipMemcpyHostToDevice, st));
    FBG_HIP_TRY(ctx, hipStreamSynchronize(st));  // code[] lives on this stack frame
    *launches += 1;
    return FBG_OK;
}

// sorted (by key) slots of the first `count` entries of the A buffers -> B buffers (+ offset)
static int sort_slots(fbg_ctx *ctx, const KeyGeom &g, uint64_t count, uint64_t out_offset)
{
    hipStream_t st = ctx->stream;
    uint64_t *ka = ctx->keysA.as<uint64_t>(), *kb = ctx->keysB.as<uint64_t>() + out_offset;
    if (g.packed) {
        // rocPRIM 7.2's merge-sort path (inputs below a few million words) mis-sorts a bit range that starts above
        // bit 0 and ends at bit 64 (scripts/micro/sortcheck.hip); whole words sort correctly and cost nothing there
        unsigned lo = (unsigned)g.pb;
        const unsigned hi = (unsigned)(g.pb + g.key_bits);
        if (hi == 64 && count < (1u << 23)) lo = 0;
        return with_tmp(ctx, [&](void *tmp, size_t &bytes) {
            return rocprim::radix_sort_keys<fbg_sort_config>(tmp, bytes, ka, kb, (size_t)count, lo, hi, st);
        });
    }
    uint32_t *va = ctx->valsA.as<uint32_t>(), *vb = ctx->valsB.as<uint32_t>() + out_offset;
    unsigned lo = g.wide ? (unsigned)g.pb : ctx->sp_key_flags ? 2u : 0u;     // (span_scan.hip: two flag bits below the key)
    const unsigned hi = lo + (unsigned)g.key_bits;
    if (hi == 64 && count < (1u << 23)) lo = 0;        // same caution as above
    return with_tmp(ctx, [&](void *tmp, size_t &bytes) {
        return rocprim::radix_sort_pairs<fbg_sort_config>(tmp, bytes, ka, kb, va, vb, (size_t)count, lo, hi, st);
    });
}

// Rows that resemble each other tie on almost every key: the rank-order scan is not for them.  Cheap look before
// the sort: twins among the keys of a sample (the exact test follows after the sort, rank_scan.hip).
static int sample_says_similar(fbg_ctx *ctx, const KeyGeom &g, bool *similar, int *launches)
{
    const uint64_t N = ctx->N;
    hipStream_t st = ctx->stream;
    *similar = false;
    if (N < (1u << 22)) return FBG_OK;
    const uint64_t S = 1u << 20, stride = N / S;
    FBG_TRY(fbg_reserve(ctx, ctx->dp_g, S * 8));
    FBG_TRY(fbg_reserve(ctx, ctx->dp_h, S * 8));
    uint64_t *smp = ctx->dp_g.as<uint64_t>(), *srt = ctx->dp_h.as<uint64_t>();
    unsigned long long *d_count = ctx->scalars.as<unsigned long long>() + 8;
    FBG_HIP_TRY(ctx, hipMemsetAsync(d_count, 0, 8, st));
    hipLaunchKernelGGL(k_sample_keys, dim3(fbg_blocks(S, 256)), dim3(256), 0, st, ctx->text.as<uint8_t>(), N, g.d_code, g.b, g.K,
                       g.compact ? 1 : 0, stride, S, smp);
    FBG_TRY(with_tmp(ctx, [&](void *tmp, size_t &bytes) {
        return rocprim::radix_sort_keys(tmp, bytes, smp, srt, (size_t)S, 0u, (unsigned)g.key_bits, st);
    }));
    hipLaunchKernelGGL(k_count_equal_neighbours, dim3(fbg_blocks(S, 256)), dim3(256), 0, st, srt, S, d_count);
    unsigned long long twins = 0;
    FBG_HIP_TRY(ctx, hipMemcpyAsync(&twins, d_count, 8, hipMemcpyDeviceToHost, st));
    FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
    *launches += 3;
    // a suffix with one twin somewhere shows up as a sample twin with probability S/N: estimated tie fraction
    const double est = (double)twins * (double)N / ((double)S * (double)S);
    *similar = est > 0.5;
    return FBG_OK;
}

int fbg_suffix_sort(fbg_ctx *ctx)
{
    const uint64_t N = ctx->N;
    hipStream_t st = ctx->stream;
    FBG_TRY(fbg_stage_begin(ctx, FBG_STAGE_SUFFIX_SORT));
    int launches = 0;
    const uint8_t *T = ctx->text.as<uint8_t>();
    ctx->part_active = false;
    ctx->ranked = false;
    ctx->granked = false;
    ctx->grs_ebits = nullptr;
    ctx->grs_flagged = false;
    KeyGeom g;

    // ---- gap-free MSAs: compact keys, sort, and the whole extension scan in rank order (rank_scan.hip) -------
    if (ctx->gapfree && !ctx->have_ignore && !ctx->opt.no_ranked) {
        // (a streamed upload has set the keys up and run pass 1 of the sort already: same geometry)
        if (!fbg_msd_pre_geom(ctx, &g)) FBG_TRY(fbg_key_setup(ctx, true, &g, &launches));
        bool similar = false;
        FBG_TRY(sample_says_similar(ctx, g, &similar, &launches));
        // packed slots of a large text: three-pass MSD sort fused with the key packing (msd_sort.hip); else, or when
        // its optimistic bucket capacities do not hold, pack and sort with rocPRIM's onesweep
        uint64_t *sorted = nullptr;
        int msd_ok = 0;
        FBG_TRY(fbg_msd_sort(ctx, g, &sorted, &msd_ok, &launches));
        if (!msd_ok) {
            FBG_TRY(fbg_reserve(ctx, ctx->keysA, N * 8));
            FBG_TRY(fbg_reserve(ctx, ctx->keysB, N * 8));
            if (!g.packed) {
                FBG_TRY(fbg_reserve(ctx, ctx->valsA, N * 4));
                FBG_TRY(fbg_reserve(ctx, ctx->valsB, N * 4));
            }
            PackArgs pa;
            pa.T = T; pa.N = N; pa.code = g.d_code; pa.b = g.b; pa.K = g.K; pa.pb = g.pb;
            pa.keys = ctx->keysA.as<uint64_t>(); pa.vals = ctx->valsA.as<uint32_t>();
            pa.lo = pa.hi = pa.cap = 0; pa.nohi = 1; pa.counter = nullptr;
            launch_pack(ctx, g, false, pa);
            launches++;
            FBG_TRY(sort_slots(ctx, g, N, 0));
            sorted = ctx->keysB.as<uint64_t>();
        }
        uint32_t *svals = g.packed ? nullptr : ctx->valsB.as<uint32_t>();
        int done = 0;
        // rows that differ: the scan slot by slot (rank_scan.hip).  Rows that resemble each other (judged from key twins in a
        // sample, or by the slot-level scan itself, which gives up where a quarter of the slots tie): group by group
        if (!similar && ctx->opt.pure_scan != 1) FBG_TRY(fbg_rank_scan_try(ctx, sorted, svals, g, &done));
        if (!done && ctx->opt.pure_scan != -1) FBG_TRY(fbg_pure_scan_try(ctx, sorted, svals, g, &done));
        if (done) {
            FBG_HIP_TRY(ctx, hipGetLastError());
            return fbg_stage_end(ctx, FBG_STAGE_SUFFIX_SORT, launches);
        }
    }

    // ---- round 0: sort all suffixes by their first K symbols -------------------------------
    FBG_TRY(fbg_key_setup(ctx, false, &g, &launches));
    int b = g.b, K = g.K, key_bits = g.key_bits;
    FBG_TRY(fbg_reserve(ctx, ctx->keysA, N * 8));
    FBG_TRY(fbg_reserve(ctx, ctx->keysB, N * 8));
    FBG_TRY(fbg_reserve(ctx, ctx->valsA, N * 4));
    FBG_TRY(fbg_reserve(ctx, ctx->valsB, N * 4));
    uint64_t *keysA = ctx->keysA.as<uint64_t>(), *keysB = ctx->keysB.as<uint64_t>();
    uint32_t *valsA = ctx->valsA.as<uint32_t>(), *valsB = ctx->valsB.as<uint32_t>();
    uint32_t *grp = nullptr;
    uint4 *rec = nullptr;
    uint8_t *flags = nullptr;
    uint32_t *sa = valsB;                     // the sorted positions ARE the suffix array
    ctx->sa_ptr = sa;
    ColTest ct;
    ct.colT = ctx->gapfree ? nullptr : ctx->colT.as<uint32_t>();
    ct.row_len = (uint32_t)(ctx->n + 1);
    ct.n = (uint32_t)ctx->n;
    ct.last = (uint32_t)(N - 1);
    // MSAs with gaps / ignore characters: the scan in rank order follows the sort (gapped_rank.hip); its table of the
    // text's irregular positions is made now, the pack kernels fold it into the values' top bit
    ctx->pairs_similar = false;
    ctx->spanned = false;
    ctx->sort_payload = nullptr;
    ctx->sp_key_flags = false;
    // rows that resemble each other (twins among the keys of a sample; option span_scan = 1: whatever the rows): the
    // group-level scan on column spans (span_scan.hip), whose sort carries cells instead of positions
    bool try_span = fbg_span_eligible(ctx, g);
    if (try_span && ctx->opt.span_scan != 1 && ctx->opt.span_scan != 3) {
        bool similar = false;
        FBG_TRY(sample_says_similar(ctx, g, &similar, &launches));
        try_span = similar;
    }
    const bool try_grs = !try_span && (!ctx->gapfree || ctx->have_ignore) && !ctx->reversed && !ctx->grs_skip && ctx->opt.gapped_rank != -1 && K <= 32;
    if (try_grs || try_span) FBG_TRY(fbg_grs_prepare(ctx, &launches));
    if (try_span) {
        ctx->grs_ebits = nullptr; ctx->grs_flagged = false;           // (the bitmap goes into the cells' flags instead)
        ctx->sp_key_flags = fbg_span_key_flags(ctx, g);               // (2^30 cells and more: may shorten the key by a symbol)
        K = g.K; key_bits = g.key_bits;
        FBG_TRY(fbg_span_prepare(ctx, g, &launches));
        ctx->sort_payload = ctx->sp_cells.as<uint32_t>();
    }
    {
        // three passes of a sample sort fused with the key packing (msd_sort_pairs.hip) where the sizes suit it; else,
        // or when a capacity does not hold, pack and sort with rocPRIM's onesweep
        int ss_ok = 0;
        FBG_TRY(fbg_sample_sort_pairs(ctx, g, &ss_ok, &launches));
        if (!ss_ok) {
            PackArgs pa;
            pa.T = T; pa.N = N; pa.code = g.d_code; pa.b = b; pa.K = K; pa.pb = 0;
            // the sample sort reserves larger buffers before it can decline: the pointers taken above may be stale
            pa.keys = ctx->keysA.as<uint64_t>(); pa.vals = ctx->valsA.as<uint32_t>();
            pa.lo = pa.hi = pa.cap = 0; pa.nohi = 1; pa.counter = nullptr;
            pa.ebits = ctx->grs_ebits;
            pa.payload = ctx->sort_payload;
            pa.key_flags = ctx->sp_key_flags ? ctx->sp_flagT.as<uint8_t>() : nullptr;
            launch_pack(ctx, g, false, pa);
            launches++;
            FBG_TRY(sort_slots(ctx, g, N, 0));
        }
        ctx->sort_payload = nullptr;
        ctx->sp_key_flags_sorted = ctx->sp_key_flags;
        ctx->sp_key_flags = false;                                    // (sort_slots serves the later rounds too)
        // the sample sort sizes its buffers itself: take the pointers again
        keysA = ctx->keysA.as<uint64_t>(); keysB = ctx->keysB.as<uint64_t>();
        valsA = ctx->valsA.as<uint32_t>(); valsB = ctx->valsB.as<uint32_t>();
        sa = valsB;
        ctx->sa_ptr = sa;
    }
    // ---- MSAs with gaps / ignore characters: the extension scan in rank order on these slots (gapped_rank.hip) ----
    ctx->grs_ebits = nullptr;
    if (try_span) {
        int done = 0;
        FBG_TRY(fbg_span_try(ctx, keysB, valsB, g, &done));
        if (done) {
            FBG_HIP_TRY(ctx, hipGetLastError());
            return fbg_stage_end(ctx, FBG_STAGE_SUFFIX_SORT, launches);
        }
        // declined: the values are text positions again, the record path goes on from here
    } else if (try_grs && ctx->pairs_similar) FBG_TRY(fbg_grs_strip(ctx, valsB));    // similar rows: tie groups of hundreds, not for that scan
    else if (try_grs) {
        int done = 0;
        FBG_TRY(fbg_grs_try(ctx, keysB, valsB, g, &done));
        if (done) {
            FBG_HIP_TRY(ctx, hipGetLastError());
            return fbg_stage_end(ctx, FBG_STAGE_SUFFIX_SORT, launches);
        }
        FBG_TRY(fbg_grs_strip(ctx, valsB));        // the record path reads the values as the suffix array
    }
    FBG_TRY(fbg_build_cell_tables(ctx));       // the per-cell tables of the record path's column scan
    FBG_TRY(fbg_reserve(ctx, ctx->grp, N * 4));
    FBG_TRY(fbg_reserve(ctx, ctx->flags, (N + FC_TILE - 1) / FC_TILE * FC_TILE));   // whole tiles: k_flag_compact loads 16 bytes at a time
    FBG_TRY(fbg_reserve(ctx, ctx->rec, N * 16));
    grp = ctx->grp.as<uint32_t>();
    rec = ctx->rec.as<uint4>();
    flags = ctx->flags.as<uint8_t>();
    // groups of equal keys -> records (rank = SA index of the group head, key-derived LCPs), unresolved flags
    hipLaunchKernelGGL(k_mark_heads, dim3(fbg_blocks(N, 256)), dim3(256), 0, st, keysB, N, (const uint32_t *)nullptr, grp);
    FBG_TRY(with_tmp(ctx, [&](void *tmp, size_t &bytes) {
        return rocprim::inclusive_scan(tmp, bytes, grp, grp, (size_t)N, rocprim::maximum<uint32_t>(), st);
    }));
    // FBG_BP_MIN lowers the size from which the records travel to their positions in passes (tests); FBG_RECORD_SCATTER
    // keeps the direct scatter
    const uint64_t bp_min = ctx->opt.bp_min >= 0 ? (uint64_t)ctx->opt.bp_min : (1ull << 24);
    const bool by_position = N >= bp_min && N > 2 * BP_LEAF && N <= 1500000000ull && !ctx->opt.record_scatter;
    if (by_position) {
        int bits = 0;
        while ((1ull << bits) < N) bits++;
        const int S = bits - BP_LEAF_BITS, d1 = (S + 1) / 2, d2 = S - d1;      // two splitting passes, then leaves
        FBG_TRY(fbg_reserve(ctx, ctx->msd_w, N * 16));
        FBG_TRY(fbg_reserve(ctx, ctx->msd_v, N * 16));
        const uint64_t nleaf = (N + BP_LEAF - 1) >> BP_LEAF_BITS, ntop = (N >> (d2 + BP_LEAF_BITS)) + 1;
        const uint64_t nwin = (N >> BP_WIN_BITS) + 1;
        FBG_TRY(fbg_reserve(ctx, ctx->list, (ntop + nleaf + 2) * 8 + 3 * nwin * 4));
        unsigned long long *cur1 = ctx->list.as<unsigned long long>(), *cur2 = cur1 + ntop + 1;
        uint2 *win = reinterpret_cast<uint2 *>(cur1 + ntop + nleaf + 2);
        uint32_t *win_lo = reinterpret_cast<uint32_t *>(win + nwin);
        if (ct.colT) {
            hipLaunchKernelGGL(k_bp_windows, dim3(fbg_blocks(nwin, 4)), dim3(256), 0, st, ct.colT, N, ct.n, win, win_lo);
            launches++;
        }
        unsigned long long *bflag = ctx->scalars.as<unsigned long long>() + 120;
        FBG_HIP_TRY(ctx, hipMemsetAsync(cur1, 0, (ntop + nleaf + 2) * 8, st));
        FBG_HIP_TRY(ctx, hipMemsetAsync(bflag, 0, 8, st));
        BpArgs a;
        a.N = N; a.flag = bflag;
        a.grp = grp; a.vals = valsB; a.keys = keysB; a.b = b; a.key_bits = key_bits; a.ct = ct; a.flags = flags;
        a.win = win; a.win_lo = win_lo;
        a.shift = d2 + BP_LEAF_BITS; a.nd = 1u << d1; a.in = nullptr; a.seg_shift = 0; a.tiles_per_seg = 0;
        a.out = ctx->msd_w.as<uint4>(); a.cursor = cur1;
        hipLaunchKernelGGL(k_bp_groups0, dim3(fbg_blocks(N, BP_TILE)), dim3(BP_THREADS), 0, st, a);
        a.in = ctx->msd_w.as<uint4>(); a.seg_shift = d2 + BP_LEAF_BITS;
        a.tiles_per_seg = (uint32_t)(((1ull << a.seg_shift) + BP_TILE - 1) / BP_TILE);
        a.shift = BP_LEAF_BITS; a.nd = 1u << d2; a.out = ctx->msd_v.as<uint4>(); a.cursor = cur2;
        hipLaunchKernelGGL(k_bp_split, dim3((unsigned)(ntop * a.tiles_per_seg)), dim3(BP_THREADS), 0, st, a);
        hipLaunchKernelGGL(k_bp_leaf, dim3((unsigned)nleaf), dim3(512), 0, st, ctx->msd_v.as<uint4>(), N, rec, bflag);
        unsigned long long hf = 0;
        FBG_HIP_TRY(ctx, hipMemcpyAsync(&hf, bflag, 8, hipMemcpyDeviceToHost, st));
        FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
        launches += 6;
        if (hf != 0) return fbg_fail(ctx, FBG_ERR_HIP, "records by position: the sorted positions are not a permutation");
    } else {
        hipLaunchKernelGGL(k_apply_groups0, dim3(fbg_blocks(N, 256)), dim3(256), 0, st, grp, valsB, keysB, N, b, key_bits,
                           ct, rec, flags);
        launches += 4;
    }

    // ---- doubling rounds on the unresolved suffixes ----------------------------------------
    // state of the compacted list (in SA order): where[k] = SA index, vals[k] = position, grp[k]
    uint64_t cnt = N;
    const uint32_t *where_cur = nullptr;      // nullptr = identity (round 0)
    uint32_t *vals_cur = valsB, *grp_cur = grp;
    FBG_TRY(fbg_reserve(ctx, ctx->list, N * 4));
    uint32_t *sel = ctx->list.as<uint32_t>();
    DevBuf *wbuf[2] = {&ctx->dp_a, &ctx->dp_b}, *vbuf[2] = {&ctx->dp_c, &ctx->dp_d}, *gbuf[2] = {&ctx->dp_e, &ctx->dp_f};
    int pp = 0;
    uint64_t h = (uint64_t)K;
    uint64_t dirty_cnt = 0;
    ctx->lcp_from_keys = false;
    uint64_t *dkeys_in = keysA, *dkeys_out = nullptr;
    for (int round = 1;; round++) {
        // select unresolved members of the current list
        unsigned long long hc = 0;
        {
            const unsigned ftiles = fbg_blocks(cnt, FC_TILE);
            FBG_TRY(fbg_reserve(ctx, ctx->ps_a, ((size_t)ftiles + 1) * 4));
            uint32_t *tile_cnt = ctx->ps_a.as<uint32_t>();
            hipLaunchKernelGGL((k_flag_compact<false>), dim3(ftiles), dim3(FC_THREADS), 0, st, flags, cnt, tile_cnt, sel);
            FBG_HIP_TRY(ctx, hipMemsetAsync(tile_cnt + ftiles, 0, 4, st));
            FBG_TRY(with_tmp(ctx, [&](void *tmp, size_t &bytes) {
                return rocprim::exclusive_scan(tmp, bytes, tile_cnt, tile_cnt, 0u, (size_t)ftiles + 1, rocprim::plus<uint32_t>(), st);
            }));
            hipLaunchKernelGGL((k_flag_compact<true>), dim3(ftiles), dim3(FC_THREADS), 0, st, flags, cnt, tile_cnt, sel);
            uint32_t h32 = 0;
            FBG_HIP_TRY(ctx, hipMemcpyAsync(&h32, tile_cnt + ftiles, 4, hipMemcpyDeviceToHost, st));
            FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
            hc = h32;
        }
        launches += 3;
        if (round == 1) {
            // few ties: the key-derived LCPs stand, only the tie groups are patched afterwards;
            // the round-0 keys (keysB) must then survive the doubling rounds
            ctx->lcp_from_keys = hc <= N / 32 && !ctx->opt.lcp_text;
            if (ctx->lcp_from_keys && hc > 0) {
                dirty_cnt = hc;
                FBG_TRY(fbg_reserve(ctx, ctx->io_d, hc * 4));
                FBG_HIP_TRY(ctx, hipMemcpyAsync(ctx->io_d.p, sel, hc * 4, hipMemcpyDeviceToDevice, st));
                FBG_TRY(fbg_reserve(ctx, ctx->dp_g, hc * 8));
                FBG_TRY(fbg_reserve(ctx, ctx->dp_h, hc * 8));
                dkeys_in = ctx->dp_g.as<uint64_t>();
                dkeys_out = ctx->dp_h.as<uint64_t>();
            }
        }
        if (hc == 0) break;
        if (round > 64) return fbg_fail(ctx, FBG_ERR_HIP, "suffix sort did not converge");
        if (!dkeys_out) dkeys_out = keysB;
        const uint64_t ncnt = hc;
        FBG_TRY(fbg_reserve(ctx, *wbuf[pp], ncnt * 4));
        FBG_TRY(fbg_reserve(ctx, *vbuf[pp], ncnt * 4));
        FBG_TRY(fbg_reserve(ctx, *gbuf[pp], ncnt * 4));
        uint32_t *where_new = wbuf[pp]->as<uint32_t>(), *vals_new = vbuf[pp]->as<uint32_t>(),
                 *grp_new = gbuf[pp]->as<uint32_t>();
        hipLaunchKernelGGL(k_gather_unresolved, dim3(fbg_blocks(ncnt, 256)), dim3(256), 0, st, sel, ncnt, where_cur,
                           vals_cur, grp_cur, where_new, vals_new, grp_new);
        int nb = 1;
        while ((1ull << nb) < N) nb++;
        hipLaunchKernelGGL(k_doubling_keys, dim3(fbg_blocks(ncnt, 256)), dim3(256), 0, st, vals_new, grp_new, ncnt,
                           rec, h, N, nb, dkeys_in);
        // sort by (group, rank at +h); the sorted positions go back to the same SA slots
        FBG_TRY(with_tmp(ctx, [&](void *tmp, size_t &bytes) {
            if (ncnt >= (1u << 23))
                return rocprim::radix_sort_pairs<fbg_sort_config>(tmp, bytes, dkeys_in, dkeys_out, vals_new, valsA, (size_t)ncnt, 0u,
                                                                  (unsigned)(2 * nb), st);
            return rocprim::radix_sort_pairs(tmp, bytes, dkeys_in, dkeys_out, vals_new, valsA, (size_t)ncnt, 0u, (unsigned)(2 * nb), st);
        }));
        hipLaunchKernelGGL(k_mark_heads, dim3(fbg_blocks(ncnt, 256)), dim3(256), 0, st, dkeys_out, ncnt, where_new, grp_new);
        FBG_TRY(with_tmp(ctx, [&](void *tmp, size_t &bytes) {
            return rocprim::inclusive_scan(tmp, bytes, grp_new, grp_new, (size_t)ncnt, rocprim::maximum<uint32_t>(), st);
        }));
        hipLaunchKernelGGL(k_apply_groups, dim3(fbg_blocks(ncnt, 256)), dim3(256), 0, st, grp_new, valsA, ncnt,
                           where_new, rec, sa, flags);
        // the sorted positions become the list's values for the next round
        FBG_HIP_TRY(ctx, hipMemcpyAsync(vals_new, valsA, ncnt * 4, hipMemcpyDeviceToDevice, st));
        launches += 7;
        where_cur = where_new; vals_cur = vals_new; grp_cur = grp_new;
        cnt = ncnt;
        pp ^= 1;
        h *= 2;
    }
    if (ctx->lcp_from_keys && dirty_cnt > 0) {
        hipLaunchKernelGGL(k_fix_dirty, dim3(fbg_blocks(dirty_cnt, 256)), dim3(256), 0, st, ctx->io_d.as<uint32_t>(),
                           dirty_cnt, sa, keysB, T, N, b, key_bits, K, ct, rec);
        launches += 1;
    }
    FBG_HIP_TRY(ctx, hipGetLastError());
    return fbg_stage_end(ctx, FBG_STAGE_SUFFIX_SORT, launches);
}

// ---- partitioned index: this GPU sorts only the suffixes whose key lies in its range ------------------------
// Every rank of a multi-GPU job holds the whole text (MSAs are small next to their index: 1 byte per symbol
// against 16-24 bytes of sort state), computes the same splitters from the same key sample, and then packs,
// sorts and scans only its own key range: sort state and sort time divide by the number of GPUs.  The ranks
// exchange nothing but FBG_PART_HALO edge slots each and the per-column maxima (rank_scan.hip, distributed.py).
int fbg_part_sort(fbg_ctx *ctx, int part, int nparts, uint8_t *d_blob, int *ok)
{
    const uint64_t N = ctx->N;
    hipStream_t st = ctx->stream;
    FBG_TRY(fbg_stage_begin(ctx, FBG_STAGE_SUFFIX_SORT));
    int launches = 0;
    const uint8_t *T = ctx->text.as<uint8_t>();
    KeyGeom g;
    ctx->granked = false; ctx->gpart = false;
    ctx->grs_ebits = nullptr; ctx->grs_flagged = false;
    int pre_ok = ctx->gapfree && !ctx->have_ignore && !ctx->opt.no_ranked;
    FBG_TRY(fbg_key_setup(ctx, pre_ok != 0, &g, &launches));
    pre_ok = pre_ok && g.compact;
    // MSAs with gaps / ignore characters: the same partitioning on (key, position) pairs in the code of any alphabet,
    // scanned by gapped_rank.hip
    const bool gapped = (!ctx->gapfree || ctx->have_ignore) && !ctx->reversed && ctx->opt.gapped_rank != -1 && !g.compact && g.K <= 32 &&
                        N < (1ull << 32);
    if (gapped) FBG_TRY(fbg_grs_prepare(ctx, &launches));
    if (pre_ok) {
        // similar rows tie almost everywhere: not for the slot-level scan of the partitions (every rank draws the same
        // sample and declines alike); the caller builds the whole index, whose group-level scan is made for them
        bool similar = false;
        FBG_TRY(sample_says_similar(ctx, g, &similar, &launches));
        if (similar) pre_ok = 0;
    }
    ctx->part = part; ctx->nparts = nparts; ctx->part_active = true;
    uint64_t count = 0;
    // splitters: quantiles of a sorted key sample -- the same on every rank, ties never straddle a boundary
    uint64_t S = N / 64;
    if (S > (1u << 20)) S = 1u << 20;
    if (S < 1024) S = N < 1024 ? N : 1024;
    const uint64_t stride = N / S;
    uint64_t lo = 0, hi = 0;
    if ((pre_ok || gapped) && nparts > 1) {
        FBG_TRY(fbg_reserve(ctx, ctx->dp_g, S * 8));
        FBG_TRY(fbg_reserve(ctx, ctx->dp_h, S * 8));
        uint64_t *smp = ctx->dp_g.as<uint64_t>(), *smp_sorted = ctx->dp_h.as<uint64_t>();
        hipLaunchKernelGGL(k_sample_keys, dim3(fbg_blocks(S, 256)), dim3(256), 0, st, T, N, g.d_code, g.b, g.K, g.compact ? 1 : 0, stride, S, smp);
        FBG_TRY(with_tmp(ctx, [&](void *tmp, size_t &bytes) {
            return rocprim::radix_sort_keys(tmp, bytes, smp, smp_sorted, (size_t)S, 0u, (unsigned)g.key_bits, st);
        }));
        launches += 2;
        if (part > 0)
            FBG_HIP_TRY(ctx, hipMemcpyAsync(&lo, smp_sorted + (uint64_t)part * S / nparts, 8, hipMemcpyDeviceToHost, st));
        if (part + 1 < nparts)
            FBG_HIP_TRY(ctx, hipMemcpyAsync(&hi, smp_sorted + (uint64_t)(part + 1) * S / nparts, 8, hipMemcpyDeviceToHost, st));
        FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
    }
    int msd_ok = 0;
    if (pre_ok)     // three passes over 12-byte slots, packing and filtering fused into the first (msd_sort_pairs.hip)
        FBG_TRY(fbg_msd_sort_part(ctx, g, lo, hi, part + 1 >= nparts, nparts, FBG_PART_HALO, &count, &msd_ok, &launches));
    if ((pre_ok || gapped) && !msd_ok) {
        unsigned long long *d_count = ctx->scalars.as<unsigned long long>() + 8;
        uint64_t cap = N / nparts + N / 8 + 65536;
        if (cap > N) cap = N;
        for (int attempt = 0; attempt < 2; attempt++) {
            FBG_TRY(fbg_reserve(ctx, ctx->keysA, cap * 8));
            if (!g.packed) FBG_TRY(fbg_reserve(ctx, ctx->valsA, cap * 4));
            FBG_HIP_TRY(ctx, hipMemsetAsync(d_count, 0, 8, st));
            PackArgs pa;
            pa.T = T; pa.N = N; pa.code = g.d_code; pa.b = g.b; pa.K = g.K; pa.pb = g.pb;
            pa.keys = ctx->keysA.as<uint64_t>(); pa.vals = ctx->valsA.as<uint32_t>();
            pa.lo = lo; pa.hi = hi; pa.cap = cap; pa.nohi = part + 1 >= nparts; pa.counter = d_count;
            pa.ebits = gapped ? ctx->grs_ebits : nullptr;
            launch_pack(ctx, g, true, pa);
            launches++;
            unsigned long long hc = 0;
            FBG_HIP_TRY(ctx, hipMemcpyAsync(&hc, d_count, 8, hipMemcpyDeviceToHost, st));
            FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
            count = hc;
            if (count <= cap) break;
            cap = count;                               // a skewed sample: once more with room for all of them
        }
    }
    const uint64_t slots = count + 2 * FBG_PART_HALO;
    FBG_TRY(fbg_reserve(ctx, ctx->keysB, slots * 8));
    if (!g.packed) FBG_TRY(fbg_reserve(ctx, ctx->valsB, slots * 4));
    if ((pre_ok || gapped) && !msd_ok && count > 0) FBG_TRY(sort_slots(ctx, g, count, FBG_PART_HALO));
    ctx->grs_ebits = nullptr;
    FBG_HIP_TRY(ctx, hipGetLastError());
    FBG_TRY(fbg_stage_end(ctx, FBG_STAGE_SUFFIX_SORT, launches));
    if (gapped)
        return fbg_grs_part_classify(ctx, ctx->keysB.as<uint64_t>(), ctx->valsB.as<uint32_t>(), count, g, 1, d_blob, ok);
    return fbg_rank_part_classify(ctx, ctx->keysB.as<uint64_t>(), g.packed ? nullptr : ctx->valsB.as<uint32_t>(), count, g,
                                  pre_ok, d_blob, ok);
}
