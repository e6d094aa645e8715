// rank_common.h -- slot layouts and row arithmetic shared by the scans that work in suffix-array order
// (rank_scan.hip: the slot-level scan; pure_scan.hip: the group-level scan for similar rows).
#pragma once
#include "fbg_internal.h"
#include "text_cmp.h"

#define RS_TG 4    // tie groups up to this size are settled inside the scan

// number of equal leading symbols of two different K-symbol keys (b bits per symbol)
__device__ __forceinline__ uint32_t rs_key_lcp(uint64_t a, uint64_t c, int b, int key_bits)
{
    const uint64_t d = a ^ c;
    const uint32_t bits = (uint32_t)(__clzll((long long)d) - (64 - key_bits));
    const uint32_t inv_b = (65536u + (uint32_t)b - 1) / (uint32_t)b;   // hoisted: b is uniform
    return (bits * inv_b) >> 16;
}

struct RankArgs {
    uint32_t magic32;          // floor(2^32 / row_len): umulhi(p, magic32) is p / row_len or one less
    uint64_t magic64;          // floor(2^64 / row_len): umul64hi(p, magic64) is p / row_len or one less, for every 64-bit p
    uint64_t *keys;            // sorted slots: keys, or packed words key << pb | position
    uint32_t *vals;            // positions in SA order (pairs layout; final once the tie groups are ordered)
    uint64_t pmask;            // packed / wide: (1 << pb) - 1, the position bits of the key word
    int pb;
    const uint8_t *T;
    uint64_t N, n;             // N: number of SA slots in keys[] / vals[]
    uint64_t Ntext;            // text length (position Ntext-1 is the sentinel)
    uint64_t own_lo, own_hi;   // slots this launch owns; the rest are halo copies of the neighbouring partitions
    int first_part, last_part; // partition holds the globally first / last suffix (no neighbour beyond)
    int part_mode;             // >1 partitions: slots next to a partition edge are re-examined once the halos are in
    int values_only;           // second pass for columns the threshold starved: no lists, no threshold
    uint32_t row_len;          // n + 1
    uint32_t g_min;            // extensions below this cannot be a column maximum (sampled; verified afterwards)
    int b, key_bits, K, reversed;
    uint32_t *gmax;            // per column: max over rows of g (0 = no row pointer seen)
    uint32_t *cand;            // SA slots that need the run treatment (see header)
    uint32_t *blk_count;       // k_rank_scan: candidates found by each workgroup (its private region of cand[])
    uint32_t region;           // capacity of one workgroup's region
    uint32_t *ties;            // heads of the small tie groups, same per-workgroup regions
    uint32_t *tie_count;
    uint32_t tie_region;
    uint2 *pairs;              // k_rank_scan_lean: the two text positions of every simple tied pair, same per-workgroup regions (k_tie_pairs)
    uint32_t *pair_count;
    uint32_t *pm;              // scratch parallel to cand: prefix minima of the forward walk
    uint32_t *big;             // tie groups too long for one thread: (head slot, size) pairs, counters[5] of them
    unsigned long long *counters;   // [1] fallback flag
};

struct Slot {
    uint64_t key;
    uint64_t pos;
    uint32_t col;              // MSA column of the position, n for '#' / sentinel (never a row pointer without gaps)
    uint32_t rem;              // symbols left in the row from this position on (0 for '#' / sentinel)
};

// MSA column of a row pointer with `rem` symbols left; n for '#' / sentinel (never a row pointer without gaps)
__device__ __forceinline__ uint32_t rs_col_of_rem(const RankArgs &a, uint32_t rem)
{
    return rem == 0 ? (uint32_t)a.n : (a.reversed ? rem - 1 : (uint32_t)a.n - rem);
}
// slot layouts (FBG_SLOTS_*): the key is always word >> pb (pb = 0 for plain pairs)
template <int L> struct PosT { typedef uint32_t type; };
template <> struct PosT<FBG_SLOTS_WIDE> { typedef uint64_t type; };
template <int L> __device__ __forceinline__ uint64_t rs_key(const RankArgs &a, uint64_t k) { return a.keys[k] >> a.pb; }
template <int L> __device__ __forceinline__ uint64_t rs_pos_of(const RankArgs &a, uint64_t w, uint32_t v)
{
    if (L == FBG_SLOTS_PACKED) return w & a.pmask;
    if (L == FBG_SLOTS_WIDE) return ((w & a.pmask) << 32) | v;
    return v;
}
template <int L> __device__ __forceinline__ uint64_t rs_pos(const RankArgs &a, uint64_t k)
{
    return rs_pos_of<L>(a, L == FBG_SLOTS_PAIRS ? 0ull : a.keys[k], L == FBG_SLOTS_PACKED ? 0u : a.vals[k]);
}
template <int L> __device__ __forceinline__ void rs_set_pos(const RankArgs &a, uint64_t k, uint64_t p)
{
    if (L == FBG_SLOTS_PACKED) a.keys[k] = (a.keys[k] & ~a.pmask) | p;
    else if (L == FBG_SLOTS_WIDE) { a.keys[k] = (a.keys[k] & ~a.pmask) | (p >> 32); a.vals[k] = (uint32_t)p; }
    else a.vals[k] = (uint32_t)p;
}
// symbols left in the row of text position p (0 for '#' / sentinel)
template <int L> __device__ __forceinline__ uint32_t rs_rem(const RankArgs &a, uint64_t p)
{
    if (L == FBG_SLOTS_WIDE) {
        // p / d - p * floor(2^64 / d) / 2^64 < p / 2^64 < 1: the quotient is at most one short, whatever p and d are
        uint64_t c = p - __umul64hi(p, a.magic64) * a.row_len;
        if (c >= a.row_len) c -= a.row_len;
        return p == a.Ntext - 1 ? 0u : (uint32_t)(a.n - c);
    }
    const uint32_t p32 = (uint32_t)p;
    uint32_t c = p32 - __umulhi(p32, a.magic32) * a.row_len;           // p mod (n+1), possibly one row_len too much
    if (c >= a.row_len) c -= a.row_len;
    return p32 == (uint32_t)(a.Ntext - 1) ? 0u : (uint32_t)a.n - c;
}
template <int L> __device__ __forceinline__ Slot rs_slot(const RankArgs &a, uint64_t k)
{
    Slot s;
    const uint64_t w = a.keys[k];
    s.key = w >> a.pb;
    s.pos = rs_pos_of<L>(a, w, L == FBG_SLOTS_PACKED ? 0u : a.vals[k]);
    s.rem = rs_rem<L>(a, s.pos);
    s.col = rs_col_of_rem(a, s.rem);
    return s;
}
template <int L> __device__ __forceinline__ uint32_t rs_col(const RankArgs &a, uint64_t k)
{
    return rs_col_of_rem(a, rs_rem<L>(a, rs_pos<L>(a, k)));
}

__device__ __forceinline__ void rs_update(const RankArgs &a, uint32_t col, uint32_t g)
{
    // stale reads are only ever too small (values grow monotonically): never skips a needed update
    if (a.gmax[col] < g) atomicMax(&a.gmax[col], g);
}


// one wave-aggregated append per list: a single update per wave of the workgroup's counter (LDS; it goes to
// global memory once, when the workgroup is done)
__device__ __forceinline__ void rs_append(bool want, uint32_t *counter, uint32_t *list, uint32_t region, uint32_t value)
{
    const unsigned long long mask = __ballot(want);
    if (!mask) return;
    const int lane = threadIdx.x & 63;
    uint32_t base = 0;
    const int leader = __ffsll((long long)mask) - 1;
    if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = __shfl(base, leader, 64);
    if (want) {
        const uint32_t slot = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1));
        if (slot < region) list[(size_t)blockIdx.x * region + slot] = value;
    }
}

#define RS_HALO 8

// Where rank_scan_slow finds the keys and symbols-left of the slots around the one it looks at: index i of a window.
struct RsLdsView {                             // k_rank_scan: a chunk staged in LDS
    const uint64_t *skey;
    const uint32_t *srem;
    __device__ __forceinline__ uint64_t key(int i) const { return skey[i]; }
    __device__ __forceinline__ uint32_t rem(int i) const { return srem[i]; }
};
struct RsWordView {                            // packed words (key << pb | position) in LDS or global memory, words[lo .. hi) readable
    const uint64_t *words;
    const RankArgs *a;
    int lo, hi;
    __device__ __forceinline__ uint64_t word(int i) const { return i >= lo && i < hi ? words[i] : 0ull; }
    __device__ __forceinline__ uint64_t key(int i) const { return word(i) >> a->pb; }
    __device__ __forceinline__ uint32_t rem(int i) const { return rs_rem<FBG_SLOTS_PACKED>(*a, word(i) & a->pmask); }
};

// Slots that tie with a neighbour on the whole key, and slots next to such a group, need to look around.
// i: LDS index of the slot (slot k = base + i - RS_HALO); valid LDS indices are [lo_i, hi_i).
template <class V>
__device__ __forceinline__ void rank_scan_slow(const RankArgs &a, const V &v, const int i,
                                               const int lo_i, const int hi_i, const uint64_t k, bool &want_cand, bool &want_tie)
{
    const uint64_t key = v.key(i);
    const uint32_t rem = v.rem(i);
    const bool has_prev = i > lo_i, has_next = i + 1 < hi_i;               // neighbours inside the owned range
    const uint64_t kp = v.key(i - 1), kn = v.key(i + 1);
    const bool tie = (has_prev && kp == key) || (has_next && kn == key);
    if (tie) {
        // bounds of the group, looking at most RS_TG slots either way
        int h = i, t = i;
        while (h > lo_i && i - h < RS_TG && v.key(h - 1) == key) h--;
        bool big = h > lo_i && v.key(h - 1) == key;
        while (t + 1 < hi_i && t - i < RS_TG && v.key(t + 1) == key) t++;
        big = big || (t + 1 < hi_i && v.key(t + 1) == key) || t - h + 1 > RS_TG;
        bool simple = !big;
        if (simple && a.part_mode && (k - (uint64_t)(i - h) < a.own_lo + 2 || k + (uint64_t)(t - i) + 2 >= a.own_hi))
            simple = false;                                                                    // partition edge
        if (simple) {
            // every member: K real symbols, columns all different, and different from the slots next to the group
            uint32_t cols[RS_TG];
            const int s = t - h + 1;
#pragma unroll
            for (int q = 0; q < RS_TG; q++) {
                cols[q] = 0xffffffffu - (uint32_t)q;
                if (q < s) {
                    cols[q] = v.rem(h + q);
                    if (cols[q] < (uint32_t)a.K) simple = false;
                }
            }
#pragma unroll
            for (int q = 0; q < RS_TG; q++)
#pragma unroll
                for (int r = q + 1; r < RS_TG; r++)
                    if (cols[q] == cols[r]) simple = false;
            if (simple && h > lo_i) {
                if (h - 1 > lo_i && v.key(h - 2) == v.key(h - 1)) simple = false;                // tie groups side by side
                const uint32_t oc = v.rem(h - 1);
#pragma unroll
                for (int q = 0; q < RS_TG; q++)
                    if (cols[q] == oc) simple = false;
            }
            if (simple && t + 1 < hi_i) {
                if (t + 2 < hi_i && v.key(t + 2) == v.key(t + 1)) simple = false;
                const uint32_t oc = v.rem(t + 1);
#pragma unroll
                for (int q = 0; q < RS_TG; q++)
                    if (cols[q] == oc) simple = false;
            }
        }
        if (simple) want_tie = i == h;
        else want_cand = true;
        return;
    }
    if (rem == 0) return;
    // not a tie itself, but next to a tie group whose final order is not known here: any member may end up next to
    // this slot.  Same column anywhere in it -> run treatment; LCP with it: the member with most symbols left
    const bool edge = a.part_mode && (k < a.own_lo + 2 || k + 2 >= a.own_hi);           // neighbour not known yet
    bool run = (has_prev && v.rem(i - 1) == rem) || (has_next && v.rem(i + 1) == rem);
    uint32_t mrp = has_prev ? v.rem(i - 1) : 0u, mrn = has_next ? v.rem(i + 1) : 0u;
    if (has_prev && i - 1 > lo_i && v.key(i - 2) == kp) {
        int j = i - 1, cnt = 0;
        uint32_t mr = 0;
        for (;;) {
            const uint32_t rj = v.rem(j);
            mr = max(mr, rj);
            run = run || rj == rem;
            cnt++;
            if (j == lo_i || v.key(j - 1) != kp) break;
            if (cnt == RS_TG) { run = true; break; }
            j--;
        }
        mrp = mr;
    }
    if (has_next && i + 2 < hi_i && v.key(i + 2) == kn) {
        int j = i + 1, cnt = 0;
        uint32_t mr = 0;
        for (;;) {
            const uint32_t rj = v.rem(j);
            mr = max(mr, rj);
            run = run || rj == rem;
            cnt++;
            if (j + 1 >= hi_i || v.key(j + 1) != kn) break;
            if (cnt == RS_TG) { run = true; break; }
            j++;
        }
        mrn = mr;
    }
    if (edge || run) { want_cand = true; return; }
    const uint32_t lp = has_prev ? min(min(rs_key_lcp(kp, key, a.b, a.key_bits), rem), mrp) : 0u;
    const uint32_t ln = has_next ? min(min(rs_key_lcp(key, kn, a.b, a.key_bits), rem), mrn) : 0u;
    const uint32_t g = max(lp, ln) + 1;
    if (g >= a.g_min || rem <= 64) rs_update(a, rs_col_of_rem(a, rem), g);
}


// ---- host side ----------------------------------------------------------------------------------------------
static inline int rs_layout(const KeyGeom &g) { return g.packed ? FBG_SLOTS_PACKED : g.wide ? FBG_SLOTS_WIDE : FBG_SLOTS_PAIRS; }

static inline void rs_args_init(fbg_ctx *ctx, RankArgs &a, uint64_t *keys, uint32_t *vals, uint64_t slots, int layout, int pb, int b,
                         int key_bits, int K)
{
    a.keys = keys; a.vals = vals; a.T = ctx->text.as<uint8_t>();
    a.pb = layout != FBG_SLOTS_PAIRS ? pb : 0; a.pmask = layout != FBG_SLOTS_PAIRS ? (1ull << pb) - 1 : 0;
    a.N = slots; a.Ntext = ctx->N; a.n = ctx->n; a.row_len = (uint32_t)(ctx->n + 1);
    a.own_lo = 0; a.own_hi = slots; a.first_part = a.last_part = 1; a.part_mode = 0; a.values_only = 0;
    a.magic32 = (uint32_t)((1ull << 32) / (ctx->n + 1));
    a.magic64 = ~0ull / (ctx->n + 1);     // floor((2^64 - 1) / d) = floor(2^64 / d) unless d divides 2^64, where it is one less:
                                          // still within the one correction rs_rem applies (d = row_len >= 2)
    a.b = b; a.key_bits = key_bits; a.K = K; a.reversed = ctx->reversed;
    a.gmax = ctx->gmax.as<uint32_t>();
    a.cand = nullptr; a.pm = nullptr; a.blk_count = nullptr; a.region = 0;
    a.ties = nullptr; a.tie_count = nullptr; a.tie_region = 0; a.pairs = nullptr; a.pair_count = nullptr;
    a.big = ctx->big_groups.as<uint32_t>();
    a.counters = ctx->scalars.as<unsigned long long>() + 32;
    a.g_min = 0;
}

static inline void rs_remember(fbg_ctx *ctx, uint64_t *keys, uint32_t *vals, const KeyGeom &g)
{
    ctx->rk_keys = keys; ctx->sa_ptr = vals;
    ctx->rk_layout = rs_layout(g); ctx->rk_pb = g.pb;
    ctx->rk_b = g.b; ctx->rk_key_bits = g.key_bits; ctx->rk_K = g.K;
}

