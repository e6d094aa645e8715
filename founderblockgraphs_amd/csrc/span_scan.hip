// span_scan.hip -- the extension scan in suffix-array order for SIMILAR rows WITH gaps and / or ignore characters
// (what a pangenome MSA looks like): the group-level scan of pure_scan.hip on the column spans of gapped_rank.hip.
//
// compute_f (fbg.cpp:1579-1695) keeps one text pointer per row; while a row shows gaps the pointer waits on the row's
// next symbol (fbg.cpp:1687-1691), so text position p is its row's pointer for a SPAN of columns [lo(p), hi(p)]: from
// the column after the row's previous symbol to its own.  The leaves coloured at column x are the positions whose span
// holds x (a row's first symbol only with the elastic tricks off: fullrow[], fbg.cpp:1605-1608), and the walk of
// fbg.cpp:1633-1678 gives a coloured leaf r the extension
//
//        g_x(r) = 1 + max over the suffixes u NOT coloured at x of lcp(r, u)                      
//
// which needs no order among the coloured ones.  After ONE sort of the suffixes by a K-symbol key (every symbol,
// '#' and the sentinel included, has a code of its own: equal keys <=> equal first K symbols) the suffixes with equal
// keys form a GROUP, and for a group G and a column x:
//   * every member coloured at x ("G is pure at x": x lies in P(G), the intersection of the members' spans): all
//     members extend by 1 + the symbols the key shares with the nearest group, on either side, that has a member not
//     coloured at x (groups pure at x in between are coloured along, fbg.cpp:1633-1641) -- keys only, no text;
//   * else the longest match of a coloured member is with a group mate that is not coloured at x (they share K
//     symbols, nobody outside does): text comparison inside the group.
// Rows that resemble each other put the m suffixes of one aligned position into one group whose members all have the
// one-column span [h, h] -- except the rows with a gap run right before h, whose span is wider ("odd" members: 0.25 %
// of the cells when 2 % of them are gap cells in runs of 8).  So almost every group is settled by one thread from two
// neighbouring keys; the groups with an odd member get a workgroup: the odd members are compared with their group
// mates (each needs its longest match with a mate for the gap columns in front of it, where it is coloured alone).
// Prefix doubling (the record path this replaces: six rounds of global sorts to order the members of every group among
// themselves, 105 ms for 1000 x 200 000) answers a question the scan never asks.
//
// What a slot must say about itself without a table lookup is its column: the sort's 32-bit payload is therefore not the
// text position but the CELL, row * (n + 1) + column (column n: the row's '#'; the sentinel: row m), plus two flags
// computed in text order: W -- the span is wider than one column or the position is a row's first symbol -- and I --
// the K symbols from here on are not K consecutive columns of the row (a gap run, the row's end, among them), i.e.
// fi = column + g - 1 (fbg.cpp:1666) does not hold and the member looks its extent up.  The text position of a cell,
// needed only for text comparisons and lookups, comes from a table of 20 bytes per 128 cells (symbols before the window,
// bitmap of its symbols).  (m + 1) * (n + 1) < 2^30; '-' among the ignore characters is not for this scan.
#include "fbg_internal.h"
#include "text_cmp.h"
#include <rocprim/rocprim.hpp>
#include <algorithm>

#define SP_THREADS 256
#define SP_ITEMS 8
#define SP_TILE (SP_THREADS * SP_ITEMS)
#define SP_CELL 0x3fffffffu
#define SP_I 0x40000000u
#define SP_W 0x80000000u
#define SPG_ODD 1u                   // not every member has the one-column span of the group's first member
#define SPG_IRR 2u                   // some member is irregular (flag I)
#define SPG_SEP 4u                   // '#' / sentinel slots: never row pointers that matter, share nothing with a symbol's key
#define SPG_REG 8u                   // some member is regular
#define SP_MAX_ODD 1024              // odd members of a group (LDS): a deletion shared by a third of 1000 rows makes hundreds
#define SP_MAX_PURE 4096             // columns of a pure interval worked off one after the other
#define SP_NONE 0xffffffffu

struct CWin { uint32_t rank0, b0, b1, b2, b3; };      // 128 cells of a row: symbols before them in the row, bitmap of the symbols

struct SpArgs {
    const uint64_t *keys;
    const uint32_t *vals;
    uint64_t N;
    uint32_t n, m, row_len, magic;   // row_len = n + 1; magic = floor(2^32 / row_len)
    int b, K, key_bits, disable_tricks;
    int ksh;                         // 0: the flags W / I in the values' top bits; 2: in the key words' low bits (2^30 cells and more)
    const uint8_t *T;
    const uint32_t *colT, *pos, *tot;   // colT = nullptr: rows without gaps (cell == text position)
    const CWin *cwin;
    uint32_t wpr;                    // windows per row
    const uint8_t *is_ignore;        // by byte; nullptr: no ignore characters
    uint32_t ign_lo, ign_hi;         // by symbol code
    unsigned long long *tile_cnt;    // per tile of slots: group heads (low word), irregular slots (high word) -> exclusive offsets
    uint32_t *gstart, *gcol, *gflags;
    uint64_t G;
    uint32_t *rtile, *rstart, *rid;  // runs of groups that are pure in one column
    uint64_t R;
    uint32_t *gplo, *gphi;           // odd groups: the columns at which every member is coloured (plo > phi: none)
    uint32_t *gval;                  // the other groups: g at their column
    uint32_t *odd;                   // lists of the odd groups, one per size class (up to 64 / 896 / 1024 members, more), odd_cap entries each
    uint32_t odd_cap;
    uint2 *irr;                      // (slot, group) of the irregular slots
    uint64_t n_irr;
    uint32_t *fmax;                  // per column: largest fi
    unsigned long long *counters;    // [0] odd groups (small), [1] odd groups (big), [2] comparisons ahead, [3] decline flag, [4] min last column,
                                     // [5] odd members in all, [6] entries of the chain list, [7] groups of the slow list
    const uint8_t *code;             // symbol codes of the sort's keys, by byte
    struct SpChain *chain;           // odd members whose longest match is found along the chain of groups (k_sp_chain)
    uint32_t chain_cap;
    uint32_t *slow;                  // odd groups that need every pair compared (k_sp_odd_slow)
    uint32_t slow_cap;
    uint32_t *mins32;                // per 32 columns: the fewest symbols any row has there (nullptr: rows without gaps)
};

// An odd member q on the chain list.  OWN: at the columns [xlo, xhi] \ [plo, phi] of its span the same mates are coloured
// with it throughout -- the nex members at the text positions ex[] (rows with the same gap run in front of them), nobody
// else.  EVENT: q is NOT coloured at the majority column evcol of its group, whose members therefore extend by their
// longest match with q (and the likes of it): worked out only if that could raise the column's maximum at all.
#define SP_CHAIN_EX 7
#define SP_CH_OWN 0x100u
#define SP_CH_EVENT 0x200u
struct SpChain { uint32_t p, row, xlo, xhi, plo, phi, group, nex, evcol, ex[SP_CHAIN_EX]; };

__device__ __forceinline__ uint32_t sp_key_lcp(uint64_t a, uint64_t c, int b, int key_bits)
{
    const uint64_t d = a ^ c;
    const uint32_t bits = (uint32_t)(__clzll((long long)d) - (64 - key_bits));
    return (bits * ((65536u + (uint32_t)b - 1) / (uint32_t)b)) >> 16;
}

// The flags of a slot.  (m + 1) * (n + 1) < 2^30: bits 30 and 31 of its value.  Beyond that the value is the cell alone and the
// flags are the two lowest bits of the key word, below the K symbols (ksh = 2: fbg_span_prepare).
__device__ __forceinline__ uint64_t sp_key(const SpArgs &a, uint64_t slot) { return a.keys[slot] >> a.ksh; }
__device__ __forceinline__ uint32_t sp_fl(const SpArgs &a, uint64_t slot, uint32_t v)
{
    return a.ksh ? (uint32_t)(a.keys[slot] & 3ull) << 30 : v & ~SP_CELL;
}

__device__ __forceinline__ void sp_decode(const SpArgs &a, uint32_t v, uint32_t &row, uint32_t &col)
{
    const uint32_t c = a.ksh ? v : v & SP_CELL;
    row = __umulhi(c, a.magic);
    col = c - row * a.row_len;
    if (col >= a.row_len) { col -= a.row_len; row++; }
}

// text position of cell (row, col); col = n: the row's '#', row = m: the sentinel
__device__ __forceinline__ uint32_t sp_pos(const SpArgs &a, uint32_t row, uint32_t col)
{
    if (row >= a.m) return (uint32_t)(a.N - 1);
    if (!a.cwin) return row * a.row_len + col;
    if (col >= a.n) return a.pos[row] + a.tot[row];
    const CWin w = a.cwin[(size_t)row * a.wpr + (col >> 7)];
    const uint32_t o = col & 127u;
    const uint32_t bits[4] = {w.b0, w.b1, w.b2, w.b3};
    uint32_t cnt = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t lo = 32u * q;
        if (o >= lo + 32) cnt += (uint32_t)__popc(bits[q]);
        else if (o > lo) cnt += (uint32_t)__popc(bits[q] & ((1u << (o - lo)) - 1));
    }
    return a.pos[row] + w.rank0 + cnt;
}

__device__ __forceinline__ uint32_t sp_col_of(const SpArgs &a, uint32_t q, uint32_t row)
{
    return a.colT ? a.colT[q] : q - row * a.row_len;
}

// the columns [lo, hi] at which the symbol of cell (row, col), flagged W, is a coloured pointer (lo > hi: never)
__device__ __forceinline__ void sp_wide_span(const SpArgs &a, uint32_t row, uint32_t col, uint32_t p, uint32_t &lo, uint32_t &hi)
{
    hi = col;
    if (p == a.pos[row]) {                                   // the row's first symbol: only with the tricks off (fbg.cpp:1605-1608)
        if (a.disable_tricks) lo = 0; else { lo = 1; hi = 0; }
    } else lo = a.colT ? a.colT[p - 1] + 1 : col;
}

// first of the `reach` symbols of the key that is an ignore character; SP_NONE if none
__device__ __forceinline__ uint32_t sp_key_first_ignore(const SpArgs &a, uint64_t key, uint32_t reach)
{
    if (!a.is_ignore) return SP_NONE;
    for (uint32_t k = 0; k < reach && k < (uint32_t)a.K; k++) {
        const uint32_t c = (uint32_t)(key >> (a.b * (a.K - 1 - (int)k))) & ((1u << a.b) - 1);
        if ((c < 32 ? a.ign_lo >> c : a.ign_hi >> (c - 32)) & 1u) return k;
    }
    return SP_NONE;
}

// fi of fbg.cpp:1657-1670 for the pointer p of row `row` extended by g; ign: offset from p of the row's first ignore
// character (at least as far as g symbols were looked at; SP_NONE: none)
__device__ __forceinline__ uint32_t sp_extent(const SpArgs &a, uint32_t p, uint32_t row, uint32_t g, uint32_t ign)
{
    const uint32_t p0 = a.pos[row], tt = a.tot[row];
    const unsigned long long gg = (unsigned long long)(p - p0) + g;                      // 1657
    uint32_t fi, reach = g;
    if (gg > tt) {                                                                       // 1659-1664
        fi = a.disable_tricks ? a.n : (tt ? sp_col_of(a, p0 + tt - 1, row) : 0u);
        reach = p0 + tt - p;
    } else fi = sp_col_of(a, p + g - 1, row);                                            // 1666
    if (ign < reach) fi = min(fi, sp_col_of(a, p + ign, row));                           // 1669-1670
    return fi;
}

// symbols of the row from p on that may be looked at for an ignore character when the extension is g
__device__ __forceinline__ uint32_t sp_reach(const SpArgs &a, uint32_t p, uint32_t row, uint32_t g)
{
    const uint32_t left = a.pos[row] + a.tot[row] - p;
    return g < left ? g : left;
}

__device__ __forceinline__ uint32_t sp_first_ignore(const SpArgs &a, uint32_t p, uint32_t reach)
{
    if (!a.is_ignore) return SP_NONE;
    for (uint32_t k = 0; k < reach; k++)
        if (a.is_ignore[a.T[(uint64_t)p + k]]) return k;
    return SP_NONE;
}

__device__ __forceinline__ void sp_update(const SpArgs &a, uint32_t x, uint32_t fi)
{
    if (fi > x && a.fmax[x] < fi) atomicMax(&a.fmax[x], fi);
}

// ---- text order: the window table of the cells and the payload of the sort --------------------------------------
__global__ __launch_bounds__(256) void k_sp_cwin(const uint8_t *__restrict__ msa, uint32_t n, uint32_t wpr, CWin *__restrict__ cwin, uint32_t *__restrict__ mins32)
{
    const uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, row = blockIdx.y;
    if (w >= wpr) return;
    const uint8_t *r = msa + (size_t)row * n;
    const uint32_t c0 = w * 128 + lane, c1 = c0 + 64;
    const unsigned long long m0 = __ballot(c0 < n && r[c0] != '-'), m1 = __ballot(c1 < n && r[c1] != '-');
    if (lane == 0) {
        CWin e;
        e.rank0 = (uint32_t)__popcll(m0) + (uint32_t)__popcll(m1);      // the count for now; k_sp_cwin_scan turns it into the prefix
        e.b0 = (uint32_t)m0; e.b1 = (uint32_t)(m0 >> 32); e.b2 = (uint32_t)m1; e.b3 = (uint32_t)(m1 >> 32);
        cwin[(size_t)row * wpr + w] = e;
        const uint32_t words[4] = {e.b0, e.b1, e.b2, e.b3};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t c = (uint32_t)__popc(words[q]);
            if (w * 128 + 32 * q < n && mins32[w * 4 + q] > c) atomicMin(&mins32[w * 4 + q], c);
        }
    }
}

__global__ __launch_bounds__(64) void k_sp_cwin_scan(uint32_t wpr, CWin *__restrict__ cwin)
{
    CWin *r = cwin + (size_t)blockIdx.x * wpr;
    const uint32_t lane = threadIdx.x;
    uint32_t carry = 0;
    for (uint32_t w0 = 0; w0 < wpr; w0 += 64) {
        const uint32_t w = w0 + lane;
        const uint32_t c = w < wpr ? r[w].rank0 : 0u;
        uint32_t inc = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d, 64); if ((int)lane >= d) inc += o; }
        if (w < wpr) r[w].rank0 = carry + inc - c;
        carry += __shfl(inc, 63, 64);
    }
}

// cell | flags of every text position; a wave takes 64 consecutive positions: the row of the first by binary search,
// the others count the separators in between
// A wave walks SPC_CHUNKS stretches of 64 positions: the row of its first position by binary search in the rows' first
// positions (ten dependent loads -- per stretch of 64 they were most of the kernel's time), from there on by counting the
// separators passed.
#define SPC_CHUNKS 8
__global__ __launch_bounds__(256) void k_sp_cells(const uint32_t *__restrict__ colT, const uint32_t *__restrict__ pos, uint64_t N, uint32_t n, uint32_t m,
                                                  int K, const unsigned long long *__restrict__ ebits, uint32_t *__restrict__ cellT,
                                                  uint8_t *__restrict__ flagT)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint64_t pw = wave * (64ull * SPC_CHUNKS);           // the wave's first position
    if (pw >= N) return;
    uint32_t lo = 0;
    if (colT) {
        uint32_t hi = m;
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if ((uint64_t)pos[mid] <= pw) lo = mid; else hi = mid; }
    }
#pragma unroll 2
    for (int c = 0; c < SPC_CHUNKS; c++) {
        const uint64_t p0 = pw + 64ull * c;
        if (p0 >= N) return;
        const uint64_t p = p0 + lane;
        const bool in = p < N;
        uint32_t row, col;
        if (colT) {
            col = in ? colT[p] : n;
            if (col > n) col = n;
            const unsigned long long seps = __ballot(in && col >= n);
            row = lo + (uint32_t)__popcll(seps & ((1ull << lane) - 1));
            lo += (uint32_t)__popcll(seps);
            if (p == N - 1) { row = m; col = n; }              // the sentinel (it would pass for a row's '#')
        } else {
            row = (uint32_t)(p / (n + 1));
            col = (uint32_t)(p - (uint64_t)row * (n + 1));
            if (p == N - 1) { row = m; col = n; }
        }
        if (!in) return;
        // irregular positions: bit p -> W, bits (p, p + K) -> I
        const unsigned long long *e = ebits + (p >> 6);
        const unsigned sh = (unsigned)(p & 63);
        unsigned long long bits = e[0] >> sh;
        if (sh) bits |= e[1] << (64 - sh);
        uint32_t w = (bits & 1ull) ? SP_W : 0u;
        uint32_t i = ((bits >> 1) & ((1ull << (K - 1)) - 1)) ? SP_I : 0u;
        if (col >= n) w = i = 0;                               // '#' / sentinel: no flags (the column says what they are)
        if (flagT) { cellT[p] = row * (n + 1) + col; flagT[p] = (uint8_t)((w | i) >> 30); }    // (2^30 cells and more: the flags join the key word)
        else cellT[p] = (row * (n + 1) + col) | w | i;
    }
}

// the values back to text positions (the record path reads them as the suffix array)
__global__ void k_sp_to_positions(SpArgs a, uint32_t *__restrict__ vals, uint64_t *__restrict__ keys)
{
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.N) return;
    uint32_t row, col;
    sp_decode(a, vals[k], row, col);
    vals[k] = sp_pos(a, row, col);
    if (a.ksh) keys[k] >>= a.ksh;                              // (and the key words without the flags)
}

__global__ void k_sp_check(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ vals, const uint32_t *__restrict__ cells, uint64_t N,
                           unsigned long long *__restrict__ out, int ksh)
{
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= N) return;
    atomicAdd(&out[0], (unsigned long long)vals[k]);
    atomicAdd(&out[1], (unsigned long long)cells[k]);
    atomicXor(&out[2], (unsigned long long)vals[k] * 0x9E3779B97F4A7C15ull);
    atomicXor(&out[3], (unsigned long long)cells[k] * 0x9E3779B97F4A7C15ull);
    if (k > 0 && (keys[k] >> ksh) < (keys[k - 1] >> ksh)) atomicAdd(&out[4], 1ull);
}

// ---- groups: maximal stretches of equal keys --------------------------------------------------------------------
__device__ __forceinline__ unsigned long long sp_block_excl(unsigned long long v, unsigned long long *total, unsigned long long *lds)
{
    // both words of v (two counts) at once: the sums stay below 2^32 each
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned long long inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const unsigned long long o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
    if (lane == 63) lds[w] = inc;
    __syncthreads();
    unsigned long long base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < SP_THREADS / 64; k++) { const unsigned long long s = lds[k]; if (k < w) base += s; tot += s; }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

template <bool FILL> __global__ __launch_bounds__(SP_THREADS) void k_sp_groups(SpArgs a)
{
    __shared__ unsigned long long lds[SP_THREADS / 64];
    const uint64_t N = a.N;
    const uint64_t k0 = (uint64_t)blockIdx.x * SP_TILE + (uint64_t)threadIdx.x * SP_ITEMS;
    uint64_t key[SP_ITEMS + 1];
    uint32_t v[SP_ITEMS + 1];
    {
        const bool ok = k0 > 0 && k0 - 1 < N;
        key[0] = ok ? a.keys[k0 - 1] >> a.ksh : 0ull;
        v[0] = (FILL && ok) ? a.vals[k0 - 1] : 0u;
    }
    uint32_t heads = 0, irr = 0, wide = 0;
#pragma unroll
    for (int j = 0; j < SP_ITEMS; j++) {
        const uint64_t k = k0 + j;
        const bool ok = k < N;
        const uint64_t word = ok ? a.keys[k] : 0ull;
        key[j + 1] = word >> a.ksh;
        v[j + 1] = ok ? a.vals[k] : 0u;
        const uint32_t fl = a.ksh ? (uint32_t)(word & 3ull) << 30 : v[j + 1] & ~SP_CELL;
        if (ok && (k == 0 || key[j + 1] != key[j])) heads |= 1u << j;
        if (ok && (fl & SP_I)) irr |= 1u << j;
        if (ok && (fl & SP_W)) wide |= 1u << j;
    }
    unsigned long long total;
    const unsigned long long before = sp_block_excl((unsigned long long)__popc(heads) | ((unsigned long long)__popc(irr) << 32), &total, lds);
    if (!FILL) {
        if (threadIdx.x == 0) a.tile_cnt[blockIdx.x] = total;
        return;
    }
    const unsigned long long base = a.tile_cnt[blockIdx.x] + before;
    uint32_t gid = (uint32_t)base - 1;                        // group of the slot before the thread's first (wraps for slot 0: unused)
    uint32_t iat = (uint32_t)(base >> 32);
    // The flags of a group are the OR over its members.  A group of hundreds of members runs over dozens of threads, and
    // every one of them adding its bits with an atomic of its own made a chain of same-address atomics per group: the part
    // a thread holds of the group it starts in (lead) and of the group it ends in (acc, group gid) meet inside the wave --
    // the groups' numbers rise from lane to lane -- and the last lane of a group's stretch adds them once.
    const uint32_t gid_first = gid;
    uint32_t acc = 0, lead = 0, prow, pcol;
    bool led = false;
    sp_decode(a, v[0], prow, pcol);
#pragma unroll
    for (int j = 0; j < SP_ITEMS; j++) {
        const uint64_t k = k0 + j;
        if (k >= N) break;
        uint32_t row, col;
        sp_decode(a, v[j + 1], row, col);
        const bool sep = col >= a.n;
        if ((heads >> j) & 1u) {
            if (!led) { lead = acc; led = true; }
            else if (acc) atomicOr(&a.gflags[gid], acc);          // a group that starts and ends inside the thread's slots
            acc = 0;
            gid++;
            a.gstart[gid] = (uint32_t)k;
            a.gcol[gid] = col;
        } else if (col != pcol) acc |= SPG_ODD;
        if (sep) acc |= SPG_SEP;
        else {
            if ((wide >> j) & 1u) acc |= SPG_ODD;
            acc |= ((irr >> j) & 1u) ? SPG_IRR : SPG_REG;
            if ((irr >> j) & 1u) a.irr[iat++] = make_uint2((uint32_t)k, gid);
        }
        pcol = col;
    }
    const uint32_t lane = threadIdx.x & 63;
    if (lane == 0 && lead) atomicOr(&a.gflags[gid_first], lead);  // (the group goes on from the wave before)
    uint32_t nlead = __shfl_down(lead, 1, 64);                    // the next lane's part of the group this one ends in
    if (lane == 63) nlead = 0;
    uint32_t x = acc | nlead;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t ox = __shfl_up(x, d, 64), og = __shfl_up(gid, d, 64);
        if ((int)lane >= d && og == gid) x |= ox;
    }
    const uint32_t ng = __shfl_down(gid, 1, 64);
    if ((lane == 63 || ng != gid) && x) atomicOr(&a.gflags[gid], x);
}

// ---- runs: maximal stretches of groups whose members all have the same one-column span ---------------------------
__device__ __forceinline__ bool sp_run_head(const SpArgs &a, uint64_t g)
{
    if (g == 0) return true;
    return ((a.gflags[g] | a.gflags[g - 1]) & (SPG_ODD | SPG_SEP)) || a.gcol[g] != a.gcol[g - 1];
}

template <bool FILL> __global__ __launch_bounds__(SP_THREADS) void k_sp_runs(SpArgs a)
{
    __shared__ unsigned long long lds[SP_THREADS / 64];
    const uint64_t g0 = (uint64_t)blockIdx.x * SP_TILE + (uint64_t)threadIdx.x * SP_ITEMS;
    uint32_t heads = 0;
#pragma unroll
    for (int j = 0; j < SP_ITEMS; j++)
        if (g0 + j < a.G && sp_run_head(a, g0 + j)) heads |= 1u << j;
    unsigned long long total;
    const unsigned long long before = sp_block_excl((unsigned long long)__popc(heads), &total, lds);
    if (!FILL) {
        if (threadIdx.x == 0) a.rtile[blockIdx.x] = (uint32_t)total;
        return;
    }
    uint32_t run = a.rtile[blockIdx.x] + (uint32_t)before - 1;
#pragma unroll
    for (int j = 0; j < SP_ITEMS; j++) {
        const uint64_t g = g0 + j;
        if (g >= a.G) break;
        if ((heads >> j) & 1u) { run++; a.rstart[run] = (uint32_t)g; }
        a.rid[g] = run;
    }
}

// the odd groups, listed by size class (one reservation per workgroup and class: 5 * 10^5 single additions to one address
// took 5 ms, one per wave still 3.7)
__global__ __launch_bounds__(1024) void k_sp_oddlist(SpArgs a)
{
    __shared__ uint32_t cnt[4];
    __shared__ unsigned long long base[4];
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (threadIdx.x < 4) cnt[threadIdx.x] = 0;
    __syncthreads();
    int cls = -1;
    if (g < a.G) {
        const uint32_t fl = a.gflags[g];
        if ((fl & SPG_ODD) && !(fl & SPG_SEP)) {
            const uint32_t s = a.gstart[g + 1] - a.gstart[g];
            cls = s <= 64 ? 0 : s <= 896 ? 1 : s <= 1024 ? 2 : 3;
        }
    }
    uint32_t mine = 0;
    if (cls >= 0) mine = atomicAdd(&cnt[cls], 1u);
    __syncthreads();
    if (threadIdx.x < 4 && cnt[threadIdx.x]) base[threadIdx.x] = atomicAdd(&a.counters[8 + threadIdx.x], (unsigned long long)cnt[threadIdx.x]);
    __syncthreads();
    if (cls >= 0) {
        const unsigned long long e = base[cls] + mine;
        if (e < a.odd_cap) a.odd[(size_t)cls * a.odd_cap + e] = (uint32_t)g; else a.counters[3] = 1;
    }
}

// the column of the first member of each listed group (sort key: the lists are worked off in column order)
__global__ void k_sp_list_cols(SpArgs a, const uint32_t *__restrict__ list, uint32_t count, uint32_t *__restrict__ cols)
{
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < count) cols[e] = a.gcol[list[e]];
}

// the pure interval of an odd group (one wave per group): the columns at which every member is coloured
__global__ __launch_bounds__(256) void k_sp_odd_spans(SpArgs a, const uint32_t *__restrict__ list, uint32_t count, uint32_t max_group)
{
    __shared__ unsigned long long work, members;
    const uint32_t lane = threadIdx.x & 63;
    if (threadIdx.x == 0) { work = 0; members = 0; }
    __syncthreads();
    unsigned long long my_work = 0, my_members = 0;            // (lane 0 of every wave)
    for (uint32_t e = blockIdx.x * 4 + (threadIdx.x >> 6); e < count; e += gridDim.x * 4) {
        const uint32_t g = list[e];
        const uint32_t s0 = a.gstart[g], s = a.gstart[g + 1] - s0, c0 = a.gcol[g];
        uint32_t lo_max = 0, hi_min = SP_NONE, nodd = 0;
        for (uint32_t i = lane; i < s; i += 64) {
            const uint32_t v = a.vals[s0 + i];
            uint32_t row, col, lo, hi;
            sp_decode(a, v, row, col);
            lo = hi = col;
            const bool wide = sp_fl(a, s0 + i, v) & SP_W;
            if (wide) sp_wide_span(a, row, col, sp_pos(a, row, col), lo, hi);
            if (wide || col != c0) nodd++;
            if (lo > hi) { lo_max = SP_NONE; hi_min = 0; }       // a member that is never coloured
            lo_max = max(lo_max, lo); hi_min = min(hi_min, hi);
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            lo_max = max(lo_max, (uint32_t)__shfl_xor(lo_max, d, 64));
            hi_min = min(hi_min, (uint32_t)__shfl_xor(hi_min, d, 64));
            nodd += (uint32_t)__shfl_xor(nodd, d, 64);
        }
        if (lane == 0) {
            const bool none = lo_max > hi_min;
            a.gplo[g] = none ? 1u : lo_max;
            a.gphi[g] = none ? 0u : hi_min;
            my_work += (unsigned long long)nodd * s;
            my_members += nodd;
            if (s > max_group || (!none && hi_min - lo_max >= SP_MAX_PURE)) a.counters[3] = 1;
        }
    }
    if (lane == 0 && my_work) { atomicAdd(&work, my_work); atomicAdd(&members, my_members); }
    __syncthreads();
    if (threadIdx.x == 0 && work) { atomicAdd(&a.counters[2], work); atomicAdd(&a.counters[5], members); }
}

// Nearest group beyond g (dir -1 / +1) with a member that is NOT coloured at column x: the symbols `key` shares with
// its key (0: none, or a separator group).  Groups pure at x are coloured along; whole runs of them are jumped.
__device__ uint32_t sp_outside(const SpArgs &a, uint64_t g, uint64_t key, uint32_t x, int dir)
{
    for (uint64_t h = g;;) {
        if (dir < 0) { if (h == 0) return 0; h--; } else { h++; if (h >= a.G) return 0; }
        const uint32_t fl = a.gflags[h];
        if (fl & SPG_SEP) return 0;
        if (fl & SPG_ODD) {
            if (a.gplo[h] <= x && x <= a.gphi[h]) continue;
        } else if (a.gcol[h] == x) {
            const uint32_t run = a.rid[h];
            h = dir < 0 ? a.rstart[run] : a.rstart[run + 1] - 1;
            continue;
        }
        return sp_key_lcp(key, sp_key(a, a.gstart[h]), a.b, a.key_bits);
    }
}

// the groups whose members all sit in one column with one-column spans: one thread each
__global__ void k_sp_values(SpArgs a)
{
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.G) return;
    const uint32_t fl = a.gflags[g];
    if (fl & SPG_SEP) return;
    if (fl & SPG_ODD) {
        // an odd group whose pure interval is one column: g there, for its workgroup to pick up (the walk is a chain of
        // dependent reads that one thread of a workgroup would make with the other 255 waiting)
        const uint32_t x = a.gplo[g];
        if (x == a.gphi[g]) { const uint64_t key = sp_key(a, a.gstart[g]); a.gval[g] = 1 + max(sp_outside(a, g, key, x, -1), sp_outside(a, g, key, x, +1)); }
        return;
    }
    const uint32_t x = a.gcol[g];
    const uint64_t key = sp_key(a, a.gstart[g]);
    const uint32_t run = a.rid[g];
    // (the groups of the run are coloured along: start beyond it)
    const uint32_t gv = 1 + max(sp_outside(a, a.rstart[run], key, x, -1), sp_outside(a, (uint64_t)a.rstart[run + 1] - 1, key, x, +1));   // 1656
    a.gval[g] = gv;
    if (fl & SPG_REG) {
        const uint32_t ign = sp_key_first_ignore(a, key, gv);
        sp_update(a, x, x + (ign < gv ? ign : gv - 1));                                  // 1666, 1669-1670
    }
}

// their irregular members: the extent by lookup
__global__ void k_sp_irr(SpArgs a)
{
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= a.n_irr) return;
    const uint2 sg = a.irr[e];
    if (a.gflags[sg.y] & (SPG_ODD | SPG_SEP)) return;
    const uint32_t gv = a.gval[sg.y];
    uint32_t row, col;
    sp_decode(a, a.vals[sg.x], row, col);
    const uint32_t p = sp_pos(a, row, col);
    sp_update(a, col, sp_extent(a, p, row, gv, sp_key_first_ignore(a, sp_key(a, sg.x), gv)));
}

// ---- the odd groups: one workgroup each ---------------------------------------------------------------------------
// k_sp_odd: the members' values in LDS, the odd members (flag W, or another column than the majority's) listed with
// their spans; the columns of the group's pure interval as above, per member.  Every other column x of an odd member q
// asks for 1 + the longest match of q with a member that is not coloured at x.  Where q is coloured ALONE at those
// columns (the usual case: a row's symbol behind a gap run that no other row shares), that is its longest match with any
// group mate, found without looking at the mates (k_sp_chain, below).  Else -- several odd members coloured together (a
// deletion common to many rows), members of the majority column that are not all coloured there (a stray suffix of
// another column in the group, a row's first symbol with the tricks on) -- every pair that matters is compared
// (k_sp_odd_slow): the texts beyond the K symbols of the key, 128 bytes at a time, 16 lanes per mate, four mates per wave
// and load, all four waves on the same q.  (All 5 * 10^5 odd members of the star phylogeny that way, 810 mates each:
// 130 GB of scattered text lines, 48 ms and more.)
__device__ __forceinline__ uint64_t sp_load8(const SpArgs &a, uint64_t p)
{
    return p + 8 <= a.N + 56 ? fbg_load8(a.T, p) : 0ull;       // (the text is padded by 64 bytes; no match runs beyond the sentinel)
}

// The column most members of a group sit in (any choice is correct: it only says who counts as "odd"; a poor one makes
// nearly every member odd -- three probes did, in a few dozen of 170 000 groups, and each of those took 100 ms): five probes,
// every one counted over all members.  votes: 8 words of LDS; all threads call, all get the answer.
__device__ __forceinline__ uint32_t sp_major(const SpArgs &a, const uint32_t *sv, uint32_t s, uint32_t *votes)
{
    uint32_t cand[5], row;
    const uint32_t at[5] = {0u, s / 4, s / 2, (3 * s) / 4, s - 1};
#pragma unroll
    for (int k = 0; k < 5; k++) sp_decode(a, sv[at[k]], row, cand[k]);
    __syncthreads();
    if (threadIdx.x < 5) votes[threadIdx.x] = 0;
    __syncthreads();
    uint32_t mine[5] = {0, 0, 0, 0, 0};
    for (uint32_t i = threadIdx.x; i < s; i += blockDim.x) {
        uint32_t col;
        sp_decode(a, sv[i], row, col);
#pragma unroll
        for (int k = 0; k < 5; k++) mine[k] += col == cand[k] ? 1u : 0u;
    }
#pragma unroll
    for (int k = 0; k < 5; k++) if (mine[k]) atomicAdd(&votes[k], mine[k]);
    __syncthreads();
    uint32_t best = 0;
#pragma unroll
    for (int k = 1; k < 5; k++) if (votes[k] > votes[best]) best = k;
    return cand[best];
}

#define SP_CHAIN_SCAN 16u             // members of a group looked at per step of the chain
template <int CAP> __global__ __launch_bounds__(SP_THREADS) void k_sp_odd(SpArgs a, const uint32_t *__restrict__ list, uint32_t count)
{
    __shared__ uint32_t sv[CAP];
    __shared__ uint16_t oidx[SP_MAX_ODD];
    __shared__ uint32_t olo[SP_MAX_ODD], ohi[SP_MAX_ODD], opos[SP_MAX_ODD];
    __shared__ uint32_t n_odd, s_gv, s_slow, votes[8];
    for (uint32_t e = blockIdx.x; e < count; e += gridDim.x) {
        const uint32_t g = list[e];
        const uint32_t s0 = a.gstart[g], s = a.gstart[g + 1] - s0;
        __syncthreads();
        if (s > CAP) continue;                                 // (flagged by k_sp_odd_spans)
        if (threadIdx.x == 0) { n_odd = 0; s_slow = 0; }
        for (uint32_t i = threadIdx.x; i < s; i += SP_THREADS) sv[i] = a.vals[s0 + i];
        __syncthreads();
        const uint32_t major = sp_major(a, sv, s, votes);
        for (uint32_t i = threadIdx.x; i < s; i += SP_THREADS) {
            uint32_t row, col;
            sp_decode(a, sv[i], row, col);
            if ((sp_fl(a, s0 + i, sv[i]) & SP_W) || col != major) {
                const uint32_t o = atomicAdd(&n_odd, 1u);
                if (o < SP_MAX_ODD) {
                    const uint32_t p = sp_pos(a, row, col);
                    uint32_t lo = col, hi = col;
                    if (sp_fl(a, s0 + i, sv[i]) & SP_W) sp_wide_span(a, row, col, p, lo, hi);
                    oidx[o] = (uint16_t)i; olo[o] = lo; ohi[o] = hi; opos[o] = p;
                }
            }
        }
        __syncthreads();
        const uint32_t no = n_odd;
        if (no > SP_MAX_ODD) { if (threadIdx.x == 0) a.counters[3] = 1; continue; }
        const uint32_t plo = a.gplo[g], phi = a.gphi[g];
        const uint64_t key = sp_key(a, s0);
        const bool has_narrow = no < s;
        // -- the pure interval (usually one column, the majority's): every member is coloured, keys decide
        for (uint32_t x = plo; x <= phi && plo <= phi; x++) {
            __syncthreads();
            if (threadIdx.x == 0) s_gv = 1 + max(sp_outside(a, g, key, x, -1), sp_outside(a, g, key, x, +1));
            __syncthreads();
            const uint32_t gv = s_gv;
            const uint32_t kign = sp_key_first_ignore(a, key, gv);
            for (uint32_t i = threadIdx.x; i < s; i += SP_THREADS) {
                uint32_t row, col;
                sp_decode(a, sv[i], row, col);
                const uint32_t fi = (sp_fl(a, s0 + i, sv[i]) & SP_I) ? sp_extent(a, sp_pos(a, row, col), row, gv, kign) : col + (kign < gv ? kign : gv - 1);
                sp_update(a, x, fi);
            }
        }
        // -- the other columns of the odd members, and the majority column where an odd member is not coloured there: the
        // chain list where the partners do not change from column to column; else the whole group -> the slow list
        const bool major_open = has_narrow && !(plo <= major && major <= phi);   // members of the majority column with a mate that is not coloured there
        bool slow = false;
        for (uint32_t o = threadIdx.x; o < no; o += SP_THREADS) {
            const uint32_t lo = olo[o], hi = ohi[o];
            SpChain c;
            c.nex = 0; c.evcol = 0;
            if (major_open && !(lo <= major && major <= hi)) { c.nex |= SP_CH_EVENT; c.evcol = major; }
            if (lo <= hi) {
                // its columns outside the pure interval: [lo, xa] and [xb, hi]
                const bool none = plo > phi;
                const uint32_t xa = none ? hi : (plo > 0 ? min(hi, plo - 1) : 0u), xb = none ? hi + 1 : max(lo, phi + 1);
                const bool left = none ? true : (plo > lo), right = !none && hi > phi;
                if (left || right) {
                    bool alone = !(has_narrow && ((left && lo <= major && major <= xa) || (right && xb <= major && major <= hi)));
                    // the other odd members: never coloured at those columns (a mate like any other), or at all of them
                    // (coloured along: no partner); one that is coloured at some of them only makes the partners change
                    uint32_t nex = 0;
                    for (uint32_t o2 = 0; o2 < no && alone; o2++) {
                        if (o2 == o || olo[o2] > ohi[o2]) continue;
                        const bool hitl = left && olo[o2] <= xa && lo <= ohi[o2], hitr = right && olo[o2] <= hi && xb <= ohi[o2];
                        if (!hitl && !hitr) continue;
                        const bool whole = (!left || (olo[o2] <= lo && xa <= ohi[o2])) && (!right || (olo[o2] <= xb && hi <= ohi[o2]));
                        if (whole && nex < SP_CHAIN_EX) c.ex[nex++] = opos[o2]; else alone = false;
                    }
                    if (alone && hi - lo <= 1024) c.nex |= nex | SP_CH_OWN; else slow = true;
                }
            }
            if (c.nex & (SP_CH_OWN | SP_CH_EVENT)) {
                uint32_t row, col;
                sp_decode(a, sv[oidx[o]], row, col);
                const unsigned long long at = atomicAdd(&a.counters[6], 1ull);
                if (at < a.chain_cap) {
                    c.p = opos[o]; c.row = row; c.xlo = lo; c.xhi = hi; c.plo = plo; c.phi = phi; c.group = g;
                    a.chain[at] = c;
                } else a.counters[3] = 1;
            }
        }
        if (slow) s_slow = 1;
        __syncthreads();
        if (threadIdx.x == 0 && s_slow) {
            const unsigned long long at = atomicAdd(&a.counters[7], 1ull);
            if (at < a.slow_cap) a.slow[at] = g; else a.counters[3] = 1;
        }
    }
}

// The longest match of the suffix at text position pq with any OTHER suffix that shares its first K symbols (its group
// mates; but for those at the positions ex[]), without looking at the mates one by one.  A mate that matches L + 1 symbols and more has, at offset L - K + 1,
// the same K symbols as pq has there: it sits L - K + 1 positions before a member of THAT key's group -- found by binary
// search in the sorted keys.  So from L = K: the members of the group of the K symbols that end at offset L are tried
// (shifted back, compared with pq from the start); the best of them lifts L to where the two part, and so on.  While the
// rows go along with each other that group is the rows' copies of a later position and any few of its members lift L by a
// mutation's distance; where pq itself leaves the others the group is small -- the rows that share the deviation -- and when
// none of ALL its members goes on, L is the answer.  *fail: a large group none of whose first members goes on (repeats
// inside the rows): every pair has to be compared.
__device__ uint32_t sp_chain_max(const SpArgs &a, uint32_t pq, const uint32_t *ex, uint32_t nex, bool *fail)
{
    uint32_t L = (uint32_t)a.K;
    *fail = false;
    for (int step = 0; step < 4096; step++) {
        const uint32_t shift = L - (uint32_t)a.K + 1;
        const uint64_t t = (uint64_t)pq + shift;
        if (t + 1 >= a.N) return L;                            // the sentinel: nothing matches beyond
        uint64_t key = 0;
        for (int k0 = 0; k0 < a.K; k0 += 8) {
            const uint64_t x = sp_load8(a, t + k0);
            for (int k = k0; k < a.K && k < k0 + 8; k++) key = (key << a.b) | (t + k < a.N ? (uint64_t)a.code[(x >> (8 * (k - k0))) & 255u] : 0ull);
        }
        uint64_t lo = 0, hi = a.N;                             // first slot whose key is not below
        while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (sp_key(a, mid) < key) lo = mid + 1; else hi = mid; }
        uint32_t best = L;
        bool all = true;
        for (uint64_t k = lo; k < a.N; k++) {
            if (sp_key(a, k) != key) break;
            if (k - lo >= SP_CHAIN_SCAN) { all = false; break; }
            uint32_t row, col;
            sp_decode(a, a.vals[k], row, col);
            const uint32_t pv = sp_pos(a, row, col);
            if (pv == t || pv < shift) continue;
            const uint32_t u = pv - shift;
            bool coloured = false;                             // a mate that is coloured along with pq is no partner
            for (uint32_t i = 0; i < nex; i++) coloured = coloured || ex[i] == u;
            if (coloured) continue;
            best = max(best, fbg_clamp_lcp(fbg_extend_match(a.T, pq, (uint64_t)u, 0)));
        }
        if (best > L) { L = best; continue; }
        if (!all) *fail = true;
        return L;
    }
    *fail = true;
    return L;
}

// a column that no row, extended by g symbols from its symbol at column x, gets beyond: whole 32-column windows with the
// fewest symbols any row has in them
__device__ uint32_t sp_upper(const SpArgs &a, uint32_t x, uint32_t g)
{
    if (!a.mins32) { const unsigned long long u = (unsigned long long)x + g - 1; return u < a.n ? (uint32_t)u : a.n; }
    const uint32_t nwin = (a.n + 31) / 32;
    uint32_t need = g - 1, w = (x >> 5) + 1;
    while (w < nwin && need > a.mins32[w]) { need -= a.mins32[w]; w++; }
    if (w >= nwin) return a.n;                                 // may run out of its row: fi = n with the tricks off
    return min(a.n - 1, w * 32 + 31);
}

// pass 0: the OWN columns of the entries; pass 1 (after pass 0 of ALL entries, and whatever else raises the columns'
// maxima): the EVENT entries -- a bound on what the members of the majority column could reach against q is compared
// with the column's maximum so far; only where it is higher does the group go to the slow list
__global__ void k_sp_chain(SpArgs a, uint32_t count, int pass)
{
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count) return;
    const SpChain c = a.chain[e];
    bool fail = false;
    if (pass == 0) {
        if (!(c.nex & SP_CH_OWN)) return;
        const uint32_t g = sp_chain_max(a, c.p, c.ex, c.nex & 0xffu, &fail) + 1;         // 1656
        if (!fail) {
            const uint32_t fi = sp_extent(a, c.p, c.row, g, sp_first_ignore(a, c.p, sp_reach(a, c.p, c.row, g)));
            for (uint64_t x = c.xlo; x <= c.xhi; x++)
                if (!(c.plo <= x && x <= c.phi)) sp_update(a, (uint32_t)x, fi);
        }
    } else {
        if (!(c.nex & SP_CH_EVENT)) return;
        const uint32_t g = sp_chain_max(a, c.p, nullptr, 0, &fail) + 1;                 // no member of the majority column extends further
        if (!fail && sp_upper(a, c.evcol, g) > a.fmax[c.evcol]) fail = true;
    }
    if (fail) {
        const unsigned long long at = atomicAdd(&a.counters[7], 1ull);
        if (at < a.slow_cap) a.slow[at] = c.group; else a.counters[3] = 1;
    }
}

#define SP_WIN 128u

// (split: a group's odd members are dealt to that many workgroups -- two columns of a long MSA whose K symbols agree make a
// group of twice the rows with half of them odd, 125 ms for one workgroup; what the members of the majority column get from
// the odd members is a maximum, which the column maxima take from the parts as well)
template <int CAP> __global__ __launch_bounds__(SP_THREADS) void k_sp_odd_slow(SpArgs a, const uint32_t *__restrict__ list, uint32_t count, uint32_t min_size,
                                                                                uint32_t split)
{
    __shared__ uint32_t sv[CAP], sp[CAP], nbest[CAP];
    __shared__ uint16_t omap[CAP];                             // member -> its place in the odd list (0xffff: a member of the majority)
    __shared__ uint16_t oidx[SP_MAX_ODD];
    __shared__ uint32_t olo[SP_MAX_ODD], ohi[SP_MAX_ODD], Lq[SP_MAX_ODD];
    __shared__ uint32_t n_odd, n_col, s_mn, s_any, s_gmax, s_ign, votes[8];
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t sub = lane >> 4, sl = lane & 15;            // mate of the wave's four, place in its 128-byte window
    for (uint64_t item = blockIdx.x; item < (uint64_t)count * split; item += gridDim.x) {
        const uint32_t e = (uint32_t)(item / split), part = (uint32_t)(item % split);
        const uint32_t g = list[e];
        if (e > 0 && list[e - 1] == g) continue;               // (the big groups' list is sorted: a group that several chains gave up on is in it once per chain)
        const uint32_t s0 = a.gstart[g], s = a.gstart[g + 1] - s0;
        __syncthreads();
        if (s > CAP || s <= min_size) continue;                // (flagged by k_sp_odd_spans / another instance's)
        if (threadIdx.x == 0) { n_odd = 0; n_col = 0; }
        for (uint32_t i = threadIdx.x; i < s; i += SP_THREADS) { sv[i] = a.vals[s0 + i]; nbest[i] = 0; omap[i] = 0xffffu; }
        __syncthreads();
        const uint32_t major = sp_major(a, sv, s, votes);
        for (uint32_t i = threadIdx.x; i < s; i += SP_THREADS) {
            uint32_t row, col;
            sp_decode(a, sv[i], row, col);
            const uint32_t p = sp_pos(a, row, col);
            sp[i] = p;
            uint32_t lo = col, hi = col;                        // (odd: as in k_sp_odd_pairs)
            if (sp_fl(a, s0 + i, sv[i]) & SP_W) sp_wide_span(a, row, col, p, lo, hi);
            if (col != major || lo != col || hi != col) {
                const uint32_t o = atomicAdd(&n_odd, 1u);
                if (lo <= hi) atomicAdd(&n_col, 1u);
                if (o < SP_MAX_ODD) {
                    oidx[o] = (uint16_t)i;
                    omap[i] = (uint16_t)o;
                    olo[o] = lo; ohi[o] = hi;
                }
            }
        }
        __syncthreads();
        const uint32_t no = n_odd;
        if (no == s && n_col == 0) continue;                   // nobody is ever coloured (the first symbols of rows, tricks on): nothing to say
        if (no > SP_MAX_ODD) { if (threadIdx.x == 0) a.counters[3] = 1; continue; }
        const uint32_t plo = a.gplo[g], phi = a.gphi[g];
        const bool has_narrow = no < s;
        // -- the other columns of the odd members, one odd member after the other
        for (uint32_t o = 0; o < no; o++) {
            const uint32_t q = oidx[o], pq = sp[q], lo = olo[o], hi = ohi[o];
            if (q % split != part) continue;                   // (by the member's place in the group: the odd list's order differs from workgroup to workgroup)
            const bool at_major = lo <= major && major <= hi;  // coloured at the majority column: no partner for its members there
            const bool own = lo <= hi && (plo > phi || lo < plo || hi > phi);   // has columns outside the pure interval
            if (!own && (at_major || !has_narrow)) continue;   // (uniform over the workgroup)
            __syncthreads();
            if (threadIdx.x == 0) { s_mn = 0; s_any = 0; s_gmax = 0; s_ign = SP_NONE; }
            __syncthreads();
            // four rounds of loads in flight per wave (a round: 4 mates x 128 bytes), window after window until every one of
            // the 16 mates has shown its first difference (a lane walking on alone, 8 bytes a step, made the other lanes of its
            // wave wait through two dependent loads a step: 7 % of the mates match beyond the first window)
            for (uint32_t u0 = wv * 4; u0 < s; u0 += 4 * (SP_THREADS / 16)) {
                uint32_t Lr[4] = {0, 0, 0, 0};
                bool lives[4], pend[4];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const uint32_t u = u0 + r * (SP_THREADS / 16) + sub;
                    lives[r] = u < s && u != q && (own || omap[u] == 0xffffu);
                    pend[r] = lives[r];
                }
                for (uint32_t off = 0;; off += SP_WIN) {
                    const uint64_t qtext = sp_load8(a, (uint64_t)pq + a.K + off + 8 * sl);
                    uint64_t xs[4];
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const uint32_t u = u0 + r * (SP_THREADS / 16) + sub;
                        xs[r] = pend[r] ? sp_load8(a, (uint64_t)sp[u] + a.K + off + 8 * sl) : qtext;
                    }
                    bool more = false;
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const uint64_t diff = xs[r] ^ qtext;
                        const uint32_t seg = (uint32_t)(__ballot(diff != 0) >> (16 * sub)) & 0xffffu;
                        const uint32_t f = seg ? (uint32_t)__ffs(seg) - 1 : 0u;
                        const uint64_t dfirst = __shfl(diff, (int)(16 * sub + f), 64);
                        if (pend[r] && seg) { Lr[r] = off + 8 * f + ((uint32_t)__ffsll((unsigned long long)dfirst) - 1) / 8; pend[r] = false; }
                        more = more || pend[r];
                    }
                    if (!__ballot(more)) break;                // (two suffixes differ at the sentinel at the latest)
                }
                if (sl != 0) continue;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    if (!lives[r]) continue;
                    const uint32_t u = u0 + r * (SP_THREADS / 16) + sub;
                    const uint32_t L = fbg_clamp_lcp(Lr[r] + (uint32_t)a.K);
                    const uint32_t ou = omap[u];
                    if (ou == 0xffffu) {
                        atomicMax(&s_mn, L);
                        s_any = 1;
                        if (!at_major) atomicMax(&nbest[u], L);
                    } else Lq[ou] = L;
                }
            }
            __syncthreads();
            if (!own) continue;
            const uint32_t mn = s_mn;
            const bool any_narrow = s_any != 0;
            // g at column x of the span: the majority members count everywhere but at their own column
            auto g_at = [&](uint32_t x) -> uint32_t {
                uint32_t best = (any_narrow && x != major) ? mn : 0u;
                for (uint32_t o2 = 0; o2 < no; o2++)
                    if (o2 != o && !(olo[o2] <= x && x <= ohi[o2])) best = max(best, Lq[o2]);
                return best + 1;                                                         // 1656 (a partner exists: x is outside the pure interval)
            };
            uint32_t gm = 0;
            for (uint64_t x = (uint64_t)lo + threadIdx.x; x <= hi; x += SP_THREADS)
                if (!(plo <= x && x <= phi)) gm = max(gm, g_at((uint32_t)x));
            if (gm) atomicMax(&s_gmax, gm);
            __syncthreads();
            uint32_t qrow, qcol;
            sp_decode(a, sv[q], qrow, qcol);
            // the row's first ignore character among the symbols any of those extensions reaches (one wave looks)
            if (a.is_ignore && wv == 0) {
                const uint32_t reach = sp_reach(a, pq, qrow, s_gmax);
                uint32_t ign = SP_NONE;
                for (uint32_t k0 = 0; k0 < reach && ign == SP_NONE; k0 += 64) {
                    const uint32_t k = k0 + lane;
                    const unsigned long long hit = __ballot(k < reach && a.is_ignore[a.T[(uint64_t)pq + k]]);
                    if (hit) ign = k0 + (uint32_t)__ffsll((unsigned long long)hit) - 1;
                }
                if (lane == 0) s_ign = ign;
            }
            __syncthreads();
            const uint32_t ign = s_ign;
            for (uint64_t x = (uint64_t)lo + threadIdx.x; x <= hi; x += SP_THREADS)
                if (!(plo <= x && x <= phi)) sp_update(a, (uint32_t)x, sp_extent(a, pq, qrow, g_at((uint32_t)x), ign));
        }
        __syncthreads();
        // -- the members of the majority column where some odd member is not coloured there
        if (has_narrow && !(plo <= major && major <= phi)) {
            for (uint32_t i = threadIdx.x; i < s; i += SP_THREADS) {
                uint32_t row, col;
                sp_decode(a, sv[i], row, col);
                if (omap[i] != 0xffffu || nbest[i] == 0) continue;
                const uint32_t gv = nbest[i] + 1;
                sp_update(a, major, sp_extent(a, sp[i], row, gv, sp_first_ignore(a, sp[i], sp_reach(a, sp[i], row, gv))));
            }
        }
    }
}

// ---- the odd groups of up to 1024 members: everything in one workgroup, a lane per member ----------------------------
// What made the kernel above slow is not the bytes but the waiting: every odd member walks its mates again, two or three
// dependent rounds of scattered loads per 64 mates, one odd member after the other.  Here every member gets a lane: the
// 128 bytes behind its key are loaded into the lane's registers, once, and the odd members -- up to SPP_MAXO of them; a
// group with more goes to the slow list -- are compared with them, their windows read from LDS (A).  The pairs that match
// beyond those 128 bytes (7 % when the rows differ in 1 % of their symbols) are listed and walked together, window after
// window (B).  Then every (odd member, column) of the spans gets a thread (C).
#define SPP_MAXO 32
#define SPP_TAILS 448
// SPP_ROWS * 32: the members an instance has room for in LDS (values, positions, best matches): 64 (the small groups, more
// than half of the odd ones: one wave per group, k_sp_odd_pairs_small), 896 or 1024; an instance takes the groups of more
// than MINS members that it has room for.  What the kernel waits for is memory -- some twenty dependent steps per group,
// 75 us -- so the groups in flight count: 6 workgroups per CU at 80 registers and 25-27 KB of LDS (the window in two halves
// of 64 bytes, 448 listed pairs).  (The round's first
// version kept ALL members' windows in registers, 16 bytes of a member in each of eight lanes: 168 registers, 3 workgroups
// per CU, and a third of its time in the instructions that find a first difference across lanes: 17 ms against 10.6.)
template <int SPP_ROWS, int MINS> __device__ __forceinline__
void sp_odd_pairs_body(const SpArgs a, const uint32_t *__restrict__ list, uint32_t count, unsigned long long *__restrict__ ticket)
{
    constexpr int CAP = SPP_ROWS * 32;
    // the small groups (up to 64 members): a wave per group -- four times the groups in flight for the same registers
    constexpr uint32_t NT = SPP_ROWS <= 2 ? 64u : (uint32_t)SP_THREADS;
    constexpr uint32_t NTAILS = SPP_ROWS <= 2 ? 256u : (uint32_t)SPP_TAILS;
    constexpr int MAXO = SPP_ROWS <= 2 ? 16 : SPP_MAXO;           // odd members a group may have here (more: k_sp_odd_slow); 16 of up to 64 members: 7 KB of LDS
    __shared__ uint32_t sv[CAP], sp[CAP], nbest[CAP];
    __shared__ uint16_t omap[CAP];
    __shared__ uint16_t oidx[MAXO];
    __shared__ uint32_t olo[MAXO], ohi[MAXO], omn[MAXO], oany[MAXO], ogmax[MAXO], oign[MAXO];
    __shared__ uint32_t LQ[MAXO][MAXO];
    __shared__ uint64_t qwin[MAXO][16];
    __shared__ uint32_t tl[NTAILS], tmin[NTAILS];
    __shared__ uint32_t n_odd, n_col, ntl, s_gv, s_need, votes[8];
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t sub = lane >> 4, sl = lane & 15;       // B: 16 lanes per pair, 8 bytes each
    // The list is in column order and the groups are handed out one by one (a ticket), so that the groups in flight are
    // neighbours: a text line holds 128 columns of a row, and what a group reads its neighbours read again -- out of the
    // XCD's L2 when they run at the same time, out of memory when the workgroups each walk a stride of the whole list
    // (a ticket is good for a few groups: the small groups are done faster than same-address atomics follow each other)
    constexpr uint32_t BATCH = SPP_ROWS <= 2 ? 16u : 1u;
    __shared__ uint32_t s_e;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) s_e = (uint32_t)atomicAdd(ticket, (unsigned long long)BATCH);
        __syncthreads();
        const uint32_t e0 = s_e;
        if (e0 >= count) break;
      for (uint32_t e = e0; e < e0 + BATCH && e < count; e++) {
        const uint32_t g = list[e];
        const uint32_t s0 = a.gstart[g], s = a.gstart[g + 1] - s0;
        __syncthreads();
#ifdef SP_PHASE_TIMERS
        long long tph[7];
        tph[0] = clock64();
#define SP_T(k) tph[k] = clock64()
#else
#define SP_T(k)
#endif
        if (s > CAP || s <= MINS) continue;                    // (another instance's)
        if (threadIdx.x == 0) { n_odd = 0; n_col = 0; s_need = 0; ntl = 0; }
        for (uint32_t i = threadIdx.x; i < s; i += NT) { sv[i] = a.vals[s0 + i]; nbest[i] = 0; omap[i] = 0xffffu; }
        __syncthreads();
        const uint32_t major = sp_major(a, sv, s, votes);
        for (uint32_t i = threadIdx.x; i < s; i += NT) {
            uint32_t row, col;
            sp_decode(a, sv[i], row, col);
            const uint32_t p = sp_pos(a, row, col);
            sp[i] = p;
            // odd: another column than the majority's, or a span that is not just its column (a W member whose span
            // IS its column -- a row's first symbol at column 0 with the tricks off -- is a member like any other)
            uint32_t lo = col, hi = col;
            if (sp_fl(a, s0 + i, sv[i]) & SP_W) sp_wide_span(a, row, col, p, lo, hi);
            if (col != major || lo != col || hi != col) {
                const uint32_t o = atomicAdd(&n_odd, 1u);
                if (lo <= hi) atomicAdd(&n_col, 1u);
                if (o < MAXO) {
                    oidx[o] = (uint16_t)i;
                    omap[i] = (uint16_t)o;
                    olo[o] = lo; ohi[o] = hi;
                }
            }
        }
        __syncthreads();
        SP_T(1);
        const uint32_t no = n_odd;
        if (no == s && n_col == 0) continue;                   // nobody is ever coloured (the first symbols of more than MAXO rows, tricks on): nothing to say
        const uint32_t plo = a.gplo[g], phi = a.gphi[g];
        const uint64_t key = sp_key(a, s0);
        const bool has_narrow = no < s;
        // -- the pure interval (usually one column, the majority's): every member is coloured, keys decide
        for (uint32_t x = plo; x <= phi && plo <= phi; x++) {
            __syncthreads();
            if (threadIdx.x == 0) s_gv = plo == phi ? a.gval[g] : 1 + max(sp_outside(a, g, key, x, -1), sp_outside(a, g, key, x, +1));
            __syncthreads();
            const uint32_t gv = s_gv;
            const uint32_t kign = sp_key_first_ignore(a, key, gv);
            const uint32_t step = kign < gv ? kign : gv - 1;
            // the regular members of the majority column all reach the same column: one update for them (hundreds of atomic
            // maxima on one address cost more than everything else the group needs)
            bool reg_major = false;
            for (uint32_t i = threadIdx.x; i < s; i += NT) {
                uint32_t row, col;
                sp_decode(a, sv[i], row, col);
                if (!(sp_fl(a, s0 + i, sv[i]) & SP_I) && col == major) { reg_major = true; continue; }
                const uint32_t fi = (sp_fl(a, s0 + i, sv[i]) & SP_I) ? sp_extent(a, sp[i], row, gv, kign) : col + step;
                sp_update(a, x, fi);
            }
            if (__ballot(reg_major) && (threadIdx.x & 63) == 0) sp_update(a, x, major + step);
        }
        SP_T(2);
        if (no > MAXO) {                                   // many odd members (a deletion common to many rows): k_sp_odd_slow
            if (threadIdx.x == 0) {
                const unsigned long long at = atomicAdd(&a.counters[7], 1ull);
                if (at < a.slow_cap) a.slow[at] = g; else a.counters[3] = 1;
                if (CAP > 1024) {                              // what the slow kernel has ahead in the large groups (the host decides)
                    atomicAdd(&a.counters[32], (unsigned long long)no * s);
                    if (no > 256) atomicAdd(&a.counters[33], 1ull);
                }
            }
            continue;
        }
        auto own_of = [&](uint32_t o) { return olo[o] <= ohi[o] && (plo > phi || olo[o] < plo || ohi[o] > phi); };   // has columns outside the pure interval
        auto at_major_of = [&](uint32_t o) { return olo[o] <= major && major <= ohi[o]; };      // coloured at the majority column: no partner for its members there
        if (threadIdx.x < no) {
            const uint32_t o = threadIdx.x;
            omn[o] = 0; oany[o] = 0; ogmax[o] = 0; oign[o] = SP_NONE;
            if (own_of(o) || (!at_major_of(o) && has_narrow)) s_need = 1;
        }
        for (uint32_t i = threadIdx.x; i < no * MAXO; i += NT) LQ[i / MAXO][i % MAXO] = 0;
        __syncthreads();
        if (!s_need) continue;
        // -- the 128 bytes behind the key of the odd members -> LDS
        for (uint32_t o = wv * 4 + sub; o < no; o += NT / 16) qwin[o][sl] = sp_load8(a, (uint64_t)sp[oidx[o]] + a.K + 8 * sl);
        __syncthreads();
        SP_T(3);
        auto note = [&](uint32_t o, uint32_t u, uint32_t Lbeyond) {   // the match of odd member o with member u beyond the key
            const uint32_t L = fbg_clamp_lcp(Lbeyond + (uint32_t)a.K);
            const uint32_t ou = omap[u];
            if (ou == 0xffffu) {
                atomicMax(&omn[o], L);
                oany[o] = 1;
                if (!at_major_of(o)) atomicMax(&nbest[u], L);
            } else LQ[o][ou] = L;
        };
        // -- A: a lane takes a member: the 128 bytes behind its key into the lane's registers (17 aligned words, shifted
        // into place), then every odd member's window (LDS, the same word for all lanes) against them.  (Sixteen bytes of a
        // member in each of eight lanes, all members of the group in registers at once, was the round's first version: a
        // third of the kernel's time went into finding the first difference across lanes, ~40 instructions per row of 32
        // members for two XORs of payload, and 168 registers held the kernel to 3 waves per SIMD.)
        // The window in two halves of 64 bytes: most pairs part in the first one (the rows differ in 2 % of their columns), and
        // 8 instead of 16 words per lane are what lets a sixth workgroup onto the CU.
        for (uint32_t u = threadIdx.x; u < ((s + NT - 1) / NT) * NT; u += NT) {
            const bool ex = u < s;
            const uint64_t p = ex ? (uint64_t)sp[u] + a.K : 0ull;
            const uint64_t *w = reinterpret_cast<const uint64_t *>(a.T) + (p >> 3);
            const uint64_t words = (a.N + 64) >> 3;             // the text buffer: N + 64 bytes, zero padded
            const unsigned sh = (unsigned)(p & 7) * 8;
            const bool narrow_u = ex && omap[u] == 0xffffu;
            uint32_t my_nbest = 0;
            uint32_t open = 0;                                  // bit o: this member and odd member o agree on the first half
#pragma unroll 1
            for (int half = 0; half < 2; half++) {
                if (half == 1 && !__ballot(open != 0)) break;  // (uniform over the wave)
                uint64_t m[8];
                {
                    const bool want = ex && (half == 0 || open != 0);
                    uint64_t x[9];
#pragma unroll
                    for (int i = 0; i < 9; i++) x[i] = (want && (p >> 3) + 8 * half + i < words) ? w[8 * half + i] : 0ull;
#pragma unroll
                    for (int i = 0; i < 8; i++) m[i] = sh ? (x[i] >> sh) | (x[i + 1] << (64 - sh)) : x[i];
                }
                for (uint32_t o = 0; o < no; o++) {
                    const bool own = own_of(o);
                    const bool atm = at_major_of(o);
                    if (!own && (atm || !has_narrow)) continue;    // (uniform over the workgroup)
                    const bool live = half == 0 ? ex && (own || narrow_u) && u != oidx[o] : ((open >> o) & 1u) != 0;
                    if (half == 1 && !__ballot(live)) continue;    // (uniform over the wave)
                    // the first differing word (descending: the lowest one wins), its first differing byte once at the end
                    uint64_t dd = 0;
                    uint32_t wi = 8;
#pragma unroll
                    for (int i = 7; i >= 0; i--) {
                        const uint64_t d = m[i] ^ qwin[o][8 * half + i];
                        if (d) { dd = d; wi = (uint32_t)i; }
                    }
                    const bool found = wi < 8;
                    uint32_t mymax = 0;                            // the lane's match with a member of the majority
                    if (live && found) {
                        const uint32_t Lk = (uint32_t)a.K + 64u * half + 8 * wi + ((uint32_t)__ffsll((unsigned long long)dd) - 1) / 8;
                        if (narrow_u) { mymax = Lk; if (!atm) my_nbest = max(my_nbest, Lk); }
                        else LQ[o][omap[u]] = Lk;
                    }
                    if (live && !found) {
                        if (half == 0) open |= 1u << o;
                        else {
                            const uint32_t at = atomicAdd(&ntl, 1u);
                            if (at < NTAILS) tl[at] = o << 16 | u;
                            else note(o, u, fbg_extend_match(a.T, (uint64_t)sp[oidx[o]] + a.K + SP_WIN, (uint64_t)sp[u] + a.K + SP_WIN, 0) + SP_WIN);
                        }
                    }
#pragma unroll
                    for (int d = 32; d >= 1; d >>= 1) mymax = max(mymax, (uint32_t)__shfl_xor(mymax, d, 64));
                    if (lane == 0 && mymax) { atomicMax(&omn[o], mymax); oany[o] = 1; }
                }
            }
            if (my_nbest) atomicMax(&nbest[u], my_nbest);
        }
        __syncthreads();
        SP_T(4);
        // -- B: the pairs that match beyond the first window: window after window, ALL of them at once -- a thread takes 8
        // bytes of a pair, the pair's first differing byte is the minimum over its 16 threads (32 pairs at a time, each waiting
        // for its own round trip, took a quarter of the kernel)
        const uint32_t nt = min(ntl, (uint32_t)NTAILS);
        for (uint32_t off = SP_WIN; nt > 0; off += SP_WIN) {
            for (uint32_t t = threadIdx.x; t < nt; t += NT) tmin[t] = SP_NONE;
            if (threadIdx.x == 0) s_gv = 0;                    // (pairs still open after this window)
            __syncthreads();
            // (four pieces per thread and turn, their loads out together: one piece per turn was one round trip per turn,
            // a dozen turns per window for the ~200 pairs of a group)
            for (uint32_t idx0 = threadIdx.x; idx0 < nt * 16; idx0 += 4 * NT) {
                uint64_t diff[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t idx = idx0 + (uint32_t)j * NT;
                    diff[j] = 0;
                    if (idx < nt * 16) {
                        const uint32_t ent = tl[idx >> 4], piece = idx & 15;
                        if (ent != SP_NONE)                    // (else: settled in an earlier window)
                            diff[j] = sp_load8(a, (uint64_t)sp[ent & 0xffffu] + a.K + off + 8 * piece) ^
                                      sp_load8(a, (uint64_t)sp[oidx[ent >> 16]] + a.K + off + 8 * piece);
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t idx = idx0 + (uint32_t)j * NT;
                    if (diff[j]) atomicMin(&tmin[idx >> 4], 8 * (idx & 15) + ((uint32_t)__ffsll((unsigned long long)diff[j]) - 1) / 8);
                }
            }
            __syncthreads();
            for (uint32_t t = threadIdx.x; t < nt; t += NT) {
                const uint32_t ent = tl[t];
                if (ent == SP_NONE) continue;
                if (tmin[t] != SP_NONE) { note(ent >> 16, ent & 0xffffu, off + tmin[t]); tl[t] = SP_NONE; }
                else s_gv = 1;                                 // (two suffixes differ at the sentinel at the latest)
            }
            __syncthreads();
            const bool more = s_gv != 0;
            __syncthreads();                                   // (everyone has read the flag before the next round resets it)
            if (!more) break;
        }
        __syncthreads();
        SP_T(5);
        // -- C: the columns of the odd members' spans outside the pure interval, a thread per (odd member, column)
        // g at column x: the majority members count everywhere but at their own column
        auto g_at = [&](uint32_t o, uint32_t x) -> uint32_t {
            uint32_t best = (oany[o] && x != major) ? omn[o] : 0u;
            for (uint32_t o2 = 0; o2 < no; o2++)
                if (o2 != o && !(olo[o2] <= x && x <= ohi[o2])) best = max(best, LQ[o][o2]);
            return best + 1;                                                             // 1656 (a partner exists: x is outside the pure interval)
        };
        for (int pass = 0; pass < 2; pass++) {
            for (uint32_t idx = threadIdx.x; idx < no * 64; idx += NT) {
                const uint32_t o = idx >> 6;
                if (!own_of(o)) continue;
                uint32_t qrow, qcol;
                sp_decode(a, sv[oidx[o]], qrow, qcol);
                uint32_t gm = 0;
                for (uint64_t x = (uint64_t)olo[o] + (idx & 63); x <= ohi[o]; x += 64) {
                    if (plo <= x && x <= phi) continue;
                    const uint32_t gx = g_at(o, (uint32_t)x);
                    if (pass == 0) gm = max(gm, gx);
                    else sp_update(a, (uint32_t)x, sp_extent(a, sp[oidx[o]], qrow, gx, oign[o]));
                }
                if (pass == 0 && gm) atomicMax(&ogmax[o], gm);
            }
            __syncthreads();
            if (pass == 0 && a.is_ignore) {                    // the row's first ignore character among the symbols any of those extensions reaches
                if (threadIdx.x < no && own_of(threadIdx.x)) {
                    const uint32_t o = threadIdx.x, pq = sp[oidx[o]];
                    uint32_t qrow, qcol;
                    sp_decode(a, sv[oidx[o]], qrow, qcol);
                    oign[o] = sp_first_ignore(a, pq, sp_reach(a, pq, qrow, ogmax[o]));
                }
                __syncthreads();
            }
        }
        // -- the members of the majority column where some odd member is not coloured there
        if (has_narrow && !(plo <= major && major <= phi)) {
            for (uint32_t i = threadIdx.x; i < s; i += NT) {
                uint32_t row, col;
                sp_decode(a, sv[i], row, col);
                if (omap[i] != 0xffffu || nbest[i] == 0) continue;
                const uint32_t gv = nbest[i] + 1;
                sp_update(a, major, sp_extent(a, sp[i], row, gv, sp_first_ignore(a, sp[i], sp_reach(a, sp[i], row, gv))));
            }
        }
#ifdef SP_PHASE_TIMERS
        __syncthreads();
        SP_T(6);
        if (threadIdx.x == 0 && SPP_ROWS == 28) {
            for (int k = 0; k < 6; k++) atomicAdd(&a.counters[24 + k], (unsigned long long)(tph[k + 1] - tph[k]));
            atomicAdd(&a.counters[30], 1ull);
        }
#endif
      }
    }
}

template <int SPP_ROWS, int MINS> __global__ __launch_bounds__(SP_THREADS) __attribute__((amdgpu_waves_per_eu(6)))
void k_sp_odd_pairs(SpArgs a, const uint32_t *__restrict__ list, uint32_t count, unsigned long long *__restrict__ ticket)
{
    sp_odd_pairs_body<SPP_ROWS, MINS>(a, list, count, ticket);
}
// the groups of more than 1024 members (LDS, not registers, limits the workgroups per CU)
template <int SPP_ROWS, int MINS> __global__ __launch_bounds__(SP_THREADS)
void k_sp_odd_pairs_big(SpArgs a, const uint32_t *__restrict__ list, uint32_t count, unsigned long long *__restrict__ ticket)
{
    sp_odd_pairs_body<SPP_ROWS, MINS>(a, list, count, ticket);
}
// the groups of up to 64 members: a wave per group
__global__ __launch_bounds__(64) void k_sp_odd_pairs_small(SpArgs a, const uint32_t *__restrict__ list, uint32_t count, unsigned long long *__restrict__ ticket)
{
    sp_odd_pairs_body<2, 0>(a, list, count, ticket);
}

// With the tricks off a row that has ended keeps its '#' as the pointer: gg > tot, fi = n (fbg.cpp:1659-1664) for every
// column behind the row's last symbol -- from the smallest such column on
__global__ void k_sp_row_ends(SpArgs a)
{
    const uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= a.m) return;
    const uint32_t tt = a.tot[row];
    const unsigned long long first = tt ? (unsigned long long)sp_col_of(a, a.pos[row] + tt - 1, row) + 1 : 0ull;
    atomicMin(&a.counters[4], first);
}
__global__ void k_sp_fill_ends(SpArgs a)
{
    const uint64_t x = a.counters[4] + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x < a.n) a.fmax[x] = a.n;
}

__global__ void k_sp_count_heads(const uint32_t *__restrict__ list, uint32_t count, unsigned long long *__restrict__ out)
{
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    const bool head = e < count && (e == 0 || list[e - 1] != list[e]);
    const unsigned long long mask = __ballot(head);
    if ((threadIdx.x & 63) == 0 && mask) atomicAdd(out, (unsigned long long)__popcll(mask));
}

template <class F> static int sp_with_tmp(fbg_ctx *ctx, F &&call)
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

static void sp_args(fbg_ctx *ctx, SpArgs &a, int disable_tricks)
{
    a.keys = ctx->rk_keys; a.vals = ctx->sa_ptr; a.N = ctx->N;
    a.n = (uint32_t)ctx->n; a.m = (uint32_t)ctx->m; a.row_len = (uint32_t)(ctx->n + 1);
    a.magic = (uint32_t)((1ull << 32) / (ctx->n + 1));
    a.b = ctx->rk_b; a.K = ctx->rk_K; a.key_bits = ctx->rk_key_bits; a.disable_tricks = disable_tricks;
    a.ksh = ctx->sp_key_flags_sorted ? 2 : 0;
    a.T = ctx->text.as<uint8_t>();
    a.colT = ctx->gapfree ? nullptr : ctx->colT.as<uint32_t>();
    a.pos = ctx->pos.as<uint32_t>(); a.tot = ctx->tot.as<uint32_t>();
    a.cwin = ctx->gapfree ? nullptr : ctx->sp_cwin.as<CWin>();
    a.wpr = (uint32_t)((ctx->n + 127) / 128);
    a.is_ignore = ctx->have_ignore ? ctx->small.as<uint8_t>() : nullptr;
    a.ign_lo = ctx->grs_ign_lo; a.ign_hi = ctx->grs_ign_hi;
    a.tile_cnt = ctx->sp_tiles.as<unsigned long long>();
    a.gstart = ctx->sp_gstart.as<uint32_t>(); a.gcol = ctx->sp_gcol.as<uint32_t>(); a.gflags = ctx->sp_gflags.as<uint32_t>();
    a.G = ctx->sp_G;
    a.rtile = nullptr; a.rstart = ctx->sp_rstart.as<uint32_t>(); a.rid = ctx->sp_rid.as<uint32_t>(); a.R = ctx->sp_R;
    a.gplo = ctx->sp_gplo.as<uint32_t>(); a.gphi = ctx->sp_gphi.as<uint32_t>(); a.gval = ctx->sp_gval.as<uint32_t>();
    a.odd = ctx->sp_odd.as<uint32_t>(); a.odd_cap = ctx->sp_odd_cap;
    a.irr = ctx->sp_irr.as<uint2>(); a.n_irr = ctx->sp_n_irr;
    a.fmax = ctx->gmax.as<uint32_t>();
    a.counters = ctx->scalars.as<unsigned long long>() + 208;     // (gapped_rank.hip: 128 .. 202)
    a.code = ctx->small.as<uint8_t>() + 2048;                     // (fbg_key_setup's table)
    a.chain = ctx->sp_chain.as<SpChain>(); a.chain_cap = (uint32_t)(ctx->sp_chain.cap / sizeof(SpChain));
    a.slow = ctx->sp_slow.as<uint32_t>(); a.slow_cap = (uint32_t)(ctx->sp_slow.cap / 8);    // (the other half: the big groups' list, sorted)
    a.mins32 = ctx->gapfree ? nullptr : ctx->sp_mins.as<uint32_t>();
}

// May this MSA go through the group-level scan on spans at all?  (the caller has decided that its rows are similar)
bool fbg_span_eligible(fbg_ctx *ctx, const KeyGeom &g)
{
    if (ctx->opt.span_scan == -1 || ctx->grs_skip || ctx->reversed) return false;   // (span_scan = 2: as 0, plus a check of the sort)
    if (ctx->gapfree && !ctx->have_ignore) return false;
    if (g.compact || g.packed || g.wide || g.K > 32 || g.K < 2) return false;
    // the cell is the sort's 32-bit value; from 2^30 cells on its two flags move into the key word (fbg_span_key_flags)
    if ((ctx->m + 1) * (ctx->n + 1) >= (1ull << 32) || ctx->m >= 65535) return false;
    if (ctx->have_ignore && ctx->ignore_tab[(unsigned char)'-']) return false;          // gap cells that clamp: the record path's per-cell table
    return true;
}

// 2^30 cells and more: the flags W / I do not fit beside the cell.  They become the two lowest bits of the key word, below
// the symbols (a key of more than 62 bits gives up its last symbol for them: the caller's geometry changes).  The sort
// orders by the symbols alone; the scan shifts them away (SpArgs::ksh).
bool fbg_span_key_flags(fbg_ctx *ctx, KeyGeom &g)
{
    const bool want = (ctx->m + 1) * (ctx->n + 1) >= (1ull << 30) || ctx->opt.span_key_flags;
    if (want && g.key_bits > 62) { g.K -= 1; g.key_bits -= g.b; }
    return want;
}

// Before the sort (fbg_grs_prepare has made the bitmap of irregular positions): the payload of every text position ->
// ctx->sp_cells (and ctx->sp_flagT), the window table of the cells.
int fbg_span_prepare(fbg_ctx *ctx, const KeyGeom &g, int *launches)
{
    const uint64_t N = ctx->N, n = ctx->n, m = ctx->m;
    hipStream_t st = ctx->stream;
    FBG_TRY(fbg_reserve(ctx, ctx->sp_cells, N * 4));
    if (ctx->sp_key_flags) FBG_TRY(fbg_reserve(ctx, ctx->sp_flagT, N));
    if (!ctx->gapfree) {
        const uint32_t wpr = (uint32_t)((n + 127) / 128);
        FBG_TRY(fbg_reserve(ctx, ctx->sp_cwin, (size_t)m * wpr * sizeof(CWin)));
        FBG_TRY(fbg_reserve(ctx, ctx->sp_mins, (size_t)wpr * 4 * 4));
        FBG_HIP_TRY(ctx, hipMemsetAsync(ctx->sp_mins.p, 0x7f, (size_t)wpr * 4 * 4, st));
        hipLaunchKernelGGL(k_sp_cwin, dim3((wpr + 3) / 4, (unsigned)m), dim3(256), 0, st, ctx->d_msa, (uint32_t)n, wpr, ctx->sp_cwin.as<CWin>(), ctx->sp_mins.as<uint32_t>());
        hipLaunchKernelGGL(k_sp_cwin_scan, dim3((unsigned)m), dim3(64), 0, st, wpr, ctx->sp_cwin.as<CWin>());
        *launches += 2;
    }
    hipLaunchKernelGGL(k_sp_cells, dim3(fbg_blocks(N, 256 * SPC_CHUNKS)), dim3(256), 0, st, ctx->gapfree ? (const uint32_t *)nullptr : ctx->colT.as<uint32_t>(),
                       ctx->pos.as<uint32_t>(), N, (uint32_t)n, (uint32_t)m, g.K, ctx->gbits.as<unsigned long long>(), ctx->sp_cells.as<uint32_t>(),
                       ctx->sp_key_flags ? ctx->sp_flagT.as<uint8_t>() : (uint8_t *)nullptr);
    *launches += 1;
    FBG_HIP_TRY(ctx, hipGetLastError());
    return FBG_OK;
}

// the tricks-dependent part: pure intervals of the odd groups, values, odd groups; *ok = 0: a capacity did not hold
static int sp_scan(fbg_ctx *ctx, int disable_tricks, int *ok, int *launches)
{
    *ok = 0;
    ctx->sp_decline = 0;
    hipStream_t st = ctx->stream;
    const uint64_t n = ctx->n, G = ctx->sp_G;
    SpArgs a;
    sp_args(ctx, a, disable_tricks);
    FBG_HIP_TRY(ctx, hipMemsetAsync(ctx->gmax.p, 0, (n + 1) * 4, st));
    FBG_HIP_TRY(ctx, hipMemsetAsync(a.counters + 2, 0, 6 * sizeof(unsigned long long), st));
    FBG_HIP_TRY(ctx, hipMemsetAsync(a.counters + 32, 0, 2 * sizeof(unsigned long long), st));
    const uint32_t *lists[4];
    uint32_t cnts[4];
    for (int c = 0; c < 4; c++) { lists[c] = a.odd + (size_t)c * a.odd_cap; cnts[c] = ctx->sp_n_odd[c]; }
    const uint32_t n_small = cnts[0] + cnts[1] + cnts[2], n_big = cnts[3];
    for (int c = 0; c < 4; c++)
        if (cnts[c]) hipLaunchKernelGGL(k_sp_odd_spans, dim3(fbg_blocks(cnts[c], 4, 4096)), dim3(256), 0, st, a, lists[c], cnts[c], c == 3 ? 8192u : 1024u);
    unsigned long long h[6] = {0, 0, 0, 0, 0, 0};
    FBG_HIP_TRY(ctx, hipMemcpyAsync(h, a.counters + 2, sizeof(h), hipMemcpyDeviceToHost, st));
    FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
    *launches += 2;
    ctx->sp_work = h[0];
    // text comparisons ahead (odd members x group size): beyond ten per suffix of the text the record path is cheaper.  Measured on
    // the star phylogeny 1000 x 200 000 with deletions that 30 % / 90 % of the rows share at 1000 and at 4000 columns
    // (scripts/gpu_stargaps.py, FBG_STAR_SHARED; profiles/r04_shared_deletions.txt): 4.2 / 5.9 / 7.1 comparisons per suffix took
    // 87 / 131 / 118 ms here against 150 / 149 / 140 on the record path, 12.7 took 326 against 128 (the bound was 32 until round 4)
    if (h[1] != 0) { ctx->sp_decline = 1; return FBG_OK; }
    if (h[0] > std::max<unsigned long long>(10 * ctx->N, 1ull << 26)) { ctx->sp_decline = 2; return FBG_OK; }
    // the larger groups' odd members: a list for those that are coloured alone, a list of the groups that need every pair compared
    const uint64_t members = h[3];
    FBG_TRY(fbg_reserve(ctx, ctx->sp_chain, (members + 1) * sizeof(SpChain)));
    FBG_TRY(fbg_reserve(ctx, ctx->sp_slow, ((uint64_t)n_small + n_big + members + 1) * 4 * 2));     // (second half: the big groups' list, sorted)
    sp_args(ctx, a, disable_tricks);
    hipLaunchKernelGGL(k_sp_values, dim3(fbg_blocks(G, 256)), dim3(256), 0, st, a);
    if (a.n_irr) hipLaunchKernelGGL(k_sp_irr, dim3(fbg_blocks(a.n_irr, 256)), dim3(256), 0, st, a);
    // the odd groups of up to 1024 members: one kernel, everything; the larger ones (more than 1024 rows): pure interval and
    // lists (k_sp_odd), the chains (k_sp_chain), every pair where that does not do (k_sp_odd_slow)
    // (workgroups that stay and take group after group: a workgroup per group spent more time being launched than working)
    FBG_HIP_TRY(ctx, hipMemsetAsync(a.counters + 12, 0, 4 * sizeof(unsigned long long), st));        // the tickets
#ifdef SP_PHASE_TIMERS
    FBG_HIP_TRY(ctx, hipMemsetAsync(a.counters + 24, 0, 8 * sizeof(unsigned long long), st));
#endif
    const bool big_chains = ctx->opt.span_scan == 3;           // (debug: the groups of more than 1024 members by chains along the later keys' groups)
    if (cnts[0]) hipLaunchKernelGGL(k_sp_odd_pairs_small, dim3(std::min<uint32_t>(cnts[0], 16384u)), dim3(64), 0, st, a, lists[0], cnts[0], a.counters + 12);
    if (cnts[1]) hipLaunchKernelGGL((k_sp_odd_pairs<28, 64>), dim3(std::min<uint32_t>(cnts[1], 3072u)), dim3(SP_THREADS), 0, st, a, lists[1], cnts[1], a.counters + 13);
    if (cnts[2]) hipLaunchKernelGGL((k_sp_odd_pairs<32, 896>), dim3(std::min<uint32_t>(cnts[2], 3072u)), dim3(SP_THREADS), 0, st, a, lists[2], cnts[2], a.counters + 14);
    if (n_big && !big_chains) {
        // more than 1024 rows: the same kernel with room for 2048 / 4096 / 8192 members (3 / 2 / 1 workgroups per CU); every
        // instance walks the list of the large groups and takes those of its size
        FBG_HIP_TRY(ctx, hipMemsetAsync(a.counters, 0, 2 * sizeof(unsigned long long), st));
        FBG_HIP_TRY(ctx, hipMemsetAsync(a.counters + 15, 0, sizeof(unsigned long long), st));
        hipLaunchKernelGGL((k_sp_odd_pairs_big<64, 1024>), dim3(std::min<uint32_t>(n_big, 1024u)), dim3(SP_THREADS), 0, st, a, lists[3], n_big, a.counters + 15);
        hipLaunchKernelGGL((k_sp_odd_pairs_big<128, 2048>), dim3(std::min<uint32_t>(n_big, 512u)), dim3(SP_THREADS), 0, st, a, lists[3], n_big, a.counters + 0);
        hipLaunchKernelGGL((k_sp_odd_pairs_big<256, 4096>), dim3(std::min<uint32_t>(n_big, 256u)), dim3(SP_THREADS), 0, st, a, lists[3], n_big, a.counters + 1);
        *launches += 3;
    }
    *launches += 3;
    ctx->sp_chain_n = 0; ctx->sp_slow_n = 0;
    if (n_small || (n_big && !big_chains)) {
        // the groups with more odd members than that kernel takes: every pair the slow way
        FBG_HIP_TRY(ctx, hipMemcpyAsync(h, a.counters + 2, sizeof(h), hipMemcpyDeviceToHost, st));
        FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
        if (h[1] != 0) { ctx->sp_decline = 3; return FBG_OK; }
        const uint32_t n_slow = (uint32_t)h[5];
        ctx->sp_slow_n = n_slow;
        // (a large group on the slow list has more odd members than the kernel above takes -- a deletion in dozens of a few
        // thousand rows -- and every one of them is compared with every mate: fine for dozens, but where hundreds of a
        // group's members are odd (a thousand identical rows that start late: every pair runs on for the rows' length) a
        // group is tens of milliseconds: those inputs, and more than 64 comparisons per suffix, go to the record path;
        // span_scan = 1 insists)
        if (n_big && !big_chains && ctx->opt.span_scan != 1 && n_slow) {
            unsigned long long ahead[2] = {0, 0};
            FBG_HIP_TRY(ctx, hipMemcpyAsync(ahead, a.counters + 32, sizeof(ahead), hipMemcpyDeviceToHost, st));
            FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
            // (ahead[1]: the large groups with more than 256 odd members.  A long MSA has some by chance -- two columns whose K
            // symbols agree make one group of twice the rows, half of them "odd"; their number grows with the square of the
            // columns -- and they run side by side, a workgroup each: up to one per 2^22 suffixes is accepted)
            if (ahead[1] > (ctx->N >> 22) || ahead[0] > 64 * ctx->N) { ctx->sp_decline = 4; return FBG_OK; }
        }
        if (n_slow) {
            hipLaunchKernelGGL((k_sp_odd_slow<1024>), dim3(std::min<uint32_t>(n_slow, 1u << 20)), dim3(SP_THREADS), 0, st, a, (const uint32_t *)a.slow, n_slow, 0u, 1u);
            if (n_big && !big_chains) {
                const uint32_t split = ctx->opt.span_slow_split > 0 ? (uint32_t)ctx->opt.span_slow_split : 32u;
                hipLaunchKernelGGL((k_sp_odd_slow<8192>), dim3((unsigned)std::min<uint64_t>((uint64_t)n_slow * split, 1u << 16)), dim3(SP_THREADS), 0, st, a, (const uint32_t *)a.slow,
                                   n_slow, 1024u, split);
            }
            *launches += 2;
        }
        FBG_HIP_TRY(ctx, hipMemsetAsync(a.counters + 7, 0, 8, st));
    }
    if (n_big && big_chains) {
        hipLaunchKernelGGL((k_sp_odd<8192>), dim3(std::min<uint32_t>(n_big, 1u << 16)), dim3(SP_THREADS), 0, st, a, lists[3], n_big);
        FBG_HIP_TRY(ctx, hipMemcpyAsync(h, a.counters + 2, sizeof(h), hipMemcpyDeviceToHost, st));
        FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
        if (h[1] != 0) { ctx->sp_decline = 5; return FBG_OK; }
        const uint32_t n_chain = (uint32_t)h[4];
        if (n_chain) {
            hipLaunchKernelGGL(k_sp_chain, dim3(fbg_blocks(n_chain, 64)), dim3(64), 0, st, a, n_chain, 0);
            hipLaunchKernelGGL(k_sp_chain, dim3(fbg_blocks(n_chain, 64)), dim3(64), 0, st, a, n_chain, 1);
        }
        FBG_HIP_TRY(ctx, hipMemcpyAsync(h, a.counters + 2, sizeof(h), hipMemcpyDeviceToHost, st));
        FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
        if (h[1] != 0) { ctx->sp_decline = 6; return FBG_OK; }
        const uint32_t n_slow = (uint32_t)h[5];
        ctx->sp_chain_n = n_chain; ctx->sp_slow_n = n_slow;
        if (n_slow) {
            // A group is in the list once per chain that gave up on it: sorted, the kernel takes every group once.  And it
            // compares every odd member of a group with every mate, window after window -- a millisecond and more per group
            // of a thousand rows that go along for thousands of symbols: beyond a few such groups per 10^6 suffixes the
            // record path is the cheaper one (the option span_scan = 1 insists)
            uint32_t *sorted = a.slow + a.slow_cap;
            FBG_TRY(sp_with_tmp(ctx, [&](void *tmp, size_t &bytes) {
                return rocprim::radix_sort_keys(tmp, bytes, (const uint32_t *)a.slow, sorted, (size_t)n_slow, 0u, 32u, st);
            }));
            FBG_HIP_TRY(ctx, hipMemsetAsync(a.counters + 15, 0, 8, st));
            hipLaunchKernelGGL(k_sp_count_heads, dim3(fbg_blocks(n_slow, 256)), dim3(256), 0, st, (const uint32_t *)sorted, n_slow, a.counters + 15);
            unsigned long long uniq = 0;
            FBG_HIP_TRY(ctx, hipMemcpyAsync(&uniq, a.counters + 15, 8, hipMemcpyDeviceToHost, st));
            FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
            ctx->sp_slow_n = (uint32_t)uniq;
            if (ctx->opt.span_scan != 1 && ctx->opt.span_scan != 3 && uniq > ctx->N / (1ull << 20) + 4) { ctx->sp_decline = 7; return FBG_OK; }
            hipLaunchKernelGGL((k_sp_odd_slow<8192>), dim3(std::min<uint32_t>(n_slow, 1u << 16)), dim3(SP_THREADS), 0, st, a, (const uint32_t *)sorted, n_slow, 0u, 1u);
            *launches += 3;
        }
        *launches += 4;
    }
    const unsigned long long none = n;                          // (function scope: the copy below may read it until the stream is synchronised further down)
    if (disable_tricks) {
        FBG_HIP_TRY(ctx, hipMemcpyAsync(a.counters + 4, &none, 8, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_sp_row_ends, dim3(fbg_blocks(ctx->m, 256)), dim3(256), 0, st, a);
        hipLaunchKernelGGL(k_sp_fill_ends, dim3(fbg_blocks(n, 256)), dim3(256), 0, st, a);
        *launches += 2;
    }
    unsigned long long flag = 0;
    FBG_HIP_TRY(ctx, hipMemcpyAsync(&flag, a.counters + 3, 8, hipMemcpyDeviceToHost, st));
    FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
    FBG_HIP_TRY(ctx, hipGetLastError());
    if (flag != 0) { ctx->sp_decline = 8; return FBG_OK; }
    ctx->grs_tricks_off = disable_tricks;
    *ok = 1;
    return FBG_OK;
}

// the ignore characters as a mask over the key's symbol codes (ranks of the bytes that occur)
static bool sp_ignore_mask(fbg_ctx *ctx, uint64_t *by_code)
{
    *by_code = 0;
    if (!ctx->have_ignore) return true;
    int code = 0;
    for (int c = 0; c < 256; c++) {
        const bool occurs = ctx->byte_hist[c] != 0;
        if (ctx->ignore_tab[c] && occurs) { if (code >= 64) return false; *by_code |= 1ull << code; }
        if (occurs) code++;
    }
    return true;
}

// After the sort of the (key, cell) pairs.  *done = 1: the index is the sorted slots, the group tables and the
// per-column maxima (ctx->granked, ctx->spanned); 0: vals hold text positions again, continue with the record path.
int fbg_span_try(fbg_ctx *ctx, uint64_t *keys, uint32_t *vals, const KeyGeom &g, int *done)
{
    *done = 0;
    ctx->granked = false; ctx->spanned = false;
    const uint64_t N = ctx->N, n = ctx->n;
    hipStream_t st = ctx->stream;
    FBG_TRY(fbg_stage_begin(ctx, FBG_STAGE_RANKSCAN));
    int launches = 0;
    uint64_t by_code = 0;
    bool good = sp_ignore_mask(ctx, &by_code);
    ctx->sp_decline = good ? 0 : 10;
    FBG_TRY(fbg_reserve(ctx, ctx->gmax, (n + 1) * 4));
    ctx->rk_keys = keys; ctx->sa_ptr = vals;
    ctx->rk_layout = FBG_SLOTS_PAIRS; ctx->rk_pb = 0; ctx->rk_b = g.b; ctx->rk_key_bits = g.key_bits; ctx->rk_K = g.K;
    ctx->grs_ign_lo = (uint32_t)by_code; ctx->grs_ign_hi = (uint32_t)(by_code >> 32);
    ctx->sp_G = 0; ctx->sp_R = 0; ctx->sp_n_irr = 0; ctx->sp_n_odd[0] = ctx->sp_n_odd[1] = ctx->sp_n_odd[2] = ctx->sp_n_odd[3] = 0; ctx->sp_odd_cap = 0;
    SpArgs a;
    if (good && ctx->opt.span_scan == 2) {
        // debugging aid: the sorted values are a permutation of the cells (sum and xor agree), the keys ascend
        sp_args(ctx, a, 0);
        FBG_HIP_TRY(ctx, hipMemsetAsync(a.counters + 16, 0, 8 * sizeof(unsigned long long), st));
        hipLaunchKernelGGL(k_sp_check, dim3(fbg_blocks(N, 256)), dim3(256), 0, st, keys, (const uint32_t *)vals, ctx->sp_cells.as<uint32_t>(), N, a.counters + 16, a.ksh);
        unsigned long long c[5];
        FBG_HIP_TRY(ctx, hipMemcpyAsync(c, a.counters + 16, sizeof(c), hipMemcpyDeviceToHost, st));
        FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
        if (c[0] != c[1] || c[2] != c[3] || c[4] != 0)
            return fbg_fail(ctx, FBG_ERR_HIP, "span scan: the sorted slots are not the cells in key order (%llu keys out of order)", c[4]);
    }
    if (good) {
        // groups
        const unsigned tiles = fbg_blocks(N, SP_TILE);
        FBG_TRY(fbg_reserve(ctx, ctx->sp_tiles, ((size_t)tiles + 1) * 8));
        sp_args(ctx, a, 0);
        hipLaunchKernelGGL((k_sp_groups<false>), dim3(tiles), dim3(SP_THREADS), 0, st, a);
        FBG_HIP_TRY(ctx, hipMemsetAsync(a.tile_cnt + tiles, 0, 8, st));
        FBG_TRY(sp_with_tmp(ctx, [&](void *tmp, size_t &bytes) {
            return rocprim::exclusive_scan(tmp, bytes, a.tile_cnt, a.tile_cnt, 0ull, (size_t)tiles + 1, rocprim::plus<unsigned long long>(), st);
        }));
        unsigned long long tot = 0;
        FBG_HIP_TRY(ctx, hipMemcpyAsync(&tot, a.tile_cnt + tiles, 8, hipMemcpyDeviceToHost, st));
        FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
        launches += 2;
        const uint64_t G = (uint32_t)tot, n_irr = tot >> 32;
        ctx->sp_G = G; ctx->sp_n_irr = n_irr;
        for (DevBuf *b : {&ctx->sp_gstart, &ctx->sp_gcol, &ctx->sp_gflags, &ctx->sp_rid, &ctx->sp_gplo, &ctx->sp_gphi, &ctx->sp_gval})
            FBG_TRY(fbg_reserve(ctx, *b, (G + 1) * 4));
        FBG_TRY(fbg_reserve(ctx, ctx->sp_irr, (n_irr + 1) * 8));
        sp_args(ctx, a, 0);
        FBG_HIP_TRY(ctx, hipMemsetAsync(a.gflags, 0, (G + 1) * 4, st));
        hipLaunchKernelGGL((k_sp_groups<true>), dim3(tiles), dim3(SP_THREADS), 0, st, a);
        const uint32_t N32 = (uint32_t)N;
        FBG_HIP_TRY(ctx, hipMemcpyAsync(a.gstart + G, &N32, 4, hipMemcpyHostToDevice, st));
        // runs
        const unsigned gtiles = fbg_blocks(G, SP_TILE);
        FBG_TRY(fbg_reserve(ctx, ctx->ps_e, ((size_t)gtiles + 1) * 4));
        a.rtile = ctx->ps_e.as<uint32_t>();
        hipLaunchKernelGGL((k_sp_runs<false>), dim3(gtiles), dim3(SP_THREADS), 0, st, a);
        FBG_HIP_TRY(ctx, hipMemsetAsync(a.rtile + gtiles, 0, 4, st));
        FBG_TRY(sp_with_tmp(ctx, [&](void *tmp, size_t &bytes) {
            return rocprim::exclusive_scan(tmp, bytes, a.rtile, a.rtile, 0u, (size_t)gtiles + 1, rocprim::plus<uint32_t>(), st);
        }));
        uint32_t R32 = 0;
        FBG_HIP_TRY(ctx, hipMemcpyAsync(&R32, a.rtile + gtiles, 4, hipMemcpyDeviceToHost, st));
        FBG_HIP_TRY(ctx, hipStreamSynchronize(st));               // (also: N32 above lives on this frame)
        launches += 3;
        ctx->sp_R = R32;
        FBG_TRY(fbg_reserve(ctx, ctx->sp_rstart, ((size_t)R32 + 1) * 4));
        a.rstart = ctx->sp_rstart.as<uint32_t>(); a.R = R32;
        hipLaunchKernelGGL((k_sp_runs<true>), dim3(gtiles), dim3(SP_THREADS), 0, st, a);
        const uint32_t G32 = (uint32_t)G;
        FBG_HIP_TRY(ctx, hipMemcpyAsync(a.rstart + R32, &G32, 4, hipMemcpyHostToDevice, st));
        // the odd groups, listed
        const uint32_t cap = (uint32_t)std::min<uint64_t>(G, 1u << 24);
        ctx->sp_odd_cap = cap;
        FBG_TRY(fbg_reserve(ctx, ctx->sp_odd, (size_t)cap * 16));
        sp_args(ctx, a, 0);
        FBG_HIP_TRY(ctx, hipMemsetAsync(a.counters, 0, 16 * sizeof(unsigned long long), st));
        hipLaunchKernelGGL(k_sp_oddlist, dim3(fbg_blocks(G, 1024)), dim3(1024), 0, st, a);
        unsigned long long h[12];
        FBG_HIP_TRY(ctx, hipMemcpyAsync(h, a.counters, sizeof(h), hipMemcpyDeviceToHost, st));
        FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
        launches += 2;
        if (h[3] != 0) { good = false; ctx->sp_decline = 9; }
        for (int c = 0; c < 4; c++) ctx->sp_n_odd[c] = (uint32_t)std::min<unsigned long long>(h[8 + c], cap);
        // The lists in COLUMN order: the workgroups of an odd group read the texts of all its members -- hundreds of rows at
        // one column, two memory lines each -- and so do the groups of the columns next to it.  In key order those meet at
        // random times; in column order the groups in flight at one time sit in neighbouring columns and share the lines
        // through the L2 (35 -> 24 GB fetched for 1000 x 200 000).
        for (int c = 0; c < 4 && good; c++) {
            const uint32_t cnt = ctx->sp_n_odd[c];
            if (cnt < 2) continue;
            FBG_TRY(fbg_reserve(ctx, ctx->ps_f, (size_t)cnt * 4 * 3));
            uint32_t *cols = ctx->ps_f.as<uint32_t>(), *cols_s = cols + cnt, *list_s = cols + 2 * (size_t)cnt;
            uint32_t *lst = a.odd + (size_t)c * a.odd_cap;
            hipLaunchKernelGGL(k_sp_list_cols, dim3(fbg_blocks(cnt, 256)), dim3(256), 0, st, a, (const uint32_t *)lst, cnt, cols);
            int nb = 1;
            while ((1ull << nb) < n + 1) nb++;
            FBG_TRY(sp_with_tmp(ctx, [&](void *tmp, size_t &bytes) {
                return rocprim::radix_sort_pairs(tmp, bytes, cols, cols_s, lst, list_s, (size_t)cnt, 0u, (unsigned)nb, st);
            }));
            FBG_HIP_TRY(ctx, hipMemcpyAsync(lst, list_s, (size_t)cnt * 4, hipMemcpyDeviceToDevice, st));
            launches += 3;
        }
    }
    int ok = 0;
    if (good) FBG_TRY(sp_scan(ctx, 0, &ok, &launches));
    if (ok) {
        ctx->granked = true; ctx->spanned = true;
        ctx->ranked = false; ctx->part_active = false;
        *done = 1;
    } else {
        sp_args(ctx, a, 0);
        hipLaunchKernelGGL(k_sp_to_positions, dim3(fbg_blocks(N, 256)), dim3(256), 0, st, a, vals, keys);
        launches++;
        FBG_HIP_TRY(ctx, hipGetLastError());
    }
    return fbg_stage_end(ctx, FBG_STAGE_RANKSCAN, launches);
}

// the scan again for the other setting of the elastic tricks (fbg_scan_f); *ok = 0: a capacity did not hold
int fbg_span_rescan(fbg_ctx *ctx, int disable_tricks, int *ok)
{
    int launches = 0;
    return sp_scan(ctx, disable_tricks, ok, &launches);
}
