// gapped_rank.hip -- the extension scan in suffix-array order for MSAs WITH gaps and / or ignore characters.
//
// compute_f (fbg.cpp:1579-1695) keeps one text pointer per row; while a row shows gaps the pointer stays on the row's
// next symbol (fbg.cpp:1687-1691), so text position p is the pointer of its row for the columns (column of the row's
// previous symbol, column of p] -- a span [lo(p), hi(p)], one column wide except behind a gap run.  The leaves
// coloured at column x are the positions whose span holds x (rows that have shown a symbol already: fullrow[],
// fbg.cpp:1605-1608, 1621), and what the walk of fbg.cpp:1633-1678 extracts for a coloured leaf is 1 + its longest
// match with any leaf NOT coloured at x.  In suffix-array order that is local: a slot none of whose two
// neighbours is coloured together with it extends by g = 1 + max(LCP with the slot before, LCP with the slot after),
// the LCPs being the equal leading symbols of the sort keys -- no text access -- and g is the same for every column of
// its span; f takes the column its row reaches after g symbols, fi = col_i(rank_i(x) + g) clamped to the row's last
// symbol and to the next ignore character (fbg.cpp:1657-1670).  Slots with a neighbour coloured in a common column
// (runs of consecutive coloured ranks, fbg.cpp:1633-1641) and slots that tie with a neighbour on the whole key become
// (column, slot) candidates: sorted, they are walked run by run like k_runs does for gap-free MSAs.
//
// Round 1 sent these MSAs down the record path: ranks, LCPs and run hints carried from suffix order to text order by
// three passes of 16-byte records, then streamed per column.  Here nothing travels to text order.  What the scan needs
// per slot is the column span and the row of a text position: a table of 16 bytes per 128 positions (first column, row,
// up to two gap runs inside) answers that -- one random access of an Infinity-Cache-sized table per question, and
// 5 * 10^8 of those cost 15 ms on an MI355X, three times the rest of the scan.  So most slots never ask:
//   * the sort's payload carries one bit per slot, computed in text order when the keys are packed: "the K symbols from
//     this position on are one row's consecutive columns" (no gap run, row start, separator among them).  For such a
//     slot the span is one column x and fi = x + g - 1 (less up to an ignore character, read from the key).
//   * of those slots only the ones with g >= t matter, t a threshold from a sample of the g values such that every
//     column should see several: g computed from the two neighbours is an upper bound of the g a run member gets
//     (mins over more LCPs), so a skipped slot contributes at most x + t - 2.  Afterwards every column whose maximum is
//     below x + t - 2 is redone exactly: the pointers of its rows found by binary search, put in a hash set, the
//     slots' positions streamed once more against it (4 bytes per slot), and the slots found treated like the others
//     (few columns: column 0, whose pointers are all row starts, is one).
// The slots that do ask -- flagged ones, ties, g >= t: about 15 % on a random MSA with 5 % gap cells -- also ask for
// their neighbours' spans and list the whole run of coloured slots around themselves as candidates (duplicates are
// skipped after the sort), because a skipped neighbour would not list itself.
//
// The scan depends on the elastic tricks (which rows count as started): it runs for the setting the index was built
// for and again, on the kept slots, when fbg_scan_f asks for the other one.  Declines (record path) for tie groups of
// more than 64 suffixes, candidate lists beyond their capacity (similar rows with gaps), '-' among the ignore
// characters, and more than 65534 rows.  A position that is the pointer for more than 64 columns (long gap runs, the
// gaps a row starts or ends with) gets a wave of its own.
#include "fbg_internal.h"
#include "text_cmp.h"
#include <rocprim/rocprim.hpp>
#include <algorithm>
#include <cmath>

#define GW_BITS 7
#define GW (1u << GW_BITS)
#define GW_IRREGULAR 0xffffu
#define GR_THREADS 256
#define GR_LONG 64u                  // a slot that is the pointer for more columns than this gets a wave of its own (k_grs_long)
#define GR_LONG_CAP (1u << 20)
#define GR_MAX_TIE 64u
#define GR_SAMPLE (1u << 16)
#define GR_HASH (1u << 15)           // slots of the position set of the columns that are redone (LDS, 128 KB)
#define GR_HASH_FILL (GR_HASH / 2)
#define GR_EMPTY 0xffffffffu
#define GR_HEAD 0x80000000u          // k_grs_classify's list: the slot is the first of a tie group (texts below 2^31 symbols)

struct GWin {                        // window of GW text positions
    uint32_t col0;                   // MSA column of its first position
    uint32_t lo0;                    // first column that position is the row pointer for (0: it is its row's first symbol)
    uint16_t row;                    // row of the window (GW_IRREGULAR: look the positions up in colT / pos)
    uint8_t off[2];                  // up to two gap runs inside: len[e] gap cells right before position off[e] (0: unused)
    uint16_t len[2];
};

struct GrsArgs {
    const uint64_t *keys;            // sorted keys
    uint32_t *vals;                  // positions in suffix order (| flag << 31); tie groups put in text order by k_grs_ties
    uint32_t vmask;                  // 0x7fffffff: bit 31 of a value = "something irregular within K symbols"; 0xffffffff: no flags
    const uint8_t *T;
    const GWin *win;                 // nullptr: rows without gaps (ignore characters only), position = row * (n + 1) + column
    const uint32_t *colT, *pos, *tot;
    const uint8_t *is_ignore;        // by byte; nullptr: no ignore characters
    uint64_t N, n, m;                // text length, columns, rows
    uint64_t own_lo, own_hi;         // the slots this scan works on (a key-range partition: the middle of the arrays) ...
    uint64_t lim_lo, lim_hi;         // ... and the slots that exist around them (halos of the neighbouring partitions)
    int open_lo, open_hi;            // more slots, unseen, beyond lim_lo / lim_hi (another partition's)
    int b, K, key_bits, disable_tricks;
    uint32_t ign_lo, ign_hi;         // bit c of (ign_hi:ign_lo): symbol code c is an ignore character
    uint32_t t;                      // slots with g < t that are regular are skipped (1: none)
    uint32_t *fmax;                  // per column: largest fi seen
    unsigned long long *cand;        // (column << 32 | slot)
    uint32_t *pm;                    // scratch parallel to cand
    unsigned long long cand_cap;
    uint32_t *longs;                 // slots with long spans
    unsigned long long *counters;    // [0] candidates, [1] decline flag, [2] long slots, [3] columns to redo, [4] their slots, [6] long runs
};

__device__ __forceinline__ uint32_t gr_key_lcp(uint64_t a, uint64_t c, int b, int key_bits)
{
    const uint64_t d = a ^ c;
    const uint32_t bits = (uint32_t)(__clzll((long long)d) - (64 - key_bits));
    return (bits * ((65536u + (uint32_t)b - 1) / (uint32_t)b)) >> 16;
}

// row of text position p: the last row that starts at or before it
__device__ __forceinline__ uint32_t gr_row_of(const uint32_t *__restrict__ pos, uint32_t m, uint32_t p)
{
    uint32_t lo = 0, hi = m;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (pos[mid] <= p) lo = mid; else hi = mid; }
    return lo;
}

__device__ __forceinline__ uint32_t gr_win_col(const GWin &w, uint32_t o, uint32_t &before)
{
    uint32_t c = w.col0 + o;
    before = 0;
    if (w.off[0] && w.off[0] <= o) { c += w.len[0]; if (w.off[0] == o) before = w.len[0]; }
    if (w.off[1] && w.off[1] <= o) { c += w.len[1]; if (w.off[1] == o) before = w.len[1]; }
    return c;
}

// The columns [lo, hi] at which text position p is a COLOURED pointer of its row (empty: lo > hi), its row, and the
// window entry that was read (w.row = GW_IRREGULAR when none applies).  A symbol points from the column after the
// row's previous symbol to its own; the '#' behind a row from the column after the row's last symbol to the last
// column (the finished row keeps its leaf, fbg.cpp:1687-1691); the sentinel never; a row's first symbol only with the
// elastic tricks off (fullrow[], fbg.cpp:1605-1608).
__device__ __forceinline__ void gr_span(const GrsArgs &a, uint32_t p, uint32_t &lo, uint32_t &hi, uint32_t &row, GWin &w)
{
    lo = 1; hi = 0; row = 0;
    w.row = GW_IRREGULAR;
    if (p >= a.N - 1) return;
    bool first;
    if (!a.win) {
        row = p / (uint32_t)(a.n + 1);
        const uint32_t c = p - row * (uint32_t)(a.n + 1);
        if (c >= a.n) return;
        lo = c; hi = c;
        first = c == 0;
    } else {
        w = a.win[p >> GW_BITS];
        if (w.row != GW_IRREGULAR) {
            const uint32_t o = p & (GW - 1);
            uint32_t before;
            const uint32_t c = gr_win_col(w, o, before);
            row = w.row; hi = c;
            lo = o ? c - before : w.lo0;
            first = o == 0 && w.lo0 == 0;
        } else {
            const uint32_t c = a.colT[p];
            row = gr_row_of(a.pos, (uint32_t)a.m, p);
            first = p == a.pos[row];
            lo = first ? 0u : a.colT[p - 1] + 1;
            hi = c < a.n ? c : (uint32_t)a.n - 1;
        }
    }
    if (first && !a.disable_tricks) { lo = 1; hi = 0; }
}

// MSA column of the symbol at text position p of row `row` (not a separator); w: the window entry of position wp, if any
__device__ __forceinline__ uint32_t gr_col(const GrsArgs &a, uint32_t p, uint32_t row, const GWin &w, uint32_t wp)
{
    if (!a.win) return p - row * (uint32_t)(a.n + 1);
    uint32_t before;
    if (w.row != GW_IRREGULAR && (wp >> GW_BITS) == (p >> GW_BITS)) return gr_win_col(w, p & (GW - 1), before);
    const GWin v = a.win[p >> GW_BITS];
    if (v.row == GW_IRREGULAR) return a.colT[p];
    return gr_win_col(v, p & (GW - 1), before);
}

// One wave per window: the table entry, and the bitmap "position p is irregular": not the column after its
// predecessor's (gap run before it, row start), or no symbol at all ('#', sentinel, beyond the text).
__global__ __launch_bounds__(256) void k_gw_build(const uint32_t *__restrict__ colT, const uint32_t *__restrict__ pos, uint64_t N, uint32_t n,
                                                  uint32_t m, const uint16_t *__restrict__ winrow, GWin *__restrict__ win,
                                                  unsigned long long *__restrict__ ebits)
{
    // four windows per wave, their loads issued together: with one window (8 bytes a lane) per wave the kernel ran at the rate
    // the waves in flight could keep bytes in flight -- 1.8 TB/s over colT (round 4)
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wfirst = ((uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
    __shared__ GWin out[4];
    uint32_t c0a[4], c1a[4], cpa[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint64_t w0 = (wfirst + k) << GW_BITS;
        const uint64_t q0 = w0 + lane, q1 = w0 + 64 + lane;
        c0a[k] = q0 < N ? colT[q0] : n;
        c1a[k] = q1 < N ? colT[q1] : n;
        cpa[k] = (lane == 0 && w0 > 0 && w0 <= N) ? colT[w0 - 1] : n;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
    const uint64_t w = wfirst + k;
    const uint64_t w0 = w << GW_BITS;
    if (w0 >= N + GW) continue;                                    // (one window of padding in the bitmap)
    const uint32_t c0 = c0a[k], c1 = c1a[k];
    // column of the position before each of the lane's two
    uint32_t cp0 = __shfl_up(c0, 1, 64), cp1 = __shfl_up(c1, 1, 64);
    const uint32_t c0_last = __shfl(c0, 63, 64);
    if (lane == 0) { cp0 = cpa[k]; cp1 = c0_last; }
    const unsigned long long e0 = __ballot(c0 >= n || cp0 >= n || c0 != cp0 + 1), e1 = __ballot(c1 >= n || cp1 >= n || c1 != cp1 + 1);
    if (lane == 0) { ebits[2 * w] = e0; ebits[2 * w + 1] = e1; }
    if (w0 >= N) continue;
    const bool sep = c0 >= n || c1 >= n;                           // a '#', the sentinel, or the end of the text inside
    // a gap run right before a position of the window (not its first: that one is in lo0)
    const bool ev0 = lane > 0 && c0 < n && cp0 < n && c0 > cp0 + 1;
    const bool ev1 = c1 < n && cp1 < n && c1 > cp1 + 1;
    const unsigned long long b0 = __ballot(ev0), b1 = __ballot(ev1);
    const uint32_t events = (uint32_t)__popcll(b0) + (uint32_t)__popcll(b1);
    const bool wide = __ballot((ev0 && c0 - cp0 - 1 > 0xffffu) || (ev1 && c1 - cp1 - 1 > 0xffffu)) != 0;
    const bool irregular = __ballot(sep) != 0 || events > 2 || wide;
    // the events in position order
    const unsigned long long below = (1ull << lane) - 1;
    const uint32_t r0 = (uint32_t)__popcll(b0 & below), r1 = (uint32_t)__popcll(b0) + (uint32_t)__popcll(b1 & below);
    GWin &e = out[threadIdx.x >> 6];
    if (lane == 0) {
        e.col0 = c0; e.off[0] = e.off[1] = 0; e.len[0] = e.len[1] = 0; e.row = GW_IRREGULAR; e.lo0 = 0;
    }
    __builtin_amdgcn_wave_barrier();
    if (!irregular && lane == 0) {
        // the window's row: noted by the text writer (k_write_text) when it placed the window's first symbol
        const uint32_t row = winrow[w];
        e.row = (uint16_t)row;
        e.lo0 = pos[row] == (uint32_t)w0 ? 0u : cp0 + 1;
    }
    __builtin_amdgcn_wave_barrier();
    if (!irregular) {
        if (ev0) { e.off[r0] = (uint8_t)lane; e.len[r0] = (uint16_t)(c0 - cp0 - 1); }
        if (ev1) { e.off[r1] = (uint8_t)(64 + lane); e.len[r1] = (uint16_t)(c1 - cp1 - 1); }
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) win[w] = e;
    __builtin_amdgcn_wave_barrier();
    }
}

// the bitmap for rows without gaps: irregular = a row's first position, its '#', the sentinel, beyond the text
__global__ void k_grs_ebits_gapfree(uint64_t N, uint64_t n, uint64_t words, unsigned long long *__restrict__ ebits)
{
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= words) return;
    unsigned long long e = 0;
    uint64_t p = w * 64, c = p % (n + 1);
    for (int i = 0; i < 64; i++, p++) {
        if (p >= N - 1 || c == 0 || c == n) e |= 1ull << i;
        c = c == n ? 0 : c + 1;
    }
    ebits[w] = e;
}

// the tie group (equal keys) that starts at slot k0 in text order: insertion sort by the text beyond the key, values
// rewritten in place
__device__ __forceinline__ void gr_order_group(const GrsArgs &a, uint64_t k0)
{
    const uint64_t key = a.keys[k0];
    uint32_t s = 2;
    while (k0 + s < a.own_hi && s <= GR_MAX_TIE && a.keys[k0 + s] == key) s++;
    if (s > GR_MAX_TIE) { a.counters[1] = 1; return; }
    // (two suffixes with equal keys both have K symbols before the sentinel, and the compare ends at the sentinel at the latest)
    for (uint32_t i = 1; i < s; i++) {
        const uint32_t cur = a.vals[k0 + i], pc = cur & a.vmask;
        uint32_t j = i;
        while (j > 0) {
            const uint32_t o = a.vals[k0 + j - 1], po = o & a.vmask;
            const uint32_t h = fbg_extend_match(a.T, (uint64_t)po + a.K, (uint64_t)pc + a.K, 0);
            if (a.T[(uint64_t)po + a.K + h] < a.T[(uint64_t)pc + a.K + h]) break;
            a.vals[k0 + j] = o;
            j--;
        }
        a.vals[k0 + j] = cur;
    }
}

// every tie group of the own slots (one streaming pass over the keys: the partitions, and scans without a threshold)
__global__ void k_grs_ties(GrsArgs a)
{
    const uint64_t k0 = a.own_lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;      // (equal keys never straddle partitions)
    if (k0 + 1 >= a.own_hi) return;
    const uint64_t key = a.keys[k0];
    if (a.keys[k0 + 1] != key) return;
    if (k0 > a.own_lo && a.keys[k0 - 1] == key) return;
    gr_order_group(a, k0);
}

// LCP of the suffixes in slots k - 1 and k (final order)
__device__ __forceinline__ uint32_t gr_slot_lcp(const GrsArgs &a, uint64_t k)
{
    if (k <= a.lim_lo || k >= a.lim_hi) return 0;
    const uint64_t x = a.keys[k - 1], y = a.keys[k];
    if (x != y) return gr_key_lcp(x, y, a.b, a.key_bits);
    return fbg_clamp_lcp(fbg_extend_match(a.T, (uint64_t)(a.vals[k - 1] & a.vmask) + a.K, (uint64_t)(a.vals[k] & a.vmask) + a.K, 0) + (uint32_t)a.K);
}

// histogram of g = 1 + max(LCP before, LCP after) over evenly spaced slots (ties: K + 1)
__global__ __launch_bounds__(256) void k_grs_sample(GrsArgs a, uint64_t stride, unsigned long long *__restrict__ hist)
{
    __shared__ uint32_t h[66];
    if (threadIdx.x < 66) h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t s = a.own_lo + i * stride + 1;
    if (i < GR_SAMPLE && s + 1 < a.own_hi) {
        const uint64_t kp = a.keys[s - 1], k = a.keys[s], kn = a.keys[s + 1];
        uint32_t g = (uint32_t)a.K + 1;
        if (kp != k && kn != k) g = max(gr_key_lcp(kp, k, a.b, a.key_bits), gr_key_lcp(k, kn, a.b, a.key_bits)) + 1;
        atomicAdd(&h[min(g, 65u)], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 66 && h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

// first symbol of the key that is an ignore character, among its first `reach` (<= K); reach if none
__device__ __forceinline__ uint32_t gr_key_first_ignore(const GrsArgs &a, uint64_t key, uint32_t reach)
{
    for (uint32_t k = 0; k < reach; k++) {
        const uint32_t c = (uint32_t)(key >> (a.b * (a.K - 1 - (int)k))) & ((1u << a.b) - 1);
        if ((c < 32 ? a.ign_lo >> c : a.ign_hi >> (c - 32)) & 1u) return k;
    }
    return reach;
}

// fi of fbg.cpp:1657-1670 for the row pointer p of row `row` extended by g.  key_ok: the key holds the first g symbols.
// w: the window entry of p (row GW_IRREGULAR: none).
__device__ __forceinline__ uint32_t gr_extent(const GrsArgs &a, uint32_t p, uint32_t row, uint32_t g, uint64_t key, bool key_ok, const GWin &w)
{
    uint32_t reach = g;                                    // symbols from p on in which an ignore character clamps
    uint32_t fi;
    const bool inside = a.win && w.row != GW_IRREGULAR && ((p + g - 1) >> GW_BITS) == (p >> GW_BITS);
    if (inside) {
        // a regular window holds no separator: the row goes on at least to the window's end
        uint32_t before;
        fi = gr_win_col(w, (p + g - 1) & (GW - 1), before);                              // 1666
    } else {
        const uint32_t p0 = a.pos[row], tt = a.tot[row];
        const unsigned long long gg = (unsigned long long)(p - p0) + g;                  // 1657
        if (gg > tt) {                                                                   // 1659-1664
            fi = a.disable_tricks ? (uint32_t)a.n : (tt ? gr_col(a, p0 + tt - 1, row, w, p) : 0u);
            reach = p0 + tt - p;
        } else fi = gr_col(a, p + g - 1, row, w, p);                                     // 1666
    }
    if (a.is_ignore && reach) {                                                          // 1669-1670: the row's first ignore column at or after x
        uint32_t first = reach;
        if (key_ok && reach <= (uint32_t)a.K) first = gr_key_first_ignore(a, key, reach);
        else {
            for (uint32_t k = 0; k < reach; k++)
                if (a.is_ignore[a.T[(uint64_t)p + k]]) { first = k; break; }
        }
        if (first < reach) fi = min(fi, gr_col(a, p + first, row, w, p));
    }
    return fi;
}

__device__ __forceinline__ void gr_update(const GrsArgs &a, uint32_t x, uint32_t fi)
{
    if (fi > x && a.fmax[x] < fi) atomicMax(&a.fmax[x], fi);
}

// The run of coloured slots around slot s, column by column: every slot before (after) s that is coloured at a column
// of [lo, hi] together with all slots between it and s.  f(x_lo, x_hi, slot) for each.
template <typename F>
__device__ __forceinline__ void gr_walk(const GrsArgs &a, uint64_t s, uint32_t lo, uint32_t hi, F &&f)
{
    for (int dir = -1; dir <= 1; dir += 2) {
        uint32_t l = lo, h = hi;
        uint64_t q = s + dir;
        for (; q >= a.lim_lo && q < a.lim_hi; q += dir) {               // (q = 0 - 1 wraps and ends the loop)
            uint32_t ql, qh, qrow;
            GWin qw;
            gr_span(a, a.vals[q] & a.vmask, ql, qh, qrow, qw);
            l = max(l, ql); h = min(h, qh);
            if (l > h) break;
            f(l, h, q);
        }
        if (l <= h && (dir < 0 ? a.open_lo : a.open_hi)) a.counters[1] = 1;   // the run goes on where this partition cannot see
    }
}

// The work for one slot per lane (all lanes of a wave come here together; `in`: the lane has a slot).  Direct updates
// where no neighbour is coloured in the same column, candidates (the slot and the run around it) where one is.
// k2p / k2n: the keys two slots before / after (only read when cheap_ok).
// cheap_ok: a neighbour that the scan skips (flag clear, no tie, g < t) need not be looked at when this slot is regular
// with g >= t: the LCP with such a neighbour is below t - 1, so it does not decide this slot's g whether or not the two
// share a column, and what the neighbour itself contributes stays within the bound the skipped slots are held to.
// stage / stage_fill: the workgroup's candidates collect in LDS (GR_STAGE entries) and leave with one reservation in
// the global list per flush -- a reservation per wave, 10^6 of them on one address, cost 12 ms.
#define GR_STAGE 2048
// The slot comes resolved: v its value (position | flag), tie / g from its keys, see_prev / see_next whether the slot
// before / after has to be looked at (it exists and is not one the scan skips next to a regular slot with g >= t).
__device__ __forceinline__ void gr_slot_core(const GrsArgs &a, bool in, uint64_t s, uint32_t v, uint64_t key, bool tie, uint32_t g, bool see_prev,
                                             bool see_next, unsigned long long *stage, uint32_t *stage_fill)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t p = v & a.vmask;
    bool work = in;
    uint32_t lo = 1, hi = 0, row = 0, fi0 = 0;
    GWin w;
    w.row = GW_IRREGULAR;
    if (work) {
        gr_span(a, p, lo, hi, row, w);
        work = lo <= hi;
    }
    if (work && hi - lo >= GR_LONG) {                                   // long gap run, leading / trailing gaps: k_grs_long
        const unsigned long long at = atomicAdd(&a.counters[2], 1ull);
        if (at < GR_LONG_CAP) a.longs[at] = (uint32_t)s; else a.counters[1] = 1;
        work = false;
    }
    // the neighbours' spans: columns shared with a coloured neighbour need the run treatment
    uint32_t pl = 1, ph = 0, nl = 1, nh = 0, ncand = 0;
    if (work) {
        uint32_t r2;
        GWin w2;
        if (see_prev) { gr_span(a, a.vals[s - 1] & a.vmask, pl, ph, r2, w2); pl = max(pl, lo); ph = min(ph, hi); }
        if (see_next) { gr_span(a, a.vals[s + 1] & a.vmask, nl, nh, r2, w2); nl = max(nl, lo); nh = min(nh, hi); }
        if (!tie) fi0 = gr_extent(a, p, row, g, key, true, w);
        if (tie) ncand = hi - lo + 1;
        else {
            if (pl <= ph) ncand += ph - pl + 1;
            if (nl <= nh) ncand += nh - nl + 1;
            if (pl <= ph && nl <= nh) { const uint32_t ol = max(pl, nl), oh = min(ph, nh); if (ol <= oh) ncand -= oh - ol + 1; }
        }
        if (pl <= ph || nl <= nh) gr_walk(a, s, lo, hi, [&](uint32_t l, uint32_t h, uint64_t) { ncand += h - l + 1; });
    }
    // room for the candidates: in the workgroup's stage, else (full) straight in the global list -- one reservation per wave
    uint32_t inc = ncand;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d, 64); if ((int)lane >= d) inc += o; }
    unsigned long long wbase = 0;
    uint32_t staged = 0;
    if (lane == 63 && inc) {
        const uint32_t at = atomicAdd(stage_fill, inc);
        if (at + inc <= GR_STAGE) { wbase = at; staged = 1; }
        else { atomicSub(stage_fill, inc); wbase = atomicAdd(&a.counters[0], (unsigned long long)inc); }
    }
    wbase = __shfl(wbase, 63, 64);
    staged = __shfl(staged, 63, 64);
    if (!work) return;
    unsigned long long o = wbase + inc - ncand;
    auto put = [&](unsigned long long e) {
        if (staged) stage[o] = e; else if (o < a.cand_cap) a.cand[o] = e;
        o++;
    };
    for (uint32_t x = lo; x <= hi; x++) {
        if (tie || (x >= pl && x <= ph) || (x >= nl && x <= nh)) put((unsigned long long)x << 32 | (uint32_t)s);
        else gr_update(a, x, fi0);
    }
    if (pl <= ph || nl <= nh)
        gr_walk(a, s, lo, hi, [&](uint32_t l, uint32_t h, uint64_t q) {
            for (uint32_t x = l; x <= h; x++) put((unsigned long long)x << 32 | (uint32_t)q);
        });
}

// the same from the slot's keys: every neighbour that exists is looked at
__device__ __forceinline__ void gr_slot(const GrsArgs &a, bool in, uint64_t s, uint64_t key, uint64_t kp, uint64_t kn, unsigned long long *stage,
                                        uint32_t *stage_fill)
{
    const bool has_prev = s > a.lim_lo, has_next = s + 1 < a.lim_hi;
    const uint32_t v = in ? a.vals[s] : 0u;
    const bool tie = in && ((has_prev && kp == key) || (has_next && kn == key));
    uint32_t g = 0;
    if (in && !tie) g = max(has_prev ? gr_key_lcp(kp, key, a.b, a.key_bits) : 0u, has_next ? gr_key_lcp(key, kn, a.b, a.key_bits) : 0u) + 1;   // 1656
    gr_slot_core(a, in, s, v, key, tie, g, in && has_prev, in && has_next, stage, stage_fill);
}

// the workgroup's staged candidates -> the global list (all threads; force: whatever the fill, else only when half full)
__device__ __forceinline__ void gr_flush(const GrsArgs &a, unsigned long long *stage, uint32_t *stage_fill, unsigned long long *gbase, bool force)
{
    __syncthreads();
    const uint32_t have = *stage_fill;
    if (have == 0 || (!force && have < GR_STAGE / 2)) return;          // uniform
    if (threadIdx.x == 0) *gbase = atomicAdd(&a.counters[0], (unsigned long long)have);
    __syncthreads();
    const unsigned long long g0 = *gbase;
    for (uint32_t i = threadIdx.x; i < have; i += blockDim.x) if (g0 + i < a.cand_cap) a.cand[g0 + i] = stage[i];
    __syncthreads();
    if (threadIdx.x == 0) *stage_fill = 0;
    __syncthreads();
}

// every slot (no threshold: texts without flag bits, few rows)
__global__ __launch_bounds__(GR_THREADS) void k_grs_scan_all(GrsArgs a)
{
    __shared__ uint64_t skey[GR_THREADS + 2];
    __shared__ unsigned long long stage[GR_STAGE], gbase;
    __shared__ uint32_t stage_fill;
    const uint64_t base = a.own_lo + (uint64_t)blockIdx.x * GR_THREADS;
    const int me = (int)threadIdx.x + 1;
    const uint64_t s = base + threadIdx.x;
    const bool in = s < a.own_hi;
    skey[me] = s < a.lim_hi ? a.keys[s] : 0ull;
    if (threadIdx.x == 0) skey[0] = base > a.lim_lo ? a.keys[base - 1] : 0ull;
    if (threadIdx.x == 1) skey[GR_THREADS + 1] = base + GR_THREADS < a.lim_hi ? a.keys[base + GR_THREADS] : 0ull;
    if (threadIdx.x == 0) stage_fill = 0;
    __syncthreads();
    gr_slot(a, in, s, skey[me], skey[me - 1], skey[me + 1], stage, &stage_fill);
    gr_flush(a, stage, &stage_fill, &gbase, true);
}

// the slots of a list: the pointers of the columns that are redone (second pass)
__global__ __launch_bounds__(GR_THREADS) void k_grs_scan_list(GrsArgs a, const uint32_t *__restrict__ list, uint32_t count)
{
    __shared__ unsigned long long stage[GR_STAGE], gbase;
    __shared__ uint32_t stage_fill;
    if (threadIdx.x == 0) stage_fill = 0;
    __syncthreads();
    const uint64_t i = (uint64_t)blockIdx.x * GR_THREADS + threadIdx.x;
    const bool in = i < count;
    const uint64_t s = in ? list[i] : 0;
    const uint64_t key = in ? a.keys[s] : 0ull;
    const uint64_t kp = in && s > a.lim_lo ? a.keys[s - 1] : 0ull, kn = in && s + 1 < a.lim_hi ? a.keys[s + 1] : 0ull;
    gr_slot(a, in, s, key, kp, kn, stage, &stage_fill);
    gr_flush(a, stage, &stage_fill, &gbase, true);
}

#define GR_SEG 16384                 // slots per workgroup of k_grs_classify = capacity of its stretch of the list
// What k_grs_classify notes of a listed slot beside its index: the key, the value and what the keys around it say
// (meta: bits 0-6 g, 7 tie, 8 / 9 the slot before / after has to be looked at).  The scan of the listed slots reads these
// in order instead of gathering five keys and three values per slot from where they lie (250 bytes of sectors each).
struct GrsRec { uint64_t key; uint32_t v, meta; };
#define GR_META_TIE 0x80u
#define GR_META_PREV 0x100u
#define GR_META_NEXT 0x200u

// the slots k_grs_classify listed, one stretch of the list per workgroup
__global__ __launch_bounds__(GR_THREADS) void k_grs_scan_seg(GrsArgs a, const uint32_t *__restrict__ list, const GrsRec *__restrict__ recs,
                                                             const uint32_t *__restrict__ segcnt)
{
    __shared__ unsigned long long stage[GR_STAGE], gbase;
    __shared__ uint32_t stage_fill;
    if (threadIdx.x == 0) stage_fill = 0;
    __syncthreads();
    const uint32_t cnt = segcnt[blockIdx.x];
    const uint32_t *mine = list + (uint64_t)blockIdx.x * GR_SEG;
    const GrsRec *mine_r = recs + (uint64_t)blockIdx.x * GR_SEG;
    for (uint32_t r = 0; r < cnt; r += GR_THREADS) {
        const uint32_t i = r + threadIdx.x;
        const bool in = i < cnt;
        const uint64_t s = in ? (mine[i] & ~GR_HEAD) : 0;
        GrsRec rec;
        rec.key = 0; rec.v = 0; rec.meta = 0;
        if (in) rec = mine_r[i];
        if (in && (rec.meta & GR_META_TIE)) rec.v = a.vals[s];         // (the tie groups were put in text order after the note was taken)
        gr_slot_core(a, in, s, rec.v, rec.key, (rec.meta & GR_META_TIE) != 0, rec.meta & 0x7fu, (rec.meta & GR_META_PREV) != 0, (rec.meta & GR_META_NEXT) != 0,
                     stage, &stage_fill);
        gr_flush(a, stage, &stage_fill, &gbase, r + GR_THREADS >= cnt);
    }
}

// The slots the scan has to work on (flag bit set, tie, g >= t), listed: in a wave of k_grs_scan_all a few lanes would
// walk through the dependent table reads while the others wait -- every wave, at 15 % of the lanes; from the list all
// lanes do.  A workgroup takes GR_SEG slots and fills its own stretch of the list (no global counter: 10^7 waves adding
// to one address took 90 ms).
__global__ __launch_bounds__(GR_THREADS) void k_grs_classify(GrsArgs a, uint32_t *__restrict__ list, GrsRec *__restrict__ recs, uint32_t *__restrict__ segcnt)
{
    // four consecutive slots per thread and round (keys and values by 16-byte loads where the slots allow); the two keys
    // and the value either side of a thread's four come from the adjacent lanes, at a wave's edges from memory
    __shared__ uint32_t fill;
    const uint32_t lane = threadIdx.x & 63;
    uint32_t *mine = list + (uint64_t)blockIdx.x * GR_SEG;
    GrsRec *mine_r = recs + (uint64_t)blockIdx.x * GR_SEG;
    if (threadIdx.x == 0) fill = 0;
    __syncthreads();
    const bool aligned = ((a.own_lo & 3) == 0) && (((uintptr_t)a.keys & 15) == 0) && (((uintptr_t)a.vals & 15) == 0);
    for (uint32_t r = 0; r < GR_SEG; r += 4 * GR_THREADS) {
        const uint64_t s0 = a.own_lo + (uint64_t)blockIdx.x * GR_SEG + r + 4 * threadIdx.x;
        // K[j]: key of slot s0 - 2 + j (j = 0 .. 7), V[j]: value of slot s0 - 1 + j (j = 0 .. 5); 0 where no slot is
        uint64_t K[8];
        uint32_t V[6];
        if (aligned && s0 + 4 <= a.lim_hi) {
            const ulonglong2 k01 = *reinterpret_cast<const ulonglong2 *>(a.keys + s0), k23 = *reinterpret_cast<const ulonglong2 *>(a.keys + s0 + 2);
            const uint4 vv = *reinterpret_cast<const uint4 *>(a.vals + s0);
            K[2] = k01.x; K[3] = k01.y; K[4] = k23.x; K[5] = k23.y;
            V[1] = vv.x; V[2] = vv.y; V[3] = vv.z; V[4] = vv.w;
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) { const bool ex = s0 + i < a.lim_hi; K[2 + i] = ex ? a.keys[s0 + i] : 0ull; V[1 + i] = ex ? a.vals[s0 + i] : 0u; }
        }
        K[1] = __shfl_up(K[5], 1, 64); K[0] = __shfl_up(K[4], 1, 64); V[0] = __shfl_up(V[4], 1, 64);
        K[6] = __shfl_down(K[2], 1, 64); K[7] = __shfl_down(K[3], 1, 64); V[5] = __shfl_down(V[1], 1, 64);
        if (lane == 0) {
            K[1] = (s0 >= a.lim_lo + 1 && s0 - 1 < a.lim_hi) ? a.keys[s0 - 1] : 0ull;
            K[0] = (s0 >= a.lim_lo + 2 && s0 - 2 < a.lim_hi) ? a.keys[s0 - 2] : 0ull;
            V[0] = (s0 >= a.lim_lo + 1 && s0 - 1 < a.lim_hi) ? a.vals[s0 - 1] : 0u;
        }
        if (lane == 63) {
            K[6] = s0 + 4 < a.lim_hi ? a.keys[s0 + 4] : 0ull;
            K[7] = s0 + 5 < a.lim_hi ? a.keys[s0 + 5] : 0ull;
            V[5] = s0 + 4 < a.lim_hi ? a.vals[s0 + 4] : 0u;
        }
        // per slot j (index into K): does it exist, does it tie with the slot before; LCP with the slot before
        bool ex[8], eq[8];
        uint32_t L[8];
#pragma unroll
        for (int j = 0; j < 8; j++) { const uint64_t sj = s0 + j - 2; ex[j] = s0 + j >= a.lim_lo + 2 && sj < a.lim_hi; }
        eq[0] = false; L[0] = 0;
#pragma unroll
        for (int j = 1; j < 8; j++) {
            const bool both = ex[j - 1] && ex[j];
            eq[j] = both && K[j - 1] == K[j];
            L[j] = (both && !eq[j]) ? gr_key_lcp(K[j - 1], K[j], a.b, a.key_bits) : 0u;
        }
        // slots 1 .. 6 of K (s0 - 1 .. s0 + 4): tie, g, "the scan skips it" (regular, no tie, g < t)
        bool tie[8], skip[8];
        uint32_t g[8];
#pragma unroll
        for (int j = 1; j < 7; j++) {
            tie[j] = eq[j] || eq[j + 1];
            g[j] = max(L[j], L[j + 1]) + 1;
            skip[j] = ex[j] && !tie[j] && !(V[j - 1] >> 31) && g[j] < a.t;
        }
        uint32_t entry[4], cnt = 0;
        GrsRec rec[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int j = i + 2;
            const uint64_t sl = s0 + i;
            const bool in = sl < a.own_hi;
            if (in && !skip[j]) {
                const bool regular_kept = !tie[j] && !(V[j - 1] >> 31);          // regular with g >= t: a skipped neighbour need not be looked at
                const bool see_prev = ex[j - 1] && !(regular_kept && skip[j - 1]);
                const bool see_next = ex[j + 1] && !(regular_kept && skip[j + 1]);
                entry[cnt] = (uint32_t)sl | ((tie[j] && !eq[j]) ? GR_HEAD : 0u);   // first slot of a tie group
                rec[cnt].key = K[j]; rec[cnt].v = V[j - 1];
                rec[cnt].meta = (tie[j] ? 0u : g[j]) | (tie[j] ? GR_META_TIE : 0u) | (see_prev ? GR_META_PREV : 0u) | (see_next ? GR_META_NEXT : 0u);
                cnt++;
            }
        }
        uint32_t inc = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d, 64); if ((int)lane >= d) inc += o; }
        uint32_t at = 0;
        if (lane == 63 && inc) at = atomicAdd(&fill, inc);
        at = __shfl(at, 63, 64) + inc - cnt;
#pragma unroll
        for (int i = 0; i < 4; i++) if (i < (int)cnt) { mine[at + i] = entry[i]; mine_r[at + i] = rec[i]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) segcnt[blockIdx.x] = fill;
}

// the tie groups whose first slots k_grs_classify marked, put in text order: the marked entries of a stretch gathered
// in LDS, then one lane per group
__global__ __launch_bounds__(GR_THREADS) void k_grs_ties_listed(GrsArgs a, uint32_t *__restrict__ list, const uint32_t *__restrict__ segcnt)
{
    __shared__ uint32_t heads[GR_SEG / 2];
    __shared__ uint32_t nheads;
    if (threadIdx.x == 0) nheads = 0;
    __syncthreads();
    const uint32_t cnt = segcnt[blockIdx.x];
    uint32_t *mine = list + (uint64_t)blockIdx.x * GR_SEG;
    for (uint32_t i = threadIdx.x; i < cnt; i += GR_THREADS) {
        const uint32_t e = mine[i];
        if (e & GR_HEAD) { heads[atomicAdd(&nheads, 1u)] = e & ~GR_HEAD; mine[i] = e & ~GR_HEAD; }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nheads; i += GR_THREADS) gr_order_group(a, heads[i]);
}

// second pass: the slots whose position is in the set (LDS copy; hbits = log2 of its size)
__global__ __launch_bounds__(256) void k_grs_find(const uint32_t *__restrict__ vals, uint32_t vmask, uint64_t N, uint32_t first_slot,
                                                  const uint32_t *__restrict__ set, uint32_t hbits, uint32_t *__restrict__ list, uint32_t cap,
                                                  unsigned long long *__restrict__ counters)
{
    extern __shared__ uint32_t set_lds[];
    const uint32_t H = 1u << hbits;
    for (uint32_t i = threadIdx.x; i < H; i += 256) set_lds[i] = set[i];
    __syncthreads();
    // four values per thread and round (the loop is bound by the latency of its one load otherwise)
    const uint64_t quads = (N + 3) / 4;
    for (uint64_t q = (uint64_t)blockIdx.x * 256 + threadIdx.x; q < quads; q += (uint64_t)gridDim.x * 256) {
        uint32_t pv[4];
        if (4 * q + 3 < N && ((uintptr_t)vals & 15) == 0) { const uint4 x = reinterpret_cast<const uint4 *>(vals)[q]; pv[0] = x.x; pv[1] = x.y; pv[2] = x.z; pv[3] = x.w; }
        else for (int i = 0; i < 4; i++) pv[i] = 4 * q + i < N ? vals[4 * q + i] : GR_EMPTY;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (pv[i] == GR_EMPTY) continue;
            const uint32_t p = pv[i] & vmask;
            uint32_t h = (p * 2654435761u) >> (32 - hbits);
            for (;;) {
                const uint32_t v = set_lds[h];
                if (v == p) {
                    const unsigned long long at = atomicAdd(&counters[4], 1ull);
                    if (at < cap) list[at] = first_slot + (uint32_t)(4 * q + i);
                    break;
                }
                if (v == GR_EMPTY) break;
                h = (h + 1) & (H - 1);
            }
        }
    }
}

// the same for one slot with a long span: a wave, lanes over the columns
__global__ __launch_bounds__(256) void k_grs_long(GrsArgs a, uint32_t count)
{
    const uint32_t wv = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (wv >= count) return;
    const uint64_t s = a.longs[wv];
    const uint64_t key = a.keys[s];
    const uint32_t p = a.vals[s] & a.vmask;
    uint32_t lo, hi, row;
    GWin w;
    gr_span(a, p, lo, hi, row, w);
    const bool has_prev = s > a.lim_lo, has_next = s + 1 < a.lim_hi;
    uint64_t kp = 0, kn = 0;
    uint32_t pl = 1, ph = 0, nl = 1, nh = 0, r2;
    GWin w2;
    if (has_prev) { kp = a.keys[s - 1]; gr_span(a, a.vals[s - 1] & a.vmask, pl, ph, r2, w2); pl = max(pl, lo); ph = min(ph, hi); }
    if (has_next) { kn = a.keys[s + 1]; gr_span(a, a.vals[s + 1] & a.vmask, nl, nh, r2, w2); nl = max(nl, lo); nh = min(nh, hi); }
    const bool tie = (has_prev && kp == key) || (has_next && kn == key);
    uint32_t fi0 = 0;
    if (!tie) {
        const uint32_t lp = has_prev ? gr_key_lcp(kp, key, a.b, a.key_bits) : 0u;
        const uint32_t ln = has_next ? gr_key_lcp(key, kn, a.b, a.key_bits) : 0u;
        fi0 = gr_extent(a, p, row, max(lp, ln) + 1, key, true, w);
    }
    // (x_lo .. x_hi, slot) -> candidates, lanes over the columns; self: direct updates where no neighbour shares the column
    auto emit = [&](uint32_t l, uint32_t h, uint64_t q, bool self) {
        for (uint64_t x0 = l; x0 <= h; x0 += 64) {
            const uint64_t x = x0 + lane;
            const bool live = x <= h;
            const bool cand = live && (!self || tie || (x >= pl && x <= ph) || (x >= nl && x <= nh));
            const unsigned long long mask = __ballot(cand);
            if (mask) {
                unsigned long long base = 0;
                if (lane == 0) base = atomicAdd(&a.counters[0], (unsigned long long)__popcll(mask));
                base = __shfl(base, 0, 64);
                const unsigned long long o = base + (unsigned long long)__popcll(mask & ((1ull << lane) - 1));
                if (cand && o < a.cand_cap) a.cand[o] = (unsigned long long)x << 32 | (uint32_t)q;
            }
            if (self && live && !cand) gr_update(a, (uint32_t)x, fi0);
        }
    };
    emit(lo, hi, s, true);
    if (pl <= ph || nl <= nh) gr_walk(a, s, lo, hi, [&](uint32_t l, uint32_t h, uint64_t q) { emit(l, h, q, false); });
}

// Candidates sorted by (column, slot), possibly listed more than once: a run = consecutive slots in one column
// (fbg.cpp:1633-1641).  Every run that holds a slot the scan worked on is listed whole (gr_walk), so the slots before
// and after a listed run are not coloured.  g = 1 + max(min LCP towards the run's head, min LCP towards its tail)
// (fbg.cpp:1644-1678) in three steps: the LCPs of every entry with the slot before and after it (parallel; these are
// the dependent reads), the two running minima along each run (one thread per run head, over the arrays just made: a
// run of the row ends' 256 suffixes walked with the reads inside took 1.3 ms), the update per entry (parallel).
__global__ void k_grs_runs_lcp(GrsArgs a, uint64_t T, uint32_t *__restrict__ lcpL, uint32_t *__restrict__ lcpR)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const uint64_t s = (uint32_t)a.cand[t];
    lcpL[t] = gr_slot_lcp(a, s);
    lcpR[t] = gr_slot_lcp(a, s + 1);
}

// gq[t] (= a.pm): g of the entry, 0 for the repeats of an entry.  A run of more than GR_RUN_SERIAL entries is left to
// k_grs_runs_long (its head's index goes to `longs`: the row ends' suffixes form runs of hundreds).
#define GR_RUN_SERIAL 48
__global__ void k_grs_runs_walk(GrsArgs a, uint64_t T, const uint32_t *__restrict__ lcpL, const uint32_t *__restrict__ lcpR, uint32_t *__restrict__ longs,
                                uint32_t longs_cap)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const unsigned long long e = a.cand[t];
    if (t > 0 && (a.cand[t - 1] == e || a.cand[t - 1] + 1 == e)) return;          // a repeat, or (x, slot - 1) is listed: not a head
    uint64_t tt = t;
    uint32_t run = lcpL[t];
    for (;;) {
        a.pm[tt] = run;
        if (tt + 1 >= T) break;
        const unsigned long long nx = a.cand[tt + 1], cu = a.cand[tt];
        if (nx == cu) { tt++; continue; }
        if (nx != cu + 1) break;
        tt++;
        run = min(run, lcpL[tt]);
        if (tt - t >= GR_RUN_SERIAL) {
            const unsigned long long at = atomicAdd(&a.counters[6], 1ull);
            if (at < longs_cap) { longs[at] = (uint32_t)t; return; }
        }
    }
    uint32_t rmin = 0xffffffffu;
    for (;;) {
        rmin = min(rmin, lcpR[tt]);
        const uint32_t g = max(a.pm[tt], rmin) + 1;
        while (tt > t && a.cand[tt - 1] == a.cand[tt]) { a.pm[tt] = 0; tt--; }
        a.pm[tt] = g;
        if (tt == t) break;
        tt--;
    }
}

// one wave per long run: the same two passes, 64 entries at a time
__global__ __launch_bounds__(64) void k_grs_runs_long(GrsArgs a, uint64_t T, const uint32_t *__restrict__ lcpL, const uint32_t *__restrict__ lcpR,
                                                      const uint32_t *__restrict__ longs)
{
    const uint64_t t0 = longs[blockIdx.x];
    const uint32_t lane = threadIdx.x;
    // forward: the run's end, pm = running minimum of lcpL over the slots (a repeat shares its slot's value)
    uint64_t end = t0;                                                 // one past the run's last entry
    uint32_t carry = 0xffffffffu;
    unsigned long long prev_last = 0;
    for (uint64_t c = t0;; c += 64) {
        const uint64_t i = c + lane;
        const bool in = i < T;
        const unsigned long long e = in ? a.cand[i] : 0ull;
        unsigned long long ep = __shfl_up(e, 1, 64);
        if (lane == 0) ep = prev_last;
        const bool cont = in && (i == t0 || e == ep || e == ep + 1);
        const unsigned long long ok = __ballot(cont);
        const uint32_t len = ok == ~0ull ? 64u : (uint32_t)__ffsll((long long)~ok) - 1;      // leading entries that belong to the run
        uint32_t v = lane < len ? lcpL[i] : 0xffffffffu;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(v, d, 64); if ((int)lane >= d) v = min(v, o); }
        v = min(v, carry);
        if (lane < len) a.pm[i] = v;
        end = c + len;
        if (len < 64) break;
        carry = __shfl(v, 63, 64);
        prev_last = __shfl(e, 63, 64);
    }
    // backward: suffix minimum of lcpR; g for the first entry of every slot's group of repeats, 0 for the others
    carry = 0xffffffffu;
    for (uint64_t hi = end; hi > t0;) {
        const uint64_t lo = hi - t0 >= 64 ? hi - 64 : t0;
        const uint32_t len = (uint32_t)(hi - lo);
        const uint64_t i = lo + lane;
        uint32_t v = lane < len ? lcpR[i] : 0xffffffffu;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_down(v, d, 64); if ((int)lane + d < 64) v = min(v, o); }
        v = min(v, carry);
        if (lane < len) {
            const bool first = i == t0 || a.cand[i - 1] != a.cand[i];
            a.pm[i] = first ? max(a.pm[i], v) + 1 : 0u;
        }
        carry = __shfl(v, 0, 64);
        hi = lo;
    }
}

__global__ void k_grs_runs_apply(GrsArgs a, uint64_t T)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const uint32_t g = a.pm[t];
    if (g == 0) return;
    const unsigned long long e = a.cand[t];
    const uint32_t x = (uint32_t)(e >> 32);
    const uint64_t s = (uint32_t)e;
    const uint32_t p = a.vals[s] & a.vmask;
    uint32_t lo, hi, row;
    GWin w;
    gr_span(a, p, lo, hi, row, w);
    gr_update(a, x, gr_extent(a, p, row, g, a.keys[s], g <= (uint32_t)a.K, w));
}

// columns whose maximum a skipped slot could beat (it contributes at most x + t - 2): listed, to be redone
// (a skipped slot has K regular symbols ahead: none points at a column beyond n - K, nor -- a row start -- at column 0
// while the tricks are on)
__global__ void k_grs_unfilled(const uint32_t *__restrict__ fmax, uint64_t n, uint32_t t, uint32_t K, int disable_tricks, uint32_t *__restrict__ cols,
                               uint32_t cap, unsigned long long *__restrict__ counters)
{
    const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n || x + K > n || (x == 0 && !disable_tricks)) return;
    if ((uint64_t)max(fmax[x], (uint32_t)x) + 2 < x + t) {
        const unsigned long long at = atomicAdd(&counters[3], 1ull);
        if (at < cap) cols[at] = (uint32_t)x;
    }
}

// the pointer of every row at every listed column (first position of the row whose column is >= x; its '#' if none)
// -> hash set of positions
__global__ void k_grs_pointers(GrsArgs a, const uint32_t *__restrict__ cols, uint32_t ncols, uint32_t *__restrict__ set, uint32_t hbits)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)ncols * a.m) return;
    const uint32_t x = cols[i / a.m], row = (uint32_t)(i % a.m);
    const uint32_t p0 = a.pos[row], tt = a.tot[row];
    uint32_t p;
    if (!a.win) p = p0 + x;
    else {
        uint32_t lo = 0, hi = tt;                                     // first k in [0, tt) with col(p0 + k) >= x, else tt
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (a.colT[p0 + mid] >= x) hi = mid; else lo = mid + 1; }
        p = p0 + lo;
    }
    uint32_t h = (p * 2654435761u) >> (32 - hbits);
    for (;;) {
        const uint32_t old = atomicCAS(&set[h], GR_EMPTY, p);
        if (old == GR_EMPTY || old == p) break;
        h = (h + 1) & ((1u << hbits) - 1);
    }
}

__global__ void k_grs_finish(const uint32_t *__restrict__ fmax, uint64_t x0, uint64_t x1, uint64_t *__restrict__ out)
{
    const uint64_t x = x0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= x1) return;
    const unsigned long long fx = max((unsigned long long)x, (unsigned long long)fmax[x]);      // fbg.cpp:1618
    out[x] = max((unsigned long long)out[x], fx);                                                // fbg.cpp:1681
}

// test / debugging aid (fbg_index_download)
__global__ void k_grs_materialize(GrsArgs a, uint32_t *sa, uint32_t *isa, uint32_t *pl, uint32_t *pr)
{
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.N) return;
    const uint32_t p = a.vals[k] & a.vmask;
    sa[k] = p;
    isa[p] = (uint32_t)k;
    pl[p] = gr_slot_lcp(a, k);
    pr[p] = gr_slot_lcp(a, k + 1);
}

__global__ void k_grs_strip(uint32_t *__restrict__ vals, uint64_t N)
{
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < N) vals[k] &= 0x7fffffffu;
}

static void grs_args(fbg_ctx *ctx, GrsArgs &a, int disable_tricks)
{
    a.keys = ctx->rk_keys; a.vals = ctx->sa_ptr; a.T = ctx->text.as<uint8_t>();
    a.vmask = ctx->grs_flagged ? 0x7fffffffu : 0xffffffffu;
    a.win = ctx->gapfree ? nullptr : ctx->gwin.as<GWin>();
    a.colT = ctx->gapfree ? nullptr : ctx->colT.as<uint32_t>();
    a.pos = ctx->pos.as<uint32_t>(); a.tot = ctx->tot.as<uint32_t>();
    a.is_ignore = ctx->have_ignore ? ctx->small.as<uint8_t>() : nullptr;
    a.N = ctx->N; a.n = ctx->n; a.m = ctx->m;
    if (ctx->part_active) {
        a.own_lo = FBG_PART_HALO; a.own_hi = FBG_PART_HALO + ctx->part_count;
        a.open_lo = ctx->part > 0; a.open_hi = ctx->part + 1 < ctx->nparts;
        a.lim_lo = a.open_lo ? 0 : a.own_lo; a.lim_hi = a.open_hi ? a.own_hi + FBG_PART_HALO : a.own_hi;
    } else {
        a.own_lo = a.lim_lo = 0; a.own_hi = a.lim_hi = ctx->N;
        a.open_lo = a.open_hi = 0;
    }
    a.b = ctx->rk_b; a.K = ctx->rk_K; a.key_bits = ctx->rk_key_bits; a.disable_tricks = disable_tricks;
    a.ign_lo = ctx->grs_ign_lo; a.ign_hi = ctx->grs_ign_hi;
    a.t = 1;
    a.fmax = ctx->gmax.as<uint32_t>();
    a.counters = ctx->scalars.as<unsigned long long>() + 128;
    a.cand = nullptr; a.pm = nullptr; a.cand_cap = 0; a.longs = nullptr;
}

// Before the sort: the window table and the bitmap of irregular positions the pack kernels turn into the flag bit of
// the sort's payload (ctx->grs_ebits; nullptr: no flags -- texts of 2^31 symbols and more have no spare bit).
int fbg_grs_prepare(fbg_ctx *ctx, int *launches)
{
    ctx->grs_ebits = nullptr;
    ctx->grs_flagged = false;
    const uint64_t N = ctx->N, n = ctx->n, m = ctx->m;
    if (N >= (1ull << 32) || m >= GW_IRREGULAR) return FBG_OK;
    hipStream_t st = ctx->stream;
    const uint64_t nwin = ((N + GW - 1) >> GW_BITS) + 1;
    FBG_TRY(fbg_reserve(ctx, ctx->gbits, nwin * 16 + 64));
    if (!ctx->gapfree) {
        FBG_TRY(fbg_reserve(ctx, ctx->gwin, nwin * sizeof(GWin)));
        hipLaunchKernelGGL(k_gw_build, dim3(fbg_blocks(nwin, 16)), dim3(256), 0, st, ctx->colT.as<uint32_t>(), ctx->pos.as<uint32_t>(), N, (uint32_t)n,
                           (uint32_t)m, ctx->gwin_rows.as<uint16_t>(), ctx->gwin.as<GWin>(), ctx->gbits.as<unsigned long long>());
    } else {
        hipLaunchKernelGGL(k_grs_ebits_gapfree, dim3(fbg_blocks(nwin * 2, 256)), dim3(256), 0, st, N, n, nwin * 2, ctx->gbits.as<unsigned long long>());
    }
    *launches += 1;
    if (N < (1ull << 31) && ctx->opt.gapped_rank != 2) { ctx->grs_ebits = ctx->gbits.as<uint64_t>(); ctx->grs_flagged = true; }
    return FBG_OK;
}

// positions without the flag bit (the record path reads the values as the suffix array)
int fbg_grs_strip(fbg_ctx *ctx, uint32_t *vals)
{
    if (!ctx->grs_flagged) return FBG_OK;
    hipLaunchKernelGGL(k_grs_strip, dim3(fbg_blocks(ctx->N, 256)), dim3(256), 0, ctx->stream, vals, ctx->N);
    ctx->grs_flagged = false;
    return FBG_OK;
}

static int grs_sort_and_runs(fbg_ctx *ctx, GrsArgs &a, uint64_t T, int *launches)
{
    hipStream_t st = ctx->stream;
    unsigned long long *sorted = ctx->ps_f.as<unsigned long long>();
    int nb = 1;
    while ((1ull << nb) < ctx->n + 1) nb++;
    size_t bytes = 0;
    hipError_t e = rocprim::radix_sort_keys(nullptr, bytes, a.cand, sorted, (size_t)T, 0u, 32u + (unsigned)nb, st);
    if (e != hipSuccess) return fbg_fail(ctx, FBG_ERR_HIP, "rocprim sort size query: %s", hipGetErrorString(e));
    FBG_TRY(fbg_reserve(ctx, ctx->tmp, bytes));
    size_t have = ctx->tmp.cap;
    e = rocprim::radix_sort_keys(ctx->tmp.p, have, a.cand, sorted, (size_t)T, 0u, 32u + (unsigned)nb, st);
    if (e != hipSuccess) return fbg_fail(ctx, FBG_ERR_HIP, "rocprim radix_sort_keys: %s", hipGetErrorString(e));
    unsigned long long *unsorted = a.cand;
    uint32_t *lcpL = reinterpret_cast<uint32_t *>(unsorted), *lcpR = lcpL + a.cand_cap;     // the unsorted list is done with
    a.cand = sorted;
    hipLaunchKernelGGL(k_grs_runs_lcp, dim3(fbg_blocks(T, 256)), dim3(256), 0, st, a, T, lcpL, lcpR);
    FBG_HIP_TRY(ctx, hipMemsetAsync(a.counters + 6, 0, 8, st));
    uint32_t *long_runs = a.longs;                                     // (k_grs_long is done with the list)
    hipLaunchKernelGGL(k_grs_runs_walk, dim3(fbg_blocks(T, 64)), dim3(64), 0, st, a, T, lcpL, lcpR, long_runs, (uint32_t)GR_LONG_CAP);
    unsigned long long nlong = 0;
    FBG_HIP_TRY(ctx, hipMemcpyAsync(&nlong, a.counters + 6, 8, hipMemcpyDeviceToHost, st));
    FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
    if (nlong > GR_LONG_CAP) nlong = GR_LONG_CAP;                     // (the heads beyond the list walked their runs themselves)
    if (nlong > 0) hipLaunchKernelGGL(k_grs_runs_long, dim3((unsigned)nlong), dim3(64), 0, st, a, T, lcpL, lcpR, long_runs);
    hipLaunchKernelGGL(k_grs_runs_apply, dim3(fbg_blocks(T, 256)), dim3(256), 0, st, a, T);
    a.cand = unsorted;
    *launches += 5;
    return FBG_OK;
}

// the scratch lists of a scan over `a`'s own slots
static int grs_buffers(fbg_ctx *ctx, GrsArgs &a)
{
    const uint64_t own = a.own_hi - a.own_lo;
    a.cand_cap = std::max<uint64_t>(own / 8, 1u << 20);
    FBG_TRY(fbg_reserve(ctx, ctx->ps_e, a.cand_cap * 8));
    FBG_TRY(fbg_reserve(ctx, ctx->ps_f, a.cand_cap * 8));
    FBG_TRY(fbg_reserve(ctx, ctx->ps_g, a.cand_cap * 4));
    FBG_TRY(fbg_reserve(ctx, ctx->ps_h, (size_t)GR_LONG_CAP * 4));
    a.cand = ctx->ps_e.as<unsigned long long>();
    a.pm = ctx->ps_g.as<uint32_t>();
    a.longs = ctx->ps_h.as<uint32_t>();
    return FBG_OK;
}

// one pass over the own slots (list = nullptr), over a list, or over the stretches k_grs_classify filled (segcnt:
// `count` stretches of GR_SEG) -- plus the long spans and the candidates' runs; *ok = 0: a capacity did not hold
static int grs_pass(fbg_ctx *ctx, GrsArgs &a, const uint32_t *list, const uint32_t *segcnt, uint32_t count, int *ok, int *launches,
                    const GrsRec *recs = nullptr)
{
    *ok = 0;
    hipStream_t st = ctx->stream;
    FBG_HIP_TRY(ctx, hipMemsetAsync(a.counters, 0, 3 * sizeof(unsigned long long), st));
    if (segcnt) hipLaunchKernelGGL(k_grs_scan_seg, dim3(count), dim3(GR_THREADS), 0, st, a, list, recs, segcnt);
    else if (list) hipLaunchKernelGGL(k_grs_scan_list, dim3(fbg_blocks(count, GR_THREADS)), dim3(GR_THREADS), 0, st, a, list, count);
    else hipLaunchKernelGGL(k_grs_scan_all, dim3(fbg_blocks(a.own_hi - a.own_lo, GR_THREADS)), dim3(GR_THREADS), 0, st, a);
    unsigned long long h[3];
    FBG_HIP_TRY(ctx, hipMemcpyAsync(h, a.counters, sizeof(h), hipMemcpyDeviceToHost, st));
    FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
    *launches += 1;
    if (h[1] != 0 || h[0] > a.cand_cap) return FBG_OK;
    if (h[2] > 0) {
        hipLaunchKernelGGL(k_grs_long, dim3(fbg_blocks(h[2], 4)), dim3(256), 0, st, a, (uint32_t)h[2]);
        FBG_HIP_TRY(ctx, hipMemcpyAsync(h, a.counters, sizeof(h), hipMemcpyDeviceToHost, st));
        FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
        *launches += 1;
        if (h[1] != 0 || h[0] > a.cand_cap) return FBG_OK;
    }
    if (h[0] > 0) {
        FBG_TRY(grs_sort_and_runs(ctx, a, h[0], launches));
        FBG_HIP_TRY(ctx, hipMemcpyAsync(h, a.counters, sizeof(h), hipMemcpyDeviceToHost, st));     // (a run beyond a partition's halo)
        FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
        if (h[1] != 0) return FBG_OK;
    }
    *ok = 1;
    return FBG_OK;
}

// the threshold from a sample of the own slots: at least ln(n) + 10 slots per column expected at or above it (a column
// without any is redone afterwards); 1: none
static int grs_pick_threshold(fbg_ctx *ctx, GrsArgs &a, uint32_t *t, int *launches)
{
    *t = 1;
    hipStream_t st = ctx->stream;
    const uint64_t own = a.own_hi - a.own_lo, N = ctx->N, n = ctx->n;
    if (!ctx->grs_flagged || ctx->opt.gapped_rank == 3 || !(own > 4 * (uint64_t)GR_SAMPLE || ctx->opt.gapped_rank == 4) || own < 4) return FBG_OK;
    unsigned long long *hist = a.counters + 8;
    FBG_HIP_TRY(ctx, hipMemsetAsync(hist, 0, 66 * sizeof(unsigned long long), st));
    const uint64_t stride = std::max<uint64_t>(1, own / GR_SAMPLE);
    hipLaunchKernelGGL(k_grs_sample, dim3(GR_SAMPLE / 256), dim3(256), 0, st, a, stride, hist);
    unsigned long long hh[66];
    FBG_HIP_TRY(ctx, hipMemcpyAsync(hh, hist, sizeof(hh), hipMemcpyDeviceToHost, st));
    FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
    *launches += 1;
    unsigned long long total = 0;
    for (int g = 0; g < 66; g++) total += hh[g];
    double need = (log((double)n) + 10.0) / ((double)N / (double)n);             // fraction of ALL the slots (every partition has its share)
    if (ctx->opt.gapped_rank == 4) need = 0.05;                                    // tests: skip a lot whatever the shape
    unsigned long long above = 0;
    if (total > 0 && need < 0.5)
        for (int g = 65; g >= 2; g--) {
            above += hh[g];
            if ((double)above >= need * (double)total) { *t = (uint32_t)g; break; }
        }
    return FBG_OK;
}

// the main pass with threshold a.t: classify + the listed slots, or every slot.  order_ties: the tie groups are not in
// text order yet (the classification marks their first slots; without one, a pass of its own finds them).
static int grs_main_pass(fbg_ctx *ctx, GrsArgs &a, bool order_ties, int *ok, int *launches)
{
    hipStream_t st = ctx->stream;
    *ok = 0;
    auto ties_fit = [&](bool *fit) -> int {        // a tie group of more than GR_MAX_TIE suffixes raises counters[1]
        unsigned long long flag = 0;
        FBG_HIP_TRY(ctx, hipMemcpyAsync(&flag, a.counters + 1, 8, hipMemcpyDeviceToHost, st));
        FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
        *fit = flag == 0;
        return FBG_OK;
    };
    bool fit = true;
    if (a.t > 1) {
        const uint32_t nseg = fbg_blocks(a.own_hi - a.own_lo, GR_SEG);
        FBG_TRY(fbg_reserve(ctx, ctx->grp, (size_t)nseg * GR_SEG * 4));
        FBG_TRY(fbg_reserve(ctx, ctx->flags, (size_t)nseg * 4));
        FBG_TRY(fbg_reserve(ctx, ctx->msd_w, (size_t)nseg * GR_SEG * sizeof(GrsRec)));
        GrsRec *recs = ctx->msd_w.as<GrsRec>();
        hipLaunchKernelGGL(k_grs_classify, dim3(nseg), dim3(GR_THREADS), 0, st, a, ctx->grp.as<uint32_t>(), recs, ctx->flags.as<uint32_t>());
        *launches += 1;
        if (order_ties) {
            hipLaunchKernelGGL(k_grs_ties_listed, dim3(nseg), dim3(GR_THREADS), 0, st, a, ctx->grp.as<uint32_t>(), ctx->flags.as<uint32_t>());
            *launches += 1;
            FBG_TRY(ties_fit(&fit));
        }
        if (!fit) return FBG_OK;
        return grs_pass(ctx, a, ctx->grp.as<uint32_t>(), ctx->flags.as<uint32_t>(), nseg, ok, launches, recs);
    }
    if (order_ties) {
        hipLaunchKernelGGL(k_grs_ties, dim3(fbg_blocks(a.own_hi - a.own_lo, 256)), dim3(256), 0, st, a);
        *launches += 1;
        FBG_TRY(ties_fit(&fit));
        if (!fit) return FBG_OK;
    }
    return grs_pass(ctx, a, nullptr, nullptr, 0, ok, launches);
}

// columns a skipped slot could still beat, given the largest threshold t any scan used -> ctx->xlist, *nc of them
static int grs_unfilled(fbg_ctx *ctx, GrsArgs &a, uint32_t t, uint64_t *nc)
{
    *nc = 0;
    if (t <= 1) return FBG_OK;
    hipStream_t st = ctx->stream;
    const uint64_t n = ctx->n;
    const uint32_t cap = (uint32_t)std::min<uint64_t>(n, 1u << 20);
    FBG_TRY(fbg_reserve(ctx, ctx->xlist, (n + 1) * 4));
    FBG_HIP_TRY(ctx, hipMemsetAsync(a.counters + 3, 0, 8, st));
    hipLaunchKernelGGL(k_grs_unfilled, dim3(fbg_blocks(n, 256)), dim3(256), 0, st, a.fmax, n, t, (uint32_t)a.K, a.disable_tricks, ctx->xlist.as<uint32_t>(),
                       cap, a.counters);
    unsigned long long h = 0;
    FBG_HIP_TRY(ctx, hipMemcpyAsync(&h, a.counters + 3, 8, hipMemcpyDeviceToHost, st));
    FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
    *nc = h;
    return FBG_OK;
}

// the nc columns of ctx->xlist exactly, on top of the maxima in ctx->gmax: their rows' pointers -> hash set -> the own
// slots that hold them -> the treatment of the main pass for those; too many for the set: every own slot again
static int grs_redo(fbg_ctx *ctx, GrsArgs &a, uint64_t nc, int *ok, int *launches)
{
    *ok = 1;
    if (nc == 0) return FBG_OK;
    hipStream_t st = ctx->stream;
    const uint64_t own = a.own_hi - a.own_lo;
    if (nc * ctx->m > GR_HASH_FILL) {
        a.t = 1;
        return grs_pass(ctx, a, nullptr, nullptr, 0, ok, launches);
    }
    uint32_t hbits = 10;
    while ((1ull << hbits) < 2 * nc * ctx->m) hbits++;
    const uint32_t lcap = (uint32_t)(nc * ctx->m);
    FBG_TRY(fbg_reserve(ctx, ctx->list, (size_t)GR_HASH * 4));
    FBG_TRY(fbg_reserve(ctx, ctx->tie_list, (size_t)lcap * 4));
    uint32_t *set = ctx->list.as<uint32_t>(), *found = ctx->tie_list.as<uint32_t>();
    FBG_HIP_TRY(ctx, hipMemsetAsync(set, 0xff, (size_t)4 << hbits, st));
    FBG_HIP_TRY(ctx, hipMemsetAsync(a.counters + 4, 0, 8, st));
    hipLaunchKernelGGL(k_grs_pointers, dim3(fbg_blocks(nc * ctx->m, 256)), dim3(256), 0, st, a, ctx->xlist.as<uint32_t>(), (uint32_t)nc, set, hbits);
    const size_t lds = (size_t)4 << hbits;
    if (lds > 48 * 1024) FBG_HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_grs_find, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_grs_find, dim3(fbg_blocks(own, 256, 2048)), dim3(256), lds, st, a.vals + a.own_lo, a.vmask, own, (uint32_t)a.own_lo, set, hbits,
                       found, lcap, a.counters);
    unsigned long long nf = 0;
    FBG_HIP_TRY(ctx, hipMemcpyAsync(&nf, a.counters + 4, 8, hipMemcpyDeviceToHost, st));
    FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
    *launches += 2;
    if (nf > lcap) return fbg_fail(ctx, FBG_ERR_HIP, "gapped rank scan: %llu slots for %u row pointers", nf, lcap);
    if (nf > 0) FBG_TRY(grs_pass(ctx, a, found, nullptr, (uint32_t)nf, ok, launches));
    return FBG_OK;
}

// the scan proper on the kept slots of a whole index, for one setting of the elastic tricks; *ok = 0: a capacity did not hold
static int grs_scan(fbg_ctx *ctx, int disable_tricks, int *ok, int *launches)
{
    *ok = 0;
    GrsArgs a;
    grs_args(ctx, a, disable_tricks);
    FBG_HIP_TRY(ctx, hipMemsetAsync(ctx->gmax.p, 0, (ctx->n + 1) * 4, ctx->stream));
    FBG_TRY(grs_buffers(ctx, a));
    FBG_TRY(grs_pick_threshold(ctx, a, &a.t, launches));
    ctx->grs_t = a.t;
    ctx->grs_redone = 0;
    FBG_TRY(grs_main_pass(ctx, a, !ctx->grs_ties_done, ok, launches));
    if (!*ok) return FBG_OK;
    ctx->grs_ties_done = true;
    uint64_t nc = 0;
    FBG_TRY(grs_unfilled(ctx, a, a.t, &nc));
    ctx->grs_redone = nc;
    FBG_TRY(grs_redo(ctx, a, nc, ok, launches));
    if (!*ok) return FBG_OK;
    FBG_HIP_TRY(ctx, hipGetLastError());
    ctx->grs_tricks_off = disable_tricks;
    return FBG_OK;
}

// the ignore characters as a mask over symbol codes (the key's alphabet: ranks of the bytes that occur); false: not for this scan
static bool grs_ignore_mask(fbg_ctx *ctx, uint64_t *by_code)
{
    *by_code = 0;
    if (!ctx->have_ignore) return true;
    int code = 0;
    for (int c = 0; c < 256; c++) {
        const bool occurs = ctx->byte_hist[c] != 0;
        if (ctx->ignore_tab[c]) {
            if (c == '-') return false;                                // gap cells that clamp: the record path's per-cell table
            if (occurs) { if (code >= 64) return false; *by_code |= 1ull << code; }
        }
        if (occurs) code++;
    }
    return true;
}

static void grs_remember(fbg_ctx *ctx, uint64_t *keys, uint32_t *vals, const KeyGeom &g, uint64_t by_code)
{
    ctx->rk_keys = keys; ctx->sa_ptr = vals;
    ctx->rk_layout = FBG_SLOTS_PAIRS; ctx->rk_pb = 0; ctx->rk_b = g.b; ctx->rk_key_bits = g.key_bits; ctx->rk_K = g.K;
    ctx->grs_ign_lo = (uint32_t)by_code; ctx->grs_ign_hi = (uint32_t)(by_code >> 32);
}

// Called by fbg_suffix_sort after the round-0 sort of the (key, position) pairs of an MSA with gaps / ignore characters
// (fbg_grs_prepare ran before it).  *done = 1: the index is the sorted slots plus the per-column maxima
// (ctx->granked); 0: continue with the record path.
int fbg_grs_try(fbg_ctx *ctx, uint64_t *keys, uint32_t *vals, const KeyGeom &g, int *done)
{
    *done = 0;
    ctx->granked = false;
    const uint64_t N = ctx->N, n = ctx->n, m = ctx->m;
    if (N >= (1ull << 32) || m >= GW_IRREGULAR || g.compact || g.packed || g.wide || ctx->reversed) return FBG_OK;
    hipStream_t st = ctx->stream;
    uint64_t by_code = 0;
    if (!grs_ignore_mask(ctx, &by_code)) return FBG_OK;
    FBG_TRY(fbg_stage_begin(ctx, FBG_STAGE_RANKSCAN));
    int launches = 0;
    FBG_TRY(fbg_reserve(ctx, ctx->gmax, (n + 1) * 4));
    grs_remember(ctx, keys, vals, g, by_code);
    GrsArgs a;
    grs_args(ctx, a, 0);
    FBG_HIP_TRY(ctx, hipMemsetAsync(a.counters, 0, 6 * sizeof(unsigned long long), st));
    ctx->grs_ties_done = false;                    // (the tie groups are put in text order by the first scan)
    int ok = 0;
    FBG_TRY(grs_scan(ctx, 0, &ok, &launches));
    if (ok) {
        ctx->granked = true;
        ctx->ranked = false;
        ctx->part_active = false;
        *done = 1;
    }
    return fbg_stage_end(ctx, FBG_STAGE_RANKSCAN, launches);
}

// ---- key-range partitions of a multi-GPU job (suffix_sort.hip, fbg_part_sort; the contract: include/fbg_hip.h) --------
// keys / vals: FBG_PART_HALO + count + FBG_PART_HALO slots, the middle sorted.  Equal keys never straddle partitions,
// so tie groups are whole; what a partition cannot see is a run of coloured slots that goes on beyond the 64 edge slots
// its neighbours publish (declined: every rank falls back together).
__global__ void k_grs_halo_export(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ vals, uint64_t own_lo, uint64_t own_hi, uint64_t word,
                                  uint8_t *__restrict__ blob)
{
    uint64_t *bk = reinterpret_cast<uint64_t *>(blob);
    uint32_t *bv = reinterpret_cast<uint32_t *>(blob + 2 * FBG_PART_HALO * 8);
    uint64_t *tail = reinterpret_cast<uint64_t *>(blob + 2 * FBG_PART_HALO * 12);
    const uint32_t t = threadIdx.x;                    // 2 * FBG_PART_HALO threads: head slots, then tail slots
    if (word & 1) {
        const uint64_t k = t < FBG_PART_HALO ? own_lo + t : own_hi - 2 * FBG_PART_HALO + t;
        bk[t] = keys[k]; bv[t] = vals[k];
    } else { bk[t] = 0; bv[t] = 0; }
    if (t == 0) { tail[0] = word; tail[1] = own_hi - own_lo; }
}

__global__ void k_grs_halo_import(const uint8_t *__restrict__ blobs, int part, int nparts, uint64_t own_hi, uint64_t *__restrict__ keys,
                                  uint32_t *__restrict__ vals)
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

// Phase 1 (fbg_part_index_build): the tie groups of the owned slots in text order, the threshold from a sample of them,
// the edge slots published (d_blob; its tail word: bit 0 = this partition can go on, bits 8.. = its threshold).
int fbg_grs_part_classify(fbg_ctx *ctx, uint64_t *keys, uint32_t *vals, uint64_t count, const KeyGeom &g, int eligible, uint8_t *d_blob, int *ok)
{
    hipStream_t st = ctx->stream;
    FBG_TRY(fbg_stage_begin(ctx, FBG_STAGE_RANKSCAN));
    int launches = 0;
    ctx->granked = false; ctx->ranked = false; ctx->gpart = true;
    ctx->part_count = count;
    uint64_t by_code = 0;
    int good = eligible && count >= 2 * FBG_PART_HALO && ctx->N < (1ull << 32) && ctx->m < GW_IRREGULAR && grs_ignore_mask(ctx, &by_code);
    FBG_TRY(fbg_reserve(ctx, ctx->gmax, (ctx->n + 1) * 4));
    grs_remember(ctx, keys, vals, g, by_code);
    GrsArgs a;
    grs_args(ctx, a, ctx->opt.part_tricks_off ? 1 : 0);
    uint32_t t = 1;
    FBG_HIP_TRY(ctx, hipMemsetAsync(a.counters, 0, 6 * sizeof(unsigned long long), st));
    if (good) {
        hipLaunchKernelGGL(k_grs_ties, dim3(fbg_blocks(count, 256)), dim3(256), 0, st, a);
        launches++;
        unsigned long long flag = 0;
        FBG_HIP_TRY(ctx, hipMemcpyAsync(&flag, a.counters + 1, 8, hipMemcpyDeviceToHost, st));
        FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
        if (flag != 0) good = 0;
    }
    if (good) FBG_TRY(grs_pick_threshold(ctx, a, &t, &launches));
    ctx->grs_t = t;
    hipLaunchKernelGGL(k_grs_halo_export, dim3(1), dim3(2 * FBG_PART_HALO), 0, st, keys, vals, a.own_lo, a.own_hi, (uint64_t)good | ((uint64_t)t << 8), d_blob);
    launches++;
    FBG_HIP_TRY(ctx, hipGetLastError());
    *ok = good;
    return fbg_stage_end(ctx, FBG_STAGE_RANKSCAN, launches);
}

// Phase 2 (fbg_part_scan): halos in, the scan of the owned slots, their column maxima out (d_gmax: n + 1 words; word n =
// 1 when a partition declined, so that the max-reduction carries the verdict).
int fbg_grs_part_scan(fbg_ctx *ctx, const uint8_t *d_blobs, uint32_t *d_gmax, int *ok)
{
    const uint64_t n = ctx->n;
    hipStream_t st = ctx->stream;
    FBG_TRY(fbg_stage_begin(ctx, FBG_STAGE_RANKSCAN));
    int launches = 0;
    std::vector<uint8_t> hb((size_t)ctx->nparts * FBG_PART_HALO_BYTES);
    FBG_HIP_TRY(ctx, hipMemcpyAsync(hb.data(), d_blobs, hb.size(), hipMemcpyDeviceToHost, st));
    FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
    int good = 1;
    uint32_t t_all = 1;
    for (int p = 0; p < ctx->nparts; p++) {
        uint64_t tail[2];
        memcpy(tail, hb.data() + (size_t)p * FBG_PART_HALO_BYTES + 2 * FBG_PART_HALO * 12, sizeof(tail));
        if (!(tail[0] & 1)) good = 0;
        t_all = std::max(t_all, (uint32_t)(tail[0] >> 8));
    }
    FBG_HIP_TRY(ctx, hipMemsetAsync(ctx->gmax.p, 0, (n + 1) * 4, st));
    if (good) {
        GrsArgs a;
        grs_args(ctx, a, ctx->opt.part_tricks_off ? 1 : 0);
        hipLaunchKernelGGL(k_grs_halo_import, dim3(1), dim3(FBG_PART_HALO), 0, st, d_blobs, ctx->part, ctx->nparts, a.own_hi, ctx->rk_keys, a.vals);
        launches++;
        FBG_TRY(grs_buffers(ctx, a));
        a.t = ctx->grs_t;
        FBG_TRY(grs_main_pass(ctx, a, false, &good, &launches));
        ctx->grs_tricks_off = a.disable_tricks;
    }
    ctx->part_gmin = t_all;         // the unfilled test of fbg_part_finish goes by the largest threshold any partition used
    FBG_HIP_TRY(ctx, hipMemcpyAsync(d_gmax, ctx->gmax.p, (n + 1) * 4, hipMemcpyDeviceToDevice, st));
    const uint32_t verdict = good ? 0u : 1u;
    FBG_HIP_TRY(ctx, hipMemcpyAsync(d_gmax + n, &verdict, 4, hipMemcpyHostToDevice, st));
    FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
    *ok = good;
    return fbg_stage_end(ctx, FBG_STAGE_RANKSCAN, launches);
}

// after the max-reduction (ctx->gmax holds the reduced maxima): the columns that have to be redone (the same list on every rank)
int fbg_grs_part_unfilled(fbg_ctx *ctx, uint64_t *unfilled)
{
    GrsArgs a;
    grs_args(ctx, a, ctx->grs_tricks_off);
    FBG_TRY(grs_unfilled(ctx, a, ctx->part_gmin, unfilled));
    ctx->grs_redone = *unfilled;
    return FBG_OK;
}

// those columns exactly from this partition's slots, on top of the reduced maxima in ctx->gmax
int fbg_grs_part_rescan(fbg_ctx *ctx)
{
    FBG_TRY(fbg_stage_begin(ctx, FBG_STAGE_RANKSCAN));
    int launches = 0, ok = 1;
    GrsArgs a;
    grs_args(ctx, a, ctx->grs_tricks_off);
    FBG_TRY(grs_buffers(ctx, a));
    a.t = ctx->grs_t;
    FBG_TRY(grs_redo(ctx, a, ctx->grs_redone, &ok, &launches));
    ctx->part_gmin = 0;
    ctx->grs_part_failed = !ok;
    return fbg_stage_end(ctx, FBG_STAGE_RANKSCAN, launches);
}

// f for columns [x0, x1) from the per-column maxima; the scan is redone first when it ran for the other setting of the
// tricks.  *ok = 0: that second scan ran out of room -- the caller rebuilds the index the record way.
int fbg_grs_finish(fbg_ctx *ctx, uint64_t x0, uint64_t x1, int disable_tricks, uint64_t *d_out, int *ok)
{
    *ok = 1;
    if ((disable_tricks != 0) != (ctx->grs_tricks_off != 0)) {
        if (ctx->gpart)
            return fbg_fail(ctx, FBG_ERR_INVALID, "the partitioned index of this MSA (gaps / ignore characters) was scanned %s the elastic tricks; "
                            "set option part_tricks_off before fbg_part_index_build for the other setting", ctx->grs_tricks_off ? "without" : "with");
        int launches = 0;
        if (ctx->spanned) FBG_TRY(fbg_span_rescan(ctx, disable_tricks ? 1 : 0, ok));    // (similar rows: the group-level scan, span_scan.hip)
        else FBG_TRY(grs_scan(ctx, disable_tricks ? 1 : 0, ok, &launches));
        if (!*ok) return FBG_OK;
    }
    hipLaunchKernelGGL(k_grs_finish, dim3(fbg_blocks(x1 - x0, 256)), dim3(256), 0, ctx->stream, ctx->gmax.as<uint32_t>(), x0, x1, d_out);
    FBG_HIP_TRY(ctx, hipGetLastError());
    return FBG_OK;
}

int fbg_grs_materialize(fbg_ctx *ctx, uint32_t *d_sa, uint32_t *d_isa, uint32_t *d_pl, uint32_t *d_pr)
{
    GrsArgs a;
    grs_args(ctx, a, ctx->grs_tricks_off);
    hipLaunchKernelGGL(k_grs_materialize, dim3(fbg_blocks(ctx->N, 256)), dim3(256), 0, ctx->stream, a, d_sa, d_isa, d_pl, d_pr);
    FBG_HIP_TRY(ctx, hipGetLastError());
    return FBG_OK;
}
