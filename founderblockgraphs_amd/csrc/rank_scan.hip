// rank_scan.hip -- the extension scan done in suffix-array order (gap-free MSAs, no ignore characters).
//
// Coloured ranks that are consecutive integers (fbg.cpp:1633-1641) are, in suffix-array order, simply
// neighbouring slots whose positions lie in the same MSA column (position mod (n+1)).  So the whole scan of
// fbg.cpp:1610-1694 can be done on the SORTED round-0 keys, without the inverse permutation:
//
//   k_rank_scan     per SA slot: its two neighbour LCPs are the numbers of equal leading symbols of its key and
//                   the neighbours' keys.  A slot whose neighbours are in other columns is a run of length one:
//                   g = 1 + max(LCP[r], LCP[r+1]) goes into the column's maximum (table read first, atomicMax
//                   only when it grows; extensions too small to be a column maximum -- judged from a sampled
//                   histogram and verified afterwards -- skip the table altogether).
//                   Suffixes that tie on the whole key: a small group none of whose members shares a column
//                   with another member or with the slots next to the group needs no ordering at all -- the
//                   extension of a member is 1 + its longest match with any other member (k_tie_simple).
//                   Everything else -- slots with a same-column neighbour, larger or entangled tie groups --
//                   goes to a (short) candidate list.
//   k_tie_simple    the small tie groups: all pairs compared beyond the key, straight into the column maxima.
//   k_tie_groups    candidate tie groups are ordered by text comparison: final SA order.
//   k_runs          candidates, in final order: every maximal run of same-column neighbouring slots gets
//                   g = 1 + max(min LCP towards the run head, min LCP towards the run tail) per member
//                   (fbg.cpp:1644-1678) by one forward and one backward walk.
//   k_rank_finish   f[x] / v[j] from the column maxima (fbg.cpp:1656-1672 with rank_i(x) = x, tot_i = n).
//
// Keys use the compact coding of suffix_sort.hip: a separator ('#', sentinel) and everything behind it inside a
// key count as code 0.  A key of a suffix with fewer than K symbols left in its row ("short") is therefore the
// smallest key with its real symbols: the suffix sorts correctly up to ties, and the number of equal leading
// symbols of two different keys is their LCP once clamped to the symbols both rows really have left
// (rem = n - column, row arithmetic).  Tie groups with a short member are ordered by comparing the text from
// the suffix start.  Slots are either (key, position) pairs or one packed word, key << pb | position.
//
// Falls back to the record path (suffix_sort.hip doubling + scan.hip) when a quarter of the slots tie
// (similar rows), a workgroup's candidate region overflows, or a tie group exceeds 64 members.
#include <algorithm>
#include <cstring>
#include <vector>
#include "fbg_internal.h"
#include "text_cmp.h"
#include <rocprim/rocprim.hpp>

#include "rank_common.h"

// k_rank_scan stages a chunk of 256 slots (+ RS_HALO either side) in LDS as (key, column): all the looking around
// that tie groups need happens there.
#define RS_ITEMS 4
#define RS_THREADS 256
#define RS_CHUNK (RS_THREADS * RS_ITEMS)

// A few thousand workgroups, each walking chunks of RS_CHUNK slots of the SA; a thread owns RS_ITEMS consecutive
// slots.  Per chunk: (1) keys and columns into LDS -- the global loads of the NEXT chunk are issued right away, so
// their latency hides behind this chunk's work; (2) every slot that neither ties nor sits next to a tie group is
// finished on the spot from registers (its neighbours are the thread's own slots or the adjacent lanes'), the
// others queue up; (3) the queue is worked off, out of LDS, by as many lanes as it has entries instead of a few
// lanes in every wave.
// PLAIN = false: the threshold lies above K, nothing that does not tie can be a column maximum -- except in the 64
// columns nearest a row end, which are never thresholded: those slots take the general path, the rest only classify.
template <int L, bool PLAIN> __global__ __launch_bounds__(RS_THREADS) __attribute__((amdgpu_waves_per_eu(7, 8))) void k_rank_scan(RankArgs a)
{
    __shared__ uint64_t skey[RS_CHUNK + 2 * RS_HALO];
    __shared__ uint32_t srem[RS_CHUNK + 2 * RS_HALO];   // symbols left in the row: names the column as well
    __shared__ uint16_t sq[RS_CHUNK];
    __shared__ uint32_t sqn, s_cand_n, s_tie_n;
    const int lane = threadIdx.x & 63;
    const uint64_t nchunks = (a.own_hi - a.own_lo + RS_CHUNK - 1) / RS_CHUNK;
    if (threadIdx.x == 0) { s_cand_n = 0; s_tie_n = 0; }
    // registers for one chunk: the thread's RS_ITEMS slots, plus one of the 2 * RS_HALO extra entries (threads 0..15)
    uint64_t w[RS_ITEMS + 1];
    uint32_t v[RS_ITEMS + 1];
    const int my_i = RS_HALO + RS_ITEMS * (int)threadIdx.x;            // LDS index of the thread's first slot
    const int halo_i = (int)threadIdx.x < RS_HALO ? (int)threadIdx.x : RS_CHUNK + (int)threadIdx.x;   // threads < 2 * RS_HALO
    auto fetch = [&](uint64_t c) {
        const uint64_t base = a.own_lo + c * RS_CHUNK;                 // slot of LDS index RS_HALO
        const int lo_i = c == 0 ? RS_HALO : 0;
        const int hi_i = (int)min((uint64_t)(RS_CHUNK + 2 * RS_HALO), a.own_hi - base + RS_HALO);
#pragma unroll
        for (int r = 0; r < RS_ITEMS; r++) {
            w[r] = 0; v[r] = 0;
            if (my_i + r < hi_i) {
                w[r] = a.keys[base + (uint64_t)(my_i + r - RS_HALO)];
                if (L != FBG_SLOTS_PACKED) v[r] = a.vals[base + (uint64_t)(my_i + r - RS_HALO)];
            }
        }
        w[RS_ITEMS] = 0; v[RS_ITEMS] = 0;
        if (threadIdx.x < 2 * RS_HALO && halo_i >= lo_i && halo_i < hi_i) {
            w[RS_ITEMS] = a.keys[base + halo_i - RS_HALO];
            if (L != FBG_SLOTS_PACKED) v[RS_ITEMS] = a.vals[base + halo_i - RS_HALO];
        }
    };
    if (blockIdx.x < nchunks) fetch(blockIdx.x);
    for (uint64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const uint64_t base = a.own_lo + c * RS_CHUNK;
        const int lo_i = c == 0 ? RS_HALO : 0;                         // LDS index of the first / one past the last owned slot
        const int hi_i = (int)min((uint64_t)(RS_CHUNK + 2 * RS_HALO), a.own_hi - base + RS_HALO);
        if (threadIdx.x == 0) sqn = 0;
        uint64_t kk[RS_ITEMS + 6];                                     // keys of slots my_i - 3 .. my_i + RS_ITEMS + 2
        uint32_t cc[RS_ITEMS + 4];                                     // columns of slots my_i - 2 .. my_i + RS_ITEMS + 1
#pragma unroll
        for (int r = 0; r < RS_ITEMS; r++) {
            uint64_t key = 0;
            uint32_t rem = 0;
            if (my_i + r < hi_i) {
                key = w[r] >> a.pb;
                rem = rs_rem<L>(a, rs_pos_of<L>(a, w[r], v[r]));
            }
            kk[r + 3] = key; cc[r + 2] = rem;
            skey[my_i + r] = key; srem[my_i + r] = rem;
        }
        if (threadIdx.x < 2 * RS_HALO) {
            uint64_t key = 0;
            uint32_t rem = 0;
            if (halo_i >= lo_i && halo_i < hi_i) {
                key = w[RS_ITEMS] >> a.pb;
                rem = rs_rem<L>(a, rs_pos_of<L>(a, w[RS_ITEMS], v[RS_ITEMS]));
            }
            skey[halo_i] = key; srem[halo_i] = rem;
        }
        __syncthreads();
        if (c + gridDim.x < nchunks) fetch(c + gridDim.x);
        // the neighbours beyond the thread's own slots: adjacent lanes, LDS at the wave's edges
#pragma unroll
        for (int j = 0; j < 3; j++) {
            kk[j] = __shfl_up(kk[RS_ITEMS + j], 1, 64);                // lane - 1's last three slots
            kk[RS_ITEMS + 3 + j] = __shfl_down(kk[3 + j], 1, 64);      // lane + 1's first three
        }
#pragma unroll
        for (int j = 0; j < 2; j++) {
            cc[j] = __shfl_up(cc[RS_ITEMS + j], 1, 64);
            cc[RS_ITEMS + 2 + j] = __shfl_down(cc[2 + j], 1, 64);
        }
        if (lane == 0) {
#pragma unroll
            for (int j = 0; j < 3; j++) kk[j] = skey[my_i - 3 + j];
            cc[0] = srem[my_i - 2]; cc[1] = srem[my_i - 1];
        }
        if (lane == 63) {
#pragma unroll
            for (int j = 0; j < 3; j++) kk[RS_ITEMS + 3 + j] = skey[my_i + RS_ITEMS + j];
            cc[RS_ITEMS + 2] = srem[my_i + RS_ITEMS]; cc[RS_ITEMS + 3] = srem[my_i + RS_ITEMS + 1];
        }
        uint32_t upd_g[RS_ITEMS], upd_c[RS_ITEMS];
#pragma unroll
        for (int r = 0; r < RS_ITEMS; r++) { upd_g[r] = 0; upd_c[r] = 0; }
        uint32_t slow4 = 0, tie4 = 0, cand4 = 0;
        uint32_t lcp[RS_ITEMS + 1];                                    // key LCP of slots my_i + r - 1 and my_i + r (K for equal keys)
#pragma unroll
        for (int r = 0; r <= RS_ITEMS; r++) lcp[r] = PLAIN ? rs_key_lcp(kk[r + 2], kk[r + 3], a.b, a.key_bits) : 0u;
#pragma unroll
        for (int r = 0; r < RS_ITEMS; r++) {
            // slot i with keys K[-3..3] = kk[r .. r+6] and columns C[-2..2] = cc[r .. r+4]
            const int i = my_i + r;
            const bool in = i < hi_i;
            const bool window = i - 3 >= lo_i && i + 3 < hi_i;         // all of it inside the owned range
            const uint64_t key = kk[r + 3];
            const uint32_t rem = cc[r + 2];
            const bool eqp = kk[r + 2] == key, eqn = kk[r + 4] == key;
            bool slow = in && !window;                                 // range ends: the general code sorts it out
            bool want_cand = false, want_cand2 = false, want_tie = false;
            if (window) {
                const bool prev_tie = kk[r + 1] == kk[r + 2], next_tie = kk[r + 5] == kk[r + 4];    // the neighbour ties with ITS neighbour
                if (!eqp && !eqn) {
                    if (rem != 0) {
                        // both LCPs come from the keys, unless a neighbour shares the column (a run).  A neighbour that
                        // is half of a tied pair: either half may end up next to this slot
                        if ((prev_tie && kk[r] == kk[r + 1]) || (next_tie && kk[r + 6] == kk[r + 5])) slow = true;   // longer groups
                        else {
                            bool run = cc[r + 1] == rem || cc[r + 3] == rem;
                            uint32_t mrp = cc[r + 1], mrn = cc[r + 3];
                            if (prev_tie) { run = run || cc[r] == rem; mrp = max(mrp, cc[r]); }
                            if (next_tie) { run = run || cc[r + 4] == rem; mrn = max(mrn, cc[r + 4]); }
                            if (run) want_cand = true;
                            else if (PLAIN) {
                                const uint32_t g = max(min(lcp[r], min(rem, mrp)), min(lcp[r + 1], min(rem, mrn))) + 1;
                                // near the end of a row few suffixes compete and extensions stay short: no threshold there
                                if (g >= a.g_min || rem <= 64) { upd_g[r] = g; upd_c[r] = rs_col_of_rem(a, rem); }
                            } else if (rem <= 64) slow = true;
                        }
                    }
                } else if (!eqp && kk[r + 5] != key && i + 4 < hi_i) {
                    // head of a tied pair (slots i, i + 1): settled here.  Simple = both with K real symbols, the two
                    // columns and those of the slots before and after all different, no tie group right next to it
                    const uint32_t r1 = cc[r + 3], rb = cc[r + 1], ra = cc[r + 4];
                    const bool simple = rem >= (uint32_t)a.K && r1 >= (uint32_t)a.K && rem != r1 &&
                                        !prev_tie && rb != rem && rb != r1 &&
                                        kk[r + 6] != kk[r + 5] && ra != rem && ra != r1;
                    if (simple) want_tie = true;
                    else { want_cand = true; want_cand2 = true; }
                } else if (!(eqp && !eqn && kk[r + 1] != key && i - 4 >= lo_i)) slow = true;    // not the tail of a pair settled by its head
            }
            // what the slot asks for is only noted here: one bit per slot of the thread and kind
            slow4 |= (slow ? 1u : 0u) << r;
            tie4 |= (want_tie ? 1u : 0u) << r;
            cand4 |= ((want_cand ? 1u : 0u) << r) | ((want_cand2 ? 16u : 0u) << r);
        }
        // The appends, once per chunk instead of once per slot row: a round takes one noted slot of every thread that
        // still has one (most threads have none, few have two of a kind), so the wave-wide ballots and the branches
        // around them -- scalar work that rivalled the vector work of this kernel -- shrink to one or two rounds per kind.
        for (uint32_t bits = slow4; __ballot(bits != 0);) {
            const bool on = bits != 0;
            const int rr = on ? __ffs((int)bits) - 1 : 0;
            bits &= bits - 1;
            const unsigned long long smask = __ballot(on);
            uint32_t qb = 0;
            const int leader = __ffsll((long long)smask) - 1;
            if (lane == leader) qb = atomicAdd(&sqn, (uint32_t)__popcll(smask));
            qb = __shfl(qb, leader, 64);
            if (on) sq[qb + (uint32_t)__popcll(smask & ((1ull << lane) - 1))] = (uint16_t)(my_i + rr);
        }
        if (!a.values_only) {
            const uint32_t k_first = (uint32_t)(base + (uint64_t)(my_i - RS_HALO));      // slot of the thread's first item
            for (uint32_t bits = tie4; __ballot(bits != 0);) {
                const bool on = bits != 0;
                const int rr = on ? __ffs((int)bits) - 1 : 0;
                bits &= bits - 1;
                rs_append(on, &s_tie_n, a.ties, a.tie_region, k_first + (uint32_t)rr);
            }
            for (uint32_t bits = cand4; __ballot(bits != 0);) {                          // bits 4..7: the slot after (the tail of a pair)
                const bool on = bits != 0;
                const int rr = on ? __ffs((int)bits) - 1 : 0;
                bits &= bits - 1;
                rs_append(on, &s_cand_n, a.cand, a.region, k_first + (uint32_t)(rr & 3) + (uint32_t)(rr >> 2));
            }
        }
        // column maxima: the table reads of the thread's slots go out together (g = 0: nothing to do) and are
        // looked at after the queue has been worked off
        uint32_t cur[RS_ITEMS];
#pragma unroll
        for (int r = 0; r < RS_ITEMS; r++) cur[r] = (PLAIN && upd_g[r]) ? a.gmax[upd_c[r]] : 0xffffffffu;
        __syncthreads();
        {
            const uint32_t qn = sqn;
            for (uint32_t q0 = 0; q0 < qn; q0 += RS_THREADS) {
                if (q0 + (threadIdx.x & ~63u) >= qn) break;            // wave-uniform
                const uint32_t q = q0 + threadIdx.x;
                bool want_cand = false, want_tie = false;
                uint64_t k = 0;
                if (q < qn) {
                    const int i = sq[q];
                    k = base + (uint64_t)(i - RS_HALO);
                    const bool is_tie = (i > lo_i && skey[i - 1] == skey[i]) || (i + 1 < hi_i && skey[i + 1] == skey[i]);
                    if (!(a.values_only && is_tie)) rank_scan_slow(a, RsLdsView{skey, srem}, i, lo_i, hi_i, k, want_cand, want_tie);
                }
                if (!a.values_only) {
                    rs_append(want_cand, &s_cand_n, a.cand, a.region, (uint32_t)k);
                    rs_append(want_tie, &s_tie_n, a.ties, a.tie_region, (uint32_t)k);
                }
            }
        }
        if (PLAIN) {
#pragma unroll
            for (int r = 0; r < RS_ITEMS; r++)
                if (cur[r] < upd_g[r]) atomicMax(&a.gmax[upd_c[r]], upd_g[r]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && !a.values_only) { a.blk_count[blockIdx.x] = s_cand_n; a.tie_count[blockIdx.x] = s_tie_n; }
}

// ---- the same scan with half the instructions per slot (packed slots below 2^31, threshold above K, no partitions) ----
// k_rank_scan issues 200 instructions per 64 slots (86 vector, 101 scalar, 10 LDS: profiles/r04_sq_counters_scan_v1.txt)
// and is bound by that -- a SIMD issues one instruction of any kind every 3 to 4 cycles at the occupancy these kernels
// reach -- not by the 8 bytes per slot it reads (4.05 ms for 10^9 slots; 1.3 ms at the rate memory delivers).  With the
// threshold above K a slot can only matter if it ties on the key, shares its column with a neighbour, sits next to a tie
// group or lies in the 64 columns nearest a row end -- one slot in fifteen -- and the others should cost as little as can be:
//   * no LDS staging, no barriers: a wave streams through a stretch of the slots of its own, 64 consecutive slots at a
//     time, a lane each, the next row's loads in flight; the neighbours' values come by DPP shifts (lanes 3 .. 60 are
//     settled, consecutive rows overlap by six slots);
//   * x = fract(position / row length) in single precision stands in for the column: four instructions instead of the
//     thirteen of the exact remainder, off by less than eps / 2 (rs_lean_setup), so that |x - x'| < eps or > 1 - eps
//     whenever two slots share a column, and x >= near_end or x < eps for a slot near a row end: filters that never say
//     no wrongly;
//   * "ties with the next slot", "may share the column of the next / the one after" are wave-wide masks (one compare and
//     ballot each); the cases of k_rank_scan's fast path are scalar logic on shifted copies of them;
//   * heads of groups of two and slots the filters let through (4 % of the slots) are noted in the wave's own list in
//     LDS; whenever 64 have come together they are worked on, a lane each, with exact arithmetic (the slots around them
//     read from global memory again: cache hits): a pair is "simple" (k_rank_scan's test) and goes to the tie list, or
//     both members go to the candidates; a slot that shares its column with a neighbour (with either member of a pair
//     next to it) becomes a candidate;
//   * the rest -- members of groups of three and more, their neighbours, slots near a row end or near the ends of the
//     array: a few in a thousand -- collect in a second list for rank_scan_slow, the general code.
// Every slot is settled by the wave that owns it (a pair by the owner of its head): same lists, same entries as
// k_rank_scan's, in another order.
#define RL_EV_LO 3
#define RL_EV_HI 60
#define RL_EV (RL_EV_HI - RL_EV_LO + 1)
#define RL_THREADS 256
#define RL_WAVES (RL_THREADS / 64)
#define RL_LIST 128                            // a wave's list: worked off whenever 64 entries have come together

struct LeanArgs {
    const uint64_t *keys;                      // what the rows need of the scan's arguments
    uint64_t own_lo, own_hi;
    uint32_t per_wave, pmask;
    int pb;
    float inv_row_len, eps, one_minus_eps, near_end;
};

// lane l gets lane l + 1's value (lane 63: 0)
__device__ __forceinline__ uint32_t rl_from_next(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, false); }
__device__ __forceinline__ uint32_t rl_lanes_below(unsigned long long m)      // bits of m below this lane's
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// the noted pairs [0, n) of a wave's list (n <= 64): simple tied pairs, their two text positions, to the workgroup's region of a.pairs
__device__ __attribute__((noinline)) void rl_noted(const RankArgs &a, const uint2 *list, uint32_t n, uint32_t *s_pair_n)
{
    const int lane = threadIdx.x & 63;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(s_pair_n, n);
    base = __shfl(base, 0, 64);
    if ((uint32_t)lane < n && base + (uint32_t)lane < a.tie_region) a.pairs[(size_t)blockIdx.x * a.tie_region + base + lane] = list[lane];
}

// the queued slots [0, n) of a wave's list (n <= 64): rank_scan_slow on the RS_HALO slots either side, from global memory
__device__ __attribute__((noinline)) void rl_queued(const RankArgs &a, const uint32_t *list, uint32_t n, uint32_t *s_cand_n, uint32_t *s_tie_n)
{
    const int lane = threadIdx.x & 63;
    bool want_cand = false, want_tie = false;
    uint64_t k = 0;
    if ((uint32_t)lane < n) {
        k = list[lane];
        const int64_t base = (int64_t)k - RS_HALO;                     // slot of window index 0
        const int lo_i = (int)max((int64_t)0, (int64_t)a.own_lo - base);
        const int hi_i = (int)min((int64_t)(2 * RS_HALO + 1), (int64_t)a.own_hi - base);
        const RsWordView view{a.keys + base, &a, lo_i, hi_i};
        rank_scan_slow(a, view, RS_HALO, lo_i, hi_i, k, want_cand, want_tie);
    }
    rs_append(want_cand, s_cand_n, a.cand, a.region, (uint32_t)k);
    rs_append(want_tie, s_tie_n, a.ties, a.tie_region, (uint32_t)k);
}

// the first 64 entries of a list are done with: the rest moves to the front
template <class E> __device__ __forceinline__ void rl_shift(E *list, uint32_t &n)
{
    const int lane = threadIdx.x & 63;
    const uint32_t rest = n - 64;
    if ((uint32_t)lane < rest) { const E v = list[64 + lane]; list[lane] = v; }
    n = rest;
}

__global__ __launch_bounds__(RL_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_rank_scan_lean(const RankArgs *__restrict__ ap, LeanArgs f)
{
    // (the scan's arguments stay in memory: the rows below need a handful of them, and in scalar registers all of them
    // together crowd out the masks)
    __shared__ uint2 s_note[RL_WAVES][RL_LIST];
    __shared__ uint32_t s_queue[RL_WAVES][RL_LIST];
    __shared__ uint32_t s_cand_n, s_tie_n, s_pair_n;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));    // (told to be uniform: the loop below runs on scalar registers)
    if (threadIdx.x == 0) { s_cand_n = 0; s_tie_n = 0; s_pair_n = 0; }
    __syncthreads();
    uint2 *my_note = s_note[wv];
    uint32_t *my_queue = s_queue[wv];
    uint32_t nn = 0, nq = 0;                                            // (wave-uniform)
    const uint32_t own_lo = (uint32_t)f.own_lo, own_hi = (uint32_t)f.own_hi;   // (below 2^31: rs_lean_setup)
    const uint32_t wlo = own_lo + ((uint32_t)blockIdx.x * RL_WAVES + (uint32_t)wv) * f.per_wave;   // per_wave: a multiple of RL_EV
    const uint32_t whi = min(own_hi, wlo + f.per_wave);
    const unsigned long long EVM = ((1ull << (RL_EV_HI + 1)) - 1) & ~((1ull << RL_EV_LO) - 1);   // the lanes that settle their slots
    const unsigned long long RNG = ~(1ull << 63);                       // the lanes whose mask bits somebody may ask for
    const uint64_t *__restrict__ keys = f.keys;
    auto load_row = [&](uint32_t k0) -> uint64_t {
        // (a row that reaches beyond the array is a row the general code takes: what its lanes out there load does not matter)
        const uint32_t k = min(max(k0 - RL_EV_LO + (uint32_t)lane, own_lo), own_hi - 1);
        return keys[k];
    };
    uint64_t xnext = wlo < whi ? load_row(wlo) : 0ull;
    for (uint32_t k0 = wlo; k0 < whi; k0 += RL_EV) {
        const uint64_t x = xnext;
        if (k0 + RL_EV < whi) xnext = load_row(k0 + RL_EV);            // in flight during this row's work
        const uint32_t slot = k0 - RL_EV_LO + (uint32_t)lane;
        // rows within reach of the ends of the array (or of this wave's last slot, where a row is not full): the general
        // code knows what a missing neighbour means
        if (k0 < own_lo + 2 * RS_HALO || k0 + RL_EV + 2 * RS_HALO > own_hi || k0 + RL_EV > whi) {
            const unsigned long long Q = EVM & __ballot(slot < whi);
            if ((Q >> lane) & 1ull) my_queue[nq + rl_lanes_below(Q)] = slot;
            nq += (uint32_t)__popcll(Q);
        } else {
            const uint32_t hi = (uint32_t)(x >> 32), lo = (uint32_t)x;
            const uint32_t lo_n = rl_from_next(lo);
            const uint32_t dh = hi ^ rl_from_next(hi), dl = (lo ^ lo_n) >> f.pb;                // (pb < 32: the position lies in the low word)
            const float fr = __builtin_amdgcn_fractf((float)(lo & f.pmask) * f.inv_row_len);
            const float f1 = __uint_as_float(rl_from_next(__float_as_uint(fr))), f2 = __uint_as_float(rl_from_next(__float_as_uint(f1)));
            const float d1 = fabsf(fr - f1), d2 = fabsf(fr - f2);
            // bit l of a mask speaks of the slot of lane l (and those after it); lane 63 has no neighbour, nobody asks
            const unsigned long long E = __ballot((dh | dl) == 0);                           // ties with the next slot
            const unsigned long long R1 = __ballot(d1 < f.eps) | __ballot(d1 > f.one_minus_eps);   // may share the next slot's column
            const unsigned long long R2 = __ballot(d2 < f.eps) | __ballot(d2 > f.one_minus_eps);   // ... that of the slot after the next
            const unsigned long long NE = __ballot(fr >= f.near_end) | __ballot(fr < f.eps);     // may lie near its row's end, or be a '#'
            unsigned long long SS, Q;                                   // heads of pairs that are simple for sure; slots for the general code
            if (((R1 | R2 | NE | (E & (E << 1))) & RNG) == 0) {
                // seven rows in ten: no slot of the row can share a column with one nearby or lie near a row end, no group of three:
                // every tie is a pair, simple unless another pair sits right next to it
                SS = EVM & E & ~((E << 2) | (E >> 2));
                Q = EVM & E & ~SS;
                Q |= Q << 1;
            } else {
                const unsigned long long TIE = E | (E << 1);
                const unsigned long long PH = E & ~((E << 1) | (E >> 1));                    // head of a group of exactly two
                const unsigned long long LONG = TIE & ~(PH | (PH << 1));                     // member of a group of three and more
                const unsigned long long ADJL = ((LONG >> 1) | (LONG << 1)) & ~TIE;          // next to one: any member may end up next to it
                // k_rank_scan's test of a pair (h, h + 1), with "may" for "does": both with K real symbols left (NE covers the K <= 64
                // columns where not), their two columns and those of the slots before and after all different, no tie group next door
                SS = EVM & PH & ~(NE | (NE >> 1) | R1 | (R1 << 1) | (R2 << 1) | R2 | (R1 >> 1) | (E << 2) | (E >> 2));
                const unsigned long long QP = EVM & PH & ~SS;                                // pairs for the general code, both members
                // a slot that does not tie: a run with a neighbour -- either half of a tied pair may end up next to it --, or near its row's end
                const unsigned long long RUN = R1 | (R1 << 1) | ((PH & R2) << 2) | ((PH >> 1) & R2);
                Q = QP | (QP << 1) | (EVM & (LONG | ADJL | (~TIE & (RUN | NE))));
            }
            if (SS) {
                if ((SS >> lane) & 1ull) my_note[nn + rl_lanes_below(SS)] = make_uint2(lo & f.pmask, lo_n & f.pmask);
                nn += (uint32_t)__popcll(SS);
            }
            if (Q) {
                if ((Q >> lane) & 1ull) my_queue[nq + rl_lanes_below(Q)] = slot;
                nq += (uint32_t)__popcll(Q);
            }
        }
        while (nq >= 64) { rl_queued(*ap, my_queue, 64, &s_cand_n, &s_tie_n); rl_shift(my_queue, nq); }
        if (nn >= 64) { rl_noted(*ap, my_note, 64, &s_pair_n); rl_shift(my_note, nn); }
    }
    if (nn) rl_noted(*ap, my_note, nn, &s_pair_n);
    if (nq) rl_queued(*ap, my_queue, nq, &s_cand_n, &s_tie_n);
    __syncthreads();
    if (threadIdx.x == 0) { ap->blk_count[blockIdx.x] = s_cand_n; ap->tie_count[blockIdx.x] = s_tie_n; ap->pair_count[blockIdx.x] = s_pair_n; }
}

// Can fractions of the row length in single precision tell the columns apart?  x = fract((float)position * (float)(1 / row length)):
// the position is rounded to 24 bits (2^(pb - 25) off at most), the product once and the inverse once (2^-23 of position / row
// length together; delta allows twice that), fract is exact: delta bounds the error.
static bool rs_lean_setup(const RankArgs &a, int pb, unsigned blocks, LeanArgs *f)
{
    const uint64_t own = a.own_hi - a.own_lo, waves = (uint64_t)blocks * RL_WAVES;
    f->per_wave = (uint32_t)(((own + waves - 1) / waves + RL_EV - 1) / RL_EV * RL_EV);
    f->keys = a.keys; f->own_lo = a.own_lo; f->own_hi = a.own_hi; f->pmask = (uint32_t)a.pmask; f->pb = pb;
    const double L = (double)a.row_len;
    const double delta = (ldexp(1.0, pb - 25) + ldexp(1.0, pb - 22) + 2.0) / L + 1e-6;
    f->inv_row_len = (float)(1.0 / L);
    f->eps = (float)(2.0 * delta);
    f->one_minus_eps = 1.0f - f->eps;
    f->near_end = (float)(1.0 - 65.0 / L - delta);
    return 2.0 * delta < 0.01 && pb <= 31 && a.own_hi < (1ull << 31);
}

// the small tie groups k_rank_scan set aside (same grid: every workgroup works off its own region).  No member
// shares a column with a neighbour, so each is a run of its own and its extension is 1 + its longest match with
// any other suffix -- which is another member of the group (they agree on K symbols, nobody else does).
template <int L> __global__ __launch_bounds__(256) void k_tie_simple(RankArgs a)
{
    const uint32_t have = a.tie_count[blockIdx.x];
    if (have > a.tie_region) { if (threadIdx.x == 0) a.counters[1] = 1; return; }
    for (uint32_t e = threadIdx.x; e < have; e += blockDim.x) {
        const uint64_t h = a.ties[(size_t)blockIdx.x * a.tie_region + e];
        // the RS_TG slots from the head on, all loads at once; the group ends where the key changes
        uint64_t key[RS_TG], pos[RS_TG];
        uint32_t best[RS_TG];
#pragma unroll
        for (int i = 0; i < RS_TG; i++) {
            const bool ok = h + i < a.own_hi;
            const uint64_t w = ok ? a.keys[h + i] : 0ull;
            key[i] = w >> a.pb;
            pos[i] = rs_pos_of<L>(a, w, (ok && L != FBG_SLOTS_PACKED) ? a.vals[h + i] : 0u);
            if (!ok) key[i] = ~key[0];
            best[i] = 0;
        }
        int s = 1;
#pragma unroll
        for (int i = 1; i < RS_TG; i++)
            if (s == i && key[i] == key[0]) s = i + 1;
        // most groups are pairs whose texts part within 8 bytes: that comparison first, the rest by the general code
        const uint64_t x01 = fbg_load8(a.T, pos[0] + a.K) ^ fbg_load8(a.T, pos[1] + a.K);
#pragma unroll
        for (int i = 0; i < RS_TG; i++)
#pragma unroll
            for (int j = i + 1; j < RS_TG; j++)
                if (j < s) {
                    uint32_t x;
                    if (i == 0 && j == 1 && x01 != 0) x = (uint32_t)(__ffsll((unsigned long long)x01) - 1) / 8;
                    else x = fbg_extend_match(a.T, pos[i] + a.K, pos[j] + a.K, 0);
                    best[i] = max(best[i], x);
                    best[j] = max(best[j], x);
                }
#pragma unroll
        for (int i = 0; i < RS_TG; i++)
            if (i < s) rs_update(a, rs_col_of_rem(a, rs_rem<L>(a, pos[i])), fbg_clamp_lcp(best[i] + (uint32_t)a.K) + 1);
    }
}

// the simple tied pairs k_rank_scan_lean set aside with their text positions: the two suffixes agree on K symbols and nobody
// else does, the extension of either is 1 + their longest common prefix (no slot is read again: two text reads per pair)
__global__ __launch_bounds__(256) void k_tie_pairs(RankArgs a)
{
    const uint32_t have = a.pair_count[blockIdx.x];
    if (have > a.tie_region) { if (threadIdx.x == 0) a.counters[1] = 1; return; }
    for (uint32_t e = threadIdx.x; e < have; e += blockDim.x) {
        const uint2 p = a.pairs[(size_t)blockIdx.x * a.tie_region + e];
        const uint64_t x = fbg_load8(a.T, (uint64_t)p.x + a.K) ^ fbg_load8(a.T, (uint64_t)p.y + a.K);
        const uint32_t h = x != 0 ? (uint32_t)(__ffsll((unsigned long long)x) - 1) / 8 : fbg_extend_match(a.T, (uint64_t)p.x + a.K, (uint64_t)p.y + a.K, 8);
        const uint32_t g = fbg_clamp_lcp(h + (uint32_t)a.K) + 1;
        rs_update(a, rs_col_of_rem(a, rs_rem<FBG_SLOTS_PACKED>(a, p.x)), g);
        rs_update(a, rs_col_of_rem(a, rs_rem<FBG_SLOTS_PACKED>(a, p.y)), g);
    }
}

// cheap regime test + extension histogram over every 1024th block of 256 SA slots:
// counters[2] ties, counters[3] slots looked at, hist[g] = non-tie slots with extension min(g, 63)
__global__ __launch_bounds__(256) void k_tie_sample(const uint64_t *__restrict__ keys, int shift, uint64_t N, int b, int key_bits,
                                                    unsigned long long *__restrict__ counters,
                                                    unsigned int *__restrict__ hist)
{
    __shared__ unsigned int sh[64];
    __shared__ unsigned int sties;
    if (threadIdx.x < 64) sh[threadIdx.x] = 0;
    if (threadIdx.x == 0) sties = 0;
    __syncthreads();
    const uint64_t k = (uint64_t)blockIdx.x * 1024 * 256 + threadIdx.x;
    uint32_t bin = 0xffffffffu;                                       // 64 = ties on the whole key
    if (k < N) {
        const uint64_t key = keys[k] >> shift;
        const uint64_t kp = k > 0 ? keys[k - 1] >> shift : ~key, kn = k + 1 < N ? keys[k + 1] >> shift : ~key;
        if (kp == key || kn == key) {
            bin = 64;
        } else {
            const uint32_t lp = k > 0 ? rs_key_lcp(kp, key, b, key_bits) : 0u;
            const uint32_t ln = k + 1 < N ? rs_key_lcp(key, kn, b, key_bits) : 0u;
            bin = min(max(lp, ln) + 1, 63u);
        }
    }
    // a wave's slots fall into a handful of bins: one LDS update per distinct bin instead of 64 on the same word
    {
        const int lane = threadIdx.x & 63;
        unsigned long long rest = __ballot(bin != 0xffffffffu);
        while (rest) {
            const int l = __ffsll((long long)rest) - 1;
            const uint32_t v = __shfl(bin, l, 64);
            const unsigned long long same = __ballot(bin == v);
            if (lane == l) { if (v == 64) atomicAdd(&sties, (uint32_t)__popcll(same)); else atomicAdd(&sh[v], (uint32_t)__popcll(same)); }
            rest &= ~same;
        }
    }
    __syncthreads();
    if (threadIdx.x < 64 && sh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], sh[threadIdx.x]);
    if (threadIdx.x == 0) {
        atomicAdd(&counters[2], (unsigned long long)sties);
        atomicAdd(&counters[3], (unsigned long long)min((uint64_t)256, N - (uint64_t)blockIdx.x * 1024 * 256));
    }
}

// columns whose maximum stayed below the threshold: extensions that were skipped for being smaller than g_min may
// exceed it -- the threshold was too optimistic for them (a maximum of g_min or more cannot be beaten by a skipped one)
__global__ void k_count_unfilled(const uint32_t *__restrict__ gmax, uint64_t n, uint32_t g_min, int reversed,
                                 unsigned long long *__restrict__ counters)
{
    const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // the 64 columns nearest the row end (in text order) are never thresholded (k_rank_scan: rem <= 64)
    const uint64_t rem = reversed ? x + 1 : n - x;
    const bool miss = x < n && rem > 64 && gmax[x] < g_min;
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

// a slot whose key equals its successor's but not its predecessor's heads a tie group: put the group in text
// order.  by_list: t indexes the sorted candidates; otherwise every owned slot (fbg_rank_materialize).
// Groups of more than 64 go to k_tie_big.
#define RS_BIG_GROUPS 1024
#define RS_BIG_MEMBERS 8192
template <int L> __global__ void k_tie_groups(RankArgs a, uint64_t T, int by_list)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const uint64_t k0 = by_list ? (uint64_t)a.cand[t] : a.own_lo + t;
    const uint64_t key = rs_key<L>(a, k0);
    if (k0 > a.own_lo && rs_key<L>(a, k0 - 1) == key) return;                      // inside a group
    if (k0 + 1 >= a.own_hi || rs_key<L>(a, k0 + 1) != key) return;                 // not a tie (ties never cross partitions)
    uint32_t s = 1;
    while (k0 + s < a.own_hi && s <= RS_BIG_MEMBERS && rs_key<L>(a, k0 + s) == key) s++;
    if (s > 64) {
        if (s > RS_BIG_MEMBERS) { a.counters[1] = 1; return; }
        const unsigned long long e = atomicAdd(&a.counters[5], 1ull);
        if (e >= RS_BIG_GROUPS) { a.counters[1] = 1; return; }
        a.big[2 * e] = (uint32_t)k0;
        a.big[2 * e + 1] = s;
        return;
    }
    // order the s suffixes by their text (insertion sort, s is tiny).  The first K symbols agree -- unless a member
    // has fewer than K symbols left in its row: then the keys agree only up to the separator coding
    typedef typename PosT<L>::type pos_t;
    pos_t pos[64];
    uint32_t from = (uint32_t)a.K;
    for (uint32_t i = 0; i < s; i++) {
        pos[i] = (pos_t)rs_pos<L>(a, k0 + i);
        if (rs_rem<L>(a, pos[i]) < (uint32_t)a.K) from = 0;
    }
    for (uint32_t i = 1; i < s; i++) {
        const pos_t cur = pos[i];
        uint32_t j = i;
        while (j > 0) {
            const pos_t o = pos[j - 1];
            const uint32_t h = fbg_extend_match(a.T, (uint64_t)o + from, (uint64_t)cur + from, 0);
            if (a.T[(uint64_t)o + from + h] < a.T[(uint64_t)cur + from + h]) break;      // o < cur: in place
            pos[j] = o;
            j--;
        }
        pos[j] = cur;
    }
    for (uint32_t i = 0; i < s; i++) rs_set_pos<L>(a, k0 + i, pos[i]);
}

// Long tie groups exist where many rows end alike: the members of a group share their real symbols and differ in
// how many they have left (r#, rA#, rAA#, ... and the suffixes with K real symbols), '#' being the smallest
// symbol.  Text order is therefore: by symbols left (capped at K) first.  Members with the same number left sit in
// one MSA column: a run whose inner order has no influence on any extension (the LCPs inside exceed those at its
// two ends, fbg.cpp:1644-1656) -- left as it is unless `exact` (fbg_index_download wants the true suffix array).
// The members with K real symbols are ordered by their text; more than 64 of those -> fallback flag.
// One workgroup per group.
template <int L> __global__ __launch_bounds__(256) void k_tie_big(RankArgs a, int exact)
{
    __shared__ uint32_t pos_lo[RS_BIG_MEMBERS];
    __shared__ uint8_t pos_hi[L == FBG_SLOTS_WIDE ? RS_BIG_MEMBERS : 1];     // positions beyond 32 bits: their top byte
    __shared__ uint8_t cls[RS_BIG_MEMBERS];
    __shared__ uint32_t offs[66];
    __shared__ uint32_t q_start, q_count;
    if (blockIdx.x >= a.counters[5]) return;
    const uint64_t k0 = a.big[2 * blockIdx.x];
    const uint32_t s = a.big[2 * blockIdx.x + 1];
    auto member = [&](uint32_t i) -> uint64_t {
        return L == FBG_SLOTS_WIDE ? ((uint64_t)pos_hi[L == FBG_SLOTS_WIDE ? i : 0] << 32) | pos_lo[i] : (uint64_t)pos_lo[i];
    };
    if (threadIdx.x < 66) offs[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < s; i += blockDim.x) {
        const uint64_t p = rs_pos<L>(a, k0 + i);
        const uint32_t c = min(rs_rem<L>(a, p), (uint32_t)a.K);
        pos_lo[i] = (uint32_t)p;
        if (L == FBG_SLOTS_WIDE) pos_hi[L == FBG_SLOTS_WIDE ? i : 0] = (uint8_t)(p >> 32);
        cls[i] = (uint8_t)c;
        atomicAdd(&offs[c + 1], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        q_count = offs[a.K + 1];
        for (int c = 1; c <= a.K + 1; c++) offs[c] += offs[c - 1];      // offs[c] = first slot of class c
        q_start = offs[a.K];
    }
    __syncthreads();
    if (!exact) {
        for (uint32_t i = threadIdx.x; i < s; i += blockDim.x) {
            const uint32_t dst = atomicAdd(&offs[cls[i]], 1u);
            rs_set_pos<L>(a, k0 + dst, member(i));
        }
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t q = q_count;
            if (q > 64) { a.counters[1] = 1; return; }
            const uint64_t kq = k0 + q_start;
            for (uint32_t i = 1; i < q; i++) {                          // insertion sort by the text beyond the key
                const uint64_t cur = rs_pos<L>(a, kq + i);
                uint32_t j = i;
                while (j > 0) {
                    const uint64_t o = rs_pos<L>(a, kq + j - 1);
                    const uint32_t h = fbg_extend_match(a.T, o + a.K, cur + a.K, 0);
                    if (a.T[o + a.K + h] < a.T[cur + a.K + h]) break;
                    rs_set_pos<L>(a, kq + j, o);
                    j--;
                }
                rs_set_pos<L>(a, kq + j, cur);
            }
        }
        return;
    }
    // exact: rank of every member among all members by text comparison (distinct suffixes: ranks are a permutation)
    for (uint32_t i = threadIdx.x; i < s; i += blockDim.x) {
        const uint64_t p = member(i);
        uint32_t rank = 0;
        for (uint32_t j = 0; j < s; j++) {
            if (j == i) continue;
            const uint64_t o = member(j);
            const uint32_t h = fbg_extend_match(a.T, o, p, 0);
            rank += a.T[o + h] < a.T[p + h] ? 1u : 0u;
        }
        rs_set_pos<L>(a, k0 + rank, p);
    }
}

// LCP of the suffixes in SA slots k-1 and k (final order, or a member of an unordered small tie group next to a
// slot outside it: all its members have K symbols left, the result does not depend on which one it is)
template <int L> __device__ __forceinline__ uint32_t rs_slot_lcp(const RankArgs &a, uint64_t k)
{
    if (k == (a.first_part ? a.own_lo : 0) || k >= (a.last_part ? a.own_hi : a.N)) return 0;   // no such neighbour
    const Slot x = rs_slot<L>(a, k - 1), y = rs_slot<L>(a, k);
    if (x.key != y.key) return min(min(rs_key_lcp(x.key, y.key, a.b, a.key_bits), x.rem), y.rem);
    const uint32_t from = (x.rem < (uint32_t)a.K || y.rem < (uint32_t)a.K) ? 0u : (uint32_t)a.K;
    return fbg_clamp_lcp(fbg_extend_match(a.T, (uint64_t)x.pos + from, (uint64_t)y.pos + from, 0) + from);
}

// test / debugging aid (fbg_index_download): suffix array, its inverse and the neighbour LCPs by text position
template <int L> __global__ void k_rank_materialize(RankArgs a, uint32_t *sa, uint32_t *isa, uint32_t *pl, uint32_t *pr)
{
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.N) return;
    const uint32_t p = (uint32_t)rs_pos<L>(a, k);                          // fbg_index_download: texts below 2^32 only
    sa[k] = p;
    isa[p] = (uint32_t)k;
    pl[p] = rs_slot_lcp<L>(a, k);
    pr[p] = rs_slot_lcp<L>(a, k + 1);
}

// candidates (sorted, final SA order): one thread per run head walks its run of same-column slots.  With
// partitions a run may begin or end in a halo (copies of the neighbouring partition's edge slots): it is walked
// in full, but only owned members update the column maxima -- the neighbour does the same from its side.
template <int L> __global__ void k_runs(RankArgs a, uint64_t T)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const uint64_t k0 = a.cand[t];
    const uint32_t col = rs_col<L>(a, k0);
    if (col == a.n) return;                                            // '#' / sentinel: not a row pointer
    if (k0 > a.own_lo && rs_col<L>(a, k0 - 1) == col) return;         // an owned predecessor heads this run
    const uint64_t lo_slot = a.first_part ? a.own_lo : 0, hi_slot = a.last_part ? a.own_hi : a.N;
    uint64_t ks = k0;                                                  // true head: possibly inside the previous halo
    while (ks > lo_slot && rs_col<L>(a, ks - 1) == col) ks--;
    if (ks == 0 && !a.first_part) { a.counters[1] = 1; return; }       // run longer than the halo
    // forward: running minimum of LCP[lb..r]   (owned members of a run are contiguous in cand[])
    uint32_t run = rs_slot_lcp<L>(a, ks);
    uint64_t s = ks;
    for (;;) {
        if (s >= k0 && s < a.own_hi) a.pm[t + (s - k0)] = run;
        if (s + 1 >= hi_slot || rs_col<L>(a, s + 1) != col) break;
        s++;
        run = min(run, rs_slot_lcp<L>(a, s));
    }
    if (s + 1 == a.N && !a.last_part) { a.counters[1] = 1; return; }   // ran through the next halo
    // backward: running minimum of LCP[r+1..rb+1], extension, column maximum   (fbg.cpp:1656)
    uint32_t rmin = 0xffffffffu;
    for (;; s--) {
        rmin = min(rmin, rs_slot_lcp<L>(a, s + 1));
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

// kernels exist for both slot layouts
#define RS_LAUNCH(kernel, layout, grid, block, st, ...)                                                     \
    do {                                                                                                    \
        if ((layout) == FBG_SLOTS_PACKED) hipLaunchKernelGGL((kernel<FBG_SLOTS_PACKED>), grid, block, 0, st, __VA_ARGS__);  \
        else if ((layout) == FBG_SLOTS_WIDE) hipLaunchKernelGGL((kernel<FBG_SLOTS_WIDE>), grid, block, 0, st, __VA_ARGS__); \
        else hipLaunchKernelGGL((kernel<FBG_SLOTS_PAIRS>), grid, block, 0, st, __VA_ARGS__);                                \
    } while (0)

#define RS_LAUNCH_SCAN(layout, plain, grid, st, args)                                                                         \
    do {                                                                                                                     \
        const dim3 blk_(RS_THREADS);                                                                                         \
        if ((layout) == FBG_SLOTS_PACKED) { if (plain) hipLaunchKernelGGL((k_rank_scan<FBG_SLOTS_PACKED, true>), grid, blk_, 0, st, args);   \
                                            else hipLaunchKernelGGL((k_rank_scan<FBG_SLOTS_PACKED, false>), grid, blk_, 0, st, args); }      \
        else if ((layout) == FBG_SLOTS_WIDE) { if (plain) hipLaunchKernelGGL((k_rank_scan<FBG_SLOTS_WIDE, true>), grid, blk_, 0, st, args);  \
                                               else hipLaunchKernelGGL((k_rank_scan<FBG_SLOTS_WIDE, false>), grid, blk_, 0, st, args); }     \
        else { if (plain) hipLaunchKernelGGL((k_rank_scan<FBG_SLOTS_PAIRS, true>), grid, blk_, 0, st, args);                               \
               else hipLaunchKernelGGL((k_rank_scan<FBG_SLOTS_PAIRS, false>), grid, blk_, 0, st, args); }                                  \
    } while (0)

// k_tie_simple only raises column maxima (atomicMax) from slots no other kernel of the scan writes: it runs on a second
// stream beside the candidate kernels (compaction, sort, tie groups, runs -- small, latency-bound launches).
static int rs_fork(fbg_ctx *ctx)
{
    if (!ctx->aux) {
        FBG_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->aux, hipStreamNonBlocking));
        FBG_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->aux_fork, hipEventDisableTiming));
        FBG_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->aux_join, hipEventDisableTiming));
    }
    FBG_HIP_TRY(ctx, hipEventRecord(ctx->aux_fork, ctx->stream));
    FBG_HIP_TRY(ctx, hipStreamWaitEvent(ctx->aux, ctx->aux_fork, 0));
    ctx->aux_pending = true;
    return FBG_OK;
}
static int rs_join(fbg_ctx *ctx)
{
    if (!ctx->aux_pending) return FBG_OK;
    ctx->aux_pending = false;
    FBG_HIP_TRY(ctx, hipEventRecord(ctx->aux_join, ctx->aux));
    FBG_HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->aux_join, 0));
    return FBG_OK;
}

// The candidates, flat and unsorted in dp_e: sorted (SA order), tie groups put in final order.  On return a.cand / a.pm name
// the sorted list and its scratch.
static int rs_order_candidates(fbg_ctx *ctx, RankArgs &a, int layout, uint64_t T, int *launches)
{
    hipStream_t st = ctx->stream;
    FBG_TRY(fbg_reserve(ctx, ctx->dp_a, T * 4));
    FBG_TRY(fbg_reserve(ctx, ctx->dp_b, T * 4));
    uint32_t *sorted = ctx->dp_a.as<uint32_t>();
    uint32_t *flat = ctx->dp_e.as<uint32_t>();
    FBG_TRY(rs_with_tmp(ctx, [&](void *tmp, size_t &bytes) {
        return rocprim::radix_sort_keys(tmp, bytes, flat, sorted, (size_t)T, 0u, 32u, st);
    }));
    a.cand = sorted;
    a.pm = ctx->dp_b.as<uint32_t>();
    RS_LAUNCH(k_tie_groups, layout, dim3(fbg_blocks(T, 64)), dim3(64), st, a, T, 1);
    RS_LAUNCH(k_tie_big, layout, dim3(RS_BIG_GROUPS), dim3(256), st, a, 0);
    *launches += 3;
    return FBG_OK;
}

// k_rank_scan over the owned slots, the small tie groups, then the candidates: compacted, sorted, tie groups put
// in final order.  On return a.cand / a.pm name the sorted list and its scratch; *T = ~0 when a workgroup's
// candidate region overflowed (similar rows: the caller takes another path).
static int rs_classify(fbg_ctx *ctx, RankArgs &a, int layout, uint64_t *T_out, int *launches)
{
    hipStream_t st = ctx->stream;
    const uint64_t own = a.own_hi - a.own_lo;
    const unsigned rs_blocks = fbg_blocks(own, RS_CHUNK, 256 * 16);
    const uint32_t region = (uint32_t)((own / rs_blocks) / 8 + 256);    // a workgroup may find 1/8 of its slots + slack
    const uint32_t tie_region = (uint32_t)((own / rs_blocks) / 6 + 256);
    FBG_TRY(fbg_reserve(ctx, ctx->list, (size_t)rs_blocks * region * 4));
    FBG_TRY(fbg_reserve(ctx, ctx->tie_list, (size_t)rs_blocks * tie_region * 4));
    FBG_TRY(fbg_reserve(ctx, ctx->big_groups, RS_BIG_GROUPS * 8));
    a.big = ctx->big_groups.as<uint32_t>();
    FBG_TRY(fbg_reserve(ctx, ctx->dp_c, (size_t)(rs_blocks + 1) * 4));
    FBG_TRY(fbg_reserve(ctx, ctx->dp_d, (size_t)(rs_blocks + 1) * 4));
    FBG_TRY(fbg_reserve(ctx, ctx->dp_f, (size_t)(rs_blocks + 1) * 4));
    FBG_HIP_TRY(ctx, hipMemsetAsync(ctx->dp_c.p, 0, (size_t)(rs_blocks + 1) * 4, st));
    FBG_HIP_TRY(ctx, hipMemsetAsync(ctx->dp_f.p, 0, (size_t)(rs_blocks + 1) * 4, st));
    a.cand = ctx->list.as<uint32_t>();
    a.blk_count = ctx->dp_c.as<uint32_t>(); a.region = region;
    a.ties = ctx->tie_list.as<uint32_t>();
    a.tie_count = ctx->dp_f.as<uint32_t>(); a.tie_region = tie_region;
    FBG_TRY(fbg_stage_begin(ctx, FBG_STAGE_RANK_KERNEL));
    LeanArgs lf;
    const bool lean = layout == FBG_SLOTS_PACKED && a.g_min > (uint32_t)a.K && !a.values_only && !a.part_mode && !ctx->opt.rank_no_lean &&
                      rs_lean_setup(a, a.pb, rs_blocks, &lf);
    if (lean)
    {
        FBG_TRY(fbg_reserve(ctx, ctx->ps_g, (size_t)rs_blocks * tie_region * 8));
        FBG_TRY(fbg_reserve(ctx, ctx->ps_h, (size_t)rs_blocks * 4));
        a.pairs = ctx->ps_g.as<uint2>(); a.pair_count = ctx->ps_h.as<uint32_t>();
        // the scan's arguments travel through memory (the host copy lives in the context: nothing to wait for)
        static_assert(sizeof(RankArgs) <= sizeof(ctx->kargs_host), "room for the scan's arguments");
        FBG_TRY(fbg_reserve(ctx, ctx->kargs, sizeof(RankArgs)));
        memcpy(ctx->kargs_host, &a, sizeof(RankArgs));
        FBG_HIP_TRY(ctx, hipMemcpyAsync(ctx->kargs.p, ctx->kargs_host, sizeof(RankArgs), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_rank_scan_lean, dim3(rs_blocks), dim3(RL_THREADS), 0, st, (const RankArgs *)ctx->kargs.p, lf);
    }
    else
        RS_LAUNCH_SCAN(layout, a.g_min <= (uint32_t)a.K || a.values_only, dim3(rs_blocks), st, a);
    FBG_TRY(fbg_stage_end(ctx, FBG_STAGE_RANK_KERNEL, 1));
    {
        hipStream_t ts = st;
        if (!ctx->opt.no_aux_stream) { FBG_TRY(rs_fork(ctx)); ts = ctx->aux; }
        if (lean) hipLaunchKernelGGL(k_tie_pairs, dim3(rs_blocks), dim3(256), 0, ts, a);
        RS_LAUNCH(k_tie_simple, layout, dim3(rs_blocks), dim3(256), ts, a);
    }
    *launches += 2;
    // from here on k_tie_simple may be running on the aux stream: an error return joins it first, so that no caller
    // ever reuses or frees the buffers under it
    const int rc_rest = [&]() -> int {
        // candidate counts per workgroup -> offsets; total and the largest count come back to the host
        uint32_t *d_counts = a.blk_count, *d_offs = ctx->dp_d.as<uint32_t>();
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
        if (mx > region) { *T_out = ~0ull; return rs_join(ctx); }
        if (T > 0) {
            // regions are in SA order already (workgroup b owns chunks b, b+G, ...: not contiguous) -> compact, then sort
            FBG_TRY(fbg_reserve(ctx, ctx->dp_e, T * 4));
            uint32_t *flat = ctx->dp_e.as<uint32_t>();
            hipLaunchKernelGGL(k_cand_compact, dim3(rs_blocks), dim3(256), 0, st, a.cand, d_counts, d_offs, region, flat);
            FBG_TRY(rs_order_candidates(ctx, a, layout, T, launches));
        }
        return FBG_OK;
    }();
    if (rc_rest != FBG_OK) { (void)rs_join(ctx); return rc_rest; }
    return FBG_OK;
}

// Sample of the sorted keys (count > 2^22 only): (a) similar rows tie almost everywhere -> *reject, the rank-order scan
// is not for them; (b) a.g_min: extensions so small that >= 32 rows of every column are expected to exceed them cannot
// be a column maximum -- skipping them removes almost all table reads.  The threshold is verified afterwards
// (k_count_unfilled), never trusted.
static int rs_pick_threshold(fbg_ctx *ctx, RankArgs &a, const uint64_t *keys, uint64_t count, const KeyGeom &geom, int *reject,
                             int *launches)
{
    *reject = 0;
    if (count <= (1u << 22)) return FBG_OK;
    hipStream_t st = ctx->stream;
    unsigned long long *cnt = a.counters;
    unsigned int *d_hist = reinterpret_cast<unsigned int *>(cnt + 8);
    FBG_HIP_TRY(ctx, hipMemsetAsync(d_hist, 0, 64 * sizeof(unsigned int), st));
    hipLaunchKernelGGL(k_tie_sample, dim3(fbg_blocks(count, 1024 * 256)), dim3(256), 0, st, keys, a.pb, count, geom.b, geom.key_bits,
                       cnt, d_hist);
    (*launches)++;
    unsigned long long hs[4];
    unsigned int hh[64];
    FBG_HIP_TRY(ctx, hipMemcpyAsync(hs, cnt, sizeof(hs), hipMemcpyDeviceToHost, st));
    FBG_HIP_TRY(ctx, hipMemcpyAsync(hh, d_hist, sizeof(hh), hipMemcpyDeviceToHost, st));
    FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
    if (hs[2] * 4 > hs[3]) { *reject = 1; return FBG_OK; }
    if (!ctx->opt.rank_no_threshold) {
        const double need = 32.0 / (double)ctx->m * (double)hs[3];    // sampled slots that must lie at or above g_min
        unsigned long long above = 0;
        for (int g = 63; g >= 1; g--) {
            above += hh[g];
            if ((double)above >= need) { a.g_min = (uint32_t)g; break; }
        }
        // suffixes that tie on the whole key extend by K + 1 at least, more than any that does not tie: where
        // every column can expect dozens of them nothing else can be a column maximum
        if ((double)hs[2] / (double)hs[3] * (double)ctx->m >= 32.0) a.g_min = (uint32_t)geom.K + 1;
    }
    return FBG_OK;
}

// Called by fbg_suffix_sort right after the round-0 sort of the compact keys.  *done = 1 when the rank-order scan
// covered the whole input (ctx->ranked set); 0 = continue with the record path.
int fbg_rank_scan_try(fbg_ctx *ctx, uint64_t *keys, uint32_t *vals, const KeyGeom &geom, int *done)
{
    *done = 0;
    ctx->ranked = false;
    ctx->part_active = false;
    const uint64_t N = ctx->N, n = ctx->n;
    const int layout = rs_layout(geom);
    hipStream_t st = ctx->stream;
    FBG_TRY(fbg_stage_begin(ctx, FBG_STAGE_RANKSCAN));
    unsigned long long *cnt = ctx->scalars.as<unsigned long long>() + 32;
    int launches = 0;
    FBG_TRY(fbg_reserve(ctx, ctx->gmax, (n + 1) * 4));             // (before rs_args_init reads the pointer)
    RankArgs a;
    rs_args_init(ctx, a, keys, vals, N, layout, geom.pb, geom.b, geom.key_bits, geom.K);
    uint64_t T = 0;
    FBG_HIP_TRY(ctx, hipMemsetAsync(cnt, 0, 8 * sizeof(unsigned long long), st));
    FBG_HIP_TRY(ctx, hipMemsetAsync(ctx->gmax.p, 0, (n + 1) * 4, st));
    int reject = 0;
    FBG_TRY(rs_pick_threshold(ctx, a, keys, N, geom, &reject, &launches));
    if (reject) return fbg_stage_end(ctx, FBG_STAGE_RANKSCAN, launches);
    FBG_TRY(rs_classify(ctx, a, layout, &T, &launches));
    if (T == ~0ull) return fbg_stage_end(ctx, FBG_STAGE_RANKSCAN, launches);      // a region overflowed: record path
    if (T > 0) {
        RS_LAUNCH(k_runs, layout, dim3(fbg_blocks(T, 64)), dim3(64), st, a, T);
        launches++;
    }
    FBG_TRY(rs_join(ctx));
    unsigned long long h[5];
    // a large tie group / an overflowing tie region -> record path; a column without a value lost all its rows
    // to the threshold -> redo without it
    if (a.g_min > 1)
        hipLaunchKernelGGL(k_count_unfilled, dim3(fbg_blocks(n, 256)), dim3(256), 0, st, a.gmax, n, a.g_min, a.reversed, cnt);
    FBG_HIP_TRY(ctx, hipMemcpyAsync(h, cnt, sizeof(h), hipMemcpyDeviceToHost, st));
    FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
    launches++;
    if (h[1] != 0) return fbg_stage_end(ctx, FBG_STAGE_RANKSCAN, launches);
    if (a.g_min > 1 && h[4] != 0) {
        a.g_min = 0;
        a.values_only = 1;
        RS_LAUNCH_SCAN(layout, true, dim3(fbg_blocks(N, RS_CHUNK, 256 * 16)), st, a);
        launches++;
    }
    FBG_HIP_TRY(ctx, hipGetLastError());
    ctx->n_exc = 0;
    ctx->ranked = true;
    rs_remember(ctx, keys, vals, geom);
    *done = 1;
    return fbg_stage_end(ctx, FBG_STAGE_RANKSCAN, launches);
}

// ---- partitioned index (suffix_sort.hip, fbg_part_sort): this GPU holds the SA slots of one key range -------
// keys / vals: FBG_PART_HALO + count + FBG_PART_HALO slots; the middle part is sorted, the halos are filled in
// by fbg_rank_part_runs once the neighbouring partitions have published their edge slots.
__global__ void k_halo_export(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ vals, uint64_t own_lo,
                              uint64_t own_hi, uint64_t ok, uint8_t *__restrict__ blob)
{
    uint64_t *bk = reinterpret_cast<uint64_t *>(blob);
    uint32_t *bv = reinterpret_cast<uint32_t *>(blob + 2 * FBG_PART_HALO * 8);
    uint64_t *tail = reinterpret_cast<uint64_t *>(blob + 2 * FBG_PART_HALO * 12);
    const uint32_t t = threadIdx.x;                    // 2 * FBG_PART_HALO threads: head slots, then tail slots
    if (ok & 1) {
        const uint64_t k = t < FBG_PART_HALO ? own_lo + t : own_hi - 2 * FBG_PART_HALO + t;
        bk[t] = keys[k]; bv[t] = vals ? vals[k] : 0u;
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
        if (vals) vals[t] = reinterpret_cast<const uint32_t *>(bl + 2 * FBG_PART_HALO * 8)[FBG_PART_HALO + t];
    }
    if (part + 1 < nparts) {                           // head of the next partition -> slots [own_hi, own_hi + H)
        const uint8_t *bl = blobs + (size_t)(part + 1) * FBG_PART_HALO_BYTES;
        keys[own_hi + t] = reinterpret_cast<const uint64_t *>(bl)[t];
        if (vals) vals[own_hi + t] = reinterpret_cast<const uint32_t *>(bl + 2 * FBG_PART_HALO * 8)[t];
    }
}

static void rs_part_args(fbg_ctx *ctx, RankArgs &a)
{
    rs_args_init(ctx, a, ctx->rk_keys, ctx->sa_ptr, ctx->part_count + 2 * FBG_PART_HALO, ctx->rk_layout, ctx->rk_pb, ctx->rk_b,
                 ctx->rk_key_bits, ctx->rk_K);
    a.own_lo = FBG_PART_HALO; a.own_hi = FBG_PART_HALO + ctx->part_count;
    a.first_part = ctx->part == 0; a.last_part = ctx->part + 1 == ctx->nparts;
    a.part_mode = ctx->nparts > 1;
}

// Phase 1: classify the owned slots, order the tie groups, publish the edge slots (d_blob, FBG_PART_HALO_BYTES).
// *ok = 0: this partition cannot be handled in rank order (the flag travels in the blob; every rank sees it).
int fbg_rank_part_classify(fbg_ctx *ctx, uint64_t *keys, uint32_t *vals, uint64_t count, const KeyGeom &geom, int pre_ok,
                           uint8_t *d_blob, int *ok)
{
    const uint64_t n = ctx->n;
    hipStream_t st = ctx->stream;
    FBG_TRY(fbg_stage_begin(ctx, FBG_STAGE_RANKSCAN));
    int launches = 0;
    ctx->ranked = false;
    ctx->part_count = count;
    ctx->part_T = 0;
    rs_remember(ctx, keys, vals, geom);
    const int layout = rs_layout(geom);
    FBG_TRY(fbg_reserve(ctx, ctx->gmax, (n + 1) * 4));
    unsigned long long *cnt = ctx->scalars.as<unsigned long long>() + 32;
    FBG_HIP_TRY(ctx, hipMemsetAsync(cnt, 0, 8 * sizeof(unsigned long long), st));
    FBG_HIP_TRY(ctx, hipMemsetAsync(ctx->gmax.p, 0, (n + 1) * 4, st));
    int good = pre_ok && count >= 2 * FBG_PART_HALO;
    RankArgs a;
    rs_part_args(ctx, a);
    if (good) {                                        // similar rows tie almost everywhere: not for this path
        int reject = 0;
        FBG_TRY(rs_pick_threshold(ctx, a, keys + a.own_lo, count, geom, &reject, &launches));
        if (reject) good = 0;
    }
    ctx->part_gmin = a.g_min;
    if (good) {
        uint64_t T = 0;
        FBG_TRY(rs_classify(ctx, a, layout, &T, &launches));
        FBG_TRY(rs_join(ctx));
        if (T == ~0ull) good = 0;
        else {
            ctx->part_T = T;
            unsigned long long h[2];                   // tie groups longer than 64 / tie regions too small raise [1]
            FBG_HIP_TRY(ctx, hipMemcpyAsync(h, cnt, sizeof(h), hipMemcpyDeviceToHost, st));
            FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
            if (h[1] != 0) good = 0;
        }
    }
    // the verdict and the threshold this partition worked with travel in the blob
    hipLaunchKernelGGL(k_halo_export, dim3(1), dim3(2 * FBG_PART_HALO), 0, st, keys, vals, a.own_lo, a.own_hi,
                       (uint64_t)good | ((uint64_t)a.g_min << 8), d_blob);
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
    uint32_t gmin_all = 0;
    for (int p = 0; p < ctx->nparts; p++) {
        uint64_t tail[2];
        memcpy(tail, hb.data() + (size_t)p * FBG_PART_HALO_BYTES + 2 * FBG_PART_HALO * 12, sizeof(tail));
        if (!(tail[0] & 1)) good = 0;
        gmin_all = std::max(gmin_all, (uint32_t)(tail[0] >> 8));
    }
    ctx->part_gmin = gmin_all;      // a column maximum of at least this cannot be beaten by anything any partition skipped
    if (good) {
        RankArgs a;
        rs_part_args(ctx, a);
        a.cand = ctx->dp_a.as<uint32_t>(); a.pm = ctx->dp_b.as<uint32_t>();
        hipLaunchKernelGGL(k_halo_import, dim3(1), dim3(FBG_PART_HALO), 0, st, d_blobs, ctx->part, ctx->nparts, a.own_lo,
                           a.own_hi, a.keys, a.vals);
        launches++;
        if (ctx->part_T > 0) {
            RS_LAUNCH(k_runs, ctx->rk_layout, dim3(fbg_blocks(ctx->part_T, 64)), dim3(64), st, a, ctx->part_T);
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

// After the column maxima of all partitions were max-reduced into ctx->gmax: columns whose maximum is below the
// largest threshold any partition used (and that are not exempt from thresholds)
int fbg_rank_part_unfilled(fbg_ctx *ctx, uint64_t *unfilled)
{
    *unfilled = 0;
    if (ctx->part_gmin <= 1) return FBG_OK;
    hipStream_t st = ctx->stream;
    unsigned long long *cnt = ctx->scalars.as<unsigned long long>() + 32;
    FBG_HIP_TRY(ctx, hipMemsetAsync(cnt + 4, 0, sizeof(unsigned long long), st));
    hipLaunchKernelGGL(k_count_unfilled, dim3(fbg_blocks(ctx->n, 256)), dim3(256), 0, st, ctx->gmax.as<uint32_t>(), ctx->n, ctx->part_gmin,
                       ctx->reversed, cnt);
    unsigned long long h = 0;
    FBG_HIP_TRY(ctx, hipMemcpyAsync(&h, cnt + 4, sizeof(h), hipMemcpyDeviceToHost, st));
    FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
    *unfilled = h;
    return FBG_OK;
}

// The exact re-scan of this partition's slots without any threshold, on top of the maxima in ctx->gmax
int fbg_rank_part_rescan(fbg_ctx *ctx)
{
    FBG_TRY(fbg_stage_begin(ctx, FBG_STAGE_RANKSCAN));
    RankArgs a;
    rs_part_args(ctx, a);
    a.g_min = 0;
    a.values_only = 1;
    RS_LAUNCH_SCAN(ctx->rk_layout, true, dim3(fbg_blocks(ctx->part_count, RS_CHUNK, 256 * 16)), ctx->stream, a);
    FBG_HIP_TRY(ctx, hipGetLastError());
    ctx->part_gmin = 0;
    return fbg_stage_end(ctx, FBG_STAGE_RANKSCAN, 1);
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

// sa / isa / lcp_prev / lcp_next by text position for fbg_index_download when the index is in rank order.
// The small tie groups the scan never had to order are ordered here first.
int fbg_rank_materialize(fbg_ctx *ctx, uint32_t *d_sa, uint32_t *d_isa, uint32_t *d_pl, uint32_t *d_pr)
{
    RankArgs a;
    rs_args_init(ctx, a, ctx->rk_keys, ctx->sa_ptr, ctx->N, ctx->rk_layout, ctx->rk_pb, ctx->rk_b, ctx->rk_key_bits, ctx->rk_K);
    FBG_TRY(fbg_reserve(ctx, ctx->big_groups, RS_BIG_GROUPS * 8));
    a.big = ctx->big_groups.as<uint32_t>();
    FBG_HIP_TRY(ctx, hipMemsetAsync(a.counters + 5, 0, sizeof(unsigned long long), ctx->stream));
    RS_LAUNCH(k_tie_groups, ctx->rk_layout, dim3(fbg_blocks(ctx->N, 256)), dim3(256), ctx->stream, a, ctx->N, 0);
    RS_LAUNCH(k_tie_big, ctx->rk_layout, dim3(RS_BIG_GROUPS), dim3(256), ctx->stream, a, 1);
    RS_LAUNCH(k_rank_materialize, ctx->rk_layout, dim3(fbg_blocks(ctx->N, 256)), dim3(256), ctx->stream, a, d_sa, d_isa, d_pl, d_pr);
    FBG_HIP_TRY(ctx, hipGetLastError());
    return FBG_OK;
}
