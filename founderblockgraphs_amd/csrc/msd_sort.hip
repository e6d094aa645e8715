// msd_sort.hip -- the round-0 sort of the packed slots (key << pb | position) in three passes over the data.
//
// An LSD radix sort moves all N slots once per 9-bit digit: 4 passes for the 34 key bits of a 10^9-symbol DNA text,
// after a pass that writes the unsorted slots and one that reads them for the digit histograms (80 bytes of HBM
// traffic per suffix with rocPRIM's onesweep).  Most-significant-digit first, each pass only has to bring a slot
// into the right bucket, in any order, and the last 16 bits can be settled inside LDS:
//
//   k_msd_pack_split   pass 1, fused with the key packing of suffix_sort.hip: the keys of 8192 text positions are
//                      built in registers, ranked by their top 9 bits with LDS atomics, regrouped in LDS and written
//                      out in runs -- one global atomic per (tile, bucket) reserves the run's place.  No histogram
//                      pass: every bucket owns a fixed stretch of cap1 slots (mean + 25 %).
//   k_msd_split        pass 2: every bucket of pass 1, tile by tile, by the next 9 bits into sub-buckets of MSD_FN_CAP
//                      slots each.
//   k_msd_finish       pass 3: one workgroup per sub-bucket (a few thousand slots): split on the next 10 bits inside
//                      LDS, then every slot counts the smaller slots of its bin -- its final place -- and the
//                      sub-bucket leaves in order, at the offset the scan of the sub-bucket sizes gives it.
//
// 8 (text) + 8 + 16 + 16 = 48 bytes of traffic per suffix on paper (a little more for the gaps between buckets).
// The fixed capacities are an optimistic bet on keys that spread evenly (dissimilar rows -- the only inputs that come
// here, see sample_says_similar); a bucket or sub-bucket that overflows raises a flag and the caller sorts with
// rocPRIM instead (suffix_sort.hip).  Equal keys end up in position order.
#include "fbg_internal.h"
#include "msd_keys.h"
#include <rocprim/rocprim.hpp>
#include <algorithm>
#include <cmath>

#define MSD_DIG 9
#define MSD_NB (1 << MSD_DIG)
#define MSD_THREADS 1024
#define MSD_TILE (MSD_THREADS * MSD_ITEMS)
#define MSD_FN_THREADS 512
#define MSD_FN_CAP 4096                        // slots per sub-bucket stretch (mean at 10^9 suffixes: 3815; the few that overflow: arena)
#define MSD_FN_ITEMS (MSD_FN_CAP / MSD_FN_THREADS)
#define MSD_FN_BITS 11
#define MSD_FN_BINS (1 << MSD_FN_BITS)
#define MSD_BIG_THREADS 1024
#define MSD_BIG_CAP 16384                      // largest sub-bucket (k_msd_finish_big)
#define MSD_ARENA (1u << 20)

struct MsdArgs {
    const uint8_t *T;
    uint64_t N;
    const uint8_t *code;
    int b, K, pb, kb;                          // bits per symbol, symbols per key, position bits, key bits
    uint64_t *buf1;                            // pass 1 output: MSD_NB stretches of cap1 slots
    uint64_t cap1;
    int xs;                                    // stretches per bucket (1, or 8: one per XCD, see k_msd_pack_split), segcap = cap1 / xs slots each
    uint64_t segcap;
    unsigned long long *count1;                // [MSD_NB * xs]
    const uint32_t *tile_start;                // [MSD_NB * xs + 1]: first tile of every stretch (pass 2)
    uint64_t tile0;                            // pass 1: the first tile of this launch
    uint32_t tiles2, tiles2_x;                 // pass 2: tiles in all; tiles per XCD when the tiles are dealt to the XCDs in ranges (0: in launch order)
    uint64_t *buf2;                            // pass 2 output: MSD_NB^2 stretches of MSD_FN_CAP slots
    uint32_t *count2;                          // [MSD_NB^2]
    const unsigned long long *off;             // [MSD_NB^2 + 1]: exclusive scan of count2
    uint64_t *out;                             // sorted slots
    uint32_t *arena_sb;                        // slots that did not fit their sub-bucket's stretch: sub-bucket ...
    uint64_t *arena_w;                         // ... and word, arena_count[0] of them (few: row ends pile up on a few keys)
    unsigned long long *arena_count;
    uint32_t arena_cap;
    unsigned long long *flag;                  // != 0: a capacity was exceeded
};

// ranks -> exclusive offsets of the MSD_NB digit counts of a tile; thread d < MSD_NB also reserves the run of digit d
// behind *cursor (one global atomic per non-empty digit) and leaves its start, relative to the bucket, minus the run's
// start in the tile, in gdelta[d] (32-bit wrap-around arithmetic: the write-out adds the slot's place in the tile -- one
// LDS lookup per slot, and these kernels are bound by their LDS operations)
__device__ __forceinline__ void msd_scan_and_reserve(uint32_t *cnt, uint32_t *loff, uint32_t *gdelta, uint32_t *wsum,
                                                     unsigned long long *cursor_d, bool wide_cursor, uint32_t *cursor32_d)
{
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t c = threadIdx.x < MSD_NB ? cnt[threadIdx.x] : 0u;
    uint32_t inc = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d, 64); if ((int)lane >= d) inc += o; }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    uint32_t pre = 0;
    for (uint32_t q = 0; q < wv; q++) pre += wsum[q];
    if (threadIdx.x < MSD_NB) {
        loff[threadIdx.x] = pre + inc - c;
        unsigned long long g = 0;
        if (c) g = wide_cursor ? atomicAdd(cursor_d, (unsigned long long)c) : (unsigned long long)atomicAdd(cursor32_d, c);
        gdelta[threadIdx.x] = (uint32_t)g - (pre + inc - c);
    }
    __syncthreads();
}

// FULL: every position of the tile lies inside the text (all tiles but the last one): none of the per-slot checks
template <bool FULL>
__device__ __forceinline__ void msd_pack_split_body(const MsdArgs &a, uint64_t *buf, const uint8_t *cd, uint32_t *cnt, uint32_t *loff, uint32_t *gdelta,
                                                    uint32_t *wsum)
{
    uint8_t *tile = reinterpret_cast<uint8_t *>(buf);          // MSD_TILE + 64 bytes
    const int b = a.b, K = a.K;
    const uint64_t base = ((uint64_t)blockIdx.x + a.tile0) * MSD_TILE;     // (tile0: pass 1 may come in pieces, fbg_msd_pre_pass1)
    msd_load_tile<MSD_TILE, FULL>(tile, cd, a.T, a.N, base);
    __syncthreads();
    // keys of the thread's MSD_ITEMS consecutive positions
    const int t0 = threadIdx.x * MSD_ITEMS;
    uint64_t w[MSD_ITEMS];
    msd_build_keys(tile, t0, b, K, w);
    __syncthreads();                                            // the byte tile is done with: buf is free
    uint32_t rk[MSD_ITEMS];
    const int dshift = a.kb - MSD_DIG;
#pragma unroll
    for (int i = 0; i < MSD_ITEMS; i++) {
        const uint64_t p = base + t0 + i;
        const bool ok = FULL || p < a.N;
        rk[i] = ok ? atomicAdd(&cnt[(uint32_t)(w[i] >> dshift)], 1u) : 0xffffffffu;
        w[i] = (w[i] << a.pb) | p;
    }
    __syncthreads();
    // Workgroups b and b + 8 run on the same XCD (dispatch deals them round-robin).  With one stretch per (bucket, XCD)
    // the runs that neighbour each other in memory were written through the same L2, which merges their partial
    // 128-byte lines before they leave for HBM; runs of different XCDs leave every line they share twice, in parts.
    const uint32_t xq = a.xs > 1 ? (blockIdx.x & (uint32_t)(a.xs - 1)) : 0u;
    msd_scan_and_reserve(cnt, loff, gdelta, wsum, a.count1 + (threadIdx.x < MSD_NB ? threadIdx.x * a.xs + xq : 0), true, nullptr);
    const int wshift = a.pb + dshift;
    uint64_t *const obase = a.buf1 + xq * a.segcap;
#pragma unroll
    for (int i = 0; i < MSD_ITEMS; i++)
        if (FULL || rk[i] != 0xffffffffu) buf[loff[(uint32_t)(w[i] >> wshift)] + rk[i]] = w[i];
    __syncthreads();
    const uint32_t have = FULL ? (uint32_t)MSD_TILE : (uint32_t)min((uint64_t)MSD_TILE, a.N - base);
    bool over = false;
#pragma unroll
    for (int r = 0; r < MSD_ITEMS; r++) {
        const uint32_t j = threadIdx.x + r * MSD_THREADS;
        if (FULL || j < have) {
            const uint64_t x = buf[j];
            const uint32_t d = (uint32_t)(x >> wshift);
            const uint32_t at = j + gdelta[d];                  // below 2^32: the host checks cap1 + N
            if (at < a.segcap) obase[(uint64_t)d * a.cap1 + at] = x;
            else over = true;
        }
    }
    if (over) atomicOr(a.flag, 2ull);
}

__global__ __launch_bounds__(MSD_THREADS) void k_msd_pack_split(MsdArgs a)
{
    __shared__ uint64_t buf[MSD_TILE];         // first the symbol codes of the tile (bytes), then the regrouped slots
    __shared__ uint8_t cd[256];
    __shared__ uint32_t cnt[MSD_NB], loff[MSD_NB];
    __shared__ uint32_t gdelta[MSD_NB];
    __shared__ uint32_t wsum[MSD_THREADS / 64];
    if (threadIdx.x < 256) cd[threadIdx.x] = a.code[threadIdx.x];
    if (threadIdx.x < MSD_NB) cnt[threadIdx.x] = 0;
    __syncthreads();
    // (a body without the end-of-text checks for the tiles inside the text is slower here: 3.35 against 3.12 ms, same box,
    // where the same specialisation takes a sixth off pass 2; not taken)
    msd_pack_split_body<false>(a, buf, cd, cnt, loff, gdelta, wsum);
}

// first tile of every stretch of pass 1 (one workgroup of 1024 threads; nseg <= 4096 stretches, four per thread)
__global__ __launch_bounds__(1024) void k_msd_tiles(const unsigned long long *__restrict__ count1, uint64_t segcap, uint32_t nseg,
                                                    uint32_t *__restrict__ tile_start, unsigned long long *__restrict__ flag)
{
    __shared__ uint32_t wsum[16];
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t t[4], tot = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t s = threadIdx.x * 4 + q;
        t[q] = 0;
        if (s < nseg) {
            const unsigned long long c = count1[s] < segcap ? count1[s] : segcap;
            if (count1[s] > segcap) atomicOr(flag, 4ull);
            t[q] = (uint32_t)((c + MSD_TILE - 1) / MSD_TILE);
        }
        tot += t[q];
    }
    uint32_t inc = tot;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d, 64); if ((int)lane >= d) inc += o; }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    uint32_t pre = inc - tot;
    for (uint32_t q = 0; q < wv; q++) pre += wsum[q];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t s = threadIdx.x * 4 + q;
        if (s < nseg) tile_start[s] = pre;
        pre += t[q];
        if (s + 1 == nseg) tile_start[nseg] = pre;
    }
}

// FULL: a whole tile of MSD_TILE slots (all tiles of a stretch but its last one): none of the per-slot checks
template <bool FULL>
__device__ __forceinline__ void msd_split_body(const MsdArgs &a, uint64_t *buf, uint32_t *cnt, uint32_t *loff, uint32_t *gdelta, uint32_t *wsum,
                                               const uint64_t *__restrict__ in, uint32_t have, uint32_t seg)
{
    const int wshift = a.pb + a.kb - 2 * MSD_DIG;
    uint64_t w[MSD_ITEMS];
    uint32_t rk[MSD_ITEMS];
#pragma unroll
    for (int r = 0; r < MSD_ITEMS; r++) {
        const uint32_t j = threadIdx.x + r * MSD_THREADS;
        w[r] = (FULL || j < have) ? in[j] : 0ull;
    }
#pragma unroll
    for (int r = 0; r < MSD_ITEMS; r++) {
        const uint32_t j = threadIdx.x + r * MSD_THREADS;
        rk[r] = (FULL || j < have) ? atomicAdd(&cnt[(uint32_t)(w[r] >> wshift) & (MSD_NB - 1)], 1u) : 0u;
    }
    __syncthreads();
    msd_scan_and_reserve(cnt, loff, gdelta, wsum, nullptr, false,
                         a.count2 + (size_t)seg * MSD_NB + (threadIdx.x < MSD_NB ? threadIdx.x : 0));
#pragma unroll
    for (int r = 0; r < MSD_ITEMS; r++) {
        const uint32_t j = threadIdx.x + r * MSD_THREADS;
        if (FULL || j < have) buf[loff[(uint32_t)(w[r] >> wshift) & (MSD_NB - 1)] + rk[r]] = w[r];
    }
    __syncthreads();
    uint64_t *const obase = a.buf2 + (uint64_t)seg * MSD_NB * MSD_FN_CAP;
#pragma unroll
    for (int r = 0; r < MSD_ITEMS; r++) {
        const uint32_t j = threadIdx.x + r * MSD_THREADS;
        if (FULL || j < have) {
            const uint64_t x = buf[j];
            const uint32_t d = (uint32_t)(x >> wshift) & (MSD_NB - 1);
            const uint32_t at = j + gdelta[d];
            if (at < MSD_FN_CAP) obase[d * MSD_FN_CAP + at] = x;
            else {
                const unsigned long long e = atomicAdd(a.arena_count, 1ull);
                if (e < a.arena_cap) { a.arena_sb[e] = seg * MSD_NB + d; a.arena_w[e] = x; }
                else atomicOr(a.flag, 8ull);
            }
        }
    }
}

__global__ __launch_bounds__(MSD_THREADS) void k_msd_split(MsdArgs a)
{
    __shared__ uint64_t buf[MSD_TILE];
    __shared__ uint32_t cnt[MSD_NB], loff[MSD_NB];
    __shared__ uint32_t gdelta[MSD_NB];
    __shared__ uint32_t wsum[MSD_THREADS / 64];
    // The tile of this workgroup.  tiles2_x != 0: the workgroups of one XCD (b, b + 8, ...) take a contiguous range of the
    // tiles, i.e. whole buckets: all the runs a sub-bucket receives come through one L2, which merges the partial lines
    // where two runs meet (k_msd_pack_split has the other half of the story).
    const uint32_t tile = a.tiles2_x ? (blockIdx.x & 7u) * a.tiles2_x + (blockIdx.x >> 3) : blockIdx.x;
    if (tile >= a.tiles2) return;
    // the stretch this tile belongs to: largest s with tile_start[s] <= tile
    uint32_t lo = 0, hi = MSD_NB * (uint32_t)a.xs;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) / 2; if (a.tile_start[mid] <= tile) lo = mid; else hi = mid; }
    const uint32_t seg = lo / (uint32_t)a.xs;                    // the bucket
    const uint64_t segn = min((uint64_t)a.count1[lo], a.segcap);
    const uint64_t first = (uint64_t)(tile - a.tile_start[lo]) * MSD_TILE;
    const uint32_t have = (uint32_t)min((uint64_t)MSD_TILE, segn - first);
    const uint64_t *in = a.buf1 + (uint64_t)seg * a.cap1 + (uint64_t)(lo % (uint32_t)a.xs) * a.segcap + first;
    if (threadIdx.x < MSD_NB) cnt[threadIdx.x] = 0;
    __syncthreads();
    if (__builtin_amdgcn_readfirstlane((int)have) == MSD_TILE) msd_split_body<true>(a, buf, cnt, loff, gdelta, wsum, in, have, seg);   // (uniform, and told so: barriers inside)
    else msd_split_body<false>(a, buf, cnt, loff, gdelta, wsum, in, have, seg);
}

__global__ void k_msd_widen(const uint32_t *__restrict__ count2, unsigned long long *__restrict__ wide, uint64_t cnt)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cnt) wide[i] = count2[i];
}

// the `have` slots in w[] (slot j = threadIdx.x + r * THREADS in w[r]) -> buf, sorted: bins on key bits inside LDS, then every
// slot counts the smaller slots of its bin (the words are distinct)
#define MSD_BIG_BIN 192                        // a bin with more slots is split on its low key bits instead of counted through
#define MSD_LOW_BITS 10                        // ... when at most this many key bits lie below the bin bits
template <int CAP, int THREADS>
__device__ __forceinline__ void msd_finish_sort(uint64_t *buf, uint32_t *cnt, uint32_t *loff, uint32_t *wsum, uint64_t (&w)[CAP / THREADS],
                                                uint32_t have, int fshift, uint32_t fmask, int lowbits, uint32_t *sub, uint16_t *biglist,
                                                uint32_t *nbig_lds)
{
    constexpr int ITEMS = CAP / THREADS;
    for (int i = threadIdx.x; i < MSD_FN_BINS; i += THREADS) cnt[i] = 0;
    __syncthreads();
    uint32_t rk[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const uint32_t j = threadIdx.x + r * THREADS;
        rk[r] = j < have ? atomicAdd(&cnt[(uint32_t)(w[r] >> fshift) & fmask], 1u) : 0u;
    }
    __syncthreads();
    {   // exclusive scan of the bin counts: consecutive bins per thread (threads beyond the bins idle)
        constexpr int PER = MSD_FN_BINS / THREADS > 0 ? MSD_FN_BINS / THREADS : 1;
        const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        const bool mine = threadIdx.x * PER < MSD_FN_BINS;
        uint32_t c[PER], tot = 0;
#pragma unroll
        for (int q = 0; q < PER; q++) { c[q] = mine ? cnt[threadIdx.x * PER + q] : 0u; tot += c[q]; }
        uint32_t inc = tot;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d, 64); if ((int)lane >= d) inc += o; }
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        uint32_t pre = inc - tot;
        for (uint32_t q = 0; q < wv; q++) pre += wsum[q];
        if (mine) {
#pragma unroll
            for (int q = 0; q < PER; q++) { loff[threadIdx.x * PER + q] = pre; pre += c[q]; }
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const uint32_t j = threadIdx.x + r * THREADS;
        if (j < have) buf[loff[(uint32_t)(w[r] >> fshift) & fmask] + rk[r]] = w[r];
    }
    __syncthreads();
    // A bin of a few slots: every slot counts the smaller slots of its bin.  Similar rows put hundreds of EQUAL keys
    // into one bin (all rows of a column), where that count is quadratic: such a bin is split once more, on the key bits
    // below the bin bits (lowbits of them), by a histogram in LDS; equal keys keep the order in which they arrive --
    // nothing downstream depends on the order of equal keys.
    uint32_t nbig = 0;
    if (lowbits >= 1 && lowbits <= MSD_LOW_BITS) {
        if (threadIdx.x == 0) *nbig_lds = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < MSD_FN_BINS; i += THREADS)
            if (cnt[i] > MSD_BIG_BIN) { const uint32_t e = atomicAdd(nbig_lds, 1u); biglist[e] = (uint16_t)i; }   // at most CAP / MSD_BIG_BIN of them
        __syncthreads();
        nbig = *nbig_lds;
    }
    const uint32_t lowmask = (1u << (lowbits > 0 ? lowbits : 1)) - 1;
    for (uint32_t e = 0; e < nbig; e++) {                        // uniform over the workgroup
        const uint32_t bbin = biglist[e], b0 = loff[bbin];
        for (uint32_t i = threadIdx.x; i <= lowmask; i += THREADS) sub[i] = 0;
        __syncthreads();
        uint32_t idx[ITEMS];
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            const uint32_t j = threadIdx.x + r * THREADS;
            idx[r] = 0;
            if (j < have && ((uint32_t)(w[r] >> fshift) & fmask) == bbin) idx[r] = atomicAdd(&sub[(uint32_t)(w[r] >> (fshift - lowbits)) & lowmask], 1u);
        }
        __syncthreads();
        {   // exclusive scan of the (at most 2^MSD_LOW_BITS) counts: consecutive entries per thread
            constexpr int PER = (1 << MSD_LOW_BITS) / THREADS > 0 ? (1 << MSD_LOW_BITS) / THREADS : 1;
            const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
            uint32_t c[PER], tot = 0;
#pragma unroll
            for (int q = 0; q < PER; q++) { const uint32_t i = threadIdx.x * PER + q; c[q] = i <= lowmask ? sub[i] : 0u; tot += c[q]; }
            uint32_t inc = tot;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d, 64); if ((int)lane >= d) inc += o; }
            if (lane == 63) wsum[wv] = inc;
            __syncthreads();
            uint32_t pre = inc - tot;
            for (uint32_t q = 0; q < wv; q++) pre += wsum[q];
#pragma unroll
            for (int q = 0; q < PER; q++) { const uint32_t i = threadIdx.x * PER + q; if (i <= lowmask) sub[i] = pre; pre += c[q]; }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            const uint32_t j = threadIdx.x + r * THREADS;
            if (j < have && ((uint32_t)(w[r] >> fshift) & fmask) == bbin)
                rk[r] = b0 + sub[(uint32_t)(w[r] >> (fshift - lowbits)) & lowmask] + idx[r];
        }
        __syncthreads();
    }
    if (nbig) {
        // the slots of the crowded bins go to their places now: nobody else reads those bins' stretches
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            const uint32_t j = threadIdx.x + r * THREADS;
            if (j < have && cnt[(uint32_t)(w[r] >> fshift) & fmask] > MSD_BIG_BIN) buf[rk[r]] = w[r];
        }
        __syncthreads();
    }
    // Every slot counts the smaller slots of its bin -- taken in the order in which they lie in buf now, bin by bin: the lanes
    // of a wave sit in the same few bins, read the same words (no bank conflicts: with the slots in arrival order those cost
    // 45 % on top of the reads) and loop as often as the largest of their 30 bins has slots, not the largest of 64 bins
    // scattered over the sub-bucket.
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const uint32_t j = threadIdx.x + r * THREADS;
        if (j < have) {
            const uint64_t y = buf[j];
            const uint32_t bin = (uint32_t)(y >> fshift) & fmask;
            const uint32_t b0 = loff[bin], c = cnt[bin];
            uint32_t at = j;                                     // a crowded bin: placed above, or (no key bits left) its keys are equal
            if (!(c > MSD_BIG_BIN && (nbig || lowbits == 0))) {
                uint32_t smaller = 0;
                for (uint32_t q = 0; q < c; q++) smaller += buf[b0 + q] < y ? 1u : 0u;
                at = b0 + smaller;
            }
            w[r] = y;
            rk[r] = at;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const uint32_t j = threadIdx.x + r * THREADS;
        if (j < have) buf[rk[r]] = w[r];
    }
    __syncthreads();
}

// `have` slots (the first n_a from in_a, the rest from in_b) -> out, sorted
template <int CAP, int THREADS>
__device__ __forceinline__ void msd_finish_body(uint64_t *buf, uint32_t *cnt, uint32_t *loff, uint32_t *wsum, const uint64_t *in_a,
                                                uint32_t n_a, const uint64_t *in_b, uint32_t have, uint64_t *out, int fshift,
                                                uint32_t fmask, int lowbits, uint32_t *sub, uint16_t *biglist, uint32_t *nbig_lds)
{
    constexpr int ITEMS = CAP / THREADS;
    uint64_t w[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const uint32_t j = threadIdx.x + r * THREADS;
        w[r] = j < have ? (j < n_a ? in_a[j] : in_b[j - n_a]) : ~0ull;
    }
    msd_finish_sort<CAP, THREADS>(buf, cnt, loff, wsum, w, have, fshift, fmask, lowbits, sub, biglist, nbig_lds);
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const uint32_t j = threadIdx.x + r * THREADS;
        if (j < have) out[j] = buf[j];
    }
}

__global__ __launch_bounds__(MSD_FN_THREADS) void k_msd_finish(MsdArgs a, int fshift, uint32_t fmask)
{
    __shared__ uint64_t buf[MSD_FN_CAP];
    __shared__ uint32_t cnt[MSD_FN_BINS], loff[MSD_FN_BINS];
    __shared__ uint32_t wsum[MSD_FN_THREADS / 64];
    __shared__ uint32_t sub[1 << MSD_LOW_BITS];
    __shared__ uint16_t biglist[MSD_FN_CAP / MSD_BIG_BIN + 1];
    __shared__ uint32_t nbig_lds;
    const uint32_t have = a.count2[blockIdx.x];
    if (have == 0 || have > MSD_FN_CAP) return;                // the larger ones: k_msd_finish_big
    const uint64_t *in = a.buf2 + (uint64_t)blockIdx.x * MSD_FN_CAP;
    msd_finish_body<MSD_FN_CAP, MSD_FN_THREADS>(buf, cnt, loff, wsum, in, have, in, have, a.out + a.off[blockIdx.x], fshift, fmask,
                                                fshift - a.pb, sub, biglist, &nbig_lds);
}

// sub-buckets whose stretch overflowed: the arena (sorted by sub-bucket) holds the slots beyond MSD_FN_CAP.  One
// workgroup per arena entry; the first entry of a sub-bucket's run does the work.
__global__ __launch_bounds__(MSD_BIG_THREADS) void k_msd_finish_big(MsdArgs a, const uint32_t *__restrict__ sb_sorted,
                                                                    const uint64_t *__restrict__ w_sorted, uint32_t entries,
                                                                    int fshift, uint32_t fmask)
{
    __shared__ uint64_t buf[MSD_BIG_CAP];
    __shared__ uint32_t cnt[MSD_FN_BINS], loff[MSD_FN_BINS];
    __shared__ uint32_t wsum[MSD_BIG_THREADS / 64];
    __shared__ uint32_t sub[1 << MSD_LOW_BITS];
    __shared__ uint16_t biglist[MSD_BIG_CAP / MSD_BIG_BIN + 1];
    __shared__ uint32_t nbig_lds;
    const uint32_t e = blockIdx.x;
    if (e >= entries) return;
    const uint32_t sb = sb_sorted[e];
    if (e > 0 && sb_sorted[e - 1] == sb) return;
    const uint32_t have = a.count2[sb];
    if (have > MSD_BIG_CAP || have <= MSD_FN_CAP) { if (threadIdx.x == 0) atomicOr(a.flag, 16ull); return; }
    msd_finish_body<MSD_BIG_CAP, MSD_BIG_THREADS>(buf, cnt, loff, wsum, a.buf2 + (uint64_t)sb * MSD_FN_CAP, MSD_FN_CAP, w_sorted + e, have,
                                                  a.out + a.off[sb], fshift, fmask, fshift - a.pb, sub, biglist, &nbig_lds);
}

// Sorts the packed slots of the current text by their key bits.  *ok = 0: a capacity was exceeded (keys spread
// unevenly) or the geometry does not suit this sort -- nothing usable was produced.  On success *sorted points at
// the N sorted words (inside ctx->keysA).
// What the three passes work with, worked out from the key geometry and the text length; buffers reserved, counters zeroed
struct MsdPlan {
    KeyGeom g;
    MsdArgs a;
    uint64_t nsub;
    uint32_t nseg;
    int rest, fbits, xcd;
    uint32_t *tile_start;
    unsigned long long *off, *flag;
};
static_assert(sizeof(MsdPlan) <= sizeof(((fbg_ctx *)nullptr)->pre_state), "room for the plan of a sort begun ahead");

// *ok = 0: the geometry does not suit this sort
static int msd_plan(fbg_ctx *ctx, const KeyGeom &g, MsdPlan *p, int *ok)
{
    *ok = 0;
    const uint64_t N = ctx->N;
    const int rest = g.key_bits - 2 * MSD_DIG;                 // key bits left for the finish
    const uint64_t min_n = ctx->opt.msd_min >= 0 ? (uint64_t)ctx->opt.msd_min : (1ull << 24);   // tests lower it
    if (!g.packed || !g.compact || N < min_n || rest < 1 || ctx->opt.no_msd_sort) return FBG_OK;
    // places inside a stretch are kept in 32 bits with wrap-around arithmetic (k_msd_pack_split: j + gdelta[d]), slot numbers of the
    // scan behind it too: texts of 2^31 symbols and more go to rocPRIM's sort (what the arithmetic itself needs is
    // cap1 + MSD_TILE < 2^32 -- the round limit is the documented one)
    if (N >= (1ull << 31)) return FBG_OK;
    const int fbits = rest < MSD_FN_BITS ? rest : MSD_FN_BITS;
    // without enough bits for the bins the counting in the finish turns quadratic: leave those to rocPRIM
    if (fbits < 6) return FBG_OK;
    hipStream_t st = ctx->stream;
    const int xcd = ctx->opt.msd_xcd < 0 ? 3 : (int)ctx->opt.msd_xcd;
    const int xs = (xcd & 2) ? 8 : 1;
    const uint64_t cap1 = (N / MSD_NB + N / (4 * MSD_NB) + 65536) & ~127ull;   // a multiple of 8 stretches of 16 slots
    const uint64_t nsub = (uint64_t)MSD_NB * MSD_NB;
    if (N / nsub + N / (32 * nsub) + 64 > MSD_FN_CAP) return FBG_OK;        // sub-buckets would not fit their stretches
    FBG_TRY(fbg_reserve(ctx, ctx->keysA, (size_t)MSD_NB * cap1 * 8));
    FBG_TRY(fbg_reserve(ctx, ctx->keysB, (size_t)nsub * MSD_FN_CAP * 8));
    const uint32_t nseg = MSD_NB * (uint32_t)xs;
    FBG_TRY(fbg_reserve(ctx, ctx->dp_a, (size_t)nseg * 8 + ((size_t)nseg + 1) * 4 + 64));
    FBG_TRY(fbg_reserve(ctx, ctx->dp_b, nsub * 4));
    FBG_TRY(fbg_reserve(ctx, ctx->dp_c, (nsub + 1) * 8));
    FBG_TRY(fbg_reserve(ctx, ctx->dp_d, (size_t)MSD_ARENA * 4 * 2));
    FBG_TRY(fbg_reserve(ctx, ctx->dp_e, (size_t)MSD_ARENA * 8 * 2));
    unsigned long long *flag = ctx->scalars.as<unsigned long long>() + 100;
    MsdArgs &a = p->a;
    a.T = ctx->text.as<uint8_t>(); a.N = N; a.code = g.d_code; a.b = g.b; a.K = g.K; a.pb = g.pb; a.kb = g.key_bits;
    a.buf1 = ctx->keysA.as<uint64_t>(); a.cap1 = cap1; a.xs = xs; a.segcap = cap1 / xs;
    a.tile0 = 0; a.tiles2 = 0; a.tiles2_x = 0;
    a.count1 = ctx->dp_a.as<unsigned long long>();
    p->tile_start = reinterpret_cast<uint32_t *>(ctx->dp_a.as<uint8_t>() + (size_t)nseg * 8);
    a.tile_start = p->tile_start;
    a.buf2 = ctx->keysB.as<uint64_t>();
    a.count2 = ctx->dp_b.as<uint32_t>();
    p->off = ctx->dp_c.as<unsigned long long>();
    a.off = p->off;
    a.out = ctx->keysA.as<uint64_t>();         // pass 1's slots are dead by then
    a.flag = flag;
    a.arena_sb = ctx->dp_d.as<uint32_t>(); a.arena_w = ctx->dp_e.as<uint64_t>(); a.arena_count = flag + 1; a.arena_cap = MSD_ARENA;
    p->g = g; p->nsub = nsub; p->nseg = nseg; p->rest = rest; p->fbits = fbits; p->xcd = xcd; p->flag = flag;
    FBG_HIP_TRY(ctx, hipMemsetAsync(flag, 0, 16, st));
    FBG_HIP_TRY(ctx, hipMemsetAsync(a.count1, 0, (size_t)nseg * 8, st));
    FBG_HIP_TRY(ctx, hipMemsetAsync(a.count2, 0, nsub * 4, st));
    *ok = 1;
    return FBG_OK;
}

// Pass 1 ahead of the rest of the index build, while the MSA is still arriving (text_build.hip, streamed upload): the keys are set
// up for the alphabet ctx->byte_hist promises (the first chunk of rows, scaled); fbg_build_text checks the promise at the end
// (ctx->pre_pass1) and fbg_msd_sort picks the plan up from ctx->pre_state.
int fbg_msd_pre_begin(fbg_ctx *ctx, int *ok)
{
    *ok = 0;
    if (ctx->have_ignore || ctx->reversed || ctx->opt.no_ranked) return FBG_OK;
    int launches = 0;
    KeyGeom g;
    FBG_TRY(fbg_key_setup(ctx, true, &g, &launches));
    MsdPlan plan;
    int good = 0;
    FBG_TRY(msd_plan(ctx, g, &plan, &good));
    if (!good) return FBG_OK;
    for (int w = 0; w < 4; w++) ctx->pre_symbols[w] = 0;
    for (int b = 0; b < 256; b++) if (ctx->byte_hist[b]) ctx->pre_symbols[b >> 6] |= 1ull << (b & 63);
    memcpy(ctx->pre_state, &plan, sizeof(plan));
    ctx->pre_tiles = 0;
    *ok = 1;
    return FBG_OK;
}

int fbg_msd_pre_pass1(fbg_ctx *ctx, uint64_t avail)
{
    MsdPlan plan;
    memcpy(&plan, ctx->pre_state, sizeof(plan));
    const uint64_t N = plan.a.N;
    // a tile reads MSD_TILE + 64 text positions
    const uint64_t upto = avail >= N ? (N + MSD_TILE - 1) / MSD_TILE : (avail >= 64 ? (avail - 64) / MSD_TILE : 0);
    if (upto <= ctx->pre_tiles) return FBG_OK;
    plan.a.tile0 = ctx->pre_tiles;
    hipLaunchKernelGGL(k_msd_pack_split, dim3((unsigned)(upto - ctx->pre_tiles)), dim3(MSD_THREADS), 0, ctx->stream, plan.a);
    ctx->pre_tiles = upto;
    return FBG_OK;
}

bool fbg_msd_pre_geom(fbg_ctx *ctx, KeyGeom *g)
{
    if (!ctx->pre_pass1) return false;
    MsdPlan plan;
    memcpy(&plan, ctx->pre_state, sizeof(plan));
    *g = plan.g;
    return true;
}

int fbg_msd_sort(fbg_ctx *ctx, const KeyGeom &g, uint64_t **sorted, int *ok, int *launches)
{
    *ok = 0;
    ctx->msd_decline = 1;                                       // the geometry does not suit this sort
    const uint64_t N = ctx->N;
    hipStream_t st = ctx->stream;
    MsdPlan plan;
    const bool ahead = ctx->pre_pass1;                          // pass 1 ran while the MSA was uploaded (same g: fbg_msd_pre_geom)
    ctx->pre_pass1 = false;
    ctx->pass1_ahead = ahead ? 1 : 0;
    if (ahead) memcpy(&plan, ctx->pre_state, sizeof(plan));
    else {
        int good = 0;
        FBG_TRY(msd_plan(ctx, g, &plan, &good));
        if (!good) return FBG_OK;
    }
    MsdArgs &a = plan.a;
    const uint64_t nsub = plan.nsub;
    const uint32_t nseg = plan.nseg;
    const int rest = plan.rest, fbits = plan.fbits, xcd = plan.xcd;
    uint32_t *tile_start = plan.tile_start;
    unsigned long long *off = plan.off, *flag = plan.flag;
    if (!ahead) {
        FBG_TRY(fbg_stage_begin(ctx, FBG_STAGE_SORT_PASS1));
        hipLaunchKernelGGL(k_msd_pack_split, dim3(fbg_blocks(N, MSD_TILE)), dim3(MSD_THREADS), 0, st, a);
        FBG_TRY(fbg_stage_end(ctx, FBG_STAGE_SORT_PASS1, 1));
    }
    hipLaunchKernelGGL(k_msd_tiles, dim3(1), dim3(1024), 0, st, a.count1, a.segcap, nseg, tile_start, flag);
    uint32_t tiles2 = 0;
    unsigned long long h_flag = 0;
    FBG_HIP_TRY(ctx, hipMemcpyAsync(&tiles2, tile_start + nseg, 4, hipMemcpyDeviceToHost, st));
    FBG_HIP_TRY(ctx, hipMemcpyAsync(&h_flag, flag, 8, hipMemcpyDeviceToHost, st));
    FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
    *launches += 2;
    ctx->msd_decline = (int64_t)h_flag;                         // (fbg_get_option "msd_decline": which capacity did not hold)
    if (h_flag != 0 || tiles2 == 0) return FBG_OK;
    a.tiles2 = tiles2;
    a.tiles2_x = (xcd & 1) ? (tiles2 + 7) / 8 : 0;
    const unsigned grid2 = a.tiles2_x ? 8 * a.tiles2_x : tiles2;
    FBG_TRY(fbg_stage_begin(ctx, FBG_STAGE_SORT_PASS2));
    hipLaunchKernelGGL(k_msd_split, dim3(grid2), dim3(MSD_THREADS), 0, st, a);
    FBG_TRY(fbg_stage_end(ctx, FBG_STAGE_SORT_PASS2, 1));
    // offsets of the sub-buckets in the sorted array: exclusive scan of their sizes
    unsigned long long *wide = reinterpret_cast<unsigned long long *>(ctx->keysA.p);   // scratch: pass 1's slots are dead
    hipLaunchKernelGGL(k_msd_widen, dim3(fbg_blocks(nsub, 256)), dim3(256), 0, st, a.count2, wide, nsub);
    {
        size_t bytes = 0;
        hipError_t e = rocprim::exclusive_scan(nullptr, bytes, wide, off, 0ull, (size_t)nsub, rocprim::plus<unsigned long long>(), st);
        if (e != hipSuccess) return fbg_fail(ctx, FBG_ERR_HIP, "rocprim scan size query: %s", hipGetErrorString(e));
        FBG_TRY(fbg_reserve(ctx, ctx->tmp, bytes));
        size_t have = ctx->tmp.cap;
        e = rocprim::exclusive_scan(ctx->tmp.p, have, wide, off, 0ull, (size_t)nsub, rocprim::plus<unsigned long long>(), st);
        if (e != hipSuccess) return fbg_fail(ctx, FBG_ERR_HIP, "rocprim exclusive_scan: %s", hipGetErrorString(e));
    }
    const int fshift = g.pb + rest - fbits;
    const uint32_t fmask = (uint32_t)((1u << fbits) - 1);
    FBG_TRY(fbg_stage_begin(ctx, FBG_STAGE_SORT_PASS3));
    hipLaunchKernelGGL(k_msd_finish, dim3((unsigned)nsub), dim3(MSD_FN_THREADS), 0, st, a, fshift, fmask);
    FBG_TRY(fbg_stage_end(ctx, FBG_STAGE_SORT_PASS3, 1));
    unsigned long long h2[2] = {0, 0};
    FBG_HIP_TRY(ctx, hipMemcpyAsync(h2, flag, 16, hipMemcpyDeviceToHost, st));
    FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
    *launches += 4;
    ctx->msd_decline = (int64_t)h2[0];
    if (h2[0] != 0) return FBG_OK;
    if (h2[1] > 0) {
        // a few sub-buckets were larger than their stretch (row ends pile up on keys that end in zeros): their
        // overflow sits in the arena; sorted by sub-bucket, every run joins the MSD_FN_CAP slots that did fit
        const uint32_t entries = (uint32_t)h2[1];
        uint32_t *sb_sorted = a.arena_sb + MSD_ARENA;
        uint64_t *w_sorted = a.arena_w + MSD_ARENA;
        size_t bytes = 0;
        hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, a.arena_sb, sb_sorted, a.arena_w, w_sorted, (size_t)entries, 0u, 32u, st);
        if (e != hipSuccess) return fbg_fail(ctx, FBG_ERR_HIP, "rocprim sort size query: %s", hipGetErrorString(e));
        FBG_TRY(fbg_reserve(ctx, ctx->tmp, bytes));
        size_t have = ctx->tmp.cap;
        e = rocprim::radix_sort_pairs(ctx->tmp.p, have, a.arena_sb, sb_sorted, a.arena_w, w_sorted, (size_t)entries, 0u, 32u, st);
        if (e != hipSuccess) return fbg_fail(ctx, FBG_ERR_HIP, "rocprim radix_sort_pairs: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(k_msd_finish_big, dim3(entries), dim3(MSD_BIG_THREADS), 0, st, a, sb_sorted, w_sorted, entries, fshift, fmask);
        FBG_HIP_TRY(ctx, hipMemcpyAsync(h2, flag, 16, hipMemcpyDeviceToHost, st));
        FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
        *launches += 2;
        ctx->msd_decline = (int64_t)h2[0];
        if (h2[0] != 0) return FBG_OK;
    }
    FBG_HIP_TRY(ctx, hipGetLastError());
    *sorted = ctx->keysA.as<uint64_t>();
    *ok = 1;
    return FBG_OK;
}
