// msd_sort_pairs.hip -- the three-pass MSD sort (msd_sort.hip) for the partitioned index: 12-byte slots
// (key word u64 + low position word u32; pairs / wide layouts) and only the suffixes whose key lies in this
// partition's range [lo, hi).
//
// Buckets cannot be bit fields of the key here (the range is arbitrary): t = floor((key - lo) * 2^28 / (hi - lo)),
// one 64x64->128 multiply, is monotone in the key, and its three 9-bit digits serve as bucket, sub-bucket and finish
// bin.  Pass 1 is fused with key packing and filtering: a workgroup walks as many 4096-position tiles as there are
// partitions, so that it collects about 4096 slots of its range before it regroups and writes them.  Passes 2 and 3
// are those of msd_sort.hip with a second array riding along.  Capacities are optimistic in the same way; a flag
// sends the caller back to k_pack<FILTER> + rocPRIM.
//
// MODE 1 (fbg_sample_sort_pairs): the same three passes for the record path's whole text, whose keys are NOT spread
// evenly (order-preserving 3-bit codes of a 5- or 7-letter alphabet use a fraction of the key space: 27 leading bits
// carry 18 bits of entropy, and a linear bucket function sends everything to a few buckets).  Buckets and sub-buckets
// are then the 512 * 512 quantiles of a sorted sample of 2^22 keys (a sample sort): a slot's bucket is the number of
// grid keys G[512], G[1024], ... not above its key (binary search in LDS), its sub-bucket likewise within the bucket's
// 511 inner grid keys, and the finish bins the keys of a sub-bucket linearly between its two grid keys.  Equal keys go
// the same way at every level; a key so frequent that it swallows grid points overflows a capacity and sends the caller
// to rocPRIM like any other overflow.
#include "fbg_internal.h"
#include "msd_keys.h"
#include <rocprim/rocprim.hpp>

#define PP_NB 512
#define PP_THREADS 512
#define PP_TILE (PP_THREADS * MSD_ITEMS)       // text positions per sub-tile of pass 1, slots per tile of pass 2
#define PP_STAGE 4608                          // slots a workgroup of pass 1 collects (expected 4096)
#define PP_FN_CAP 4608                         // slots per sub-bucket
#define PP_ITEMS (PP_FN_CAP / PP_THREADS)
#define PP_BIG_CAP 9216                        // largest sub-bucket (k_pp_finish_big)
#define PP_ARENA (1u << 20)
#define PP_FN_SMALL 3072                      // LDS capacity of the finish variant for sparsely filled stretches (k_pp_finish)
#define PP_FN_TINY 2048                       // ... and of the one for stretches filled to a fifth (similar rows, texts of 2 * 10^8 symbols: four workgroups per CU)
#define PP_CROWD 160                           // MODE 1 finish: a bin of more slots than this is split once more

struct PpArgs {
    const uint8_t *T;
    uint64_t N;
    const uint8_t *code;
    int b, K, pb, nparts;
    int wide;                                  // key word = key << pb | position >> 32 (else the plain key, pb = 0)
    int packed;                                // 8-byte slots: key << pb | position, no second array (texts below 2^32)
    uint64_t lo, hi, mul;                      // key range (hi ignored when nohi), t = umul64hi(key - lo, mul) < 2^28
    int nohi;
    uint64_t *w1; uint32_t *v1; uint64_t cap1; unsigned long long *count1;      // pass 1 output: PP_NB stretches
    int xs;                                    // ... each in xs parts of segcap = cap1 / xs slots (8: one per XCD, as in msd_sort.hip; count1[PP_NB * xs])
    uint64_t segcap;
    uint32_t tiles2, tiles2_x;                 // pass 2: tiles in all; tiles per XCD when the workgroups of an XCD take a range of tiles (0: launch order)
    const uint32_t *tile_start;
    uint64_t *w2; uint32_t *v2; uint32_t *count2;                               // pass 2 output: PP_NB^2 stretches of PP_FN_CAP
    const unsigned long long *off;
    uint64_t *wout; uint32_t *vout;
    uint32_t *arena_sb; uint64_t *arena_w; uint32_t *arena_v; unsigned long long *arena_count;
    unsigned long long *flag;
    const uint64_t *grid;                      // MODE 1: G[PP_NB * PP_NB] quantiles of the key sample (G[0] is not used: below every key)
    uint64_t top;                              // MODE 1: 2^key_bits, the end of the last sub-bucket's key range
    int sigma;                                 // MODE 1: number of symbol codes in use (codes are 0 .. sigma - 1)
    const uint64_t *ebits;                     // pairs: bit 31 of the value = an irregular position among the K from p on (gapped_rank.hip)
    const uint32_t *payload;                   // pairs, MODE 1: the value of position p is payload[p] (span_scan.hip: cell | flags), not p
    int any_order;                             // MODE 1: slots with equal keys may come out in any order (the consumer works on groups of equal keys)
    int probe;                                 // timing probes of k_pp_finish (option msd_probe): 1 copy only, 2 no pass over the crowded bins, 4 no counting
    // MODE 1 finish: bins from the symbols behind the prefix a sub-bucket's keys share, every symbol code replaced by its
    // rank among the FREQUENT symbols (rb bits; 0: bins from sampled keys instead).  rtab: ew bits per code = the rank;
    // mtab: 2 bits per code = 0 a frequent symbol, 1 / 2 a rare one below / above the frequent symbol whose rank it shares
    uint64_t rtab, mtab;
    int rb, ew, nd, key_bits;
};

// The bin of a key in the finish of the sample sort (MODE 1), sub-bucket keys in [glo, ghi].  The nd symbols from symbol
// cp on (the symbols before are the same in glo and ghi, hence in every key between), each as its rank among the frequent
// symbols, read as one number: for a random text those numbers are spread evenly where the keys themselves are not (the
// 3-bit codes of a 4-letter text use 4 of 8 values per digit), so a sub-bucket's keys fill the range between the
// numbers of glo and ghi evenly and the bin is (number - number(glo)) >> shift.  Monotone in the key: a rare symbol
// ends the number (the remaining digits all 0 when it sorts below the frequent symbol it shares the rank with, all ones
// when above).  nd * rb = 30 bits: 18 for the 2^18 sub-buckets, 12 to cut one into bins.
__device__ __forceinline__ uint64_t pp_rank_number(const PpArgs &a, uint64_t key, int cp)
{
    const uint32_t dmask = (1u << a.b) - 1, emask = (1u << a.ew) - 1, full = (1u << a.rb) - 1;
    uint64_t num = 0;
    uint32_t stop = 0;
#pragma unroll 5
    for (int i = 0; i < a.nd; i++) {
        const int sym = cp + i;
        uint32_t digit = 0;
        if (stop == 0 && sym < a.K) {
            const uint32_t d = (uint32_t)(key >> (a.b * (a.K - 1 - sym))) & dmask;
            digit = (uint32_t)(a.rtab >> (a.ew * d)) & emask;
            stop = (uint32_t)(a.mtab >> (2 * d)) & 3u;
        } else if (stop == 2) digit = full;
        num = (num << a.rb) | digit;
    }
    return num;
}

// The same number, three symbols per step: tab3[the 3 b bits of three symbols] = their 3 rb digits | the stop state behind them << 12
// (pp_rank_table3 fills it; b <= 3).  15 symbols of a DNA key in 5 table reads instead of 15 rounds of shifts and selects --
// the bin function was two thirds of k_pp_finish's instructions.
__device__ __forceinline__ void pp_rank_table3(const PpArgs &a, uint16_t *tab3)
{
    const uint32_t dmask = (1u << a.b) - 1, emask = (1u << a.ew) - 1, full = (1u << a.rb) - 1;
    for (uint32_t t = threadIdx.x; t < (1u << (3 * a.b)); t += blockDim.x) {
        uint32_t num = 0, stop = 0;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            uint32_t digit = 0;
            if (stop == 0) {
                const uint32_t d = (t >> (a.b * (2 - i))) & dmask;
                digit = (uint32_t)(a.rtab >> (a.ew * d)) & emask;
                stop = (uint32_t)(a.mtab >> (2 * d)) & 3u;
            } else if (stop == 2) digit = full;
            num = (num << a.rb) | digit;
        }
        tab3[t] = (uint16_t)(num | (stop << 12));
    }
}
__device__ __forceinline__ uint64_t pp_rank_number3(const PpArgs &a, const uint16_t *tab3, uint64_t key, int cp)
{
    const int valid = min(a.nd, a.K - cp);                      // the symbols there are from cp on (uniform over the workgroup)
    const uint32_t dmask = (1u << a.b) - 1, emask = (1u << a.ew) - 1, full = (1u << a.rb) - 1;
    const uint32_t tmask = (1u << (3 * a.b)) - 1, full3 = (1u << (3 * a.rb)) - 1;
    uint64_t num = 0;
    uint32_t stop = 0;
    int i = 0;
    for (; i + 3 <= valid; i += 3) {
        const uint32_t e = tab3[(uint32_t)(key >> (a.b * (a.K - 3 - (cp + i)))) & tmask];
        const uint32_t dig = stop == 0 ? (e & full3) : (stop == 2 ? full3 : 0u);
        stop = stop == 0 ? (e >> 12) : stop;
        num = (num << (3 * a.rb)) | dig;
    }
    for (; i < a.nd; i++) {                                     // what three do not divide, and the digits beyond the key's end
        const int sym = cp + i;
        uint32_t digit = 0;
        if (stop == 0 && sym < a.K) {
            const uint32_t d = (uint32_t)(key >> (a.b * (a.K - 1 - sym))) & dmask;
            digit = (uint32_t)(a.rtab >> (a.ew * d)) & emask;
            stop = (uint32_t)(a.mtab >> (2 * d)) & 3u;
        } else if (stop == 2) digit = full;
        num = (num << a.rb) | digit;
    }
    return num;
}

// t < 2^28: bucket (9 bits), sub-bucket (9 bits), finish bin (10 bits)
__device__ __forceinline__ uint32_t pp_t28(const PpArgs &a, uint64_t word) { return (uint32_t)__umul64hi((word >> a.pb) - a.lo, a.mul); }
#define PP_FBINS 1024
#define PP_SMP 127                             // MODE 1: sample keys (bin boundaries) of a sub-bucket's finish: 128 stretches x 8 bins

// MODE 1: number of grid keys sp[1..511] not above the key (sp[0] counts as below every key): 9 steps, no branches
__device__ __forceinline__ uint32_t pp_rank511(const uint64_t *sp, uint64_t key)
{
    uint32_t at = 0;
#pragma unroll
    for (uint32_t step = 256; step >= 1; step >>= 1)
        if (sp[at + step] <= key) at += step;
    return at;
}

// the digit of regrouped slot j: the last d with loff[d] <= j (its successor starts beyond j, so it is not empty)
__device__ __forceinline__ uint32_t pp_digit_of_slot(const uint32_t *loff, uint32_t j)
{
    uint32_t at = 0;
#pragma unroll
    for (uint32_t step = 256; step >= 1; step >>= 1)
        if (loff[at + step] <= j) at += step;
    return at;
}

// scan of PP_NB counts (one per thread) + run reservation, as msd_scan_and_reserve
__device__ __forceinline__ void pp_scan_and_reserve(uint32_t *cnt, uint32_t *loff, unsigned long long *gbase, uint32_t *wsum,
                                                    unsigned long long *cursor64, uint32_t *cursor32, int stride64 = 1)
{
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t c = cnt[threadIdx.x];                       // PP_THREADS == PP_NB
    uint32_t inc = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d, 64); if ((int)lane >= d) inc += o; }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    uint32_t pre = 0;
    for (uint32_t q = 0; q < wv; q++) pre += wsum[q];
    loff[threadIdx.x] = pre + inc - c;
    unsigned long long g = 0;
    if (c) g = cursor64 ? atomicAdd(cursor64 + (size_t)threadIdx.x * stride64, (unsigned long long)c) : (unsigned long long)atomicAdd(cursor32 + threadIdx.x, c);
    gbase[threadIdx.x] = g - (pre + inc - c);                  // minus the run's start in the tile: the write-out adds the slot's place (wraps)
    __syncthreads();
}

// `have` slots sitting in (sw, sv) in any order -> regrouped by digit d(word) in place (through registers), counts in
// cnt (zeroed by the caller), offsets in loff
template <int CAP, class Digit>
__device__ __forceinline__ void pp_regroup(uint64_t *sw, uint32_t *sv, uint32_t *cnt, uint32_t have, Digit digit, uint64_t *w, uint32_t *v,
                                           uint32_t *rk, uint32_t *dg)
{
    constexpr int ITEMS = CAP / PP_THREADS;
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const uint32_t j = threadIdx.x + r * PP_THREADS;
        w[r] = j < have ? sw[j] : ~0ull;
        v[r] = j < have ? sv[j] : 0u;
        dg[r] = j < have ? digit(w[r]) : 0u;
        rk[r] = j < have ? atomicAdd(&cnt[dg[r]], 1u) : 0u;
    }
    __syncthreads();
}

template <int MODE> __global__ __launch_bounds__(PP_THREADS) void k_pp_pack_split(PpArgs a)
{
    __shared__ uint64_t sp[MODE == 1 ? PP_NB : 1];         // MODE 1: the bucket splitters G[512 j]
    __shared__ uint64_t sw[PP_STAGE];
    __shared__ uint32_t sv[PP_STAGE];
    __shared__ uint16_t sd[MODE == 1 ? PP_STAGE : 1];
    __shared__ uint8_t tile[PP_TILE + 64];
    __shared__ uint8_t cd[256];
    __shared__ uint32_t cnt[PP_NB], loff[PP_NB];
    __shared__ unsigned long long gbase[PP_NB];
    __shared__ uint32_t wsum[PP_THREADS / 64];
    __shared__ uint32_t staged;
    if (threadIdx.x < 256) cd[threadIdx.x] = a.code[threadIdx.x];
    cnt[threadIdx.x] = 0;
    if (threadIdx.x == 0) staged = 0;
    if (MODE == 1) sp[threadIdx.x] = threadIdx.x ? a.grid[(size_t)PP_NB * threadIdx.x] : 0ull;
    const int lane = threadIdx.x & 63;
    const uint64_t base0 = (uint64_t)blockIdx.x * a.nparts * PP_TILE;
    uint64_t raw = 0, raw2 = 0;
    if (base0 < a.N) msd_fetch_raw<PP_TILE>(a.T, a.N, base0, raw, raw2);
    for (int sub = 0; sub < a.nparts; sub++) {
        const uint64_t base = base0 + (uint64_t)sub * PP_TILE;
        __syncthreads();                                        // the tile of the previous round is done with
        if (base >= a.N) continue;                              // uniform
        msd_store_tile<PP_TILE>(tile, cd, a.N, base, raw, raw2);
        if (sub + 1 < a.nparts && base + PP_TILE < a.N) msd_fetch_raw<PP_TILE>(a.T, a.N, base + PP_TILE, raw, raw2);   // in flight during this tile's work
        __syncthreads();
        uint64_t key[MSD_ITEMS];
        msd_build_keys(tile, threadIdx.x * MSD_ITEMS, a.b, a.K, key);
        // the slots this thread keeps, then one place reservation per wave (not one per item)
        uint32_t km = 0;
#pragma unroll
        for (int i = 0; i < MSD_ITEMS; i++) {
            const uint64_t p = base + (uint64_t)threadIdx.x * MSD_ITEMS + i;
            if (p < a.N && key[i] >= a.lo && (a.nohi || key[i] < a.hi)) km |= 1u << i;
        }
        const uint32_t mine = (uint32_t)__popc(km);
        uint32_t inc = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
        uint32_t at = 0;
        if (lane == 63 && inc) at = atomicAdd(&staged, inc);
        at = __shfl(at, 63, 64) + inc - mine;
        uint64_t eb = 0;                                        // the irregular-position bits of [p0, p0 + 64), p0 the thread's first position
        if (a.ebits) {
            const uint64_t p0 = base + (uint64_t)threadIdx.x * MSD_ITEMS;
            const uint64_t *e = a.ebits + (p0 >> 6);
            const unsigned sh = (unsigned)(p0 & 63);
            eb = e[0] >> sh;
            if (sh) eb |= e[1] << (64 - sh);
        }
#pragma unroll
        for (int i = 0; i < MSD_ITEMS; i++) {
            if ((km >> i) & 1u) {
                const uint64_t p = base + (uint64_t)threadIdx.x * MSD_ITEMS + i;
                if (at < PP_STAGE) {
                    sw[at] = a.packed ? (key[i] << a.pb) | p : a.wide ? (key[i] << a.pb) | (p >> 32) : key[i];
                    sv[at] = a.packed ? 0u : a.payload ? a.payload[p] : (uint32_t)p | (((eb >> i) & ((1ull << a.K) - 1)) ? 0x80000000u : 0u);
                } else *a.flag = 1;
                at++;
            }
        }
    }
    __syncthreads();
    const uint32_t have = min(staged, (uint32_t)PP_STAGE);
    uint64_t w[PP_STAGE / PP_THREADS];
    uint32_t v[PP_STAGE / PP_THREADS], rk[PP_STAGE / PP_THREADS];
    uint32_t dg[PP_STAGE / PP_THREADS];
    pp_regroup<PP_STAGE>(sw, sv, cnt, have, [&](uint64_t x) { return MODE == 1 ? pp_rank511(sp, x >> a.pb) : pp_t28(a, x) >> 19; }, w, v, rk, dg);
    // a stretch per (bucket, XCD): workgroups b and b + 8 share an XCD, whose L2 merges the partial lines where their runs
    // meet (msd_sort.hip, k_msd_pack_split)
    const uint32_t xq = a.xs > 1 ? (blockIdx.x & (uint32_t)(a.xs - 1)) : 0u;
    pp_scan_and_reserve(cnt, loff, gbase, wsum, a.count1 + xq, nullptr, a.xs);
#pragma unroll
    for (int r = 0; r < PP_STAGE / PP_THREADS; r++) {
        const uint32_t j = threadIdx.x + r * PP_THREADS;
        if (j < have) { const uint32_t at = loff[dg[r]] + rk[r]; sw[at] = w[r]; sv[at] = v[r]; if (MODE == 1) sd[at] = (uint16_t)dg[r]; }
    }
    __syncthreads();
    const uint64_t xoff = (uint64_t)xq * a.segcap;
#pragma unroll
    for (int r = 0; r < PP_STAGE / PP_THREADS; r++) {
        const uint32_t j = threadIdx.x + r * PP_THREADS;
        if (j < have) {
            const uint64_t x = sw[j];
            const uint32_t d = MODE == 1 ? sd[j] : pp_t28(a, x) >> 19;       // (MODE 1: the digit travels with the slot; finding it again in loff was nine LDS reads)
            const uint64_t at = gbase[d] + j;
            if (at < a.segcap) { a.w1[(uint64_t)d * a.cap1 + xoff + at] = x; if (!a.packed) a.v1[(uint64_t)d * a.cap1 + xoff + at] = sv[j]; }
            else *a.flag = 1;
        }
    }
}

// first tile of every stretch of pass 1 and the number of slots in all (one workgroup of 1024 threads, nseg <= 4096)
__global__ __launch_bounds__(1024) void k_pp_tiles(const unsigned long long *__restrict__ count1, uint64_t segcap, uint32_t nseg,
                                                   uint32_t *__restrict__ tile_start, unsigned long long *__restrict__ total,
                                                   unsigned long long *__restrict__ flag)
{
    __shared__ uint32_t wsum[16];
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t t[4], tot = 0;
    unsigned long long sum = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t s = threadIdx.x * 4 + q;
        t[q] = 0;
        if (s < nseg) {
            if (count1[s] > segcap) *flag = 1;
            const unsigned long long c = count1[s] < segcap ? count1[s] : segcap;
            sum += c;
            t[q] = (uint32_t)((c + PP_TILE - 1) / PP_TILE);
        }
        tot += t[q];
    }
    if (sum) atomicAdd(total, sum);                            // zeroed by the caller
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

template <int MODE> __global__ __launch_bounds__(PP_THREADS) void k_pp_split(PpArgs a)
{
    __shared__ uint64_t sp[MODE == 1 ? PP_NB : 1];         // MODE 1: the inner grid keys of this tile's bucket
    __shared__ uint64_t sw[PP_TILE];
    __shared__ uint32_t sv[PP_TILE];
    __shared__ uint16_t sd[MODE == 1 ? PP_TILE : 1];
    __shared__ uint32_t cnt[PP_NB], loff[PP_NB];
    __shared__ unsigned long long gbase[PP_NB];
    __shared__ uint32_t wsum[PP_THREADS / 64];
    // tiles2_x != 0: the workgroups of one XCD take a range of the tiles, whole buckets (msd_sort.hip, k_msd_split)
    const uint32_t tile = a.tiles2_x ? (blockIdx.x & 7u) * a.tiles2_x + (blockIdx.x >> 3) : blockIdx.x;
    if (tile >= a.tiles2) return;
    uint32_t lo = 0, hi = PP_NB * (uint32_t)a.xs;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) / 2; if (a.tile_start[mid] <= tile) lo = mid; else hi = mid; }
    const uint32_t seg = lo / (uint32_t)a.xs;
    const uint64_t segn = min((uint64_t)a.count1[lo], a.segcap);
    const uint64_t first = (uint64_t)(tile - a.tile_start[lo]) * PP_TILE;
    const uint32_t have = (uint32_t)min((uint64_t)PP_TILE, segn - first);
    const uint64_t in0 = (uint64_t)seg * a.cap1 + (uint64_t)(lo % (uint32_t)a.xs) * a.segcap + first;
    const uint64_t *inw = a.w1 + in0;
    const uint32_t *inv = a.v1 + in0;
    cnt[threadIdx.x] = 0;
    if (MODE == 1) sp[threadIdx.x] = threadIdx.x ? a.grid[(size_t)PP_NB * seg + threadIdx.x] : 0ull;
    __syncthreads();
    uint64_t w[MSD_ITEMS];
    uint32_t v[MSD_ITEMS], rk[MSD_ITEMS], dg[MSD_ITEMS];
#pragma unroll
    for (int r = 0; r < MSD_ITEMS; r++) {
        const uint32_t j = threadIdx.x + r * PP_THREADS;
        w[r] = j < have ? inw[j] : 0ull;
        v[r] = (j < have && !a.packed) ? inv[j] : 0u;
    }
#pragma unroll
    for (int r = 0; r < MSD_ITEMS; r++) {
        const uint32_t j = threadIdx.x + r * PP_THREADS;
        dg[r] = j < have ? (MODE == 1 ? pp_rank511(sp, w[r] >> a.pb) : (pp_t28(a, w[r]) >> 10) & (PP_NB - 1)) : 0u;
        rk[r] = j < have ? atomicAdd(&cnt[dg[r]], 1u) : 0u;
    }
    __syncthreads();
    pp_scan_and_reserve(cnt, loff, gbase, wsum, nullptr, a.count2 + (size_t)seg * PP_NB);
#pragma unroll
    for (int r = 0; r < MSD_ITEMS; r++) {
        const uint32_t j = threadIdx.x + r * PP_THREADS;
        if (j < have) { const uint32_t at = loff[dg[r]] + rk[r]; sw[at] = w[r]; sv[at] = v[r]; if (MODE == 1) sd[at] = (uint16_t)dg[r]; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < MSD_ITEMS; r++) {
        const uint32_t j = threadIdx.x + r * PP_THREADS;
        if (j < have) {
            const uint64_t x = sw[j];
            const uint32_t d = MODE == 1 ? sd[j] : (pp_t28(a, x) >> 10) & (PP_NB - 1);
            const uint64_t at = gbase[d] + j;
            const uint64_t sb = (uint64_t)seg * PP_NB + d;
            if (at < PP_FN_CAP) { a.w2[sb * PP_FN_CAP + at] = x; if (!a.packed) a.v2[sb * PP_FN_CAP + at] = sv[j]; }
            else {
                const unsigned long long e = atomicAdd(a.arena_count, 1ull);
                if (e < PP_ARENA) { a.arena_sb[e] = (uint32_t)sb; a.arena_w[e] = x; a.arena_v[e] = sv[j]; }      // sv: zeros when packed
                else *a.flag = 1;
            }
        }
    }
}

__global__ void k_pp_widen(const uint32_t *__restrict__ count2, unsigned long long *__restrict__ wide, uint64_t cnt)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cnt) wide[i] = count2[i];
}

// slots (first n_a from a, the rest from b) -> out, sorted by (word, low position): bins on the last 9 bits of t
// inside LDS, then every slot counts the smaller slots of its bin
template <int CAP, int MODE, bool ANY>
__device__ __forceinline__ void pp_finish_body(const PpArgs &a, uint64_t *sw, uint32_t *sv, uint32_t *cnt, uint32_t *loff, uint32_t *wsum,
                                               const uint64_t *wa, const uint32_t *va, uint32_t n_a, const uint64_t *wb,
                                               const uint32_t *vb, uint32_t have, uint64_t *wout, uint32_t *vout, uint64_t *smp, uint64_t *srt,
                                               uint32_t *place, uint32_t sb)
{
    constexpr int ITEMS = CAP / PP_THREADS;
    // MODE 1: the keys of a sub-bucket follow no formula (they cluster at the two ends of its key range, deep below the
    // symbols its grid keys share), so the bins come from the sub-bucket itself: PP_SMP of its keys, sorted, are the bin
    // boundaries (the slots arrive in no particular order: every (have / PP_SMP)-th is a fair sample) and a slot's bin
    // is the number of boundaries not above its key (binary search in LDS).  The loads happen below; the sample is
    // taken from a staged copy of the keys in sw.
    int cp = 0, bshift = 0;
    uint64_t base = 0;
    __shared__ uint16_t tab3[MODE == 1 ? 512 : 1];
    const bool by3 = MODE == 1 && a.rb > 0 && a.b <= 3;
    if (by3) pp_rank_table3(a, tab3);                          // (the barrier behind the zeroing of cnt below covers it)
    if (MODE == 1 && a.rb > 0) {
        const uint64_t glo = sb ? a.grid[sb] : 0ull, ghi = (sb + 1 < (uint32_t)(PP_NB * PP_NB) ? a.grid[sb + 1] : a.top) - 1;
        const uint64_t x = glo ^ ghi;
        cp = (x == 0 || ghi < glo) ? a.K : (int)((uint32_t)(__clzll((long long)x) - (64 - a.key_bits)) / (uint32_t)a.b);
        base = pp_rank_number(a, glo, cp);
        const uint64_t span = pp_rank_number(a, ghi, cp) - base;          // the largest difference a key of the sub-bucket can show
        bshift = span >= PP_FBINS ? 64 - __clzll((long long)span) - 10 : 0;
    }
    auto bin_of = [&](uint64_t word) -> uint32_t {
        if (by3) return (uint32_t)((pp_rank_number3(a, tab3, word >> a.pb, cp) - base) >> bshift);
        if (MODE == 1 && a.rb > 0) return (uint32_t)((pp_rank_number(a, word >> a.pb, cp) - base) >> bshift);
        if (MODE == 1) {
            // PP_SMP boundaries -> PP_SMP + 1 stretches; a stretch between two boundaries is cut into 8 more bins
            // linearly in the key (float arithmetic: monotone), which thins the bins out wherever the keys of the
            // stretch do not sit at one end of it
            const uint64_t key = word >> a.pb;
            uint32_t at = 0;
#pragma unroll
            for (uint32_t step = (PP_SMP + 1) / 2; step >= 1; step >>= 1)
                if (srt[at + step - 1] <= key) at += step;         // boundaries not above the key: 0 .. PP_SMP
            uint32_t fine = 0;
            if (at > 0 && at < PP_SMP) {
                const uint64_t lo = srt[at - 1], hi = srt[at];
                fine = min(7u, (uint32_t)((float)(key - lo) * (8.0f / (float)(hi - lo))));
            }
            return at * 8 + fine;
        }
        return pp_t28(a, word) & (PP_FBINS - 1);
    };
    uint32_t bn[ITEMS];
    // SIMPLE: the variants without a second look at crowded bins (the one most sub-buckets of a large text go through, and the
    // partitions' sort): the counting runs in binned order, as in msd_sort.hip -- the bins' first slots are marked in a bitmap
    constexpr bool SIMPLE = !(MODE == 1 && (ANY || CAP > PP_FN_SMALL));
    __shared__ uint32_t bm[SIMPLE ? CAP / 32 + 2 : 1];
    if (SIMPLE) for (uint32_t i = threadIdx.x; i < CAP / 32 + 2; i += PP_THREADS) bm[i] = 0;
    cnt[2 * threadIdx.x] = 0; cnt[2 * threadIdx.x + 1] = 0;    // PP_FBINS bins, two per thread
    __syncthreads();
    uint64_t w[ITEMS];
    uint32_t v[ITEMS], rk[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const uint32_t j = threadIdx.x + r * PP_THREADS;
        w[r] = j < have ? (j < n_a ? wa[j] : wb[j - n_a]) : ~0ull;
        v[r] = (j < have && !a.packed) ? (j < n_a ? va[j] : vb[j - n_a]) : 0u;
    }
    if (a.probe & 1) {
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            const uint32_t j = threadIdx.x + r * PP_THREADS;
            if (j < have) { wout[j] = w[r]; if (!a.packed) vout[j] = v[r]; }
        }
        return;
    }
    if (MODE == 1 && a.rb == 0) {
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            const uint32_t j = threadIdx.x + r * PP_THREADS;
            if (j < have) sw[j] = w[r];
        }
        __syncthreads();
        const uint32_t ns = min((uint32_t)PP_SMP, have);
        if (threadIdx.x < PP_SMP) smp[threadIdx.x] = threadIdx.x < ns ? sw[(uint64_t)threadIdx.x * have / ns] >> a.pb : ~0ull;
        if (threadIdx.x < PP_SMP) place[threadIdx.x] = 0;
        __syncthreads();
        {   // rank of every sample key among the sample = its place; four threads share a key's comparisons
            const uint32_t t = threadIdx.x >> 2, part = threadIdx.x & 3;
            if (t < PP_SMP) {
                const uint64_t mine = smp[t];
                uint32_t c = 0;
                for (uint32_t u = part; u < PP_SMP; u += 4) { const uint64_t o = smp[u]; c += (o < mine || (o == mine && u < t)) ? 1u : 0u; }
                if (c) atomicAdd(&place[t], c);
            }
        }
        __syncthreads();
        if (threadIdx.x < PP_SMP) srt[place[threadIdx.x]] = smp[threadIdx.x];
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const uint32_t j = threadIdx.x + r * PP_THREADS;
        bn[r] = j < have ? bin_of(w[r]) : 0u;
        rk[r] = j < have ? atomicAdd(&cnt[bn[r]], 1u) : 0u;
    }
    __syncthreads();
    {
        const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        const uint32_t c0 = cnt[2 * threadIdx.x], c = c0 + cnt[2 * threadIdx.x + 1];
        uint32_t inc = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d, 64); if ((int)lane >= d) inc += o; }
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        uint32_t pre = 0;
        for (uint32_t q = 0; q < wv; q++) pre += wsum[q];
        loff[2 * threadIdx.x] = pre + inc - c;
        loff[2 * threadIdx.x + 1] = pre + inc - c + c0;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const uint32_t j = threadIdx.x + r * PP_THREADS;
        if (j < have) { const uint32_t at = loff[bn[r]] + rk[r]; sw[at] = w[r]; sv[at] = v[r]; }
    }
    if (SIMPLE) {
        // the first slot of every bin that has one, and the end of the slots
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const uint32_t b = 2 * threadIdx.x + q;
            if (cnt[b]) atomicOr(&bm[loff[b] >> 5], 1u << (loff[b] & 31));
        }
        if (threadIdx.x == 0) atomicOr(&bm[have >> 5], 1u << (have & 31));
    }
    __syncthreads();
    if (SIMPLE) {
        // Every slot counts the smaller slots of its bin, taken in the order in which they lie in sw now: the lanes of a wave
        // sit in the same few bins (the same words for all of them: no bank conflicts) and loop as often as the largest of
        // those has slots.  (key, value) pairs compare as one 96-bit number: no branch for equal keys.
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            const uint32_t j = threadIdx.x + r * PP_THREADS;
            if (j < have) {
                const uint64_t y = sw[j];
                const uint32_t yv = sv[j];
                uint32_t wi = j >> 5, m = bm[wi] & (0xffffffffu >> (31 - (j & 31)));
                while (!m) m = bm[--wi];                        // (slot 0 starts a bin: the walk ends)
                const uint32_t b0 = wi * 32 + 31 - (uint32_t)__clz((int)m);
                wi = (j + 1) >> 5; m = bm[wi] & (0xffffffffu << ((j + 1) & 31));
                while (!m) m = bm[++wi];                        // (the end of the slots is marked: the walk ends)
                const uint32_t e = wi * 32 + (uint32_t)__ffs((int)m) - 1;
                const unsigned __int128 Y = ((unsigned __int128)y << 32) | yv;
                uint32_t smaller = 0;
                for (uint32_t q = b0; q < e; q++) smaller += ((((unsigned __int128)sw[q]) << 32) | sv[q]) < Y ? 1u : 0u;
                w[r] = y; v[r] = yv; rk[r] = b0 + smaller;
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            const uint32_t j = threadIdx.x + r * PP_THREADS;
            if (j < have) { sw[rk[r]] = w[r]; sv[rk[r]] = v[r]; }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            const uint32_t j = threadIdx.x + r * PP_THREADS;
            if (j < have) { wout[j] = sw[j]; if (!a.packed) vout[j] = sv[j]; }
        }
        return;
    }
    // where every slot counts the smaller slots of its bin a bin of thousands is quadratic work -- MODE 1: the keys with a
    // rare symbol at the edge of a sub-bucket share one number.  Such a bin is split once more, on the 8 key bits from the
    // highest bit in which its smallest and largest key differ (monotone; even enough for a few hundred to a few thousand keys)
    uint32_t rb0[ITEMS], rc[ITEMS], radd[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; r++) { rb0[r] = loff[bn[r]]; rc[r] = cnt[bn[r]]; radd[r] = 0; }
    // ANY (similar rows: a bin may be one key held by hundreds of suffixes -- the rows' copies of one position -- plus a few
    // neighbours, and nobody asks for an order among equal keys): four keys picked from the bin are pivots; a slot whose key
    // is a pivot takes the place "slots below the pivot + its turn among the pivot's copies", both from counters, and only
    // the few others count the smaller slots of the bin.  (Every slot counting them took 52 ms for 2 * 10^8 suffixes; a
    // split on key bits does not part equal keys.)
    if (MODE == 1 && ANY) {
        __shared__ uint32_t nbig2, biglist2[CAP / PP_CROWD + 1];
        __shared__ uint64_t pv[4];
        __shared__ uint32_t pless[4], peq[4], nother;
        if (threadIdx.x == 0) nbig2 = 0;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < PP_FBINS; i += PP_THREADS)
            if (cnt[i] > PP_CROWD) biglist2[atomicAdd(&nbig2, 1u)] = i;
        __syncthreads();
        const uint32_t nb2 = (a.probe & 2) ? 0u : nbig2;
        for (uint32_t e = 0; e < nb2; e++) {                            // uniform over the workgroup
            const uint32_t bbin = biglist2[e], b0 = loff[bbin], c = cnt[bbin];
            if (threadIdx.x < 4) { pv[threadIdx.x] = sw[b0 + (threadIdx.x * c) / 4]; pless[threadIdx.x] = 0; peq[threadIdx.x] = 0; }
            if (threadIdx.x == 0) nother = 0;
            __syncthreads();
            const uint64_t p0 = pv[0], p1 = pv[1], p2 = pv[2], p3 = pv[3];
            const bool act[4] = {true, p1 != p0, p2 != p0 && p2 != p1, p3 != p0 && p3 != p1 && p3 != p2};
            const uint64_t pk[4] = {p0, p1, p2, p3};
            __syncthreads();                                            // (the pivots are read: the bin's stretch of sw / sv is free)
            uint32_t turn[ITEMS];
            int which[ITEMS];
#pragma unroll
            for (int r = 0; r < ITEMS; r++) {
                const uint32_t j = threadIdx.x + r * PP_THREADS;
                // (one atomic per wave and counter: hundreds of lanes adding 1 to the same LDS word take a cycle each)
                const bool inbin = j < have && bn[r] == bbin;
                which[r] = inbin ? -1 : -2; turn[r] = 0;
                if (!__ballot(inbin)) continue;                         // (uniform over the wave)
                const uint32_t wlane = threadIdx.x & 63;
                const unsigned long long below_me = (1ull << wlane) - 1;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (!act[k]) continue;
                    const bool eq = inbin && w[r] == pk[k];
                    const unsigned long long mless = __ballot(inbin && w[r] < pk[k]), meq = __ballot(eq);
                    uint32_t tb = 0;
                    if (wlane == 0) {
                        if (mless) atomicAdd(&pless[k], (uint32_t)__popcll(mless));
                        if (meq) tb = atomicAdd(&peq[k], (uint32_t)__popcll(meq));
                    }
                    tb = __shfl(tb, 0, 64);
                    if (eq) { which[r] = k; turn[r] = tb + (uint32_t)__popcll(meq & below_me); }
                }
                {   // no pivot's copy: into the compact list at the head of the bin's stretch
                    const bool other = inbin && which[r] < 0;
                    const unsigned long long mo = __ballot(other);
                    uint32_t ob = 0;
                    if (wlane == 0 && mo) ob = atomicAdd(&nother, (uint32_t)__popcll(mo));
                    ob = __shfl(ob, 0, 64);
                    if (other) { const uint32_t oi = ob + (uint32_t)__popcll(mo & below_me); sw[b0 + oi] = w[r]; sv[b0 + oi] = v[r]; }
                }
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < ITEMS; r++) {
                if (which[r] >= 0) { rb0[r] = b0 + pless[which[r]] + turn[r]; rc[r] = 0; }
                else if (which[r] == -1) {
                    // the pivots' copies below this slot are counted here, the other slots below it by the loop over the list
                    uint32_t below = 0;
#pragma unroll
                    for (int k = 0; k < 4; k++) if (act[k] && pk[k] < w[r]) below += peq[k];
                    radd[r] = below; rc[r] = nother;
                }
            }
            __syncthreads();
        }
    }
    if (MODE == 1 && CAP > PP_FN_SMALL && !ANY) {                        // (not in the variant most sub-buckets go through: it costs that one 2 ms)
        __shared__ uint32_t nbig, biglist[CAP / PP_CROWD + 1], sub[256];
        __shared__ unsigned long long kmin, kmax;
        if (threadIdx.x == 0) nbig = 0;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < PP_FBINS; i += PP_THREADS)
            if (cnt[i] > PP_CROWD) biglist[atomicAdd(&nbig, 1u)] = i;
        __syncthreads();
        const uint32_t nb = nbig;
        for (uint32_t e = 0; e < nb; e++) {                             // uniform over the workgroup
            const uint32_t bbin = biglist[e], b0 = loff[bbin];
            if (threadIdx.x == 0) { kmin = ~0ull; kmax = 0ull; }
            if (threadIdx.x < 256) sub[threadIdx.x] = 0;
            __syncthreads();
#pragma unroll
            for (int r = 0; r < ITEMS; r++) {
                const uint32_t j = threadIdx.x + r * PP_THREADS;
                if (j < have && bn[r] == bbin) { atomicMin(&kmin, (unsigned long long)w[r]); atomicMax(&kmax, (unsigned long long)w[r]); }
            }
            __syncthreads();
            const uint64_t lo = kmin, x = kmin ^ kmax;
            if (x == 0) { __syncthreads(); continue; }                  // one key, many positions: the counting below orders them by position
            const int hb = 63 - __clzll((long long)x), shift = hb >= 7 ? hb - 7 : 0;
            uint32_t idx[ITEMS];
#pragma unroll
            for (int r = 0; r < ITEMS; r++) {
                const uint32_t j = threadIdx.x + r * PP_THREADS;
                idx[r] = 0;
                if (j < have && bn[r] == bbin) idx[r] = atomicAdd(&sub[(uint32_t)((w[r] >> shift) - (lo >> shift))], 1u);
            }
            __syncthreads();
            if (threadIdx.x < 64) {                                      // exclusive scan of the 256 counts by one wave, four per lane; counts stay in the upper half word
                uint32_t c4[4], tot = 0;
#pragma unroll
                for (int q = 0; q < 4; q++) { c4[q] = sub[4 * threadIdx.x + q]; tot += c4[q]; }
                uint32_t inc = tot;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d, 64); if ((int)threadIdx.x >= d) inc += o; }
                uint32_t pre = inc - tot;
#pragma unroll
                for (int q = 0; q < 4; q++) { sub[4 * threadIdx.x + q] = pre | (c4[q] << 16); pre += c4[q]; }
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < ITEMS; r++) {
                const uint32_t j = threadIdx.x + r * PP_THREADS;
                if (j < have && bn[r] == bbin) {
                    const uint32_t pc = sub[(uint32_t)((w[r] >> shift) - (lo >> shift))];
                    rb0[r] = b0 + (pc & 0xffffu); rc[r] = pc >> 16;
                    sw[rb0[r] + idx[r]] = w[r]; sv[rb0[r] + idx[r]] = v[r];
                }
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const uint32_t j = threadIdx.x + r * PP_THREADS;
        if (j < have) {
            const uint32_t b0 = rb0[r], c = (a.probe & 4) ? 0u : rc[r];
            uint32_t smaller = 0;
            // (four loads in flight: a bin of thousands -- keys with a rare symbol at the edge of a sub-bucket share one
            // number -- is a long loop, and one load at a time made such a workgroup the tail of the launch)
#pragma unroll 4
            for (uint32_t q = 0; q < c; q++) {
                const uint64_t x = sw[b0 + q];
                bool less = x < w[r];
                if (x == w[r]) less = sv[b0 + q] < v[r];           // equal keys are rare: the second array is seldom read
                smaller += less ? 1u : 0u;
            }
            rk[r] = b0 + smaller + radd[r];
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const uint32_t j = threadIdx.x + r * PP_THREADS;
        if (j < have) { sw[rk[r]] = w[r]; sv[rk[r]] = v[r]; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const uint32_t j = threadIdx.x + r * PP_THREADS;
        if (j < have) { wout[j] = sw[j]; if (!a.packed) vout[j] = sv[j]; }
    }
}

// CAP: slots the workgroup has room for in LDS -- PP_FN_CAP (the stretch of a sub-bucket), or PP_FN_SMALL for the
// sub-buckets that hold no more than that (a text of 5 * 10^8 symbols fills a stretch to 40 %: with the smaller arrays
// four workgroups share a CU instead of two and the per-thread loops are half as long); a launch of each
// list = nullptr: one workgroup per sub-bucket, those beyond CAP (up to the stretch) noted in `later`; else the listed ones
template <int MODE, int CAP, bool ANY> __global__ __launch_bounds__(PP_THREADS) void k_pp_finish(PpArgs a, const uint32_t *__restrict__ list, uint32_t *__restrict__ later,
                                                                                       unsigned long long *__restrict__ later_count)
{
    __shared__ uint64_t sw[CAP];
    __shared__ uint32_t sv[CAP];
    __shared__ uint32_t cnt[PP_FBINS], loff[PP_FBINS];
    __shared__ uint32_t wsum[PP_THREADS / 64];
    const uint32_t sb = list ? list[blockIdx.x] : blockIdx.x;
    const uint32_t have = a.count2[sb];
    if (have == 0 || have > PP_FN_CAP) return;                 // the larger ones: k_pp_finish_big
    if (have > CAP) {                                          // uniform
        if (threadIdx.x == 0 && later) later[atomicAdd(later_count, 1ull)] = sb;
        return;
    }
    const uint64_t *wa = a.w2 + (uint64_t)sb * PP_FN_CAP;
    const uint32_t *va = a.v2 + (uint64_t)sb * PP_FN_CAP;
    __shared__ uint64_t smp[MODE == 1 ? PP_SMP + 1 : 1], srt[MODE == 1 ? PP_SMP + 1 : 1];
    __shared__ uint32_t place[MODE == 1 ? PP_SMP + 1 : 1];
    if (MODE == 1 && threadIdx.x == 0) srt[PP_SMP] = ~0ull;
    pp_finish_body<CAP, MODE, ANY>(a, sw, sv, cnt, loff, wsum, wa, va, have, wa, va, have, a.wout + a.off[sb], a.vout + a.off[sb], smp, srt, place, sb);
}

template <int MODE, bool ANY> __global__ __launch_bounds__(PP_THREADS) void k_pp_finish_big(PpArgs a, const uint32_t *__restrict__ sb_sorted,
                                                              const uint32_t *__restrict__ idx_sorted, uint32_t entries,
                                                              uint64_t *__restrict__ gw, uint32_t *__restrict__ gv)
{
    // gw / gv: the arena entries gathered in sb_sorted order (k_pp_gather)
    __shared__ uint64_t sw[PP_BIG_CAP];
    __shared__ uint32_t sv[PP_BIG_CAP];
    __shared__ uint32_t cnt[PP_FBINS], loff[PP_FBINS];
    __shared__ uint32_t wsum[PP_THREADS / 64];
    const uint32_t e = blockIdx.x;
    if (e >= entries) return;
    const uint32_t sb = sb_sorted[e];
    if (e > 0 && sb_sorted[e - 1] == sb) return;
    const uint32_t have = a.count2[sb];
    if (have > PP_BIG_CAP || have <= PP_FN_CAP) { if (threadIdx.x == 0) *a.flag = 1; return; }
    __shared__ uint64_t smp[MODE == 1 ? PP_SMP + 1 : 1], srt[MODE == 1 ? PP_SMP + 1 : 1];
    __shared__ uint32_t place[MODE == 1 ? PP_SMP + 1 : 1];
    if (MODE == 1 && threadIdx.x == 0) srt[PP_SMP] = ~0ull;
    pp_finish_body<PP_BIG_CAP, MODE, ANY>(a, sw, sv, cnt, loff, wsum, a.w2 + (uint64_t)sb * PP_FN_CAP, a.v2 + (uint64_t)sb * PP_FN_CAP, PP_FN_CAP,
                                     gw + e, gv + e, have, a.wout + a.off[sb], a.vout + a.off[sb], smp, srt, place, sb);
}

__global__ void k_pp_gather(const uint32_t *__restrict__ idx_sorted, const uint64_t *__restrict__ aw, const uint32_t *__restrict__ av,
                            uint32_t entries, uint64_t *__restrict__ gw, uint32_t *__restrict__ gv)
{
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < entries) { gw[e] = aw[idx_sorted[e]]; gv[e] = av[idx_sorted[e]]; }
}

__global__ void k_pp_iota(uint32_t *__restrict__ idx, uint32_t n)
{
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) idx[e] = e;
}

// The three passes.  a: key range / bucket function filled in by the caller; est: slots expected.  *ok = 0: not done
// (a capacity was exceeded): nothing usable.
template <int MODE>
static int pp_sort(fbg_ctx *ctx, const KeyGeom &g, PpArgs &a, uint64_t est, int nparts, uint64_t out_offset, uint64_t *count, int *ok,
                   int *launches)
{
    const uint64_t N = ctx->N;
    hipStream_t st = ctx->stream;
    const int xcd = ctx->opt.msd_xcd < 0 ? 3 : (int)ctx->opt.msd_xcd;
    const int xs = (xcd & 2) ? 8 : 1;
    const uint32_t nseg = PP_NB * (uint32_t)xs;
    const uint64_t cap1 = (est / PP_NB + est / (4 * PP_NB) + 65536 + 127) & ~127ull;     // 8 parts of whole 128-byte lines
    const uint64_t nsub = (uint64_t)PP_NB * PP_NB;
    FBG_TRY(fbg_reserve(ctx, ctx->keysA, (size_t)PP_NB * cap1 * 8));
    if (!g.packed) FBG_TRY(fbg_reserve(ctx, ctx->valsA, (size_t)PP_NB * cap1 * 4));
    FBG_TRY(fbg_reserve(ctx, ctx->msd_w, (size_t)nsub * PP_FN_CAP * 8));
    if (!g.packed) FBG_TRY(fbg_reserve(ctx, ctx->msd_v, (size_t)nsub * PP_FN_CAP * 4));
    FBG_TRY(fbg_reserve(ctx, ctx->dp_a, (size_t)nseg * 8 + ((size_t)nseg + 1) * 4 + 64));
    FBG_TRY(fbg_reserve(ctx, ctx->dp_b, nsub * 4));
    FBG_TRY(fbg_reserve(ctx, ctx->dp_c, (nsub + 1) * 8 * 2));
    FBG_TRY(fbg_reserve(ctx, ctx->dp_d, (size_t)PP_ARENA * 4 * 5));
    FBG_TRY(fbg_reserve(ctx, ctx->dp_e, (size_t)PP_ARENA * 8 * 2));
    unsigned long long *flag = ctx->scalars.as<unsigned long long>() + 100;      // [0] flag, [1] arena count, [2] total
    a.T = ctx->text.as<uint8_t>(); a.N = N; a.code = g.d_code; a.b = g.b; a.K = g.K; a.pb = (g.wide || g.packed) ? g.pb : 0; a.nparts = nparts;
    a.wide = g.wide ? 1 : 0; a.packed = g.packed ? 1 : 0;
    a.ebits = (MODE == 1 && !g.wide && !g.packed && g.K <= 32) ? ctx->grs_ebits : nullptr;
    a.payload = (MODE == 1 && !g.wide && !g.packed) ? ctx->sort_payload : nullptr;
    a.any_order = a.payload ? 1 : 0;
    a.probe = 0;
    const bool any = MODE == 1 && a.any_order;
    a.w1 = ctx->keysA.as<uint64_t>(); a.v1 = ctx->valsA.as<uint32_t>(); a.cap1 = cap1; a.xs = xs; a.segcap = cap1 / xs;
    a.tiles2 = 0; a.tiles2_x = 0;
    a.count1 = ctx->dp_a.as<unsigned long long>();
    uint32_t *tile_start = reinterpret_cast<uint32_t *>(ctx->dp_a.as<uint8_t>() + (size_t)nseg * 8);
    a.tile_start = tile_start;
    a.w2 = ctx->msd_w.as<uint64_t>(); a.v2 = ctx->msd_v.as<uint32_t>(); a.count2 = ctx->dp_b.as<uint32_t>();
    unsigned long long *wide = ctx->dp_c.as<unsigned long long>(), *off = wide + (nsub + 1);
    a.off = off;
    a.arena_sb = ctx->dp_d.as<uint32_t>(); a.arena_v = a.arena_sb + PP_ARENA;
    uint32_t *a_idx = a.arena_sb + 2 * PP_ARENA, *sb_sorted = a.arena_sb + 3 * PP_ARENA, *idx_sorted = a.arena_sb + 4 * PP_ARENA;
    a.arena_w = ctx->dp_e.as<uint64_t>();
    a.arena_count = flag + 1;
    a.flag = flag;
    a.wout = nullptr; a.vout = nullptr;
    FBG_HIP_TRY(ctx, hipMemsetAsync(flag, 0, 24, st));
    FBG_HIP_TRY(ctx, hipMemsetAsync(a.count1, 0, (size_t)nseg * 8, st));
    FBG_HIP_TRY(ctx, hipMemsetAsync(a.count2, 0, nsub * 4, st));
    hipLaunchKernelGGL((k_pp_pack_split<MODE>), dim3(fbg_blocks(N, (uint64_t)PP_TILE * nparts)), dim3(PP_THREADS), 0, st, a);
    hipLaunchKernelGGL(k_pp_tiles, dim3(1), dim3(1024), 0, st, a.count1, a.segcap, nseg, tile_start, flag + 2, flag);
    uint32_t tiles2 = 0;
    unsigned long long h3[3] = {0, 0, 0};
    FBG_HIP_TRY(ctx, hipMemcpyAsync(&tiles2, tile_start + nseg, 4, hipMemcpyDeviceToHost, st));
    FBG_HIP_TRY(ctx, hipMemcpyAsync(h3, flag, 24, hipMemcpyDeviceToHost, st));
    FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
    *launches += 2;
    if (h3[0] != 0 || tiles2 == 0) return FBG_OK;
    const uint64_t total = h3[2];
    FBG_TRY(fbg_reserve(ctx, ctx->keysB, (total + 2 * out_offset) * 8));
    if (!g.packed) FBG_TRY(fbg_reserve(ctx, ctx->valsB, (total + 2 * out_offset) * 4));
    a.wout = ctx->keysB.as<uint64_t>() + out_offset; a.vout = ctx->valsB.as<uint32_t>() + out_offset;
    a.tiles2 = tiles2;
    a.tiles2_x = (xcd & 1) ? (tiles2 + 7) / 8 : 0;
    hipLaunchKernelGGL((k_pp_split<MODE>), dim3(a.tiles2_x ? 8 * a.tiles2_x : tiles2), dim3(PP_THREADS), 0, st, a);
    hipLaunchKernelGGL(k_pp_widen, dim3(fbg_blocks(nsub, 256)), dim3(256), 0, st, a.count2, wide, nsub);
    {
        size_t bytes = 0;
        hipError_t e = rocprim::exclusive_scan(nullptr, bytes, wide, off, 0ull, (size_t)nsub, rocprim::plus<unsigned long long>(), st);
        if (e != hipSuccess) return fbg_fail(ctx, FBG_ERR_HIP, "rocprim scan size query: %s", hipGetErrorString(e));
        FBG_TRY(fbg_reserve(ctx, ctx->tmp, bytes));
        size_t have = ctx->tmp.cap;
        e = rocprim::exclusive_scan(ctx->tmp.p, have, wide, off, 0ull, (size_t)nsub, rocprim::plus<unsigned long long>(), st);
        if (e != hipSuccess) return fbg_fail(ctx, FBG_ERR_HIP, "rocprim exclusive_scan: %s", hipGetErrorString(e));
    }
    if (total / nsub + total / (4 * nsub) <= PP_FN_SMALL) {
        // sparsely filled stretches: the small-capacity variant for all, the full one for the few it notes down
        FBG_TRY(fbg_reserve(ctx, ctx->ps_a, (size_t)nsub * 4));
        unsigned long long *later_count = flag + 3;
        FBG_HIP_TRY(ctx, hipMemsetAsync(later_count, 0, 8, st));
        // (any order, stretches filled to a fifth: the variant with room for PP_FN_TINY slots)
        const bool tiny = any && MODE == 1 && total / nsub + total / (4 * nsub) <= PP_FN_TINY / 2;
        if (any && ctx->opt.msd_probe) {
            for (int v : {1, 2, 4, 6}) {
                a.probe = v;
                hipLaunchKernelGGL((k_pp_finish<MODE, PP_FN_SMALL, MODE == 1>), dim3((unsigned)nsub), dim3(PP_THREADS), 0, st, a, (const uint32_t *)nullptr, ctx->ps_a.as<uint32_t>(),
                                   later_count);
                FBG_HIP_TRY(ctx, hipMemsetAsync(later_count, 0, 8, st));
            }
            a.probe = 0;
            if (tiny) {                                        // the small variant's time beside the tiny one's
                hipLaunchKernelGGL((k_pp_finish<MODE, PP_FN_SMALL, MODE == 1>), dim3((unsigned)nsub), dim3(PP_THREADS), 0, st, a, (const uint32_t *)nullptr, ctx->ps_a.as<uint32_t>(),
                                   later_count);
                FBG_HIP_TRY(ctx, hipMemsetAsync(later_count, 0, 8, st));
            }
        }
        if (tiny) hipLaunchKernelGGL((k_pp_finish<MODE, PP_FN_TINY, MODE == 1>), dim3((unsigned)nsub), dim3(PP_THREADS), 0, st, a, (const uint32_t *)nullptr, ctx->ps_a.as<uint32_t>(),
                                     later_count);
        else if (any) hipLaunchKernelGGL((k_pp_finish<MODE, PP_FN_SMALL, MODE == 1>), dim3((unsigned)nsub), dim3(PP_THREADS), 0, st, a, (const uint32_t *)nullptr, ctx->ps_a.as<uint32_t>(),
                                    later_count);
        else hipLaunchKernelGGL((k_pp_finish<MODE, PP_FN_SMALL, false>), dim3((unsigned)nsub), dim3(PP_THREADS), 0, st, a, (const uint32_t *)nullptr, ctx->ps_a.as<uint32_t>(),
                                later_count);
        unsigned long long nlater = 0;
        FBG_HIP_TRY(ctx, hipMemcpyAsync(&nlater, later_count, 8, hipMemcpyDeviceToHost, st));
        FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
        if (nlater > 0) {
            if (any) hipLaunchKernelGGL((k_pp_finish<MODE, PP_FN_CAP, MODE == 1>), dim3((unsigned)nlater), dim3(PP_THREADS), 0, st, a, (const uint32_t *)ctx->ps_a.as<uint32_t>(),
                                        (uint32_t *)nullptr, (unsigned long long *)nullptr);
            else hipLaunchKernelGGL((k_pp_finish<MODE, PP_FN_CAP, false>), dim3((unsigned)nlater), dim3(PP_THREADS), 0, st, a, (const uint32_t *)ctx->ps_a.as<uint32_t>(),
                                    (uint32_t *)nullptr, (unsigned long long *)nullptr);
        }
        *launches += 1;
    } else {
        if (any) hipLaunchKernelGGL((k_pp_finish<MODE, PP_FN_CAP, MODE == 1>), dim3((unsigned)nsub), dim3(PP_THREADS), 0, st, a, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                    (unsigned long long *)nullptr);
        else hipLaunchKernelGGL((k_pp_finish<MODE, PP_FN_CAP, false>), dim3((unsigned)nsub), dim3(PP_THREADS), 0, st, a, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                (unsigned long long *)nullptr);
    }
    FBG_HIP_TRY(ctx, hipMemcpyAsync(h3, flag, 24, hipMemcpyDeviceToHost, st));
    FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
    *launches += 4;
    if (h3[0] != 0) return FBG_OK;
    if (h3[1] > 0) {
        const uint32_t entries = (uint32_t)h3[1];
        hipLaunchKernelGGL(k_pp_iota, dim3(fbg_blocks(entries, 256)), dim3(256), 0, st, a_idx, entries);
        size_t bytes = 0;
        hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, a.arena_sb, sb_sorted, a_idx, idx_sorted, (size_t)entries, 0u, 32u, st);
        if (e != hipSuccess) return fbg_fail(ctx, FBG_ERR_HIP, "rocprim sort size query: %s", hipGetErrorString(e));
        FBG_TRY(fbg_reserve(ctx, ctx->tmp, bytes));
        size_t have = ctx->tmp.cap;
        e = rocprim::radix_sort_pairs(ctx->tmp.p, have, a.arena_sb, sb_sorted, a_idx, idx_sorted, (size_t)entries, 0u, 32u, st);
        if (e != hipSuccess) return fbg_fail(ctx, FBG_ERR_HIP, "rocprim radix_sort_pairs: %s", hipGetErrorString(e));
        uint64_t *gw = a.arena_w + PP_ARENA;
        uint32_t *gv = a_idx;                                  // the iota is consumed by the sort
        hipLaunchKernelGGL(k_pp_gather, dim3(fbg_blocks(entries, 256)), dim3(256), 0, st, idx_sorted, a.arena_w, a.arena_v, entries, gw, gv);
        if (any) hipLaunchKernelGGL((k_pp_finish_big<MODE, MODE == 1>), dim3(entries), dim3(PP_THREADS), 0, st, a, sb_sorted, idx_sorted, entries, gw, gv);
        else hipLaunchKernelGGL((k_pp_finish_big<MODE, false>), dim3(entries), dim3(PP_THREADS), 0, st, a, sb_sorted, idx_sorted, entries, gw, gv);
        FBG_HIP_TRY(ctx, hipMemcpyAsync(h3, flag, 24, hipMemcpyDeviceToHost, st));
        FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
        *launches += 4;
        if (h3[0] != 0) return FBG_OK;
    }
    FBG_HIP_TRY(ctx, hipGetLastError());
    *count = total;
    *ok = 1;
    return FBG_OK;
}

// Packs, filters and sorts the slots of key range [lo, hi) of the current text into keysB / valsB at out_offset (count
// of them in *count).  *ok = 0: not done (a capacity was exceeded, or the sizes do not suit this sort): nothing usable.
int fbg_msd_sort_part(fbg_ctx *ctx, const KeyGeom &g, uint64_t lo, uint64_t hi, int nohi, int nparts, uint64_t out_offset,
                      uint64_t *count, int *ok, int *launches)
{
    *ok = 0;
    const uint64_t N = ctx->N;
    const uint64_t min_n = ctx->opt.msd_min >= 0 ? (uint64_t)ctx->opt.msd_min : (1ull << 24);
    if (!g.compact || N / nparts < min_n || ctx->opt.no_msd_sort) return FBG_OK;
    const uint64_t top = g.key_bits >= 64 ? ~0ull : (1ull << g.key_bits);
    const uint64_t span = (nohi ? top : hi) - lo;
    if (span < (1ull << 29) || g.key_bits >= 64) return FBG_OK;       // t needs 28 bits of resolution below the span
    const unsigned __int128 one = 1;
    const uint64_t est = N / nparts + N / (64 * (uint64_t)nparts) + 65536;   // the partitions are quantiles of a sample: sizes within a percent
    const uint64_t nsub = (uint64_t)PP_NB * PP_NB;
    if (est / nsub + est / (8 * nsub) + 64 > PP_FN_CAP) return FBG_OK;
    PpArgs a;
    a.lo = lo; a.hi = hi; a.mul = (uint64_t)((one << 92) / span); a.nohi = nohi;      // mul < 2^64 since span > 2^28
    a.grid = nullptr; a.top = top; a.sigma = 1 << g.b;
    a.rb = 0; a.ew = 2; a.nd = 0; a.rtab = a.mtab = 0; a.key_bits = g.key_bits;
    return pp_sort<0>(ctx, g, a, est, nparts, out_offset, count, ok, launches);
}

// keys of every stride-th text position, as k_pack builds them without the compact coding (general alphabet)
__global__ void k_ss_sample(const uint8_t *__restrict__ T, uint64_t N, const uint8_t *__restrict__ code, int b, int K, uint64_t stride,
                            uint64_t S, uint64_t *__restrict__ out)
{
    __shared__ uint8_t cd[256];
    if (threadIdx.x < 256) cd[threadIdx.x] = code[threadIdx.x];
    __syncthreads();
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S) return;
    const uint64_t p = i * stride;
    // the K <= 32 bytes behind p as four 8-byte loads (the text is padded with 64 zero bytes; byte by byte the loop waited
    // for memory K times: 0.78 ms for 4 * 10^6 samples at C5)
    uint64_t key = 0;
    if (K > 32) {                              // one-bit symbols: up to 64 of them
        for (int k = 0; k < K; k++) key = (key << b) | (p + k < N ? (uint64_t)cd[T[p + k]] : 0ull);
        out[i] = key;
        return;
    }
    uint64_t w[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { w[j] = 0; if (8 * j < K) __builtin_memcpy(&w[j], T + p + 8 * j, 8); }
#pragma unroll
    for (int k = 0; k < 32; k++)
        if (k < K) key = (key << b) | (p + k < N ? (uint64_t)cd[(uint8_t)(w[k >> 3] >> (8 * (k & 7)))] : 0ull);
    out[i] = key;
}

__global__ void k_ss_twins(const uint64_t *__restrict__ sorted, uint64_t S, unsigned long long *__restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool twin = i + 1 < S && sorted[i] == sorted[i + 1];
    __shared__ uint32_t blk;                               // (one addition per workgroup: 65 000 on one address took 0.5 ms)
    if (threadIdx.x == 0) blk = 0;
    __syncthreads();
    const unsigned long long mask = __ballot(twin);
    if ((threadIdx.x & 63) == 0 && mask) atomicAdd(&blk, (uint32_t)__popcll(mask));
    __syncthreads();
    if (threadIdx.x == 0 && blk) atomicAdd(out, (unsigned long long)blk);
}

__global__ void k_ss_grid(const uint64_t *__restrict__ sorted, uint64_t S, uint64_t *__restrict__ grid)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < (uint64_t)PP_NB * PP_NB) grid[t] = sorted[t * S / ((uint64_t)PP_NB * PP_NB)];
}

// The record path's round-0 sort (suffix_sort.hip): all N (key, position) pairs of the current text, keys in the
// order-preserving code of any alphabet, into keysB / valsB.  Sample sort: see the head of this file.
int fbg_sample_sort_pairs(fbg_ctx *ctx, const KeyGeom &g, int *ok, int *launches)
{
    *ok = 0;
    ctx->pairs_similar = false;
    const uint64_t N = ctx->N;
    const uint64_t min_n = ctx->opt.msd_min >= 0 ? (uint64_t)ctx->opt.msd_min : (1ull << 24);
    const uint64_t nsub = (uint64_t)PP_NB * PP_NB;
    // sub-bucket sizes follow the sample (2^22 keys, 16 per sub-bucket: +-25 %): they must fit their stretches twice over
    if (g.compact || g.packed || g.wide || N < min_n || N >= (1ull << 32) || ctx->opt.no_msd_sort || g.key_bits >= 64) return FBG_OK;
    if (ctx->sp_key_flags) return FBG_OK;                  // (span_scan.hip beyond 2^30 cells: flag bits below the key; such texts are beyond the next line anyway)
    if (2 * (N / nsub) + 64 > PP_FN_CAP) return FBG_OK;
    hipStream_t st = ctx->stream;
    uint64_t S = 1ull << 22;
    while (S > N / 4 && S > 256) S >>= 1;                  // small texts (tests): a coarser grid, buckets stay far below capacity
    if (N / S == 0) return FBG_OK;
    FBG_TRY(fbg_reserve(ctx, ctx->ps_b, S * 8));
    FBG_TRY(fbg_reserve(ctx, ctx->ps_c, S * 8));
    FBG_TRY(fbg_reserve(ctx, ctx->ps_d, nsub * 8));
    uint64_t *smp = ctx->ps_b.as<uint64_t>(), *srt = ctx->ps_c.as<uint64_t>(), *grid = ctx->ps_d.as<uint64_t>();
    hipLaunchKernelGGL(k_ss_sample, dim3(fbg_blocks(S, 256)), dim3(256), 0, st, ctx->text.as<uint8_t>(), N, g.d_code, g.b, g.K, N / S, S, smp);
    {
        size_t bytes = 0;
        hipError_t e = rocprim::radix_sort_keys(nullptr, bytes, smp, srt, (size_t)S, 0u, (unsigned)g.key_bits, st);
        if (e != hipSuccess) return fbg_fail(ctx, FBG_ERR_HIP, "rocprim sort size query: %s", hipGetErrorString(e));
        FBG_TRY(fbg_reserve(ctx, ctx->tmp, bytes));
        size_t have = ctx->tmp.cap;
        e = rocprim::radix_sort_keys(ctx->tmp.p, have, smp, srt, (size_t)S, 0u, (unsigned)g.key_bits, st);
        if (e != hipSuccess) return fbg_fail(ctx, FBG_ERR_HIP, "rocprim radix_sort_keys: %s", hipGetErrorString(e));
    }
    {
        // rows that resemble each other tie on most keys: a suffix with a twin somewhere shows up as a twin in the sample
        // with probability S / N.  Sub-buckets full of equal keys make the finish quadratic (52 ms for 2 * 10^8 suffixes of
        // a star phylogeny with gaps), and the scan in suffix order that would follow declines such inputs anyway:
        // leave them to the library sort and say so (ctx->pairs_similar)
        unsigned long long *d_twins = ctx->scalars.as<unsigned long long>() + 104;
        FBG_HIP_TRY(ctx, hipMemsetAsync(d_twins, 0, 8, st));
        hipLaunchKernelGGL(k_ss_twins, dim3(fbg_blocks(S, 256)), dim3(256), 0, st, srt, S, d_twins);
        unsigned long long twins = 0;
        FBG_HIP_TRY(ctx, hipMemcpyAsync(&twins, d_twins, 8, hipMemcpyDeviceToHost, st));
        FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
        *launches += 1;
        ctx->pairs_similar = (double)twins * (double)N / ((double)S * (double)S) > 0.5 && !ctx->opt.msd_min_force;
        if (ctx->pairs_similar && !ctx->sort_payload) return FBG_OK;    // (with a payload the consumer is the group-level scan: equal keys in any order)
    }
    hipLaunchKernelGGL(k_ss_grid, dim3(fbg_blocks(nsub, 256)), dim3(256), 0, st, srt, S, grid);
    *launches += 3;
    PpArgs a;
    a.lo = 0; a.hi = 0; a.mul = 0; a.nohi = 1;
    a.grid = grid; a.top = 1ull << g.key_bits;
    a.sigma = 0;
    for (int c = 0; c < 256; c++) a.sigma += ctx->byte_hist[c] != 0;     // fbg_key_setup's order-preserving codes: 0 .. sigma - 1
    if (a.sigma < 2) a.sigma = 2;
    {
        // the finish bins by the ranks of the symbols among the frequent ones (pp_rank_number)
        uint64_t freq[256];
        int ncodes = 0;
        for (int c = 0; c < 256; c++) if (ctx->byte_hist[c]) freq[ncodes++] = ctx->byte_hist[c];
        int nf = 0;
        for (int c = 0; c < ncodes; c++) nf += freq[c] >= N / 64;
        a.rb = nf <= 2 ? 1 : nf <= 4 ? 2 : nf <= 8 ? 3 : 0;
        a.ew = a.rb <= 2 ? 2 : 4;
        if (g.b > 5 || (1 << g.b) * a.ew > 64 || ctx->opt.msd_sample_bins) a.rb = 0;
        a.rtab = a.mtab = 0; a.nd = 0; a.key_bits = g.key_bits;
        if (a.rb > 0) {
            a.nd = 30 / a.rb;
            // the 2^rb most frequent codes, in code order, get the ranks; every other code joins the frequent one above it
            // (mode 1), the last one if there is none (mode 2)
            const int want = 1 << a.rb;
            bool main_[256] = {false};
            for (int k = 0; k < want && k < ncodes; k++) {
                int best = -1;
                for (int c = 0; c < ncodes; c++) if (!main_[c] && (best < 0 || freq[c] > freq[best])) best = c;
                main_[best] = true;
            }
            int rank_of[256], nmain = 0;
            for (int c = 0; c < ncodes; c++) if (main_[c]) rank_of[c] = nmain++;
            for (int c = 0; c < (1 << g.b); c++) {
                int r = nmain - 1, mode = 2;
                if (c < ncodes && main_[c]) { r = rank_of[c]; mode = 0; }
                else for (int d = c + 1; d < ncodes; d++) if (main_[d]) { r = rank_of[d]; mode = 1; break; }
                a.rtab |= (uint64_t)r << (a.ew * c);
                a.mtab |= (uint64_t)mode << (2 * c);
            }
        }
    }
    uint64_t count = 0;
    const uint64_t est = N + N / 64 + 65536;
    FBG_TRY(pp_sort<1>(ctx, g, a, est, 1, 0, &count, ok, launches));
    if (*ok && count != N) { *ok = 0; }
    return FBG_OK;
}
