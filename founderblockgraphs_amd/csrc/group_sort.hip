// group_sort.hip -- second half of the hybrid suffix sort.
//
// An LSD radix sort moves every (key, position) pair once per digit.  With ~log2(N)-3 leading key bits sorted
// (3 digit passes for N = 1e9) the pairs already sit in groups of a handful of entries that share those bits,
// so the remaining ~30 key bits do not need four more passes over all data: one pass that finishes every group
// locally is enough.  k_group_sort streams the partially sorted arrays through LDS tile by tile (coalesced in,
// coalesced out), finds the group heads inside the tile, and lets one thread order each small group by its
// full key with an insertion sort in LDS.  A tile begins and ends at group heads, found in a look-ahead region,
// so no group is split between workgroups.  Groups longer than GS_LOOK (skewed key distributions) raise a flag
// and the caller falls back to sorting all key bits with the radix sort.
#include "fbg_internal.h"

#define GS_THREADS 256
#define GS_TILE 2048
#define GS_LOOK 512            // look-ahead / longest group handled here

struct GroupSortArgs {
    const uint64_t *keys_in;
    const uint32_t *vals_in;
    uint64_t *keys_out;
    uint32_t *vals_out;
    uint64_t N;
    int top_shift;             // groups = runs of equal (key >> top_shift)
    unsigned long long *flag;  // [0] != 0: a group exceeded the look-ahead
};

__global__ __launch_bounds__(GS_THREADS) void k_group_sort(GroupSortArgs a)
{
    __shared__ uint64_t sk[GS_TILE + GS_LOOK];
    __shared__ uint32_t sv[GS_TILE + GS_LOOK];
    __shared__ uint16_t ghead[GS_TILE + GS_LOOK + 1];   // start offsets of the groups of this tile
    __shared__ uint32_t wsum[GS_THREADS / 64];
    __shared__ uint32_t s_first, s_end, s_ngroups;
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint64_t base = (uint64_t)blockIdx.x * GS_TILE;
    if (base >= a.N) return;
    const uint32_t avail = (uint32_t)min((uint64_t)(GS_TILE + GS_LOOK), a.N - base);
    for (uint32_t k = tid; k < avail; k += GS_THREADS) { sk[k] = a.keys_in[base + k]; sv[k] = a.vals_in[base + k]; }
    if (tid == 0) { s_first = 0xffffffffu; s_end = 0xffffffffu; }
    __syncthreads();
    // head(k): first entry of a group.  The entry before the tile comes from global memory.
    const uint64_t prev_top = base > 0 ? (a.keys_in[base - 1] >> a.top_shift) : ~0ull;
    auto is_head = [&](uint32_t k) -> bool {
        const uint64_t t = sk[k] >> a.top_shift;
        return k == 0 ? (base == 0 || t != prev_top) : t != (sk[k - 1] >> a.top_shift);
    };
    // the tile owns the groups whose head lies in [first head >= 0, first head >= GS_TILE)
    for (uint32_t k = tid; k < avail; k += GS_THREADS) {
        if (is_head(k)) {
            if (k < GS_TILE) atomicMin(&s_first, k);
            else atomicMin(&s_end, k);
        }
    }
    __syncthreads();
    uint32_t first = s_first, end = s_end;
    if (end == 0xffffffffu) {
        if (base + avail >= a.N) end = avail;                       // the text ends inside the look-ahead
        else { if (tid == 0) atomicAdd(a.flag, 1ull); return; }     // a group longer than the look-ahead
    }
    if (first == 0xffffffffu) return;   // no group starts here: the tile belongs to an earlier group (its owner flags it)
    // ---- compact the group heads of [first, end) -------------------------------------------------
    uint32_t carry = 0;
    for (uint32_t k0 = first; k0 < end; k0 += GS_THREADS) {
        const uint32_t k = k0 + tid;
        const bool h = k < end && is_head(k);
        const unsigned long long m = __ballot(h);
        if (lane == 0) wsum[w] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t off = carry, tot = 0;
        for (uint32_t q = 0; q < GS_THREADS / 64; q++) { if (q < w) off += wsum[q]; tot += wsum[q]; }
        if (h) ghead[off + __popcll(m & ((1ull << lane) - 1))] = (uint16_t)k;
        carry += tot;
        __syncthreads();
    }
    if (tid == 0) { ghead[carry] = (uint16_t)end; s_ngroups = carry; }
    __syncthreads();
    const uint32_t ngroups = s_ngroups;
    // ---- one thread per group: insertion sort by full key (groups are tiny) --------------------------
    for (uint32_t g = tid; g < ngroups; g += GS_THREADS) {
        const uint32_t lo = ghead[g], hi = ghead[g + 1];
        for (uint32_t i = lo + 1; i < hi; i++) {
            const uint64_t key = sk[i];
            const uint32_t val = sv[i];
            uint32_t j = i;
            while (j > lo && sk[j - 1] > key) { sk[j] = sk[j - 1]; sv[j] = sv[j - 1]; j--; }
            sk[j] = key; sv[j] = val;
        }
    }
    __syncthreads();
    for (uint32_t k = first + tid; k < end; k += GS_THREADS) { a.keys_out[base + k] = sk[k]; a.vals_out[base + k] = sv[k]; }
}

// keys_in/vals_in sorted by (key >> top_shift); writes the fully sorted pairs to keys_out/vals_out.
// *ok = 0 when a group was too long (outputs are then incomplete).
int fbg_group_sort(fbg_ctx *ctx, const uint64_t *keys_in, const uint32_t *vals_in, uint64_t *keys_out,
                   uint32_t *vals_out, uint64_t N, int top_shift, int *ok)
{
    hipStream_t st = ctx->stream;
    unsigned long long *flag = ctx->scalars.as<unsigned long long>() + 100;
    FBG_HIP_TRY(ctx, hipMemsetAsync(flag, 0, sizeof(unsigned long long), st));
    GroupSortArgs a;
    a.keys_in = keys_in; a.vals_in = vals_in; a.keys_out = keys_out; a.vals_out = vals_out;
    a.N = N; a.top_shift = top_shift; a.flag = flag;
    hipLaunchKernelGGL(k_group_sort, dim3(fbg_blocks(N, GS_TILE)), dim3(GS_THREADS), 0, st, a);
    FBG_HIP_TRY(ctx, hipGetLastError());
    unsigned long long h = 0;
    FBG_HIP_TRY(ctx, hipMemcpyAsync(&h, flag, sizeof(h), hipMemcpyDeviceToHost, st));
    FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
    *ok = h == 0;
    return FBG_OK;
}
