// msdsort.hip -- prototype of the three-pass MSD sort of packed 64-bit slots (key << pb | position):
//   split on the top 9 key bits, split every bucket on the next 9, finish every sub-bucket inside LDS.
// Stand-alone: random words, checks the result against rocPRIM's and times the kernels.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

#define DIG 9
#define NB (1 << DIG)            // buckets per split
#define SP_THREADS 1024
#define SP_ITEMS 8
#define SP_TILE (SP_THREADS * SP_ITEMS)
#define FN_THREADS 256
#define FN_CAP 4608              // largest sub-bucket finished in LDS
#define FN_ITEMS (FN_CAP / FN_THREADS)

__global__ void fill(uint64_t *k, size_t n, int kb, int pb)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t x = i + 0x9E3779B97F4A7C15ull;
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; x ^= x >> 31;
        k[i] = ((x >> (64 - kb)) << pb) | (uint64_t)i;
    }
}

// hist[seg * NB + digit] += ...   (segments: tiles never straddle them)
struct SplitArgs {
    const uint64_t *in;
    uint64_t *out;
    const uint64_t *seg_start;   // nseg + 1 offsets into in / out
    const uint32_t *tile_start;  // nseg + 1: first tile of every segment
    uint32_t nseg;
    int shift;                   // digit = (word >> shift) & (NB - 1)
    unsigned long long *cursor;  // [nseg * NB]: next free slot of every bucket (absolute)
    uint32_t *hist;              // [nseg * NB] (k_hist)
};

__device__ __forceinline__ uint32_t find_seg(const uint32_t *tile_start, uint32_t nseg, uint32_t tile)
{
    uint32_t lo = 0, hi = nseg;            // largest s with tile_start[s] <= tile
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) / 2; if (tile_start[mid] <= tile) lo = mid; else hi = mid; }
    return lo;
}

__global__ __launch_bounds__(SP_THREADS) void k_hist(SplitArgs a)
{
    __shared__ uint32_t sh[NB];
    const uint32_t seg = find_seg(a.tile_start, a.nseg, blockIdx.x);
    const uint64_t s0 = a.seg_start[seg], s1 = a.seg_start[seg + 1];
    const uint64_t base = s0 + (uint64_t)(blockIdx.x - a.tile_start[seg]) * SP_TILE;
    for (int i = threadIdx.x; i < NB; i += SP_THREADS) sh[i] = 0;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SP_ITEMS; r++) {
        const uint64_t k = base + threadIdx.x + (uint64_t)r * SP_THREADS;
        if (k < s1) atomicAdd(&sh[(a.in[k] >> a.shift) & (NB - 1)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NB; i += SP_THREADS)
        if (sh[i]) atomicAdd(&a.hist[(size_t)seg * NB + i], sh[i]);
}

__global__ __launch_bounds__(SP_THREADS) void k_split(SplitArgs a)
{
    __shared__ uint64_t buf[SP_TILE];
    __shared__ uint32_t cnt[NB], loff[NB];
    __shared__ unsigned long long gbase[NB];
    __shared__ uint32_t wsum[SP_THREADS / 64];
    const uint32_t seg = find_seg(a.tile_start, a.nseg, blockIdx.x);
    const uint64_t s0 = a.seg_start[seg], s1 = a.seg_start[seg + 1];
    const uint64_t base = s0 + (uint64_t)(blockIdx.x - a.tile_start[seg]) * SP_TILE;
    const uint32_t have = (uint32_t)min((uint64_t)SP_TILE, s1 - base);
    for (int i = threadIdx.x; i < NB; i += SP_THREADS) cnt[i] = 0;
    __syncthreads();
    uint64_t w[SP_ITEMS];
    uint32_t rk[SP_ITEMS];
#pragma unroll
    for (int r = 0; r < SP_ITEMS; r++) {
        const uint32_t j = threadIdx.x + r * SP_THREADS;
        w[r] = j < have ? a.in[base + j] : 0ull;
    }
#pragma unroll
    for (int r = 0; r < SP_ITEMS; r++) {
        const uint32_t j = threadIdx.x + r * SP_THREADS;
        rk[r] = j < have ? atomicAdd(&cnt[(w[r] >> a.shift) & (NB - 1)], 1u) : 0u;
    }
    __syncthreads();
    // exclusive scan of cnt[NB] (NB <= SP_THREADS): one value per thread, wave scans + wave totals
    {
        const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        const uint32_t c = threadIdx.x < NB ? cnt[threadIdx.x] : 0u;
        uint32_t inc = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d, 64); if ((int)lane >= d) inc += o; }
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        uint32_t pre = 0;
        for (uint32_t q = 0; q < wv; q++) pre += wsum[q];
        if (threadIdx.x < NB) {
            loff[threadIdx.x] = pre + inc - c;
            gbase[threadIdx.x] = c ? atomicAdd(&a.cursor[(size_t)seg * NB + threadIdx.x], (unsigned long long)c) : 0ull;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SP_ITEMS; r++) {
        const uint32_t j = threadIdx.x + r * SP_THREADS;
        if (j < have) buf[loff[(w[r] >> a.shift) & (NB - 1)] + rk[r]] = w[r];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SP_ITEMS; r++) {
        const uint32_t j = threadIdx.x + r * SP_THREADS;
        if (j < have) {
            const uint64_t x = buf[j];
            const uint32_t d = (uint32_t)(x >> a.shift) & (NB - 1);
            a.out[gbase[d] + (j - loff[d])] = x;
        }
    }
}

// one workgroup per sub-bucket [off[b], off[b+1]): split on the next FN_BITS bits inside LDS (a few items per bin),
// then every item counts the smaller items of its bin -- its final place; flag when a sub-bucket is too large
#define FN_BITS 10
#define FN_BINS (1 << FN_BITS)
__global__ __launch_bounds__(FN_THREADS) void k_finish(const uint64_t *__restrict__ in, uint64_t *__restrict__ out,
                                                       const unsigned long long *__restrict__ off, int shift,
                                                       unsigned long long *__restrict__ flag)
{
    __shared__ uint64_t buf[FN_CAP];
    __shared__ uint32_t cnt[FN_BINS], loff[FN_BINS];
    __shared__ uint32_t wsum[FN_THREADS / 64];
    const uint64_t s0 = off[blockIdx.x], s1 = off[blockIdx.x + 1];
    const uint32_t have = (uint32_t)(s1 - s0);
    if (have == 0) return;
    if (have > FN_CAP) { if (threadIdx.x == 0) atomicAdd(flag, 1ull); return; }
    for (int i = threadIdx.x; i < FN_BINS; i += FN_THREADS) cnt[i] = 0;
    __syncthreads();
    uint64_t w[FN_ITEMS];
    uint32_t rk[FN_ITEMS];
#pragma unroll
    for (int r = 0; r < FN_ITEMS; r++) {
        const uint32_t j = threadIdx.x + r * FN_THREADS;
        w[r] = j < have ? in[s0 + j] : ~0ull;
    }
#pragma unroll
    for (int r = 0; r < FN_ITEMS; r++) {
        const uint32_t j = threadIdx.x + r * FN_THREADS;
        rk[r] = j < have ? atomicAdd(&cnt[(w[r] >> shift) & (FN_BINS - 1)], 1u) : 0u;
    }
    __syncthreads();
    {   // exclusive scan of cnt[FN_BINS]: FN_BINS / FN_THREADS consecutive bins per thread
        constexpr int PER = FN_BINS / FN_THREADS;
        const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        uint32_t c[PER], tot = 0;
#pragma unroll
        for (int q = 0; q < PER; q++) { c[q] = cnt[threadIdx.x * PER + q]; tot += c[q]; }
        uint32_t inc = tot;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d, 64); if ((int)lane >= d) inc += o; }
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        uint32_t pre = inc - tot;
        for (uint32_t q = 0; q < wv; q++) pre += wsum[q];
#pragma unroll
        for (int q = 0; q < PER; q++) { loff[threadIdx.x * PER + q] = pre; pre += c[q]; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < FN_ITEMS; r++) {
        const uint32_t j = threadIdx.x + r * FN_THREADS;
        if (j < have) buf[loff[(w[r] >> shift) & (FN_BINS - 1)] + rk[r]] = w[r];
    }
    __syncthreads();
    // final place of every item: start of its bin + the number of smaller items in the bin (words are distinct)
#pragma unroll
    for (int r = 0; r < FN_ITEMS; r++) {
        const uint32_t j = threadIdx.x + r * FN_THREADS;
        if (j < have) {
            const uint32_t bin = (uint32_t)(w[r] >> shift) & (FN_BINS - 1);
            const uint32_t b0 = loff[bin], c = cnt[bin];
            uint32_t smaller = 0;
            for (uint32_t q = 0; q < c; q++) smaller += buf[b0 + q] < w[r] ? 1u : 0u;
            rk[r] = b0 + smaller;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < FN_ITEMS; r++) {
        const uint32_t j = threadIdx.x + r * FN_THREADS;
        if (j < have) buf[rk[r]] = w[r];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < FN_ITEMS; r++) {
        const uint32_t j = threadIdx.x + r * FN_THREADS;
        if (j < have) out[s0 + j] = buf[j];
    }
}

// tile_start / seg_start / cursor for a set of segments whose sizes are hist[] (one thread: <= 512 segments)
__global__ void k_segments(const uint32_t *__restrict__ sizes, uint32_t nseg, uint64_t first, uint64_t *__restrict__ seg_start,
                           uint32_t *__restrict__ tile_start)
{
    if (blockIdx.x || threadIdx.x) return;
    uint64_t s = first;
    uint32_t t = 0;
    for (uint32_t i = 0; i < nseg; i++) {
        seg_start[i] = s; tile_start[i] = t;
        s += sizes[i]; t += (sizes[i] + SP_TILE - 1) / SP_TILE;
    }
    seg_start[nseg] = s; tile_start[nseg] = t;
}

__global__ void k_inv(const uint64_t *o, size_t n, unsigned long long *bad)
{
    for (size_t i = 1 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if (o[i - 1] > o[i]) atomicAdd(bad, 1ull);
}
__global__ void k_xor(const uint64_t *o, size_t n, unsigned long long *acc)
{
    unsigned long long x = 0, s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { x ^= o[i]; s += o[i] * 0x9E3779B97F4A7C15ull; }
    atomicXor(&acc[0], x); atomicAdd(&acc[1], s);
}

int main(int argc, char **argv)
{
    const size_t n = argc > 1 ? strtoull(argv[1], 0, 10) : (1ull << 26);
    const int kb = 34, pb = 30;
    uint64_t *A, *B;
    CK(hipMalloc(&A, n * 8)); CK(hipMalloc(&B, n * 8));
    uint32_t *hist1, *hist2, *tile1, *tile2;
    uint64_t *seg1, *seg2;
    unsigned long long *cur1, *cur2, *off2, *flag, *chk;
    CK(hipMalloc(&hist1, NB * 4)); CK(hipMalloc(&hist2, ((size_t)NB * NB + 1) * 4));
    CK(hipMalloc(&tile1, 2 * 4)); CK(hipMalloc(&tile2, (NB + 1) * 4));
    CK(hipMalloc(&seg1, 2 * 8)); CK(hipMalloc(&seg2, (NB + 1) * 8));
    CK(hipMalloc(&cur1, NB * 8)); CK(hipMalloc(&cur2, (size_t)NB * NB * 8)); CK(hipMalloc(&off2, ((size_t)NB * NB + 1) * 8));
    CK(hipMalloc(&flag, 8)); CK(hipMalloc(&chk, 4 * 8));
    size_t tb = 0;
    CK((rocprim::exclusive_scan(nullptr, tb, hist2, off2, 0ull, (size_t)NB * NB + 1, rocprim::plus<unsigned long long>(), 0)));
    void *tmp; CK(hipMalloc(&tmp, tb + 256));
    hipEvent_t ev[8];
    for (auto &e : ev) CK(hipEventCreate(&e));
    float best[6] = {1e9f, 1e9f, 1e9f, 1e9f, 1e9f, 1e9f};
    for (int it = 0; it < 3; it++) {
        fill<<<4096, 256>>>(A, n, kb, pb);
        CK(hipMemset(hist1, 0, NB * 4)); CK(hipMemset(hist2, 0, ((size_t)NB * NB + 1) * 4)); CK(hipMemset(flag, 0, 8));
        CK(hipDeviceSynchronize());
        // ---- pass 1: one segment, digit = top 9 key bits ---------------------------------------------
        const uint32_t tiles1 = (uint32_t)((n + SP_TILE - 1) / SP_TILE);
        const uint64_t h_seg1[2] = {0, n};
        const uint32_t h_tile1[2] = {0, tiles1};
        CK(hipMemcpy(seg1, h_seg1, 16, hipMemcpyHostToDevice)); CK(hipMemcpy(tile1, h_tile1, 8, hipMemcpyHostToDevice));
        SplitArgs a1{A, B, seg1, tile1, 1, pb + kb - DIG, cur1, hist1};
        CK(hipEventRecord(ev[0]));
        k_hist<<<tiles1, SP_THREADS>>>(a1);
        // cursors of pass 1 = exclusive scan of hist1; the buckets become the segments of pass 2
        k_segments<<<1, 1>>>(hist1, NB, 0, seg2, tile2);
        CK(hipMemcpyAsync(cur1, seg2, NB * 8, hipMemcpyDeviceToDevice));
        CK(hipEventRecord(ev[1]));
        k_split<<<tiles1, SP_THREADS>>>(a1);
        CK(hipEventRecord(ev[2]));
        // ---- pass 2: 512 segments, digit = next 9 bits -------------------------------------------------
        uint32_t tiles2 = 0;
        CK(hipMemcpy(&tiles2, tile2 + NB, 4, hipMemcpyDeviceToHost));
        SplitArgs a2{B, A, seg2, tile2, NB, pb + kb - 2 * DIG, cur2, hist2};
        k_hist<<<tiles2, SP_THREADS>>>(a2);
        {   // sub-bucket offsets: scan of hist2 (as u64)
            // widen in place via a tiny kernel-free trick: scan u32 -> u64 with a transform iterator
            auto in = rocprim::make_transform_iterator(hist2, [] __device__(uint32_t v) { return (unsigned long long)v; });
            size_t have = tb + 256;
            CK((rocprim::exclusive_scan(tmp, have, in, off2, 0ull, (size_t)NB * NB + 1, rocprim::plus<unsigned long long>(), 0)));
        }
        CK(hipMemcpyAsync(cur2, off2, (size_t)NB * NB * 8, hipMemcpyDeviceToDevice));
        CK(hipEventRecord(ev[3]));
        k_split<<<tiles2, SP_THREADS>>>(a2);
        CK(hipEventRecord(ev[4]));
        // ---- pass 3: every sub-bucket in LDS --------------------------------------------------------------
        k_finish<<<NB * NB, FN_THREADS>>>(A, B, off2, pb + kb - 2 * DIG - FN_BITS, flag);
        CK(hipEventRecord(ev[5]));
        CK(hipDeviceSynchronize());
        for (int k = 0; k < 5; k++) { float ms; CK(hipEventElapsedTime(&ms, ev[k], ev[k + 1])); if (ms < best[k]) best[k] = ms; }
        float tot; CK(hipEventElapsedTime(&tot, ev[0], ev[5])); if (tot < best[5]) best[5] = tot;
    }
    unsigned long long h_flag = 0, h_bad = 0, c1[4] = {0, 0, 0, 0}, c2[4] = {0, 0, 0, 0};
    CK(hipMemcpy(&h_flag, flag, 8, hipMemcpyDeviceToHost));
    CK(hipMemset(chk, 0, 32));
    k_inv<<<4096, 256>>>(B, n, chk + 2);
    k_xor<<<4096, 256>>>(B, n, chk);
    CK(hipMemcpy(c1, chk, 32, hipMemcpyDeviceToHost));
    h_bad = c1[2];
    fill<<<4096, 256>>>(A, n, kb, pb);
    CK(hipMemset(chk, 0, 32));
    k_xor<<<4096, 256>>>(A, n, chk);
    CK(hipMemcpy(c2, chk, 32, hipMemcpyDeviceToHost));
    printf("n=%zu  hist1 %.2f  split1 %.2f  hist2+scan %.2f  split2 %.2f  finish %.2f  total %.2f ms\n", n, best[0], best[1], best[2],
           best[3], best[4], best[5]);
    printf("fallback flag %llu, inversions %llu, multiset %s\n", h_flag, h_bad, (c1[0] == c2[0] && c1[1] == c2[1]) ? "same" : "DIFFERENT");
    return (h_bad || h_flag || c1[0] != c2[0] || c1[1] != c2[1]) ? 1 : 0;
}
