// block_graph.hip -- nodes and edges of the elastic founder graph from a segmentation.
//
// output_efg (fbg.cpp:1185-1301) walks the blocks of the segmentation twice, hashing the gap-stripped label of
// every (row, block) into a std::unordered_map to number the distinct labels (nodes) in order of first appearance
// by row, and collects the (node in block j-1, node in block j) pairs of every row into a std::set (edges).  Here:
//
//   k_label_hash    one thread per (row, block), consecutive threads on consecutive blocks of a row (contiguous
//                   bytes): 128-bit hash of the gap-stripped label;
//   k_block_group   one workgroup per block: rows with the same hash meet in an LDS table, the smallest row index
//                   of a group is its representative; every other member compares its label with the
//                   representative's byte by byte (a difference -- a hash collision -- raises a flag and the caller
//                   uses its own hashing instead: results are exact, never probabilistic); representatives in row
//                   order are the block's nodes in the reference's numbering;
//   k_block_edges   one workgroup per block: (previous node, node) pairs of the rows, bitonic sort + unique in LDS.
//
// The host writer then only formats: B counts, S lines from the representative rows, L lines, P lines.
#include "fbg_internal.h"
#include <rocprim/rocprim.hpp>
#include <algorithm>

#define BG_NONE 0xffffffffu
#define BG_THREADS 256

struct BgArgs {
    const uint8_t *msa;
    uint64_t m, n, nb;
    const uint64_t *bounds;        // [nb] block ends (fbg.cpp:2026-2039: the last one is n)
    uint64_t *h1, *h2;             // [nb * m] label hashes, block-major; h2 == 0 && h1 == 0: empty label
    uint32_t *node_of;             // [nb * m]
    uint32_t *rep_row;             // [nb * m]
    uint32_t *count;               // [nb] nodes per block
    const unsigned long long *first;   // [nb + 1] exclusive scan of count
    unsigned long long *edge_count;    // [nb]
    unsigned long long *edges;         // [nb * m]
    unsigned long long *flag;      // != 0: two different labels with the same hash
};

__device__ __forceinline__ void bg_block_range(const BgArgs &a, uint64_t j, uint64_t &x0, uint64_t &x1)
{
    x0 = j ? a.bounds[j - 1] + 1 : 0;
    x1 = min(a.bounds[j] + 1, a.n);                // [x0, x1): std::string::substr clamps at the row end
    if (x0 > x1) x0 = x1;
}

__global__ void k_label_hash(BgArgs a)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.m * a.nb) return;
    const uint64_t i = t / a.nb, j = t % a.nb;
    uint64_t x0, x1;
    bg_block_range(a, j, x0, x1);
    const uint8_t *row = a.msa + i * a.n;
    uint64_t u = 0x9E3779B97F4A7C15ull, v = 0xC2B2AE3D27D4EB4Full, len = 0;
    for (uint64_t x = x0; x < x1; x++) {
        const uint64_t c = row[x];
        if (c == '-') continue;
        u = (u ^ c) * 0x100000001B3ull;
        u ^= u >> 29;
        v = (v + c + 1) * 0xD6E8FEB86659FD93ull;
        v ^= v >> 32;
        len++;
    }
    if (len == 0) { u = 0; v = 0; }
    else { u ^= len * 0xFF51AFD7ED558CCDull; if (u == 0 && v == 0) v = 1; }
    a.h1[j * a.m + i] = u;
    a.h2[j * a.m + i] = v;
}

// gap-stripped labels of rows p and q over [x0, x1) equal?
__device__ __forceinline__ bool bg_same_label(const BgArgs &a, uint64_t p, uint64_t q, uint64_t x0, uint64_t x1)
{
    const uint8_t *rp = a.msa + p * a.n, *rq = a.msa + q * a.n;
    uint64_t xp = x0, xq = x0;
    for (;;) {
        while (xp < x1 && rp[xp] == '-') xp++;
        while (xq < x1 && rq[xq] == '-') xq++;
        if (xp >= x1 || xq >= x1) return xp >= x1 && xq >= x1;
        if (rp[xp] != rq[xq]) return false;
        xp++; xq++;
    }
}

// dynamic LDS: table[ts] (owner row of a slot), minrow[ts], slot_of[m], flags/scan scratch -- up to 4096 rows; beyond, the
// same arrays live in global memory (scratch: one stretch per workgroup, L2-resident), and the workgroups stay and take
// block after block
__global__ __launch_bounds__(BG_THREADS) void k_block_group(BgArgs a, uint32_t ts, uint32_t *__restrict__ scratch)
{
    extern __shared__ uint32_t bg_lds[];
    uint32_t *base = scratch ? scratch + (size_t)blockIdx.x * (2 * (size_t)ts + 2 * a.m) : bg_lds;
    uint32_t *table = base, *minrow = base + ts, *slot_of = base + 2 * ts, *rank = slot_of + a.m;
    __shared__ uint32_t wsum[BG_THREADS / 64];
    __shared__ uint32_t carry;
    const uint32_t m = (uint32_t)a.m, mask = ts - 1;
  for (uint64_t j = blockIdx.x; j < a.nb; j += gridDim.x) {
    __syncthreads();
    const uint64_t *h1 = a.h1 + j * a.m, *h2 = a.h2 + j * a.m;
    for (uint32_t s = threadIdx.x; s < ts; s += BG_THREADS) { table[s] = BG_NONE; minrow[s] = BG_NONE; }
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < m; i += BG_THREADS) {
        const uint64_t u = h1[i], v = h2[i];
        uint32_t where = BG_NONE;
        if (u | v) {
            uint32_t s = (uint32_t)(u ^ (u >> 32)) & mask;
            for (;;) {
                const uint32_t old = atomicCAS(&table[s], BG_NONE, i);
                if (old == BG_NONE || (h1[old] == u && h2[old] == v)) { where = s; break; }
                s = (s + 1) & mask;
            }
            atomicMin(&minrow[where], i);
        }
        slot_of[i] = where;
    }
    __syncthreads();
    uint64_t x0, x1;
    bg_block_range(a, j, x0, x1);
    // representatives in row order = the block's nodes; exact check of every other member against its representative
    for (uint32_t i0 = 0; i0 < m; i0 += BG_THREADS) {
        const uint32_t i = i0 + threadIdx.x;
        uint32_t rep = BG_NONE;
        if (i < m && slot_of[i] != BG_NONE) {
            rep = minrow[slot_of[i]];
            if (rep != i && !bg_same_label(a, i, rep, x0, x1)) *a.flag = 1;
        }
        const bool is_rep = rep == i && i < m;
        // exclusive prefix count of representatives over the rows
        const unsigned long long bal = __ballot(is_rep);
        const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        if (lane == 0) wsum[wv] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t pre = carry, tot = 0;
        for (uint32_t q = 0; q < BG_THREADS / 64; q++) { if (q < wv) pre += wsum[q]; tot += wsum[q]; }
        const uint32_t mine = pre + (uint32_t)__popcll(bal & ((1ull << lane) - 1));
        if (i < m) rank[i] = is_rep ? mine : BG_NONE;
        if (is_rep) a.rep_row[j * a.m + mine] = i;
        __syncthreads();
        if (threadIdx.x == 0) carry += tot;
        __syncthreads();
    }
    for (uint32_t i = threadIdx.x; i < m; i += BG_THREADS)
        a.node_of[j * a.m + i] = slot_of[i] == BG_NONE ? BG_NONE : rank[minrow[slot_of[i]]];
    if (threadIdx.x == 0) a.count[j] = carry;
  }
}

// local node numbers -> global ones; edges into block j: sorted distinct (node of block j-1, node of block j)
__global__ __launch_bounds__(BG_THREADS) void k_block_edges(BgArgs a, uint32_t cap, unsigned long long *__restrict__ scratch)
{
    extern __shared__ unsigned long long bg_pairs_lds[];   // cap >= m, a power of two (more than 4096 rows: global scratch)
    unsigned long long *bg_pairs = scratch ? scratch + (size_t)blockIdx.x * cap : bg_pairs_lds;
    __shared__ uint32_t wsum[BG_THREADS / 64];
    __shared__ uint32_t carry;
    const uint32_t m = (uint32_t)a.m;
  for (uint64_t j = blockIdx.x; j < a.nb; j += gridDim.x) {
    __syncthreads();
    const unsigned long long f1 = a.first[j], f0 = j ? a.first[j - 1] : 0;
    for (uint32_t i = threadIdx.x; i < cap; i += BG_THREADS) {
        unsigned long long pr = ~0ull;
        if (i < m && j > 0) {
            const uint32_t c = a.node_of[j * a.m + i], p = a.node_of[(j - 1) * a.m + i];
            if (c != BG_NONE && p != BG_NONE) pr = ((f0 + p) << 32) | (f1 + c);     // block j-1 still holds local numbers
        }
        bg_pairs[i] = pr;
    }
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t k = 2; k <= cap; k <<= 1)
        for (uint32_t s = k >> 1; s > 0; s >>= 1) {
            for (uint32_t i = threadIdx.x; i < cap; i += BG_THREADS) {
                const uint32_t q = i ^ s;
                if (q > i) {
                    const unsigned long long x = bg_pairs[i], y = bg_pairs[q];
                    const bool up = (i & k) == 0;
                    if ((x > y) == up) { bg_pairs[i] = y; bg_pairs[q] = x; }
                }
            }
            __syncthreads();
        }
    for (uint32_t i0 = 0; i0 < cap; i0 += BG_THREADS) {
        const uint32_t i = i0 + threadIdx.x;
        const unsigned long long x = i < cap ? bg_pairs[i] : ~0ull;
        const bool keep = x != ~0ull && (i == 0 || bg_pairs[i - 1] != x);
        const unsigned long long bal = __ballot(keep);
        const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        if (lane == 0) wsum[wv] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t pre = carry, tot = 0;
        for (uint32_t q = 0; q < BG_THREADS / 64; q++) { if (q < wv) pre += wsum[q]; tot += wsum[q]; }
        if (keep) a.edges[j * a.m + pre + (uint32_t)__popcll(bal & ((1ull << lane) - 1))] = x;
        __syncthreads();
        if (threadIdx.x == 0) carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) a.edge_count[j] = carry;
  }
}

// second launch, after all edges are out: node_of local -> global (k_block_edges reads the local numbers of the block before)
__global__ void k_block_globalize(BgArgs a)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.m * a.nb) return;
    const uint32_t c = a.node_of[t];
    if (c != BG_NONE) a.node_of[t] = (uint32_t)(a.first[t / a.m] + c);
}

__global__ void k_bg_widen(const uint32_t *__restrict__ c, unsigned long long *__restrict__ w, uint64_t cnt)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cnt) w[i] = c[i];
}

int fbg_block_graph(fbg_ctx *ctx, const uint64_t *boundaries, uint64_t nb, uint32_t *node_of, uint64_t *first_node,
                    uint32_t *rep_row, uint64_t *edge_count, uint64_t *edges)
{
    if (!ctx) return FBG_ERR_INVALID;
    if (!ctx->d_msa) return fbg_fail(ctx, FBG_ERR_INVALID, "fbg_block_graph: no MSA set");
    if (!boundaries || nb == 0 || !node_of || !first_node || !rep_row || !edge_count || !edges)
        return fbg_fail(ctx, FBG_ERR_INVALID, "fbg_block_graph: bad arguments");
    const uint64_t m = ctx->m, n = ctx->n;
    if (m * nb >= (1ull << 32)) return fbg_fail(ctx, FBG_ERR_TOO_LARGE, "fbg_block_graph: more than 2^32 (row, block) cells");
    if (m > FBG_MAX_ROWS) return fbg_fail(ctx, FBG_ERR_TOO_LARGE, "fbg_block_graph: more than %d rows", FBG_MAX_ROWS);
    for (uint64_t j = 0; j < nb; j++)
        if (boundaries[j] > n || (j && boundaries[j] <= boundaries[j - 1]))
            return fbg_fail(ctx, FBG_ERR_INVALID, "fbg_block_graph: boundaries must increase and end at most at n");
    FBG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const uint64_t cells = m * nb;
    FBG_TRY(fbg_reserve(ctx, ctx->io_a, nb * 8));
    FBG_TRY(fbg_reserve(ctx, ctx->dp_a, cells * 8));       // h1
    FBG_TRY(fbg_reserve(ctx, ctx->dp_b, cells * 8));       // h2, later the edges
    FBG_TRY(fbg_reserve(ctx, ctx->dp_c, cells * 4));       // node_of
    FBG_TRY(fbg_reserve(ctx, ctx->dp_d, cells * 4));       // rep_row
    FBG_TRY(fbg_reserve(ctx, ctx->dp_e, nb * 4));          // count
    FBG_TRY(fbg_reserve(ctx, ctx->dp_f, (nb + 1) * 8 * 2)); // widened counts, first
    FBG_TRY(fbg_reserve(ctx, ctx->dp_g, nb * 8));          // edge_count
    FBG_TRY(fbg_reserve(ctx, ctx->scalars, 256 * sizeof(unsigned long long)));
    BgArgs a;
    a.msa = ctx->d_msa; a.m = m; a.n = n; a.nb = nb;
    a.bounds = ctx->io_a.as<uint64_t>();
    a.h1 = ctx->dp_a.as<uint64_t>(); a.h2 = ctx->dp_b.as<uint64_t>();
    a.node_of = ctx->dp_c.as<uint32_t>(); a.rep_row = ctx->dp_d.as<uint32_t>(); a.count = ctx->dp_e.as<uint32_t>();
    unsigned long long *wide = ctx->dp_f.as<unsigned long long>(), *first = wide + (nb + 1);
    a.first = first;
    a.edge_count = ctx->dp_g.as<unsigned long long>();
    a.edges = nullptr;
    a.flag = ctx->scalars.as<unsigned long long>() + 110;
    FBG_HIP_TRY(ctx, hipMemcpyAsync(ctx->io_a.p, boundaries, nb * 8, hipMemcpyHostToDevice, st));
    FBG_HIP_TRY(ctx, hipMemsetAsync(a.flag, 0, 8, st));
    hipLaunchKernelGGL(k_label_hash, dim3(fbg_blocks(cells, 256)), dim3(256), 0, st, a);
    uint32_t ts = 64;
    while (ts < 2 * m) ts <<= 1;
    const size_t lds1 = ((size_t)2 * ts + 2 * m) * 4;
    const bool in_lds = m <= 4096;
    const unsigned wgs = in_lds ? (unsigned)nb : (unsigned)std::min<uint64_t>(nb, 512);   // beyond LDS: workgroups that stay, a stretch of scratch each
    uint32_t *scratch1 = nullptr;
    if (in_lds) {
        if (lds1 > 64 * 1024)
            FBG_HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_block_group, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
    } else {
        uint32_t cap2 = 64;
        while (cap2 < m) cap2 <<= 1;
        FBG_TRY(fbg_reserve(ctx, ctx->dp_h, std::max((size_t)wgs * lds1, (size_t)wgs * cap2 * 8)));   // (k_block_edges' stretches too)
        scratch1 = ctx->dp_h.as<uint32_t>();
    }
    hipLaunchKernelGGL(k_block_group, dim3(wgs), dim3(BG_THREADS), in_lds ? lds1 : 0, st, a, ts, scratch1);
    // first node of every block: exclusive scan of the counts
    hipLaunchKernelGGL(k_bg_widen, dim3(fbg_blocks(nb, 256)), dim3(256), 0, st, a.count, wide, nb);
    FBG_HIP_TRY(ctx, hipMemsetAsync(wide + nb, 0, 8, st));
    {
        size_t bytes = 0;
        hipError_t e = rocprim::exclusive_scan(nullptr, bytes, wide, first, 0ull, (size_t)(nb + 1), rocprim::plus<unsigned long long>(), st);
        if (e != hipSuccess) return fbg_fail(ctx, FBG_ERR_HIP, "rocprim scan size query: %s", hipGetErrorString(e));
        FBG_TRY(fbg_reserve(ctx, ctx->tmp, bytes));
        size_t have = ctx->tmp.cap;
        e = rocprim::exclusive_scan(ctx->tmp.p, have, wide, first, 0ull, (size_t)(nb + 1), rocprim::plus<unsigned long long>(), st);
        if (e != hipSuccess) return fbg_fail(ctx, FBG_ERR_HIP, "rocprim exclusive_scan: %s", hipGetErrorString(e));
    }
    a.edges = ctx->dp_b.as<unsigned long long>();          // the hashes are not needed any more
    uint32_t cap = 64;
    while (cap < m) cap <<= 1;
    unsigned long long *scratch2 = nullptr;
    if (!in_lds) {
        scratch2 = ctx->dp_h.as<unsigned long long>();                      // (k_block_group is done with its stretches)
    }
    hipLaunchKernelGGL(k_block_edges, dim3(wgs), dim3(BG_THREADS), in_lds ? (size_t)cap * 8 : 0, st, a, cap, scratch2);
    hipLaunchKernelGGL(k_block_globalize, dim3(fbg_blocks(cells, 256)), dim3(256), 0, st, a);
    unsigned long long h_flag = 0;
    FBG_HIP_TRY(ctx, hipMemcpyAsync(&h_flag, a.flag, 8, hipMemcpyDeviceToHost, st));
    FBG_HIP_TRY(ctx, hipMemcpyAsync(first_node, first, (nb + 1) * 8, hipMemcpyDeviceToHost, st));
    FBG_HIP_TRY(ctx, hipMemcpyAsync(edge_count, a.edge_count, nb * 8, hipMemcpyDeviceToHost, st));
    FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
    // the three per-(row, block) arrays are hundreds of MB at a million columns: pinned bounce buffers, several threads
    FBG_TRY(fbg_download(ctx, node_of, a.node_of, cells * 4));
    FBG_TRY(fbg_download(ctx, rep_row, a.rep_row, cells * 4));
    FBG_TRY(fbg_download(ctx, edges, a.edges, cells * 8));
    FBG_HIP_TRY(ctx, hipGetLastError());
    if (h_flag != 0)
        return fbg_fail(ctx, FBG_ERR_HASH_COLLISION, "two different block labels share a 128-bit hash; use the host-side numbering");
    return FBG_OK;
}
