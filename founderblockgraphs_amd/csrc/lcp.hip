// lcp.hip -- neighbour LCPs by text comparison (the robust path).
//
// For every text position p, with r = rank of suffix p:
//     lcp_prev = lcp(T[p..], T[SA[r-1]..])      (= LCP[ISA[p]],     0 for the first suffix)
//     lcp_next = lcp(T[p..], T[SA[r+1]..])      (= LCP[ISA[p]+1],   0 for the last suffix)
// These two numbers are all the reference's suffix-tree walk ever extracts for a leaf at the edge of a
// run: depth(parent(w)) in fbg.cpp:1656 is a minimum of such values.  They are
// written into words y/z of the position's record (suffix_sort.hip), bit 31 carrying the run hint:
// "the SA neighbour can be a coloured row pointer in the same column as p".
//
// suffix_sort.hip already derives both words from the sorted keys when few suffixes tie on their first
// K symbols and the MSA has no gaps.  This kernel is used otherwise (similar rows: long matches; gaps:
// the hint needs the column interval of each pointer).  Each thread walks a chunk of consecutive text
// positions and carries the match length from p to p+1 (lcp(p+1, .) >= lcp(p, .) - 1, Kasai et al.),
// so long matches are extended, never re-read.  Text is compared 8 bytes at a time.
#include "fbg_internal.h"
#include "text_cmp.h"

#define LCP_THREADS 256
#define LCP_CHUNK 32

struct HintArgs {
    const uint32_t *colT;   // gapped MSAs: column of every text position ('#' and sentinel: n); else nullptr
    uint64_t n, N;
    uint32_t row_len;       // gap-free: n + 1
};

// can positions p and q both be the pointer of an active row in one column?
__device__ __forceinline__ uint32_t run_hint(const HintArgs &h, uint32_t p, uint32_t q)
{
    if (p == h.N - 1 || q == h.N - 1) return 0;
    if (!h.colT) return (p % h.row_len) == (q % h.row_len) ? 0x80000000u : 0u;
    // row pointer p is current for the columns (column of the previous symbol, column of p], fbg.cpp:1687-1691
    const uint32_t n = (uint32_t)h.n;
    const uint32_t cp = h.colT[p], cq = h.colT[q];
    const uint32_t hp = cp < n ? cp : n - 1, hq = cq < n ? cq : n - 1;
    uint32_t lp = 0, lq = 0;
    if (p > 0) { const uint32_t c = h.colT[p - 1]; lp = c >= n ? 0 : c + 1; }
    if (q > 0) { const uint32_t c = h.colT[q - 1]; lq = c >= n ? 0 : c + 1; }
    const uint32_t lo = lp > lq ? lp : lq, hi = hp < hq ? hp : hq;
    return lo <= hi ? 0x80000000u : 0u;
}

__global__ __launch_bounds__(LCP_THREADS) void k_neighbour_lcp(const uint8_t *__restrict__ T, uint64_t N,
                                                               const uint32_t *__restrict__ sa,
                                                               uint4 *__restrict__ rec, HintArgs ha)
{
    const uint64_t t = (uint64_t)blockIdx.x * LCP_THREADS + threadIdx.x;
    const uint64_t p0 = t * LCP_CHUNK;
    if (p0 >= N) return;
    const uint64_t p1 = p0 + LCP_CHUNK < N ? p0 + LCP_CHUNK : N;
    uint32_t hl = 0, hr = 0;
    for (uint64_t p = p0; p < p1; p++) {
        const uint32_t r = rec[p].x;
        uint32_t wl = 0, wr = 0;
        if (r > 0) {
            const uint32_t q = sa[r - 1];
            hl = fbg_extend_match(T, p, q, hl);
            wl = fbg_clamp_lcp(hl) | run_hint(ha, (uint32_t)p, q);
        } else {
            hl = 0;
        }
        if ((uint64_t)r + 1 < N) {
            const uint32_t q = sa[r + 1];
            hr = fbg_extend_match(T, p, q, hr);
            wr = fbg_clamp_lcp(hr) | run_hint(ha, (uint32_t)p, q);
        } else {
            hr = 0;
        }
        rec[p].y = wl;
        rec[p].z = wr;
        hl = hl > 0 ? hl - 1 : 0;
        hr = hr > 0 ? hr - 1 : 0;
    }
}

int fbg_neighbour_lcp(fbg_ctx *ctx)
{
    const uint64_t N = ctx->N;
    FBG_TRY(fbg_stage_begin(ctx, FBG_STAGE_LCP));
    int launches = 0;
    if (!ctx->lcp_from_keys && !ctx->ranked && !ctx->granked) {
        HintArgs ha;
        ha.colT = ctx->gapfree ? nullptr : ctx->colT.as<uint32_t>();
        ha.n = ctx->n; ha.N = N; ha.row_len = (uint32_t)(ctx->n + 1);
        const uint64_t threads = (N + LCP_CHUNK - 1) / LCP_CHUNK;
        hipLaunchKernelGGL(k_neighbour_lcp, dim3(fbg_blocks(threads, LCP_THREADS)), dim3(LCP_THREADS), 0, ctx->stream,
                           ctx->text.as<uint8_t>(), N, ctx->sa_ptr, ctx->rec.as<uint4>(), ha);
        FBG_HIP_TRY(ctx, hipGetLastError());
        launches = 1;
    }
    return fbg_stage_end(ctx, FBG_STAGE_LCP, launches);
}
