// lcp.hip -- for every text position p the longest common prefix of suffix p with its predecessor
// and with its successor in suffix-array order:
//     pl[p] = lcp(T[p..], T[SA[ISA[p]-1]..])      (= LCP[ISA[p]],     0 for the first suffix)
//     pr[p] = lcp(T[p..], T[SA[ISA[p]+1]..])      (= LCP[ISA[p]+1],   0 for the last suffix)
// These two numbers are all the reference's suffix-tree walk ever extracts for a leaf at the edge of
// a run: depth(parent(w)) in fbg.cpp:1656 is a minimum of such values.
//
// Each thread walks a chunk of consecutive text positions and carries the match length from p to
// p+1 (lcp(p+1, .) >= lcp(p, .) - 1, Kasai et al.), so long matches are extended, never re-read.
// Text is compared 8 bytes at a time; the text buffer is zero padded past the unique sentinel.
#include "fbg_internal.h"

#define LCP_THREADS 256
#define LCP_CHUNK 32

__device__ __forceinline__ uint64_t load8(const uint8_t *__restrict__ T, uint64_t p)
{
    // unaligned 8-byte read assembled from two aligned words
    const uint64_t *w = reinterpret_cast<const uint64_t *>(T) + (p >> 3);
    const unsigned s = (unsigned)(p & 7) * 8;
    uint64_t lo = w[0];
    if (s == 0) return lo;
    uint64_t hi = w[1];
    return (lo >> s) | (hi << (64 - s));
}

__device__ __forceinline__ uint32_t extend_match(const uint8_t *__restrict__ T, uint64_t p, uint64_t q, uint32_t h)
{
    for (;;) {
        uint64_t x = load8(T, p + h) ^ load8(T, q + h);
        if (x) return h + (uint32_t)(__ffsll((unsigned long long)x) - 1) / 8;
        h += 8;
    }
}

__global__ __launch_bounds__(LCP_THREADS) void k_neighbour_lcp(const uint8_t *__restrict__ T, uint64_t N,
                                                               const uint32_t *__restrict__ sa,
                                                               const uint32_t *__restrict__ isa,
                                                               uint32_t *__restrict__ pl, uint32_t *__restrict__ pr)
{
    const uint64_t t = (uint64_t)blockIdx.x * LCP_THREADS + threadIdx.x;
    const uint64_t p0 = t * LCP_CHUNK;
    if (p0 >= N) return;
    const uint64_t p1 = p0 + LCP_CHUNK < N ? p0 + LCP_CHUNK : N;
    uint32_t hl = 0, hr = 0;
    for (uint64_t p = p0; p < p1; p++) {
        const uint32_t r = isa[p];
        if (r > 0) {
            hl = extend_match(T, p, sa[r - 1], hl);
            pl[p] = hl;
        } else {
            hl = 0; pl[p] = 0;
        }
        if ((uint64_t)r + 1 < N) {
            hr = extend_match(T, p, sa[r + 1], hr);
            pr[p] = hr;
        } else {
            hr = 0; pr[p] = 0;
        }
        hl = hl > 0 ? hl - 1 : 0;
        hr = hr > 0 ? hr - 1 : 0;
    }
}

int fbg_neighbour_lcp(fbg_ctx *ctx)
{
    const uint64_t N = ctx->N;
    FBG_TRY(fbg_stage_begin(ctx, FBG_STAGE_LCP));
    FBG_TRY(fbg_reserve(ctx, ctx->pl, N * 4));
    FBG_TRY(fbg_reserve(ctx, ctx->pr, N * 4));
    const uint64_t threads = (N + LCP_CHUNK - 1) / LCP_CHUNK;
    hipLaunchKernelGGL(k_neighbour_lcp, dim3(fbg_blocks(threads, LCP_THREADS, 0x7fffffffu)), dim3(LCP_THREADS), 0,
                       ctx->stream, ctx->text.as<uint8_t>(), N, ctx->sa.as<uint32_t>(), ctx->isa.as<uint32_t>(),
                       ctx->pl.as<uint32_t>(), ctx->pr.as<uint32_t>());
    FBG_HIP_TRY(ctx, hipGetLastError());
    return fbg_stage_end(ctx, FBG_STAGE_LCP, 1);
}
