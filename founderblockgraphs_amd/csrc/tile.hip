// tile.hip -- lay the per-cell index values out column by column.
//
// The scan needs, for column x, the rank / predecessor-LCP / successor-LCP of the m row pointers
// leaves[i] of fbg.cpp:1594-1600,1687-1691.  In text order those live at p(i,x) = pos_i + rank_i(x),
// i.e. m addresses n apart.  This pass gathers them once and stores them as RT/PLT/PRT[x*mp + i]
// (mp = m rounded up to 64), so that one column is one contiguous run of m values and the scan's
// loads are fully coalesced.  64x64 tiles go through LDS (row pitch 65 words: conflict-free both
// ways); reads follow the rows of the text-order arrays, writes follow the columns of the tiled ones.
#include "fbg_internal.h"

#define TL 64
#define TL_THREADS 256

template <int MODE>  // 0: gap-free forward, 1: gap-free reversed, 2: gapped (prow)
__global__ __launch_bounds__(TL_THREADS) void k_tile(const uint32_t *__restrict__ isa, const uint32_t *__restrict__ pl,
                                                     const uint32_t *__restrict__ pr, const uint32_t *__restrict__ prow,
                                                     const uint32_t *__restrict__ igrow, uint64_t m, uint64_t n,
                                                     uint32_t mp, uint32_t *__restrict__ RT, uint32_t *__restrict__ PLT,
                                                     uint32_t *__restrict__ PRT, uint32_t *__restrict__ PT,
                                                     uint32_t *__restrict__ IGT)
{
    __shared__ uint32_t sT[TL][TL + 1];
    const uint64_t x0 = (uint64_t)blockIdx.x * TL, i0 = (uint64_t)blockIdx.y * TL;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    constexpr int ROWS = TL / (TL_THREADS / 64);
    // text pointer of this thread's cells (row i0 + w + 4k, column x0 + lane)
    uint32_t p[ROWS];
#pragma unroll
    for (int k = 0; k < ROWS; k++) {
        const uint64_t i = i0 + w + k * (TL_THREADS / 64), x = x0 + lane;
        p[k] = 0;
        if (i < m && x < n) {
            if (MODE == 0) p[k] = (uint32_t)(i * (n + 1) + x);
            else if (MODE == 1) p[k] = (uint32_t)(i * (n + 1) + (n - 1 - x));
            else p[k] = prow[i * n + x];
        }
    }
    const int narr = 3 + (MODE == 2 ? 1 : 0) + (igrow ? 1 : 0);
    for (int a = 0; a < narr; a++) {
        const uint32_t *src = a == 0 ? isa : a == 1 ? pl : a == 2 ? pr : nullptr;
        uint32_t *dst = a == 0 ? RT : a == 1 ? PLT : a == 2 ? PRT : (MODE == 2 && a == 3) ? PT : IGT;
        const bool is_ptr = MODE == 2 && a == 3;
        const bool is_ign = !src && !is_ptr;
#pragma unroll
        for (int k = 0; k < ROWS; k++) {
            const int rr = w + k * (TL_THREADS / 64);
            const uint64_t i = i0 + rr, x = x0 + lane;
            if (i < m && x < n) sT[lane][rr] = is_ptr ? p[k] : is_ign ? igrow[i * n + x] : src[p[k]];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < ROWS; k++) {
            const int cc = w + k * (TL_THREADS / 64);
            const uint64_t x = x0 + cc, i = i0 + lane;
            if (x < n && i < m) dst[x * mp + i] = sT[cc][lane];
        }
        __syncthreads();
    }
}

int fbg_tile_columns(fbg_ctx *ctx)
{
    const uint64_t m = ctx->m, n = ctx->n;
    const uint32_t mp = ctx->mp;
    FBG_TRY(fbg_stage_begin(ctx, FBG_STAGE_TILE));
    const size_t bytes = (size_t)n * mp * 4;
    FBG_TRY(fbg_reserve(ctx, ctx->RT, bytes));
    FBG_TRY(fbg_reserve(ctx, ctx->PLT, bytes));
    FBG_TRY(fbg_reserve(ctx, ctx->PRT, bytes));
    if (!ctx->gapfree) FBG_TRY(fbg_reserve(ctx, ctx->PT, bytes));
    if (ctx->have_ignore) FBG_TRY(fbg_reserve(ctx, ctx->IGT, bytes));
    dim3 grid((unsigned)((n + TL - 1) / TL), (unsigned)((m + TL - 1) / TL));
    const uint32_t *isa = ctx->isa.as<uint32_t>(), *pl = ctx->pl.as<uint32_t>(), *pr = ctx->pr.as<uint32_t>();
    const uint32_t *ig = ctx->have_ignore ? ctx->igrow.as<uint32_t>() : nullptr;
    uint32_t *RT = ctx->RT.as<uint32_t>(), *PLT = ctx->PLT.as<uint32_t>(), *PRT = ctx->PRT.as<uint32_t>();
    uint32_t *PT = ctx->gapfree ? nullptr : ctx->PT.as<uint32_t>();
    uint32_t *IGT = ctx->have_ignore ? ctx->IGT.as<uint32_t>() : nullptr;
    if (!ctx->gapfree)
        hipLaunchKernelGGL((k_tile<2>), grid, dim3(TL_THREADS), 0, ctx->stream, isa, pl, pr, ctx->prow.as<uint32_t>(),
                           ig, m, n, mp, RT, PLT, PRT, PT, IGT);
    else if (ctx->reversed)
        hipLaunchKernelGGL((k_tile<1>), grid, dim3(TL_THREADS), 0, ctx->stream, isa, pl, pr, (const uint32_t *)nullptr,
                           ig, m, n, mp, RT, PLT, PRT, PT, IGT);
    else
        hipLaunchKernelGGL((k_tile<0>), grid, dim3(TL_THREADS), 0, ctx->stream, isa, pl, pr, (const uint32_t *)nullptr,
                           ig, m, n, mp, RT, PLT, PRT, PT, IGT);
    FBG_HIP_TRY(ctx, hipGetLastError());
    return fbg_stage_end(ctx, FBG_STAGE_TILE, 1);
}
