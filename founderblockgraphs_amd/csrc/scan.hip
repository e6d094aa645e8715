// scan.hip -- the per-column extension scan: compute_f (fbg.cpp:1579-1695) and, on the reversed
// text, the v[j] scan of segment() (fbg.cpp:552-611).
//
// What the reference does per column x with a suffix-tree walk is, in array terms:
//   * the m row pointers leaves[i] have SA ranks r_i; the "coloured" leaves are those of the active
//     rows (all rows for v[]; rows that already emitted a character for f[], fbg.cpp:1621);
//   * coloured ranks that are consecutive integers form a run [lb..rb] (fbg.cpp:1633-1641);
//   * for a member r of the run, depth(parent(exclusive ancestor)) (fbg.cpp:1656) equals
//         max( min(LCP[lb..r]), min(LCP[r+1..rb+1]) )
//     and g = that + 1 is how far row i must be extended to the right of x.
// One workgroup owns one column at a time.  Its m (rank, lcp-prev, lcp-next) triples are one
// contiguous run of the column-tiled tables (tile.hip).  Runs are found without sorting: the ranks
// go into an LDS hash set, each member looks up rank-1 / rank+1, and the two running minima are
// propagated along the run by pointer jumping (log2(run length) rounds, skipped entirely when the
// column has no two consecutive ranks -- the common case on dissimilar rows).
#include "fbg_internal.h"

#define SC_THREADS 256
#define SC_MAX_RPT 16            // rows per thread: m <= SC_THREADS * SC_MAX_RPT = 4096
#define SC_EMPTY 0xffffffffu
#define SC_NONE 0xffffu

struct ScanArgs {
    const uint32_t *RT, *PLT, *PRT, *PT, *IGT;  // column-tiled tables (PT, IGT optional)
    const uint32_t *pos, *tot, *colT;          // per-row / per-text-position tables
    uint64_t m, n, N;
    uint32_t mp;
    uint32_t H, logH;                          // hash slots (power of two >= 2m)
    int mode, disable_tricks;
    uint64_t x0, x1;
    uint64_t *out;
};

__device__ __forceinline__ uint32_t sc_hash(uint32_t r, uint32_t logH) { return (r * 2654435761u) >> (32 - logH); }

__global__ __launch_bounds__(SC_THREADS) void k_scan_columns(ScanArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t *hkey = reinterpret_cast<uint32_t *>(smem);          // H
    uint32_t *rk = hkey + a.H;                                    // m   rank (SC_EMPTY = inactive row)
    uint32_t *vL = rk + a.m;                                      // m   running min towards the run head
    uint32_t *vR = vL + a.m;                                      // m   running min towards the run tail
    uint16_t *hrow = reinterpret_cast<uint16_t *>(vR + a.m);      // H
    uint16_t *ptrL = hrow + a.H;                                  // m
    uint16_t *ptrR = ptrL + a.m;                                  // m
    __shared__ unsigned long long red[SC_THREADS / 64];

    const uint32_t tid = threadIdx.x, Hm = a.H - 1;
    const uint32_t m = (uint32_t)a.m;

    for (uint64_t x = a.x0 + blockIdx.x; x < a.x1; x += gridDim.x) {
        const uint64_t colbase = x * a.mp;
        for (uint32_t s = tid; s < a.H; s += SC_THREADS) hkey[s] = SC_EMPTY;
        __syncthreads();

        // ---- load the column, colour the active rows, insert their ranks -------------------
        for (uint32_t i = tid; i < m; i += SC_THREADS) {
            uint32_t r = a.RT[colbase + i];
            bool active = true;
            if (a.mode == FBG_SCAN_F && !a.disable_tricks) {
                // fullrow[i] (fbg.cpp:1605-1608,1621): row i has not emitted a character yet
                uint32_t nz = a.PT ? a.PT[colbase + i] - a.pos[i] : (uint32_t)x;
                active = nz != 0;
            }
            vL[i] = a.PLT[colbase + i];
            vR[i] = a.PRT[colbase + i];
            if (active) {
                uint32_t s = sc_hash(r, a.logH);
                for (;;) {
                    uint32_t prev = atomicCAS(&hkey[s], SC_EMPTY, r);
                    if (prev == SC_EMPTY) { hrow[s] = (uint16_t)i; break; }
                    s = (s + 1) & Hm;
                }
            } else {
                r = SC_EMPTY;
            }
            rk[i] = r;
        }
        __syncthreads();

        // ---- neighbours in rank order --------------------------------------------------------
        int anylink = 0;
        for (uint32_t i = tid; i < m; i += SC_THREADS) {
            const uint32_t r = rk[i];
            uint16_t pl_ = SC_NONE, pr_ = SC_NONE;
            if (r != SC_EMPTY) {
                if (r > 0) {
                    uint32_t key = r - 1, s = sc_hash(key, a.logH);
                    for (;;) {
                        uint32_t k = hkey[s];
                        if (k == key) { pl_ = hrow[s]; break; }
                        if (k == SC_EMPTY) break;
                        s = (s + 1) & Hm;
                    }
                }
                if ((uint64_t)r + 1 < a.N) {
                    uint32_t key = r + 1, s = sc_hash(key, a.logH);
                    for (;;) {
                        uint32_t k = hkey[s];
                        if (k == key) { pr_ = hrow[s]; break; }
                        if (k == SC_EMPTY) break;
                        s = (s + 1) & Hm;
                    }
                }
            }
            ptrL[i] = pl_;
            ptrR[i] = pr_;
            anylink |= (pl_ != SC_NONE);
        }
        anylink = __syncthreads_or(anylink);

        // ---- running minima along the runs (pointer jumping) -------------------------------
        if (anylink) {
            for (;;) {
                uint32_t nvL[SC_MAX_RPT], nvR[SC_MAX_RPT];
                uint16_t naL[SC_MAX_RPT], naR[SC_MAX_RPT];
                int more = 0;
#pragma unroll
                for (int k = 0; k < SC_MAX_RPT; k++) {
                    const uint32_t i = tid + k * SC_THREADS;
                    if (i < m) {
                        const uint16_t pa = ptrL[i], pb = ptrR[i];
                        nvL[k] = vL[i]; naL[k] = pa;
                        nvR[k] = vR[i]; naR[k] = pb;
                        if (pa != SC_NONE) { nvL[k] = min(nvL[k], vL[pa]); naL[k] = ptrL[pa]; more |= naL[k] != SC_NONE; }
                        if (pb != SC_NONE) { nvR[k] = min(nvR[k], vR[pb]); naR[k] = ptrR[pb]; more |= naR[k] != SC_NONE; }
                    }
                }
                __syncthreads();
#pragma unroll
                for (int k = 0; k < SC_MAX_RPT; k++) {
                    const uint32_t i = tid + k * SC_THREADS;
                    if (i < m) { vL[i] = nvL[k]; ptrL[i] = naL[k]; vR[i] = nvR[k]; ptrR[i] = naR[k]; }
                }
                more = __syncthreads_or(more);
                if (!more) break;
            }
        }

        // ---- extension per row, column maximum ------------------------------------------------
        unsigned long long best = 0;
        for (uint32_t i = tid; i < m; i += SC_THREADS) {
            if (rk[i] == SC_EMPTY) continue;
            const unsigned long long g = (unsigned long long)max(vL[i], vR[i]) + 1;   // fbg.cpp:1656
            if (a.mode == FBG_SCAN_V) {
                best = max(best, g);
            } else {
                unsigned long long fi;
                if (a.PT) {
                    const uint32_t p = a.PT[colbase + i], p0 = a.pos[i], tt = a.tot[i];
                    const unsigned long long gg = (unsigned long long)(p - p0) + g;   // fbg.cpp:1657
                    if (gg > tt) fi = a.disable_tricks ? a.n : a.colT[p0 + tt - 1];   // 1659-1664
                    else fi = a.colT[p0 + gg - 1];                                    // 1666
                } else {
                    const unsigned long long gg = x + g;
                    if (gg > a.n) fi = a.disable_tricks ? a.n : a.n - 1;
                    else fi = gg - 1;
                }
                if (a.IGT) {                                                          // 1669-1670
                    const uint32_t ig = a.IGT[colbase + i];
                    if (ig < a.n) fi = min(fi, (unsigned long long)ig);
                }
                best = max(best, fi);
            }
        }
        for (int d = 32; d >= 1; d >>= 1) best = max(best, (unsigned long long)__shfl_down(best, d, 64));
        if ((tid & 63) == 0) red[tid >> 6] = best;
        __syncthreads();
        if (tid == 0) {
            for (int k = 1; k < SC_THREADS / 64; k++) best = max(best, red[k]);
            if (a.mode == FBG_SCAN_V) {
                // v[j] = j+1-L when the block [v..j] fits in the row, else j+1
                a.out[x] = best <= x + 1 ? x + 1 - best : x + 1;
            } else {
                unsigned long long fx = max((unsigned long long)x, best);             // fbg.cpp:1618
                a.out[x] = max((unsigned long long)a.out[x], fx);                     // fbg.cpp:1681
            }
        }
        __syncthreads();
    }
}

int fbg_scan_columns(fbg_ctx *ctx, uint64_t x0, uint64_t x1, int mode, int disable_tricks, uint64_t *d_out)
{
    if (ctx->m > (uint64_t)SC_THREADS * SC_MAX_RPT)
        return fbg_fail(ctx, FBG_ERR_TOO_LARGE, "scan kernel supports m <= %d rows (got %llu)", SC_THREADS * SC_MAX_RPT,
                        (unsigned long long)ctx->m);
    FBG_TRY(fbg_stage_begin(ctx, FBG_STAGE_SCAN));
    if (x1 > x0) {
        ScanArgs a;
        a.RT = ctx->RT.as<uint32_t>(); a.PLT = ctx->PLT.as<uint32_t>(); a.PRT = ctx->PRT.as<uint32_t>();
        a.PT = ctx->gapfree ? nullptr : ctx->PT.as<uint32_t>();
        a.IGT = (mode == FBG_SCAN_F && ctx->have_ignore) ? ctx->IGT.as<uint32_t>() : nullptr;
        a.pos = ctx->pos.as<uint32_t>(); a.tot = ctx->tot.as<uint32_t>();
        a.colT = ctx->gapfree ? nullptr : ctx->colT.as<uint32_t>();
        a.m = ctx->m; a.n = ctx->n; a.N = ctx->N; a.mp = ctx->mp;
        uint32_t logH = 7;
        while ((1u << logH) < 2 * ctx->m) logH++;
        a.H = 1u << logH; a.logH = logH;
        a.mode = mode; a.disable_tricks = disable_tricks;
        a.x0 = x0; a.x1 = x1; a.out = d_out;
        const size_t lds = (size_t)a.H * 6 + (size_t)a.m * 16;
        if (lds > 150 * 1024) return fbg_fail(ctx, FBG_ERR_TOO_LARGE, "scan kernel LDS budget exceeded (%zu bytes)", lds);
        if (lds > 64 * 1024)
            FBG_HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_scan_columns, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        // enough workgroups to fill 256 CUs several times over; each strides over columns
        unsigned blocks = fbg_blocks(x1 - x0, 1, 256 * 8);
        hipLaunchKernelGGL(k_scan_columns, dim3(blocks), dim3(SC_THREADS), lds, ctx->stream, a);
        FBG_HIP_TRY(ctx, hipGetLastError());
    }
    return fbg_stage_end(ctx, FBG_STAGE_SCAN, x1 > x0 ? 1 : 0);
}
