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
//
// Input is the record array of suffix_sort.hip / lcp.hip: rec[p] = {rank, LCP[rank] | hint, LCP[rank+1] |
// hint, -} per text position, the hint bit saying "my SA neighbour may be coloured in my column".
//
//   k_scan_stream      one THREAD per column, lanes along x: for row i the wave reads 64 consecutive
//                      records (1 KiB, fully coalesced -- row-major MSA order IS text order), so the whole
//                      scan is one pass over the records at HBM rate.  A column without any hint has only
//                      runs of length one, where the two minima are the record's own two LCPs: done.
//                      Columns with a hint are appended to an exception list instead of being written.
//   k_scan_exceptions  one WORKGROUP per listed column: the m ranks go into an LDS hash set, every member
//                      looks up rank-1 / rank+1, and the two running minima are propagated along the runs by
//                      pointer jumping (log2(run length) rounds).  No sorting.
#include "fbg_internal.h"
#include "text_cmp.h"

#define ST_THREADS 256
#define SC_THREADS 256
#define SC_MAX_RPT 16            // rows per thread of the LDS kernel: m <= SC_THREADS * SC_MAX_RPT = 4096 (more: k_scan_exceptions_big)
#define SC_EMPTY 0xffffffffu
#define SC_NONE 0xffffu

struct ScanArgs {
    const uint4 *rec;
    const uint4 *exc;                           // rank-order index: dense rows of the exception columns, else nullptr
    const uint32_t *prow, *igrow;               // row-major per-cell tables (gapped / ignore chars), optional
    const uint32_t *pos, *tot, *colT;           // per-row / per-text-position tables
    uint64_t m, n, N;
    int mode, disable_tricks, reversed;
    uint64_t x0, x1;
    uint64_t *out;
    uint32_t *xlist;                            // exception columns
    unsigned long long *xcount;
    uint32_t H, logH;                           // hash slots (power of two >= 2m)
};

// text pointer of cell (i, x): pos_i + rank_i(x)  (fbg.cpp:1596-1600,1687-1691)
__device__ __forceinline__ uint32_t cell_ptr(const ScanArgs &a, uint64_t i, uint64_t x)
{
    if (a.prow) return a.prow[i * a.n + x];
    return (uint32_t)(i * (a.n + 1) + (a.reversed ? a.n - 1 - x : x));
}

// fi of one active row for extension g (fbg.cpp:1656-1672)
__device__ __forceinline__ unsigned long long row_extent(const ScanArgs &a, uint64_t i, uint64_t x, uint32_t p,
                                                         unsigned long long g)
{
    unsigned long long fi;
    if (a.prow) {
        const uint32_t p0 = a.pos[i], tt = a.tot[i];
        const unsigned long long gg = (unsigned long long)(p - p0) + g;              // 1657
        if (gg > tt) fi = a.disable_tricks ? a.n : a.colT[p0 + tt - 1];              // 1659-1664
        else fi = a.colT[p0 + gg - 1];                                               // 1666
    } else {
        const unsigned long long gg = x + g;
        if (gg > a.n) fi = a.disable_tricks ? a.n : a.n - 1;
        else fi = gg - 1;
    }
    if (a.igrow) {                                                                   // 1669-1670
        const uint32_t ig = a.igrow[i * a.n + x];
        if (ig < a.n) fi = min(fi, (unsigned long long)ig);
    }
    return fi;
}

__device__ __forceinline__ bool row_active(const ScanArgs &a, uint64_t i, uint64_t x, uint32_t p)
{
    if (a.mode != FBG_SCAN_F || a.disable_tricks) return true;
    // fullrow[i] (fbg.cpp:1605-1608,1621): row i has not emitted a character yet
    return a.prow ? p != a.pos[i] : x != 0;
}

__device__ __forceinline__ void write_column(const ScanArgs &a, uint64_t x, unsigned long long best)
{
    if (a.mode == FBG_SCAN_V) {
        // v[j] = j+1-L when the block [v..j] fits in the row, else j+1
        a.out[x] = best <= x + 1 ? x + 1 - best : x + 1;
    } else {
        const unsigned long long fx = max((unsigned long long)x, best);              // fbg.cpp:1618
        a.out[x] = max((unsigned long long)a.out[x], fx);                            // fbg.cpp:1681
    }
}

__global__ __launch_bounds__(ST_THREADS) void k_scan_stream(ScanArgs a)
{
    const uint64_t x = a.x0 + (uint64_t)blockIdx.x * ST_THREADS + threadIdx.x;
    if (x >= a.x1) return;
    unsigned long long best = 0;
    uint32_t hint = 0;
    const uint64_t m = a.m;
#pragma unroll 4
    for (uint64_t i = 0; i < m; i++) {
        const uint32_t p = cell_ptr(a, i, x);
        const uint4 r = a.rec[p];
        if (!row_active(a, i, x, p)) continue;
        hint |= r.y | r.z;
        const unsigned long long g = (unsigned long long)max(r.y & FBG_LCP_MASK, r.z & FBG_LCP_MASK) + 1;   // 1656
        best = max(best, a.mode == FBG_SCAN_V ? g : row_extent(a, i, x, p, g));
    }
    if (hint & 0x80000000u) {
        const unsigned long long slot = atomicAdd(a.xcount, 1ull);
        a.xlist[slot] = (uint32_t)x;
    } else {
        write_column(a, x, best);
    }
}

__device__ __forceinline__ uint32_t sc_hash(uint32_t r, uint32_t logH) { return (r * 2654435761u) >> (32 - logH); }

__global__ __launch_bounds__(SC_THREADS) void k_scan_exceptions(ScanArgs a, uint64_t ncols)
{
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t *hkey = reinterpret_cast<uint32_t *>(smem);          // H
    uint32_t *rk = hkey + a.H;                                    // m   rank (SC_EMPTY = inactive row)
    uint32_t *vL = rk + a.m;                                      // m   running min towards the run head
    uint32_t *vR = vL + a.m;                                      // m   running min towards the run tail
    uint32_t *pp = vR + a.m;                                      // m   text pointer of the row
    uint16_t *hrow = reinterpret_cast<uint16_t *>(pp + a.m);      // H
    uint16_t *ptrL = hrow + a.H;                                  // m
    uint16_t *ptrR = ptrL + a.m;                                  // m
    __shared__ unsigned long long red[SC_THREADS / 64];

    const uint32_t tid = threadIdx.x, Hm = a.H - 1;
    const uint32_t m = (uint32_t)a.m;

    for (uint64_t c = blockIdx.x; c < ncols; c += gridDim.x) {
        const uint64_t x = a.xlist[c];
        if (x < a.x0 || x >= a.x1) continue;       // block-uniform: outside the requested column range
        for (uint32_t s = tid; s < a.H; s += SC_THREADS) hkey[s] = SC_EMPTY;
        __syncthreads();

        // ---- load the column, colour the active rows, insert their ranks -------------------
        for (uint32_t i = tid; i < m; i += SC_THREADS) {
            const uint32_t p = cell_ptr(a, i, x);
            const uint4 r4 = a.exc ? a.exc[c * a.m + i] : a.rec[p];
            uint32_t r = r4.x;
            vL[i] = r4.y & FBG_LCP_MASK;
            vR[i] = r4.z & FBG_LCP_MASK;
            pp[i] = p;
            if (row_active(a, i, x, p)) {
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
            best = max(best, a.mode == FBG_SCAN_V ? g : row_extent(a, i, x, pp[i], g));
        }
        for (int d = 32; d >= 1; d >>= 1) best = max(best, (unsigned long long)__shfl_down(best, d, 64));
        if ((tid & 63) == 0) red[tid >> 6] = best;
        __syncthreads();
        if (tid == 0) {
            for (int k = 1; k < SC_THREADS / 64; k++) best = max(best, red[k]);
            write_column(a, x, best);
        }
        __syncthreads();
    }
}

// The same for MSAs of more than SC_THREADS * SC_MAX_RPT rows: a column no longer fits one workgroup's LDS, so the hash
// set, the running minima and the jump pointers live in a scratch region of global memory per workgroup (L2-resident:
// a few hundred KB), row indices are 32 bits wide, and pointer jumping ping-pongs between two copies instead of
// keeping a row's new state in registers.  Exception columns are rare (hinted columns only): latency, not bandwidth.
#define SCB_NONE 0xffffffffu
__global__ __launch_bounds__(SC_THREADS) void k_scan_exceptions_big(ScanArgs a, uint64_t ncols, uint32_t *__restrict__ scratch,
                                                                    uint64_t scratch_words)
{
    __shared__ unsigned long long red[SC_THREADS / 64];
    const uint32_t tid = threadIdx.x, Hm = a.H - 1;
    const uint64_t m = a.m;
    uint32_t *base = scratch + (uint64_t)blockIdx.x * scratch_words;
    uint32_t *hkey = base;                                 // H
    uint32_t *hrow = hkey + a.H;                           // H
    uint32_t *rk = hrow + a.H;                             // m
    uint32_t *pp = rk + m;                                 // m
    uint32_t *vL[2] = {pp + m, pp + 2 * m}, *vR[2] = {pp + 3 * m, pp + 4 * m};
    uint32_t *pL[2] = {pp + 5 * m, pp + 6 * m}, *pR[2] = {pp + 7 * m, pp + 8 * m};
    for (uint64_t c = blockIdx.x; c < ncols; c += gridDim.x) {
        const uint64_t x = a.xlist[c];
        if (x < a.x0 || x >= a.x1) continue;
        __syncthreads();
        for (uint32_t s = tid; s < a.H; s += SC_THREADS) hkey[s] = SC_EMPTY;
        __syncthreads();
        for (uint64_t i = tid; i < m; i += SC_THREADS) {
            const uint32_t p = cell_ptr(a, i, x);
            const uint4 r4 = a.exc ? a.exc[c * a.m + i] : a.rec[p];
            uint32_t r = r4.x;
            vL[0][i] = r4.y & FBG_LCP_MASK;
            vR[0][i] = r4.z & FBG_LCP_MASK;
            pp[i] = p;
            if (row_active(a, i, x, p)) {
                uint32_t s = sc_hash(r, a.logH);
                for (;;) {
                    const uint32_t prev = atomicCAS(&hkey[s], SC_EMPTY, r);
                    if (prev == SC_EMPTY) { hrow[s] = (uint32_t)i; break; }
                    s = (s + 1) & Hm;
                }
            } else {
                r = SC_EMPTY;
            }
            rk[i] = r;
        }
        __syncthreads();
        int anylink = 0;
        for (uint64_t i = tid; i < m; i += SC_THREADS) {
            const uint32_t r = rk[i];
            uint32_t l = SCB_NONE, rr = SCB_NONE;
            if (r != SC_EMPTY) {
                if (r > 0) {
                    const uint32_t key = r - 1;
                    for (uint32_t s = sc_hash(key, a.logH);; s = (s + 1) & Hm) {
                        const uint32_t k = hkey[s];
                        if (k == key) { l = hrow[s]; break; }
                        if (k == SC_EMPTY) break;
                    }
                }
                if ((uint64_t)r + 1 < a.N) {
                    const uint32_t key = r + 1;
                    for (uint32_t s = sc_hash(key, a.logH);; s = (s + 1) & Hm) {
                        const uint32_t k = hkey[s];
                        if (k == key) { rr = hrow[s]; break; }
                        if (k == SC_EMPTY) break;
                    }
                }
            }
            pL[0][i] = l; pR[0][i] = rr;
            anylink |= l != SCB_NONE;
        }
        anylink = __syncthreads_or(anylink);
        int cur = 0;
        while (anylink) {
            int more = 0;
            for (uint64_t i = tid; i < m; i += SC_THREADS) {
                const uint32_t pa = pL[cur][i], pb = pR[cur][i];
                uint32_t nl = vL[cur][i], nr = vR[cur][i], na = pa, nb = pb;
                if (pa != SCB_NONE) { nl = min(nl, vL[cur][pa]); na = pL[cur][pa]; more |= na != SCB_NONE; }
                if (pb != SCB_NONE) { nr = min(nr, vR[cur][pb]); nb = pR[cur][pb]; more |= nb != SCB_NONE; }
                vL[cur ^ 1][i] = nl; vR[cur ^ 1][i] = nr; pL[cur ^ 1][i] = na; pR[cur ^ 1][i] = nb;
            }
            cur ^= 1;
            anylink = __syncthreads_or(more);
        }
        unsigned long long best = 0;
        for (uint64_t i = tid; i < m; i += SC_THREADS) {
            if (rk[i] == SC_EMPTY) continue;
            const unsigned long long g = (unsigned long long)max(vL[cur][i], vR[cur][i]) + 1;
            best = max(best, a.mode == FBG_SCAN_V ? g : row_extent(a, i, x, pp[i], g));
        }
        for (int d = 32; d >= 1; d >>= 1) best = max(best, (unsigned long long)__shfl_down(best, d, 64));
        if ((tid & 63) == 0) red[tid >> 6] = best;
        __syncthreads();
        if (tid == 0) {
            for (int k = 1; k < SC_THREADS / 64; k++) best = max(best, red[k]);
            write_column(a, x, best);
        }
    }
}

// k_scan_exceptions over `ncols` listed columns: in LDS while a column fits, else through global scratch
static int launch_exceptions(fbg_ctx *ctx, ScanArgs &a, uint64_t ncols)
{
    hipStream_t st = ctx->stream;
    const size_t lds = (size_t)a.H * 6 + (size_t)a.m * 20;
    if (a.m <= (uint64_t)SC_THREADS * SC_MAX_RPT && lds <= 150 * 1024) {
        if (lds > 64 * 1024)
            FBG_HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_scan_exceptions, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_scan_exceptions, dim3(fbg_blocks(ncols, 1, 256 * 8)), dim3(SC_THREADS), lds, st, a, ncols);
        return FBG_OK;
    }
    const unsigned blocks = fbg_blocks(ncols, 1, 512);
    const uint64_t words = 2ull * a.H + 10ull * a.m;
    FBG_TRY(fbg_reserve(ctx, ctx->exc_scratch, (size_t)blocks * words * 4));
    hipLaunchKernelGGL(k_scan_exceptions_big, dim3(blocks), dim3(SC_THREADS), 0, st, a, ncols, ctx->exc_scratch.as<uint32_t>(), words);
    return FBG_OK;
}

int fbg_scan_columns(fbg_ctx *ctx, uint64_t x0, uint64_t x1, int mode, int disable_tricks, uint64_t *d_out)
{
    FBG_TRY(fbg_stage_begin(ctx, FBG_STAGE_SCAN));
    int launches = 0;
    if (x1 > x0 && ctx->granked) {
        int ok = 0;
        FBG_TRY(fbg_grs_finish(ctx, x0, x1, disable_tricks, d_out, &ok));
        launches++;
        if (!ok) {
            // the scan for this setting of the tricks ran out of room: the index again, the record way
            ctx->grs_skip = true;
            int rc = fbg_suffix_sort(ctx);
            ctx->grs_skip = false;
            FBG_TRY(rc);
            FBG_TRY(fbg_neighbour_lcp(ctx));
        }
    }
    if (x1 > x0 && ctx->granked) {
    } else if (x1 > x0 && ctx->ranked) {
        // rank-order index: column maxima are ready, only the exception columns need the per-column kernel
        FBG_TRY(fbg_rank_finish(ctx, x0, x1, mode, disable_tricks, d_out));
        launches++;
        if (ctx->n_exc > 0) {
            ScanArgs a;
            a.rec = nullptr; a.exc = ctx->exc.as<uint4>();
            a.prow = nullptr; a.igrow = nullptr;
            a.pos = ctx->pos.as<uint32_t>(); a.tot = ctx->tot.as<uint32_t>(); a.colT = nullptr;
            a.m = ctx->m; a.n = ctx->n; a.N = ctx->N;
            a.mode = mode; a.disable_tricks = disable_tricks; a.reversed = ctx->reversed;
            a.x0 = x0; a.x1 = x1; a.out = d_out;
            a.xlist = ctx->xlist.as<uint32_t>(); a.xcount = nullptr;
            uint32_t logH = 7;
            while ((1u << logH) < 2 * ctx->m) logH++;
            a.H = 1u << logH; a.logH = logH;
            FBG_TRY(launch_exceptions(ctx, a, (uint64_t)ctx->n_exc));
            launches++;
        }
        FBG_HIP_TRY(ctx, hipGetLastError());
    } else if (x1 > x0) {
        hipStream_t st = ctx->stream;
        FBG_TRY(fbg_reserve(ctx, ctx->xlist, (ctx->n + 1) * 4));
        unsigned long long *xcount = ctx->scalars.as<unsigned long long>() + 24;
        FBG_HIP_TRY(ctx, hipMemsetAsync(xcount, 0, sizeof(unsigned long long), st));
        ScanArgs a;
        a.rec = ctx->rec.as<uint4>(); a.exc = nullptr;
        a.prow = ctx->gapfree ? nullptr : ctx->prow.as<uint32_t>();
        a.igrow = (mode == FBG_SCAN_F && ctx->have_ignore) ? ctx->igrow.as<uint32_t>() : nullptr;
        a.pos = ctx->pos.as<uint32_t>(); a.tot = ctx->tot.as<uint32_t>();
        a.colT = ctx->gapfree ? nullptr : ctx->colT.as<uint32_t>();
        a.m = ctx->m; a.n = ctx->n; a.N = ctx->N;
        a.mode = mode; a.disable_tricks = disable_tricks; a.reversed = ctx->reversed;
        a.x0 = x0; a.x1 = x1; a.out = d_out;
        a.xlist = ctx->xlist.as<uint32_t>(); a.xcount = xcount;
        uint32_t logH = 7;
        while ((1u << logH) < 2 * ctx->m) logH++;
        a.H = 1u << logH; a.logH = logH;
        hipLaunchKernelGGL(k_scan_stream, dim3(fbg_blocks(x1 - x0, ST_THREADS)), dim3(ST_THREADS), 0, st, a);
        launches++;
        unsigned long long nx = 0;
        FBG_HIP_TRY(ctx, hipMemcpyAsync(&nx, xcount, sizeof(nx), hipMemcpyDeviceToHost, st));
        FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
        if (nx > 0) {
            FBG_TRY(launch_exceptions(ctx, a, (uint64_t)nx));
            launches++;
        }
        FBG_HIP_TRY(ctx, hipGetLastError());
    }
    return fbg_stage_end(ctx, FBG_STAGE_SCAN, launches);
}
