// dp.hip -- the two segmentation recurrences and their backtracks, on the device.
//
//  * fbg_dp_minmax:     sort of (x, f[x]+1) by second + min-max-length sweep + backtrack of
//                       segment_elastic_minmaxlength (fbg.cpp:1940-2039);
//  * fbg_dp_repeatfree: s[]/prev[] recurrence + backtrack of segment() (fbg.cpp:616-664).
//
// The std::sort (fbg.cpp:1953) is a counting sort here: a histogram of f[x]+1, an exclusive scan
// and a scatter.  Ties may land in any order; the sweep's outcome does not depend on it, because
// every update inside one step j is a strict min/max over distinct x or a counter.
// The sweep itself is inherently sequential (step j reads minmaxlength[] of earlier columns); one
// lane walks it with all state in device memory, exactly the statements of fbg.cpp:1968-2014.
#include "fbg_internal.h"
#include <algorithm>
#include <rocprim/rocprim.hpp>

#define DP_NONE 0xffffffffu

// e[x] = f[x] + 1, the count of entries outside [x, n] and the longest minimal extension (one atomic each per workgroup)
__global__ __launch_bounds__(256) void k_dp_keys(const uint64_t *__restrict__ f, uint64_t n, uint32_t *__restrict__ e,
                                                 unsigned long long *__restrict__ bad)
{
    __shared__ unsigned long long s_ext[4], s_bad[4];
    unsigned long long ext = 0, nbad = 0;
    for (uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; x < n; x += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t fx = f[x];
        if (fx > n || fx < x) { nbad++; fx = fx > n ? n : x; }            // f[x] in [x, n] by construction
        e[x] = (uint32_t)(fx + 1);
        // longest minimal extension f[x]+1-x.  A column with f[x] = n (a row runs out of symbols; only without the elastic
        // tricks, fbg.cpp:1659-1663) can start no block at all -- its entry would be read at step n+1 -- and does not count
        if (fx < n) ext = max(ext, (unsigned long long)(fx + 1 - x));
    }
    for (int d = 32; d >= 1; d >>= 1) {
        ext = max(ext, (unsigned long long)__shfl_down(ext, d, 64));
        nbad += (unsigned long long)__shfl_down(nbad, d, 64);
    }
    if ((threadIdx.x & 63) == 0) { s_ext[threadIdx.x >> 6] = ext; s_bad[threadIdx.x >> 6] = nbad; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long mx = max(max(s_ext[0], s_ext[1]), max(s_ext[2], s_ext[3])), nb = s_bad[0] + s_bad[1] + s_bad[2] + s_bad[3];
        if (nb) atomicAdd(bad, nb);
        if (mx) atomicMax(bad + 1, mx);
    }
}

// hist[e]++ : the counting sort of fbg.cpp:1941-1953 (only the wave-parallel and the literal sweep read its order)
__global__ void k_dp_hist(const uint32_t *__restrict__ e, uint64_t n, uint32_t *__restrict__ hist)
{
    const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x < n) atomicAdd(&hist[e[x]], 1u);
}

__global__ void k_dp_scatter(const uint32_t *__restrict__ e, uint64_t n, uint32_t *__restrict__ cursor,
                             uint32_t *__restrict__ items)
{
    uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n) return;
    uint32_t slot = atomicAdd(&cursor[e[x]], 1u);
    items[slot] = (uint32_t)x;
}

// fbg.cpp:1960-2014, one lane.  bstart[j]..bstart[j+1] delimit the entries with f[x]+1 == j.
__global__ void k_dp_minmax(const uint32_t *__restrict__ bstart, const uint32_t *__restrict__ items, uint32_t n,
                            uint32_t *__restrict__ count, uint32_t *__restrict__ bcount,
                            uint32_t *__restrict__ thead, uint32_t *__restrict__ tnext,
                            uint32_t *__restrict__ mml, uint32_t *__restrict__ bt)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    // count[], bcount[], mml[0], bt[0] arrive zeroed; thead[] arrives filled with DP_NONE
    uint32_t I = 0, S = n + 1, bS = DP_NONE;                                    // 1967
    for (uint32_t j = 1; j <= n; j++) {
        const uint32_t b0 = bstart[j], b1 = bstart[j + 1];
        for (uint32_t y = b0; y < b1; y++) {                                     // 1969
            const uint32_t xy = items[y];
            const uint32_t rec = mml[xy];
            if (rec > n) {
                // no recursive solution (1973)
            } else if (j <= xy + rec) {                                          // 1975
                count[rec] += 1;
                I = min(I, rec);
                const uint32_t cx = bcount[rec];
                if (xy + rec > cx + mml[cx]) bcount[rec] = xy;                   // 1979
                if (xy + rec + 1 <= n) {                                         // 1982
                    const uint32_t slot = xy + rec + 1;
                    tnext[xy] = thead[slot];
                    thead[slot] = xy;
                }
            } else {
                if (j - xy < S) { bS = xy; S = j - xy; }                         // 1986-1989
            }
        }
        for (uint32_t x = thead[j]; x != DP_NONE; x = tnext[x]) {                // 1993
            const uint32_t v = mml[x];
            const uint32_t c = count[v] - 1;
            count[v] = c;
            if (j - x < S) { S = j - x; bS = x; }
            if (c == 0) bcount[v] = 0;
        }
        const uint32_t cI = count[I];
        if (cI > 0 && I < S) { mml[j] = I; bt[j] = bcount[I]; }                  // 2004
        else { mml[j] = S; bt[j] = bS; }
        S += 1;
        if (cI == 0) I += 1;                                                     // 2012
    }
}


// ---- wave-parallel sweep ---------------------------------------------------------------------------
//
// Equivalent statement of fbg.cpp:1968-2014 used when f[0] == 0 (always true with the elastic
// "tricks", fbg.cpp:1605-1608: no row is active at column 0).  Then S == 1 after step 1 and the lazy
// index I of the reference can never hide a live count (whenever count[I] == 0, S <= I+1 holds, so the
// S branch it takes is the true minimum), hence
//     minmaxlength[j] = min over x with f[x]+1 <= j of max(minmaxlength[x], j - x),
// ties resolved towards the "S" candidate exactly as the strict `I < S` test does (fbg.cpp:2004).
// Lane L of the wave keeps cand[L] = max{ x : f[x]+1 <= j, minmaxlength[x] <= L }; then
//     minmaxlength[j] = min{ L : cand[L] + L >= j }                       (one ballot + find-first)
//     backtrack[j]    = j - L*      if cand[L*-1] == j - L*   (the reference's S / backtrack_S branch)
//                     = cand[L*]    otherwise                 (its count_solutions / backtrack_count branch)
// R registers per lane cover block lengths up to 64*R-1; minmaxlength never exceeds twice the longest
// minimal extension + 1, which is how R is chosen.  A step that finds no feasible lane raises `flag`
// and the host reruns the literal kernel (never observed; it guards the bound).
#define DPW 1024   // steps / items staged in LDS per refill (multiple of 64)

template <int R>
__global__ __launch_bounds__(64) void k_dp_wave(const uint32_t *__restrict__ bstart, const uint32_t *__restrict__ items,
                                                uint32_t n, uint32_t *__restrict__ mml, uint32_t *__restrict__ bt,
                                                unsigned long long *__restrict__ flag)
{
    constexpr int RING = 64 * R;              // covers x in (j - 64R, j): extensions are < 32R
    __shared__ uint32_t ring[RING];
    __shared__ uint32_t s_bs[DPW];            // bstart[jb+1 .. jb+DPW]: bucket end of each staged step
    __shared__ uint32_t s_it[DPW];            // items[itbase .. itbase+DPW)
    const int lane = threadIdx.x;
    int cand[R];
#pragma unroll
    for (int r = 0; r < R; r++) cand[r] = 0;  // f[0] == 0: column 0 is a candidate for every length from step 1 on
    if (R > 1) { for (int k = lane; k < RING; k += 64) ring[k] = 0; }
    int minLs = 1 << 30;

    // items reach the steps through two windows: DPW entries in LDS, 64 of them in one VGPR
    uint32_t y = 0, itbase = 0, itw = 0;      // itw: items index of lane 0 of the register window
    for (int k = lane; k < DPW; k += 64) s_it[k] = (uint32_t)k < n ? items[k] : 0;
    __syncthreads();
    uint32_t itv = s_it[lane];
    // steps are walked in 64-aligned groups so that lane (j & 63) of one register holds minmaxlength[j]
    int ringv = 0, outb = 0;                  // lane 0 of the first group is column 0: both 0
    for (uint32_t jb = 0; jb <= n; jb += DPW) {
        for (uint32_t k = lane; k < DPW; k += 64) s_bs[k] = jb + k + 1 <= n + 1 ? bstart[jb + k + 1] : 0;
        __syncthreads();
        for (uint32_t g = 0; g < DPW && jb + g <= n; g += 64) {
            const uint32_t bsv = s_bs[g + lane];
            const uint32_t k0 = (jb + g == 0) ? 1u : 0u;
            const uint32_t k1 = min(64u, n - (jb + g) + 1);
            for (uint32_t k = k0; k < k1; k++) {
                const uint32_t j = jb + g + k;
                const uint32_t b1 = (uint32_t)__builtin_amdgcn_readlane((int)bsv, (int)k);
                // ---- entries with f[x]+1 == j (fbg.cpp:1969) ----
                while (y < b1) {
                    if (y - itw >= 64) {          // next register window
                        itw += 64;
                        if (itw - itbase >= DPW) {   // next LDS window
                            itbase += DPW;
                            __syncthreads();
                            for (int q = lane; q < DPW; q += 64) s_it[q] = itbase + q < n ? items[itbase + q] : 0;
                            __syncthreads();
                        }
                        itv = s_it[itw - itbase + lane];
                    }
                    const int x = __builtin_amdgcn_readlane((int)itv, (int)(y - itw));
                    y++;
                    int v;
                    if (R == 1) v = __builtin_amdgcn_readlane(ringv, x & 63);
                    else v = __builtin_amdgcn_readfirstlane((int)ring[x & (RING - 1)]);
#pragma unroll
                    for (int r = 0; r < R; r++) cand[r] = (64 * r + lane >= v) ? max(cand[r], x) : cand[r];
                }
                // ---- smallest feasible block length: cand[L] + L is increasing in L ----
                int Ls = -1;
#pragma unroll
                for (int r = R - 1; r >= 0; r--) {
                    const unsigned long long bal = __builtin_amdgcn_ballot_w64((uint32_t)(cand[r] + 64 * r + lane) >= j);
                    if (bal) Ls = 64 * r + (int)__builtin_ctzll(bal);
                }
                minLs = min(minLs, Ls);
                int cprev, ccur;
                if (R == 1) {
                    cprev = __builtin_amdgcn_readlane(cand[0], (Ls - 1) & 63);
                    ccur = __builtin_amdgcn_readlane(cand[0], Ls & 63);
                } else {
                    cprev = ccur = 0;
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        const int a = __builtin_amdgcn_readlane(cand[r], (Ls - 1) & 63);
                        const int b = __builtin_amdgcn_readlane(cand[r], Ls & 63);
                        if (((Ls - 1) >> 6) == r) cprev = a;
                        if ((Ls >> 6) == r) ccur = b;
                    }
                }
                const int jl = (int)j - Ls;
                const int back = cprev == jl ? jl : ccur;
                const bool mine = lane == (int)k;
                ringv = mine ? Ls : ringv;
                outb = mine ? back : outb;
                if (R > 1 && lane == 0) ring[j & (RING - 1)] = (uint32_t)Ls;
            }
            if ((uint32_t)lane < k1) { mml[jb + g + lane] = (uint32_t)ringv; bt[jb + g + lane] = (uint32_t)outb; }
        }
        __syncthreads();
    }
    if (minLs < 1 && lane == 0) flag[4] = 1;
}

// ---- tiled sweep: 8 steps per iteration (block lengths < 64) ---------------------------------------
//
// Same recurrence as k_dp_wave, arranged so that one wave instruction works on 8 steps x 8 candidates:
//     minmaxlength[j] = min over candidates x < j with f[x]+1 <= j of max(minmaxlength[x], j - x).
// A chunk of c <= 8 consecutive steps j..j+c-1 can be evaluated together when no column inside the chunk
// is itself a usable candidate inside it, i.e. f[x]+1 >= j+c for every x >= j (clen[], precomputed).
// Lane (t, c8) scores candidates of age 8*c8+1 .. 8*c8+8 (relative to j) for step j+t; the eight partial
// minima of a step are combined with three DPP row operations.  The reference's tie-breaking is folded
// into the key: cost*128 + 0 for a candidate of its "S" kind (age > minmaxlength[x], fbg.cpp:1985-1989,
// 1996-1999), cost*128 + 1 + age for its count_solutions kind (youngest first = backtrack_count's
// largest x, fbg.cpp:1976-1981); S wins ties like the strict `I < S` (fbg.cpp:2004).
// The last 64 columns' (extension, minmaxlength) pairs live in an LDS ring; ages beyond 64 cannot win
// because minmaxlength <= 63 here (checked: a step without a feasible candidate raises the flag).
#define DPT_BIG 0xffffu

__global__ void k_dp_prep(const uint32_t *__restrict__ e, uint32_t n, uint32_t ext_cap, uint8_t *__restrict__ ext7,
                          uint8_t *__restrict__ clen)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > n) return;
    ext7[j] = j < n ? (uint8_t)min(e[j] - j, ext_cap) : (uint8_t)ext_cap;
    uint32_t c = 8;
    for (uint32_t s = 0; s < 8 && j + s < n; s++) c = min(c, e[j + s] - j);   // e[j+s] - j >= s + 1 >= 1
    clen[j] = (uint8_t)c;
}

__device__ __forceinline__ uint32_t dpp_min8(uint32_t v)
{
    // minimum over each group of 8 consecutive lanes (half a DPP row)
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false));  // row_half_mirror
    return v;
}

__global__ __launch_bounds__(64) void k_dp_tile(const uint8_t *__restrict__ ext7, const uint8_t *__restrict__ clen,
                                                uint32_t n, uint32_t *__restrict__ mml, uint32_t *__restrict__ bt,
                                                unsigned long long *__restrict__ flag)
{
    __shared__ uint16_t ring[64];            // (extension << 7) | minmaxlength of column (x & 63)
    __shared__ uint8_t s_ext[DPW + 16], s_len[DPW + 16];
    const uint32_t lane = threadIdx.x, t = lane >> 3, c8 = lane & 7;
    ring[lane] = (uint16_t)(127u << 7);      // columns < 0 do not exist: never valid
    if (lane == 0) { mml[0] = 0; bt[0] = 0; }
    uint32_t wbase = 0xffffffffu, j = 1;
    uint32_t bad = 0;
    while (j <= n) {
        if (wbase == 0xffffffffu || j + 8 > wbase + DPW) {           // stage the next window of inputs
            __syncthreads();
            wbase = j;
            for (uint32_t k = lane; k < DPW + 8; k += 64) {
                const uint32_t x = wbase + k;
                s_ext[k] = x <= n ? ext7[x] : (uint8_t)127;
                s_len[k] = x <= n ? clen[x] : (uint8_t)1;
            }
            if (j == 1 && lane == 0) ring[0] = (uint16_t)(((uint32_t)ext7[0] << 7) | 0u);   // column 0, minmaxlength 0
            __syncthreads();
        }
        uint32_t cl = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_len[j - wbase]);
        cl = min(cl, n - j + 1);
        uint32_t best = DPT_BIG;
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) {
            const uint32_t a0 = 1 + 8 * c8 + u;                      // age of the candidate at step j
            const uint32_t pk = ring[(j - a0) & 63];
            const uint32_t v = pk & 127u, ext = pk >> 7;
            const uint32_t age = a0 + t;                             // its age at this lane's step j + t
            const uint32_t key = (max(v, age) << 7) | (age > v ? 0u : age + 1);
            best = min(best, ext <= age ? key : DPT_BIG);
        }
        best = dpp_min8(best);
        const uint32_t jt = j + t;
        const uint32_t cost = best >> 7, tb = best & 127u;
        const uint32_t back = tb == 0 ? jt - cost : jt - (tb - 1);
        const bool live = t < cl;
        bad |= (live && best == DPT_BIG) ? 1u : 0u;
        if (live && c8 == 0) {
            mml[jt] = cost;
            bt[jt] = back;
            if (jt < n) ring[jt & 63] = (uint16_t)(((uint32_t)s_ext[jt - wbase] << 7) | (cost & 127u));
        }
        j += cl;
    }
    if (__ballot(bad != 0) && lane == 0) flag[4] = 1;
}

// ---- block-matrix sweep: the recurrence as a (min,max) matrix chain ----------------------------------
//
// minmaxlength[j] = min_x max(minmaxlength[x], j-x) is a product in the (min,max) semiring, so a block of 64
// steps acts on the 64 values before it through a 64x64 matrix W_b:
//     minmaxlength[64b+1+t] = min_k max( minmaxlength[64b-63+k], W_b[k][t] ),
// W_b[k][t] = the smallest achievable longest block over all ways of cutting [64b-63+k, 64b+1+t) into valid
// blocks (first block starting at the old column, the rest inside the new 64 columns).  Only candidates of
// age <= 64 are considered, which is exact while minmaxlength <= 63 (flagged otherwise).
//   k_dp_blockW  builds every W_b independently: one wave per block, lane = old column k, 64 sequential
//                steps of the same recurrence started from a one-hot state.  Embarrassingly parallel.
//   k_dp_chain   the only sequential part left: one (min,max) matrix-vector product per block.
//   k_dp_bt      backtrack[j] from minmaxlength[] alone, one thread per column, with the reference's
//                tie-breaking (S kind first, else the youngest count_solutions kind; fbg.cpp:1976-2009).
#define DPB_INF 255u
typedef unsigned short dp_u16x2 __attribute__((ext_vector_type(2)));      // v_pk_max_u16 / v_pk_min_u16 operands
__device__ __forceinline__ dp_u16x2 dp_bits(uint32_t x) { return __builtin_bit_cast(dp_u16x2, x); }
__device__ __forceinline__ uint32_t dp_word(dp_u16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ dp_u16x2 dp_pair(uint32_t x) { return dp_bits(x | (x << 16)); }

// R = window / 64: candidates up to 64R columns back, blocks of 64R steps, matrices of (64R)^2 bytes.
// Step t of source k: w = min over the inside candidates tp < t of max(W[tp][k], t - tp), a candidate counting from the
// step on at which its block is long enough (t >= tp + ext[tp]) -- 64R steps x 64R candidates x 64R sources per block.
// The thread's row of W lies along tp in LDS (rows padded by 4 bytes: the lanes' rows start in different banks), so one
// load brings 4 candidates; their bytes as two pairs of 16-bit lanes go through packed max (with the ages, which are
// the same for all sources) and packed min; which candidates count is a byte mask per wave, kept up to date by the
// lanes themselves (lane j clears the bytes of its candidates at the step they become usable): 0xff on a byte makes
// the maximum at least 255 = no solution.  11 vector operations per 4 candidates instead of 24.
#define DPB_PADDED(WN) ((WN) + 4)
template <int R>
__global__ __launch_bounds__(64 * R) void k_dp_blockW(const uint8_t *__restrict__ ext7, const uint8_t *__restrict__ amin,
                                                      uint32_t n, uint32_t nblocks, uint8_t *__restrict__ Wt)
{
    // amin (optional): per-step minimal block length -- the non-elastic recurrence, where a block ending at
    // step j is valid iff it is at least j - v[j-1] long (fbg.cpp:625-627); 255 = no valid block ends there
    // one thread per source column k; the R waves of a block never need each other's values
    constexpr uint32_t WN = 64 * R, WP = DPB_PADDED(WN);
    extern __shared__ uint8_t dyn_lds[];
    uint8_t *wl = dyn_lds;                   // wl[k * WP + t]: W of inside column (block start + 1 + t) for source k
    uint8_t *s_ext = dyn_lds + WN * WP;      // extensions of the inside columns
    uint8_t *s_amin = s_ext + WN;            // minimal block length per step
    uint8_t *mk = s_amin + WN + (threadIdx.x >> 6) * 2 * WN;   // this wave's masks: per group of 4 candidates two words (even / odd candidates as 16-bit lanes)
    const uint32_t k = threadIdx.x, lane = threadIdx.x & 63;
    for (uint32_t b = blockIdx.x; b < nblocks; b += gridDim.x) {
        const uint32_t jb = WN * b;
        const int64_t xs = (int64_t)jb - (WN - 1) + k;                    // old column of source k
        const uint32_t ext_src = xs >= 0 ? ext7[xs] : 255u;
        __syncthreads();
        s_ext[k] = jb + 1 + k < n ? ext7[jb + 1 + k] : (uint8_t)255;
        s_amin[k] = amin ? (jb + 1 + k <= n ? amin[jb + 1 + k] : (uint8_t)255) : (uint8_t)1;
        __syncthreads();
        uint32_t minext = 255;
        for (uint32_t q = 0; q < WN; q++) minext = min(minext, (uint32_t)s_ext[q]);
        // all candidates masked out; candidate tp = 4g + j sits in byte 8g + 4(j & 1) + 2(j >> 1)
        for (uint32_t q = lane; q < WN / 4; q += 64) reinterpret_cast<uint2 *>(mk)[q] = make_uint2(0x00ff00ffu, 0x00ff00ffu);
        uint32_t ready[R];                                                // the step from which the lane's candidates count
#pragma unroll
        for (int r = 0; r < R; r++) { const uint32_t tp = lane + 64 * r; ready[r] = tp + max(1u, (uint32_t)s_ext[tp]); }
        uint8_t *row = wl + k * WP;
        uint8_t *out = Wt + (size_t)b * WN * WN;
        for (uint32_t t = 0; t < WN; t++) {
#pragma unroll
            for (int r = 0; r < R; r++)
                if (ready[r] == t) { const uint32_t tp = lane + 64 * r; mk[8 * (tp >> 2) + 4 * (tp & 1) + 2 * ((tp >> 1) & 1)] = 0; }
            __builtin_amdgcn_wave_barrier();
            const uint32_t age_src = t + WN - k;                          // (jb+1+t) - xs
            const uint32_t am = s_amin[t];
            uint32_t w = (ext_src <= age_src && age_src <= WN && age_src >= am) ? age_src : DPB_INF;
            // inside candidates need age >= their extension >= minext and age >= the step's minimum
            const uint32_t need = max(minext, am);
            const uint32_t tp_end = t >= need ? t - need + 1 : 0;
            dp_u16x2 acc_e = dp_pair(0xffffu), acc_o = acc_e;
            uint32_t g4 = 0;
            // sixteen candidates a turn without a bound check in between (round 4: the compiler's own unrolling kept a compare
            // and a branch after every four), then the rest four at a time
            for (; g4 + 16 <= tp_end; g4 += 16) {
#pragma unroll
                for (uint32_t h = 0; h < 16; h += 4) {
                    const uint32_t word = *reinterpret_cast<const uint32_t *>(row + g4 + h);
                    const uint2 m = *reinterpret_cast<const uint2 *>(mk + 2 * (g4 + h));
                    const uint32_t base = t - g4 - h;
                    const dp_u16x2 age_e = dp_bits(base | ((base - 2) << 16)), age_o = dp_bits((base - 1) | ((base - 3) << 16));
                    acc_e = __builtin_elementwise_min(acc_e, __builtin_elementwise_max(dp_bits((word & 0x00ff00ffu) | m.x), age_e));
                    acc_o = __builtin_elementwise_min(acc_o, __builtin_elementwise_max(dp_bits(((word >> 8) & 0x00ff00ffu) | m.y), age_o));
                }
            }
            for (; g4 + 4 <= tp_end; g4 += 4) {
                const uint32_t word = *reinterpret_cast<const uint32_t *>(row + g4);
                const uint2 m = *reinterpret_cast<const uint2 *>(mk + 2 * g4);
                const uint32_t base = t - g4;
                const dp_u16x2 age_e = dp_bits(base | ((base - 2) << 16)), age_o = dp_bits((base - 1) | ((base - 3) << 16));
                acc_e = __builtin_elementwise_min(acc_e, __builtin_elementwise_max(dp_bits((word & 0x00ff00ffu) | m.x), age_e));
                acc_o = __builtin_elementwise_min(acc_o, __builtin_elementwise_max(dp_bits(((word >> 8) & 0x00ff00ffu) | m.y), age_o));
            }
            if (g4 < tp_end) {                                            // the last, partial group: candidates from tp_end on do not count
                const uint32_t word = *reinterpret_cast<const uint32_t *>(row + g4);
                uint2 m = *reinterpret_cast<const uint2 *>(mk + 2 * g4);
                const uint32_t live = tp_end - g4;                        // 1 .. 3
                if (live < 2) m.y |= 0x000000ffu;
                if (live < 3) m.x |= 0x00ff0000u;
                m.y |= 0x00ff0000u;
                const uint32_t base = t - g4;
                const dp_u16x2 age_e = dp_bits((base & 0xffffu) | ((base - 2) << 16)), age_o = dp_bits(((base - 1) & 0xffffu) | ((base - 3) << 16));
                acc_e = __builtin_elementwise_min(acc_e, __builtin_elementwise_max(dp_bits((word & 0x00ff00ffu) | m.x), age_e));
                acc_o = __builtin_elementwise_min(acc_o, __builtin_elementwise_max(dp_bits(((word >> 8) & 0x00ff00ffu) | m.y), age_o));
            }
            const dp_u16x2 acc = __builtin_elementwise_min(acc_e, acc_o);
            w = min(w, min((uint32_t)acc.x, (uint32_t)acc.y));
            w = min(w, DPB_INF);
            row[t] = (uint8_t)w;
            out[(size_t)t * WN + k] = (uint8_t)w;                         // Wt[b][t][k]
        }
    }
}

template <int R>
__global__ __launch_bounds__(64) void k_dp_chain(const uint8_t *__restrict__ Wt, uint32_t n, uint32_t nblocks,
                                                 uint32_t F, uint8_t *__restrict__ Sg,
                                                 uint32_t *__restrict__ mml, unsigned long long *__restrict__ flag)
{
    // steps from group to group: block index of step g is the last block of group g (its matrix holds the
    // product of the whole group after k_dp_compose); Sg[g] receives the state entering group g
    constexpr uint32_t WN = 64 * R;
    constexpr bool PREFETCH = R <= 2;        // next block's rows held in registers (16 R^2 VGPRs)
    constexpr int NQ = R * R * 4;             // uint4 loads per lane per block
    const uint32_t lane = threadIdx.x;
    uint32_t S[R];                                                        // S[r]: state of source k = 64r + lane
#pragma unroll
    for (int r = 0; r < R; r++) S[r] = (r == R - 1 && lane == 63) ? 0u : DPB_INF;   // before block 0: only column 0
    if (lane == 0) mml[0] = 0;
    uint32_t bad = 0;
    uint4 nx[PREFETCH ? NQ : 1];
    const uint32_t ngroups = (nblocks + F - 1) / F;
    auto row_ptr = [&](uint32_t g, int rt) {
        const uint32_t b = min(g * F + F - 1, nblocks - 1);
        return reinterpret_cast<const uint4 *>(Wt + (size_t)b * WN * WN + (size_t)(64 * rt + lane) * WN);
    };
    if (PREFETCH) {
#pragma unroll
        for (int rt = 0; rt < R; rt++)
#pragma unroll
            for (int i = 0; i < 4 * R; i++) nx[rt * 4 * R + i] = row_ptr(0, rt)[i];
    }
    for (uint32_t b = 0; b < ngroups; b++) {
#pragma unroll
        for (int r = 0; r < R; r++) Sg[(size_t)b * WN + 64 * r + lane] = (uint8_t)min(S[r], DPB_INF);
        uint4 cur[NQ];
        if (PREFETCH) {
#pragma unroll
            for (int i = 0; i < NQ; i++) cur[i] = nx[i];
            if (b + 1 < ngroups) {
#pragma unroll
                for (int rt = 0; rt < R; rt++)
#pragma unroll
                    for (int i = 0; i < 4 * R; i++) nx[rt * 4 * R + i] = row_ptr(b + 1, rt)[i];
            }
        } else {
#pragma unroll
            for (int rt = 0; rt < R; rt++)
#pragma unroll
                for (int i = 0; i < 4 * R; i++) cur[rt * 4 * R + i] = row_ptr(b, rt)[i];
        }
        uint32_t best[R];
#pragma unroll
        for (int rt = 0; rt < R; rt++) {                                   // target t = 64*rt + lane
            uint32_t acc = DPB_INF;
#pragma unroll
            for (int rs = 0; rs < R; rs++) {                               // sources 64*rs .. 64*rs+63
#pragma unroll
                for (int k = 0; k < 64; k++) {
                    const uint4 q = cur[rt * 4 * R + rs * 4 + (k >> 4)];
                    const uint32_t word = ((k >> 2) & 3) == 0 ? q.x : ((k >> 2) & 3) == 1 ? q.y : ((k >> 2) & 3) == 2 ? q.z : q.w;
                    const uint32_t sk = (uint32_t)__builtin_amdgcn_readlane((int)S[rs], k);
                    acc = min(acc, max(sk, (word >> (8 * (k & 3))) & 255u));
                }
            }
            best[rt] = acc;
        }
#pragma unroll
        for (int rt = 0; rt < R; rt++) S[rt] = best[rt];   // minmaxlength values are written by k_dp_expand
    }
    if (__ballot(bad != 0) && lane == 0) flag[4] = 1;
}

// A barrier between phases that exchange through LDS only: __syncthreads() also waits for every global load and store in flight
// (its fence covers global memory), which would put the latency of the rows fetched ahead back on every step of a chain
__device__ __forceinline__ void dpw_lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// The same walk on six waves (round 4).  One wave alone spends 4 instructions per (source, target) pair -- 1.2 us per group of
// 128 x 128, all of it instruction issue.  Waves 0..3 take a quarter of the sources each (their part of every target's row,
// fetched a group ahead), leave their partial minima in LDS and, behind ONE barrier per group (the table alternates between
// two copies), every wave folds the four parts into the next state.  Wave 4 keeps the state too and is the only one that stores
// (Sg): a store in flight would make every use of a fetched row wait for all of the wave's loads (one counter for both).
// Wave 5 touches the lines of the matrix PF groups ahead (see k_dpw_chain2).
template <int R>
__global__ __launch_bounds__(384) void k_dp_chain6(const uint8_t *__restrict__ Wt, uint32_t n, uint32_t nblocks, uint32_t F, uint8_t *__restrict__ Sg,
                                                   uint32_t *__restrict__ mml)
{
    constexpr uint32_t WN = 64 * R, PF = 6;
    __shared__ uint32_t part[2][4][WN];
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t ngroups = (nblocks + F - 1) / F;
    auto last_block = [&](uint32_t g) { return min(g * F + F - 1, nblocks - 1); };
    if (wv == 5) {
        constexpr uint32_t NL = R == 1 ? 1 : R * R / 2;          // loads of 64 lines each that cover WN * WN bytes
        for (uint32_t g = 0; g < ngroups; g++) {
            const char *p = reinterpret_cast<const char *>(Wt) + (size_t)last_block(min(g + PF, ngroups - 1)) * WN * WN + (size_t)(R == 1 ? lane & 31 : lane) * 128;
            for (uint32_t i = 0; i < NL; i++) asm volatile("global_load_dword v127, %0, off" : : "v"(p + (size_t)i * 8192) : "v127", "memory");
            dpw_lds_barrier();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    uint32_t S[R];                                                        // S[r]: state of source k = 64r + lane
#pragma unroll
    for (int r = 0; r < R; r++) S[r] = (r == R - 1 && lane == 63) ? 0u : DPB_INF;   // before block 0: only column 0
    if (wv == 4) {
        if (lane == 0) mml[0] = 0;
        for (uint32_t g = 0; g < ngroups; g++) {
#pragma unroll
            for (int r = 0; r < R; r++) Sg[(size_t)g * WN + 64 * r + lane] = (uint8_t)min(S[r], DPB_INF);
            dpw_lds_barrier();
#pragma unroll
            for (int r = 0; r < R; r++)
                S[r] = min(min(part[g & 1][0][64 * r + lane], part[g & 1][1][64 * r + lane]), min(part[g & 1][2][64 * r + lane], part[g & 1][3][64 * r + lane]));
        }
        return;
    }
    // waves 0..3: sources 16 R wv .. 16 R wv + 16 R - 1, i.e. the uint4 words R wv .. R wv + R - 1 of every target's row
    auto row_ptr = [&](uint32_t g, int rt) {
        return reinterpret_cast<const uint4 *>(Wt + (size_t)last_block(g) * WN * WN + (size_t)(64 * rt + lane) * WN) + R * wv;
    };
    uint4 nx[R * R];
#pragma unroll
    for (int rt = 0; rt < R; rt++)
#pragma unroll
        for (int i = 0; i < R; i++) nx[rt * R + i] = row_ptr(0, rt)[i];
    const uint32_t k0 = 16 * R * wv;                                      // first source of the wave: lane (k0 & 63) of S[k0 >> 6]
    for (uint32_t g = 0; g < ngroups; g++) {
        uint4 cur[R * R];
#pragma unroll
        for (int i = 0; i < R * R; i++) cur[i] = nx[i];
        {
            const uint32_t gn = min(g + 1, ngroups - 1);
#pragma unroll
            for (int rt = 0; rt < R; rt++)
#pragma unroll
                for (int i = 0; i < R; i++) nx[rt * R + i] = row_ptr(gn, rt)[i];
        }
        uint32_t Ssel = S[0];                                             // the state register that holds the wave's sources
#pragma unroll
        for (int r = 1; r < R; r++) Ssel = (k0 >> 6) == (uint32_t)r ? S[r] : Ssel;
        const uint32_t l0 = k0 & 63;
#pragma unroll
        for (int rt = 0; rt < R; rt++) {
            uint32_t acc = DPB_INF;
#pragma unroll
            for (int j = 0; j < 16 * R; j++) {
                const uint4 q = cur[rt * R + (j >> 4)];
                const uint32_t word = ((j >> 2) & 3) == 0 ? q.x : ((j >> 2) & 3) == 1 ? q.y : ((j >> 2) & 3) == 2 ? q.z : q.w;
                const uint32_t sk = (uint32_t)__builtin_amdgcn_readlane((int)Ssel, (int)(l0 + j));
                acc = min(acc, max(sk, (word >> (8 * (j & 3))) & 255u));
            }
            part[g & 1][wv][64 * rt + lane] = acc;
        }
        dpw_lds_barrier();
#pragma unroll
        for (int r = 0; r < R; r++)
            S[r] = min(min(part[g & 1][0][64 * r + lane], part[g & 1][1][64 * r + lane]), min(part[g & 1][2][64 * r + lane], part[g & 1][3][64 * r + lane]));
    }
}

// Shorten the sequential chain: inside every group of F consecutive blocks the matrices are replaced by
// their running (min,max) products P_i = W_first (x) ... (x) W_i (in place, all groups in parallel), so the
// chain only has to step from group to group (through the last product) and every block's values follow
// from its group's entry state with one independent matrix-vector product (k_dp_expand).
//     (P (x) W)[t][k] = min_u max( P[u][k], W[t][u] )      t: target of W, k: source of P
template <int R>
__global__ __launch_bounds__(256) void k_dp_compose(uint8_t *__restrict__ Wt, uint32_t nblocks, uint32_t F)
{
    constexpr uint32_t WN = 64 * R;
    extern __shared__ uint8_t dyn_lds[];
    uint8_t *P = dyn_lds, *W = dyn_lds + WN * WN;       // [row][col] = [target][source]
    const uint32_t tid = threadIdx.x;
    const uint32_t b0 = blockIdx.x * F;
    if (b0 >= nblocks) return;
    const uint32_t b1 = min(b0 + F, nblocks);
    for (uint32_t q = tid * 16; q < WN * WN; q += 256 * 16)
        *reinterpret_cast<uint4 *>(P + q) = *reinterpret_cast<const uint4 *>(Wt + (size_t)b0 * WN * WN + q);
    for (uint32_t b = b0 + 1; b < b1; b++) {
        uint8_t *G = Wt + (size_t)b * WN * WN;
        __syncthreads();
        for (uint32_t q = tid * 16; q < WN * WN; q += 256 * 16)
            *reinterpret_cast<uint4 *>(W + q) = *reinterpret_cast<const uint4 *>(G + q);
        __syncthreads();
        // thread -> a tile of 4 rows t x 8 consecutive sources k, four u per step: the four W words (4 u each) and the four P
        // rows' 8 bytes are read once for 128 (t, k, u) triples; bytes as two pairs of 16-bit lanes per word (even / odd bytes:
        // packed 16-bit max / min do two sources per instruction), a W byte in both lanes by one v_perm.  1.4 instructions per
        // triple (round 4; a row t x 8 sources per thread, one u per step: 2.1)
        for (uint32_t tile = tid; tile < WN * WN / 32; tile += 256) {
            const uint32_t t0 = 4 * (tile / (WN / 8)), k8 = (tile % (WN / 8)) * 8;
            dp_u16x2 acc[4][4];
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int c = 0; c < 4; c++) acc[a][c] = dp_pair(DPB_INF);
#pragma unroll 2
            for (uint32_t u = 0; u < WN; u += 4) {
                uint32_t w[4];
#pragma unroll
                for (int a = 0; a < 4; a++) w[a] = *reinterpret_cast<const uint32_t *>(W + (t0 + a) * WN + u);
#pragma unroll
                for (int uu = 0; uu < 4; uu++) {
                    const uint2 p8 = *reinterpret_cast<const uint2 *>(P + (u + uu) * WN + k8);
                    const dp_u16x2 pe0 = dp_bits(p8.x & 0x00ff00ffu), po0 = dp_bits((p8.x >> 8) & 0x00ff00ffu),
                                   pe1 = dp_bits(p8.y & 0x00ff00ffu), po1 = dp_bits((p8.y >> 8) & 0x00ff00ffu);
#pragma unroll
                    for (int a = 0; a < 4; a++) {
                        const dp_u16x2 ww = dp_bits(__builtin_amdgcn_perm(w[a], w[a], 0x0c000c00u | (uint32_t)uu | ((uint32_t)uu << 16)));
                        acc[a][0] = __builtin_elementwise_min(acc[a][0], __builtin_elementwise_max(ww, pe0));
                        acc[a][1] = __builtin_elementwise_min(acc[a][1], __builtin_elementwise_max(ww, po0));
                        acc[a][2] = __builtin_elementwise_min(acc[a][2], __builtin_elementwise_max(ww, pe1));
                        acc[a][3] = __builtin_elementwise_min(acc[a][3], __builtin_elementwise_max(ww, po1));
                    }
                }
            }
#pragma unroll
            for (int a = 0; a < 4; a++) {
                uint2 o;
                o.x = dp_word(acc[a][0]) | (dp_word(acc[a][1]) << 8);
                o.y = dp_word(acc[a][2]) | (dp_word(acc[a][3]) << 8);
                *reinterpret_cast<uint2 *>(G + (t0 + a) * WN + k8) = o;
            }
        }
        __syncthreads();
        // the product becomes P for the next block of the group
        for (uint32_t q = tid * 16; q < WN * WN; q += 256 * 16)
            *reinterpret_cast<uint4 *>(P + q) = *reinterpret_cast<const uint4 *>(G + q);
    }
}

// state entering block b (64R values) (x) matrix of block b -> minmaxlength of the block's columns
template <int R>
__global__ __launch_bounds__(64) void k_dp_expand(const uint8_t *__restrict__ Wt, const uint8_t *__restrict__ Sg,
                                                  uint32_t n, uint32_t nblocks, uint32_t F, uint32_t *__restrict__ mml,
                                                  unsigned long long *__restrict__ flag, uint32_t first_valid)
{
    // first_valid = f[0] + 1: no block ends before that step, and the reference's sweep leaves its running S there:
    // minmaxlength[j] = n + j, backtrack[j] = -1 (fbg.cpp:1967, 2007-2010 with backtrack_S still size_type(-1))
    constexpr uint32_t WN = 64 * R;
    __shared__ uint8_t s_state[WN];
    const uint32_t lane = threadIdx.x;
    for (uint32_t b = blockIdx.x; b < nblocks; b += gridDim.x) {
        __syncthreads();
        for (uint32_t q = lane; q < WN; q += 64) s_state[q] = Sg[(size_t)(b / F) * WN + q];
        __syncthreads();
        const uint8_t *Wb = Wt + (size_t)b * WN * WN;
#pragma unroll
        for (int rt = 0; rt < R; rt++) {
            const uint32_t t = 64 * rt + lane;
            const uint4 *row = reinterpret_cast<const uint4 *>(Wb + (size_t)t * WN);
            uint32_t acc = DPB_INF;
            for (uint32_t q = 0; q < WN / 16; q++) {
                const uint4 w4 = row[q];
                const uint32_t ws[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                for (int i = 0; i < 16; i++)
                    acc = min(acc, max((uint32_t)s_state[16 * q + i], (ws[i >> 2] >> (8 * (i & 3))) & 255u));
            }
            const uint32_t j = WN * b + 1 + t;
            if (j <= n) {
                if (j < first_valid) mml[j] = n + j;
                else {
                    mml[j] = acc;
                    // exact as long as every value stays below the window: a block longer than the window, or one that
                    // starts at a column whose minimal extension exceeds it, costs more than any value seen (fbg_dp_minmax)
                    if (acc >= (R == 4 ? 255u : 64u * R)) flag[4] = 1;
                }
            }
        }
    }
}

__global__ void k_dp_bt(const uint32_t *__restrict__ mml, const uint8_t *__restrict__ ext7, uint32_t n, uint32_t window,
                        uint32_t *__restrict__ bt, unsigned long long *__restrict__ flag, uint32_t first_valid)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > n) return;
    if (j == 0) { bt[0] = 0; return; }
    if (j < first_valid) { bt[j] = DP_NONE; return; }
    const uint32_t L = mml[j];
    uint32_t best_age = 0;                    // youngest count_solutions-kind candidate (age <= its value == L)
    bool s_kind = false;                      // candidate of age L with a smaller value: the S / backtrack_S branch
    const uint32_t amax = min(window, j);
    for (uint32_t a = 1; a <= amax; a++) {
        const uint32_t x = j - a;
        if (ext7[x] > a) continue;            // block [x, j) not valid yet: f[x]+1 > j
        const uint32_t v = mml[x];
        if (v > n) continue;                  // a column before first_valid: no solution ends there (fbg.cpp:1973)
        if (a > v) { if (a == L) s_kind = true; }
        else if (v == L && best_age == 0) best_age = a;
    }
    if (s_kind) bt[j] = j - L;
    else if (best_age) bt[j] = j - best_age;
    else { bt[j] = 0; flag[4] = 1; }
}

// backtrack (fbg.cpp:2026-2039): walks backtrack[] backwards through an 8192-column LDS window (one bulk
// load per window instead of one global round trip per hop); the boundaries are collected in reverse in
// LDS and flushed 1024 at a time.  rev[] receives them, result[0] the count.
__global__ __launch_bounds__(64) void k_dp_backtrack_wave(const uint32_t *__restrict__ bt, uint32_t n,
                                                          uint32_t *__restrict__ rev,
                                                          unsigned long long *__restrict__ result)
{
    constexpr uint32_t WIN = 8192;           // backtrack[] window held in LDS, reloaded when walked out of
    __shared__ uint32_t s_bt[WIN];
    __shared__ uint32_t obuf[1024];
    const uint32_t lane = threadIdx.x;
    uint32_t j = n, cnt = 1, flushed = 0;
    uint32_t wlo = 0xffffffffu, whi = 0;      // window covers columns [wlo, whi]
    if (lane == 0) obuf[0] = n;
    bool err = false;
    for (;;) {
        if (j < wlo || j > whi) {             // (re)load the window ending at j
            __syncthreads();
            whi = j;
            wlo = j >= WIN - 1 ? j - (WIN - 1) : 0;
            for (uint32_t k = lane; k <= whi - wlo; k += 64) s_bt[k] = bt[wlo + k];
            __syncthreads();
        }
        const uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_bt[j - wlo]);
        if (b == 0) break;
        if (b > n || cnt > n + 1) { err = true; break; }
        if (cnt - flushed == 1024) {
            __syncthreads();
            for (uint32_t k = lane; k < 1024; k += 64) rev[flushed + k] = obuf[k];
            flushed += 1024;
            __syncthreads();
        }
        if (lane == 0) obuf[cnt - flushed] = b - 1;
        cnt++;
        j = b;
    }
    __syncthreads();
    for (uint32_t k = lane; k < cnt - flushed; k += 64) rev[flushed + k] = obuf[k];
    if (lane == 0) { result[0] = err ? 0 : cnt; result[1] = err ? 1 : 0; }
}

// ---- wide windows: the matrix chain for block lengths of up to sixteen thousand columns ----------------------------
//
// Rows that resemble each other, with gaps: behind a deletion a row's string occurs in the other rows some columns on, the
// minimal extensions reach hundreds of columns and so do the blocks -- beyond the byte entries and LDS-sized square
// matrices above, and the statement-by-statement sweep walks such an input at 0.5 us per column.  The same product with
// 16-bit entries and RECTANGULAR matrices: a block of DPW_B = 128 steps acts on the WS (1024, 2048, ... 16384) values before it
// through M_b[t][k] -- the recurrence of a source k along the block's steps only involves the block's own 128 columns
// as intermediate candidates, so a thread owns a source and keeps 128 values in LDS, 256 sources per workgroup.
//   k_dpw_blockM   all M_b, independently: (n / 128) * (WS / 256) workgroups; two candidates per 32-bit LDS word go
//                  through packed 16-bit max / min, their validity comes from a per-wave mask as in k_dp_blockW
//   k_dpw_chain    one workgroup walks the blocks: minmaxlength of a block's 128 columns = M_b (x) the ring of the last WS
//                  values (lanes along the sources: coalesced rows, one reduction per target); sources too old to win
//                  in a block are not read
//   k_dpw_bt       backtrack[j] from minmaxlength[] as k_dp_bt, 16-bit extensions
// A value that reaches WS raises flag[4] (the exactness argument of fbg_dp_minmax): next size, then the literal sweep.
#define DPW_B 128u
#define DPW_INF 0xffffu
typedef uint16_t __attribute__((may_alias)) dpw_u16;     // the LDS rows are written as 16-bit entries and read as pairs / quads
typedef uint32_t __attribute__((may_alias)) dpw_u32;
__global__ void k_dpw_prep(const uint32_t *__restrict__ e, uint32_t n, uint16_t *__restrict__ ext16)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > n) return;
    ext16[j] = j < n ? (uint16_t)min(e[j] - j, DPW_INF) : (uint16_t)DPW_INF;
}

template <uint32_t WS>
__global__ __launch_bounds__(256) void k_dpw_blockM(const uint16_t *__restrict__ ext16, uint32_t n, uint16_t *__restrict__ M)
{
    constexpr uint32_t RP = DPW_B + 2;            // row stride in 16-bit entries: 65 words, the lanes' rows start in different banks
    __shared__ uint32_t rows_w[256 * RP / 2];     // row k, entry t: M of inside column (block start + 1 + t) for source k
    __shared__ uint16_t s_ext[DPW_B];
    __shared__ uint32_t mk_w[4][DPW_B / 2];       // per wave: candidates 2g, 2g + 1 as the 16-bit lanes of word g; 0xffff = does not count yet
    const uint32_t b = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t k = blockIdx.y * 256 + threadIdx.x;                   // source: prefix length xs = jb - WS + 1 + k
    const uint32_t jb = DPW_B * b;
    const int64_t xs = (int64_t)jb - (int64_t)(WS - 1) + k;
    const uint32_t ext_src = xs >= 0 ? ext16[xs] : DPW_INF;
    if (threadIdx.x < DPW_B) s_ext[threadIdx.x] = jb + 1 + threadIdx.x < n ? ext16[jb + 1 + threadIdx.x] : (uint16_t)DPW_INF;
    __syncthreads();
    uint32_t minext = DPW_INF;
    for (uint32_t q = 0; q < DPW_B; q++) minext = min(minext, (uint32_t)s_ext[q]);
    dpw_u32 *mk = reinterpret_cast<dpw_u32 *>(mk_w[wv]);
    dpw_u16 *mk16 = reinterpret_cast<dpw_u16 *>(mk_w[wv]);
    mk[lane] = 0xffffffffu;
    uint32_t ready[2];                                                   // the step from which the lane's candidates (lane, lane + 64) count
#pragma unroll
    for (int r = 0; r < 2; r++) { const uint32_t tp = lane + 64 * r; ready[r] = tp + max(1u, (uint32_t)s_ext[tp]); }
    dpw_u16 *row = reinterpret_cast<dpw_u16 *>(rows_w) + threadIdx.x * RP;
    const dpw_u32 *row32 = reinterpret_cast<const dpw_u32 *>(rows_w) + threadIdx.x * (RP / 2);
    uint16_t *out = M + (size_t)b * DPW_B * WS + k;
    for (uint32_t t = 0; t < DPW_B; t++) {
#pragma unroll
        for (int r = 0; r < 2; r++)
            if (ready[r] == t) mk16[lane + 64 * r] = 0;
        __builtin_amdgcn_wave_barrier();
        const uint32_t age_src = t + WS - k;                             // (jb + 1 + t) - xs
        uint32_t w = (ext_src <= age_src && age_src <= WS) ? age_src : DPW_INF;
        const uint32_t tp_end = t >= minext ? t - minext + 1 : 0;        // inside candidates need age >= their extension >= minext
        dp_u16x2 acc = dp_pair(DPW_INF);
        uint32_t g2 = 0;
#pragma unroll 4
        for (; 2 * g2 + 2 <= tp_end; g2++) {
            const uint32_t base = t - 2 * g2;                            // ages of candidates 2 g2 and 2 g2 + 1
            acc = __builtin_elementwise_min(acc, __builtin_elementwise_max(dp_bits(row32[g2] | mk[g2]), dp_bits(base | ((base - 1) << 16))));
        }
        if (2 * g2 < tp_end) {                                           // one candidate left: the odd lane does not count
            const uint32_t base = t - 2 * g2;
            acc = __builtin_elementwise_min(acc, __builtin_elementwise_max(dp_bits(row32[g2] | mk[g2] | 0xffff0000u), dp_bits(base | 0xffff0000u)));
        }
        w = min(w, min((uint32_t)acc.x, (uint32_t)acc.y));
        row[t] = (uint16_t)w;
        out[(size_t)t * WS] = (uint16_t)w;                               // M[b][t][k]
    }
}

template <uint32_t WS>
__global__ __launch_bounds__(1024) void k_dpw_chain(const uint16_t *__restrict__ M, uint32_t n, uint32_t nblocks, uint32_t *__restrict__ mml,
                                                    unsigned long long *__restrict__ flag, uint32_t first_valid)
{
    // first_valid = f[0] + 1 as in k_dp_expand: the steps before it have no value (the ring keeps "none") and the reference's
    // arrays hold the running S there
    // ring[(x - 1) & (WS - 1)] = minmaxlength of prefix length x, for the WS prefix lengths before the current block:
    // source k of block b (x = jb - WS + 1 + k) sits at (jb + k) & (WS - 1) -- eight sources stay 16-byte aligned.
    // 16 waves of 8 targets each: a wave's 8 rows of M_b are contiguous, a lane holds 8 sources per 16-byte load, and the
    // rows are loaded one stage ahead of their use (across block ends too), so the walk runs at the rate one CU takes in M
    // what it has to read of it
    constexpr uint32_t NQ = WS / 512;            // chunks of 512 sources: one 16-byte load per lane and row
    __shared__ uint4 ring4[WS / 8];
    __shared__ uint32_t fresh[DPW_B];
    dpw_u16 *ring = reinterpret_cast<dpw_u16 *>(ring4);
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (uint32_t i = threadIdx.x; i < WS / 8; i += 1024) ring4[i] = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
    __syncthreads();
    if (threadIdx.x == 0) { ring[WS - 1] = 0; mml[0] = 0; }              // prefix length 0
    __syncthreads();
    bool bad = false;
    const uint4 *M4 = reinterpret_cast<const uint4 *>(M) + (size_t)(8 * wv) * (WS / 8) + lane;
    // Sources that cannot win are not read: every way from source k into the block starts with a block of at least
    // WS - k columns (up to the block's first column), and minmaxlength grows by at most 1 per column (the last block of
    // an optimal segmentation, one column longer, is still valid) -- so with L = minmaxlength at the column before the
    // block no target of the block lies above L + 128, and no source with WS - k > L + 128 can give its minimum.  The
    // chunks go from the youngest sources down; whole chunks below the bound are left out, in the chunk that holds it the
    // lanes below it skip their load (a lane holds sources 512 c + 8 lane .. + 7; the youngest chunk of the NEXT block is
    // asked for one block early: L + 256).  The chain reads in proportion to the block lengths, not to the window.
    // A stage = one chunk of the wave's 8 rows; the next stage's rows are loaded while this one is worked on.
    const uint4 none = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
    uint32_t L = 0;                              // minmaxlength[jb]; "none" (f[0] > 0, or past a flagged value): no bound
    uint4 buf[2][8];
#pragma unroll
    for (uint32_t i = 0; i < 8; i++) buf[0][i] = M4[(size_t)i * (WS / 8) + 64 * (NQ - 1)];
    for (uint32_t b = 0; b < nblocks; b++) {
        const uint32_t jb = DPW_B * b;
        const uint32_t kmin_here = WS - min(L + DPW_B, WS), kmin_next = WS - min(L + 2 * DPW_B, WS);
        const uint32_t ns = NQ - min(kmin_here / 512, NQ - 1);           // chunks NQ-1 .. NQ-ns are read
        const uint4 *Mb = M4 + (size_t)b * DPW_B * (WS / 8);
        dp_u16x2 acc[8];
#pragma unroll
        for (uint32_t i = 0; i < 8; i++) acc[i] = dp_pair(DPW_INF);
#pragma unroll
        for (uint32_t sg = 0; sg < NQ; sg++) {
            if (sg < ns) {                                               // uniform
                const uint32_t c = NQ - 1 - sg;
                if (sg + 1 < ns) {                                       // the next chunk of this block (c >= 1 here)
                    const bool need = 512 * (c - 1) + 8 * lane + 7 >= kmin_here;
#pragma unroll
                    for (uint32_t i = 0; i < 8; i++) buf[(sg + 1) & 1][i] = need ? Mb[(size_t)i * (WS / 8) + 64 * (c - 1)] : none;
                } else if (b + 1 < nblocks) {                            // the youngest chunk of the next block: M does not depend on the state
                    const bool need = 512 * (NQ - 1) + 8 * lane + 7 >= kmin_next;
#pragma unroll
                    for (uint32_t i = 0; i < 8; i++)
                        buf[(sg + 1) & 1][i] = need ? Mb[(size_t)(DPW_B + i) * (WS / 8) + 64 * (NQ - 1)] : none;
                }
                const uint4 s4 = ring4[((jb >> 3) + 64 * c + lane) & (WS / 8 - 1)];
#pragma unroll
                for (uint32_t i = 0; i < 8; i++) {
                    const uint4 r = buf[sg & 1][i];
                    acc[i] = __builtin_elementwise_min(acc[i], __builtin_elementwise_max(dp_bits(s4.x), dp_bits(r.x)));
                    acc[i] = __builtin_elementwise_min(acc[i], __builtin_elementwise_max(dp_bits(s4.y), dp_bits(r.y)));
                    acc[i] = __builtin_elementwise_min(acc[i], __builtin_elementwise_max(dp_bits(s4.z), dp_bits(r.z)));
                    acc[i] = __builtin_elementwise_min(acc[i], __builtin_elementwise_max(dp_bits(s4.w), dp_bits(r.w)));
                }
            }
        }
        if (ns & 1) {                                                    // the next block starts on buffer 0
#pragma unroll
            for (uint32_t i = 0; i < 8; i++) buf[0][i] = buf[1][i];
        }
        uint32_t part[8];                        // the lane's minimum per target of the wave
#pragma unroll
        for (uint32_t i = 0; i < 8; i++) part[i] = min((uint32_t)acc[i].x, (uint32_t)acc[i].y);
        // the eight minima over the wave in 7 exchanges instead of 48: two targets per word, and each exchange halves what a
        // lane still carries (lanes 32.. keep targets 4..7, then lanes with bit 4 set the odd pair), before the last four
        // steps run on one word
        uint32_t w0 = part[0] | (part[1] << 16), w1 = part[2] | (part[3] << 16), w2 = part[4] | (part[5] << 16), w3 = part[6] | (part[7] << 16);
        {
            const bool up = (lane & 32) != 0;
            const uint32_t s0 = up ? w0 : w2, s1 = up ? w1 : w3;           // what goes to the other half
            const uint32_t k0 = up ? w2 : w0, k1 = up ? w3 : w1;
            w0 = dp_word(__builtin_elementwise_min(dp_bits(k0), dp_bits((uint32_t)__shfl_xor((int)s0, 32, 64))));
            w1 = dp_word(__builtin_elementwise_min(dp_bits(k1), dp_bits((uint32_t)__shfl_xor((int)s1, 32, 64))));
            const bool odd = (lane & 16) != 0;
            const uint32_t s = odd ? w0 : w1, k = odd ? w1 : w0;
            w0 = dp_word(__builtin_elementwise_min(dp_bits(k), dp_bits((uint32_t)__shfl_xor((int)s, 16, 64))));
#pragma unroll
            for (int d = 8; d >= 1; d >>= 1) w0 = dp_word(__builtin_elementwise_min(dp_bits(w0), dp_bits((uint32_t)__shfl_xor((int)w0, d, 64))));
            if ((lane & 15) == 0) {                                        // lane 0: targets 0, 1; 16: 2, 3; 32: 4, 5; 48: 6, 7
                const uint32_t t0 = 8 * wv + 4 * (lane >> 5) + 2 * ((lane >> 4) & 1);
                fresh[t0] = w0 & 0xffffu;
                fresh[t0 + 1] = w0 >> 16;
            }
        }
        dpw_lds_barrier();
        if (threadIdx.x < DPW_B) {
            const uint32_t j = jb + 1 + threadIdx.x;
            const uint32_t v = fresh[threadIdx.x];
            ring[(j - 1) & (WS - 1)] = (uint16_t)v;                      // takes the place of prefix length j - WS
            if (j <= n) { if (j < first_valid) mml[j] = n + j; else { mml[j] = v; if (v >= WS) bad = true; } }
        }
        L = fresh[DPW_B - 1];                                            // stays until the next block's minima are written, after the barrier below
        dpw_lds_barrier();
    }
    if (bad) flag[4] = 1;
}

// ---- round 4: the wide-window chain without the wide matrices ------------------------------------------------------
//
// A source that lies more than 128 columns before a block can enter it only by ONE block that is longer than anything that
// follows inside (the inner blocks are at most 127 long): the cost of reaching target t from such a source x through the
// inner cut point c is max(minmaxlength[x], c - x) whenever t can be reached from c at all.  So the block has to know two
// things about its inside -- the 128 x 128 matrix of the 128 sources right before it (k_dpw_blockY, as k_dpw_blockM), and
// the BIT matrix "t is reachable from c" -- and for the older sources
//     G[c] = min over the old sources x with f[x] + 1 <= c of max(minmaxlength[x], c - x),   out[t] = min over c that reach t of G[c]
// is a lower envelope that needs no matrix at all.  A source that is valid from the block's first column on contributes its
// value while that dominates (a prefix of the block: a table of 128 minima and one suffix-minimum scan) and its age from then
// on (a second table, one prefix-minimum scan); the sources whose minimal extension ends inside the block (each source once in
// its life) are listed and evaluated against the targets one by one.  What the walk reads per block: 32 KB of matrix instead
// of 2 * 128 * (block length + 256) bytes, whatever the window; the matrices take 256 + 16 bytes per column instead of 2 * WS.
// (tests/test_wide_chain_model.py restates this in numpy against the oracle.)
__global__ __launch_bounds__(256) void k_dpw_blockY(const uint16_t *__restrict__ ext16, uint32_t n, uint16_t *__restrict__ My,
                                                    unsigned long long *__restrict__ Rb)
{
    // threads 0..127: the sources of the block before (prefix length x = jb - 127 + k); 128..255: the block's own columns as
    // sources (the same formula), of whose rows only "finite or not" is kept: bit c of Rb[b][t]
    constexpr uint32_t RP = DPW_B + 2;
    __shared__ uint32_t rows_w[256 * RP / 2];
    __shared__ uint16_t s_ext[DPW_B];
    __shared__ uint32_t mk_w[4][DPW_B / 2];
    const uint32_t b = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6, k = threadIdx.x;
    const uint32_t jb = DPW_B * b;
    const int64_t xs = (int64_t)jb - (int64_t)(DPW_B - 1) + k;
    const uint32_t ext_src = (xs >= 0 && xs <= (int64_t)n) ? ext16[xs] : DPW_INF;
    if (threadIdx.x < DPW_B) s_ext[threadIdx.x] = jb + 1 + threadIdx.x < n ? ext16[jb + 1 + threadIdx.x] : (uint16_t)DPW_INF;
    __syncthreads();
    uint32_t minext = DPW_INF;
    for (uint32_t q = 0; q < DPW_B; q++) minext = min(minext, (uint32_t)s_ext[q]);
    dpw_u32 *mk = reinterpret_cast<dpw_u32 *>(mk_w[wv]);
    dpw_u16 *mk16 = reinterpret_cast<dpw_u16 *>(mk_w[wv]);
    mk[lane] = 0xffffffffu;
    uint32_t ready[2];
#pragma unroll
    for (int r = 0; r < 2; r++) { const uint32_t tp = lane + 64 * r; ready[r] = tp + max(1u, (uint32_t)s_ext[tp]); }
    dpw_u16 *row = reinterpret_cast<dpw_u16 *>(rows_w) + threadIdx.x * RP;
    const dpw_u32 *row32 = reinterpret_cast<const dpw_u32 *>(rows_w) + threadIdx.x * (RP / 2);
    uint16_t *out = My + (size_t)b * DPW_B * DPW_B + k;
    for (uint32_t t = 0; t < DPW_B; t++) {
#pragma unroll
        for (int r = 0; r < 2; r++)
            if (ready[r] == t) mk16[lane + 64 * r] = 0;
        __builtin_amdgcn_wave_barrier();
        const uint32_t age_src = t + DPW_B - k;                          // wraps for the block's own columns at and behind the target
        uint32_t w = (age_src - 1 < 2 * DPW_B && ext_src <= age_src) ? age_src : DPW_INF;
        const uint32_t tp_end = t >= minext ? t - minext + 1 : 0;
        dp_u16x2 acc = dp_pair(DPW_INF);
        uint32_t g2 = 0;
#pragma unroll 4
        for (; 2 * g2 + 2 <= tp_end; g2++) {
            const uint32_t base = t - 2 * g2;
            acc = __builtin_elementwise_min(acc, __builtin_elementwise_max(dp_bits(row32[g2] | mk[g2]), dp_bits(base | ((base - 1) << 16))));
        }
        if (2 * g2 < tp_end) {
            const uint32_t base = t - 2 * g2;
            acc = __builtin_elementwise_min(acc, __builtin_elementwise_max(dp_bits(row32[g2] | mk[g2] | 0xffff0000u), dp_bits(base | 0xffff0000u)));
        }
        w = min(w, min((uint32_t)acc.x, (uint32_t)acc.y));
        row[t] = (uint16_t)w;
        if (wv < 2) out[(size_t)t * DPW_B] = (uint16_t)w;                // My[b][t][k]
        else {
            const unsigned long long bits = __ballot(w != DPW_INF);
            if (lane == 0) Rb[((size_t)b * DPW_B + t) * 2 + (wv - 2)] = bits;
        }
    }
}

#define DPW_LIST 2048u                                 // listed sources per block; more are worked off by their own threads
#define DPW_NT 512u                                    // threads of k_dpw_chain2
// minimum over the 16 lanes of a DPP row, in every lane; the packed form takes the two 16-bit halves apart
__device__ __forceinline__ uint32_t dpw_row_min(uint32_t v)
{
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false));   // row_half_mirror
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, false));   // row_mirror
    return v;
}
__device__ __forceinline__ uint32_t dpw_row_min_pk(uint32_t v)
{
    v = dp_word(__builtin_elementwise_min(dp_bits(v), dp_bits((uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false))));
    v = dp_word(__builtin_elementwise_min(dp_bits(v), dp_bits((uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false))));
    v = dp_word(__builtin_elementwise_min(dp_bits(v), dp_bits((uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false))));
    v = dp_word(__builtin_elementwise_min(dp_bits(v), dp_bits((uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, false))));
    return v;
}
__device__ __forceinline__ uint32_t dpw_wave_min(uint32_t x)                                     // uniform result
{
    x = dpw_row_min(x);
    return min(min((uint32_t)__builtin_amdgcn_readlane((int)x, 0), (uint32_t)__builtin_amdgcn_readlane((int)x, 16)),
               min((uint32_t)__builtin_amdgcn_readlane((int)x, 32), (uint32_t)__builtin_amdgcn_readlane((int)x, 48)));
}
// inclusive prefix minimum over the 64 lanes (row_shr inside the rows of 16, then the rows' last lanes passed on)
__device__ __forceinline__ uint32_t dpw_wave_prefix_min(uint32_t x)
{
    const int inf = (int)DPW_INF;
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(inf, (int)x, 0x111, 0xF, 0xF, false));
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(inf, (int)x, 0x112, 0xF, 0xF, false));
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(inf, (int)x, 0x114, 0xF, 0xF, false));
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(inf, (int)x, 0x118, 0xF, 0xF, false));
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(inf, (int)x, 0x142, 0xA, 0xF, false));    // row_bcast:15 into rows 1, 3
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(inf, (int)x, 0x143, 0xC, 0xF, false));    // row_bcast:31 into rows 2, 3
    return x;
}
// tab[idx] = min(tab[idx], val) for the active lanes; when they all name one entry (a plateau of f: one step of the block
// for a run of sources) the wave sends a single atomic
__device__ __forceinline__ void dpw_table_min(uint32_t *tab, uint32_t idx, uint32_t val, bool active)
{
    const unsigned long long m = __ballot(active);
    if (!m) return;
    const uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)idx, (int)__builtin_ctzll(m));
    if (__ballot(active && idx == first) == m) {
        const uint32_t mv = dpw_wave_min(active ? val : DPW_INF);
        if ((threadIdx.x & 63) == 0) atomicMin(&tab[first], mv);
    } else if (active) atomicMin(&tab[idx], val);
}
// dynamic LDS of k_dpw_chain2<WS>: the two rings (2 * WS bytes each), then the tables and the list below
#define DPW_CHAIN2_LDS(WS) (4u * (WS) + 4u * (128u * 9u + 4u) + 6u * DPW_LIST + 16u)

template <uint32_t WS>
__global__ __launch_bounds__(DPW_NT + 128) void k_dpw_chain2(const uint16_t *__restrict__ My, const unsigned long long *__restrict__ Rb,
                                                       const uint16_t *__restrict__ ext16, uint32_t n, uint32_t nblocks,
                                                       uint32_t *__restrict__ mml, unsigned long long *__restrict__ flag, uint32_t first_valid, uint32_t probe)
{
    // ringv[(x - 1) & (WS - 1)] = minmaxlength of prefix length x, ringe[...] = minimal extension of column x (a block [x, j)
    // needs j - x >= it), for the WS prefix lengths before the current block; 0xffff = none.
    // Ten waves.  Waves 0..7 compute: wave w owns targets 16 w .. 16 w + 15 of the matrix part (four loads of four rows, 16 lanes
    // a row, fetched one block ahead: the matrices do not depend on the state); every thread owns old sources (age at the block's first
    // column a0 = 129 + q, q = thread + 512 r, up to L + 128: older ones cannot win, see k_dpw_chain); thread (t, s) = (tid / 4,
    // tid % 4) evaluates target t against the s-th part of the listed sources, then against 32 of the inner cut points.
    // An old source x that may end a block at the block's first column already (lo = 0) gives max(v, a0 + t) at step t:
    // v up to step rr = v - a0 -- pfx, a table by rr read as a suffix minimum (kept reversed) -- and a0 + t from rr + 1 on --
    // slope, a table by the first such step read as a prefix minimum.  One that becomes valid at step lo > 0 uses slope the
    // same way from max(lo, rr + 1), csfx (by lo, prefix minimum) when its value dominates to the block's end, and is listed
    // (lo, rr, v) when it dominates for a part [lo, rr] only.  Wave 8 takes the block before from the ring to memory, wave 9
    // touches the lines PF blocks ahead (both below).  `probe` (0 in the library): bits that leave parts out, for timing only
    // (wrong results; DESIGN_HISTORY.md has what they measured).
    constexpr uint32_t PF = 6;                         // blocks the tenth wave runs ahead
    constexpr uint32_t MASK = WS - 1;
    extern __shared__ uint4 dpw_dyn[];
    uint4 *ringv4 = dpw_dyn;
    dpw_u16 *ringv = reinterpret_cast<dpw_u16 *>(dpw_dyn);
    dpw_u16 *ringe = ringv + WS;
    uint32_t *pfx = reinterpret_cast<uint32_t *>(ringe + WS);      // [127 - rr]
    uint32_t *slope = pfx + 128;
    uint32_t *csfx = slope + 128;
    uint32_t *spfx = csfx + 128, *sslope = spfx + 128, *scsfx = sslope + 128;   // the three after their scans
    uint32_t *gl = scsfx + 128;                                    // [2][128]: the listed sources' part of G, one table per block parity
    uint32_t *fresh_y = gl + 256;                                  // the matrix part of the block's values
    uint32_t *lcount = fresh_y + 128;                              // [0] listed sources, [1] minmaxlength of the block's last column
    dpw_u16 *lst = reinterpret_cast<dpw_u16 *>(lcount + 4);        // per 8 listed sources: 8 x v, 8 x (65536 - lo), 8 x rr
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t tt = tid >> 2, sl = tid & 3;
    for (uint32_t i = tid; i < WS / 8; i += DPW_NT) {
        ringv4[i] = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
        reinterpret_cast<uint4 *>(ringe)[i] = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
    }
    for (uint32_t i = tid; i < 128 * 9; i += DPW_NT) pfx[i] = DPW_INF;
    __syncthreads();
    if (tid == 0) { ringv[WS - 1] = 0; ringe[WS - 1] = ext16[0]; lcount[0] = 0; lcount[1] = 0; }   // prefix length 0
    __syncthreads();
    if (wv == DPW_NT / 64) {
        // the ninth wave takes the blocks' values from the ring to memory, a block behind the others.  The eight that compute issue
        // no store at all: loads and stores share one counter (vmcnt) that only stays in order for loads alone -- with a store
        // in flight every use of a row fetched ahead would wait for ALL the wave's loads, the newest included
        bool bad = false;
        if (lane == 0) mml[0] = 0;
        for (uint32_t b = 0; b < nblocks; b++) {
            dpw_lds_barrier();
            if (!(probe & 128)) { dpw_lds_barrier(); dpw_lds_barrier(); }
            const uint32_t j0 = DPW_B * b + 1 + 2 * lane;
            const uint32_t w = *reinterpret_cast<const dpw_u32 *>(ringv + ((j0 - 1) & MASK));
#pragma unroll
            for (uint32_t h = 0; h < 2; h++) {
                const uint32_t j = j0 + h, v = h ? w >> 16 : w & 0xffffu;
                if (j <= n) { if (j < first_valid) mml[j] = n + j; else { mml[j] = v; if (v >= WS) bad = true; } }
            }
        }
        if (bad) flag[4] = 1;
        return;
    }
    if (wv == DPW_NT / 64 + 1) {
        // the tenth wave touches one word of every 128-byte line the block PF steps ahead will read (matrix, bits, extensions),
        // so that the loads of the eight -- issued one block ahead, as far as the compiler's wait counts follow a value through
        // registers -- find their lines in the L2.  The loads are written as assembly with one fixed destination register that nothing reads:
        // the compiler would make the wave wait for a value it loads before it lets go of it, and this wave shares the others'
        // barriers -- a wave that waits for memory there makes every block as long as a trip to memory
        const char *mp = reinterpret_cast<const char *>(My) + (size_t)lane * 128;
        const char *rp = reinterpret_cast<const char *>(Rb) + (size_t)(lane & 15) * 128;
        for (uint32_t b = 0; b < nblocks; b++) {
            const uint32_t bn = min(b + PF, nblocks - 1);
            const char *m0 = mp + (size_t)bn * 32768, *r0 = rp + (size_t)bn * 2048;
            const uint16_t *e0 = ext16 + min(DPW_B * bn + 1 + 64 * (lane & 1), n);
            asm volatile("global_load_dword v127, %0, off\n\t"
                         "global_load_dword v127, %1, off\n\t"
                         "global_load_dword v127, %2, off\n\t"
                         "global_load_dword v127, %3, off\n\t"
                         "global_load_dword v127, %4, off\n\t"
                         "global_load_ushort v127, %5, off"
                         :
                         : "v"(m0), "v"(m0 + 8192), "v"(m0 + 16384), "v"(m0 + 24576), "v"(r0), "v"(e0)
                         : "v127", "memory");
            dpw_lds_barrier();
            if (!(probe & 128)) { dpw_lds_barrier(); dpw_lds_barrier(); }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    const uint4 *M4 = reinterpret_cast<const uint4 *>(My) + (size_t)(16 * wv) * (DPW_B / 8) + lane;
    const uint32_t *R32 = reinterpret_cast<const uint32_t *>(Rb) + (size_t)tt * 4 + sl;
    uint4 nxt[4];                                      // the rows, bits and extension of the next block: loaded a block ahead
#pragma unroll
    for (uint32_t i = 0; i < 4; i++) nxt[i] = M4[64 * i];
    uint32_t nrb = R32[0], nex = ext16[min(1 + tt, n)];
    uint32_t L = 0;                                    // minmaxlength at the column before the block; 0xffff (none): no bound
    {
        for (uint32_t b = 0; b < nblocks; b++) {
            const uint32_t jb = DPW_B * b;
            uint4 cur[4];
#pragma unroll
            for (uint32_t i = 0; i < 4; i++) cur[i] = nxt[i];
            const uint32_t crb = nrb, cex = nex;
            {                                          // (a block beyond the last is asked for as the last: no branches; ext16[n] = none)
                const uint32_t bn = (probe & 64) ? 0u : min(b + 1, nblocks - 1);
                if (!(probe & 64) || b == 0)
#pragma unroll
                for (uint32_t i = 0; i < 4; i++) nxt[i] = M4[(size_t)bn * DPW_B * (DPW_B / 8) + 64 * i];
                nrb = R32[(size_t)bn * DPW_B * 4];
                nex = ext16[min(DPW_B * bn + 1 + tt, n)];
            }
            uint32_t *glb = gl + 128 * (b & 1);
            // ---- phase 0 (a): the 128 sources before the block through their matrix
            if (!(probe & 2)) {
                const uint4 s4 = ringv4[(((jb - DPW_B) >> 3) + (lane & 15)) & (WS / 8 - 1)];
                uint32_t part[4];
#pragma unroll
                for (uint32_t i = 0; i < 4; i++) {
                    const uint4 r = cur[i];
                    dp_u16x2 a = __builtin_elementwise_max(dp_bits(s4.x), dp_bits(r.x));
                    a = __builtin_elementwise_min(a, __builtin_elementwise_max(dp_bits(s4.y), dp_bits(r.y)));
                    a = __builtin_elementwise_min(a, __builtin_elementwise_max(dp_bits(s4.z), dp_bits(r.z)));
                    a = __builtin_elementwise_min(a, __builtin_elementwise_max(dp_bits(s4.w), dp_bits(r.w)));
                    part[i] = min((uint32_t)a.x, (uint32_t)a.y);
                }
                const uint32_t w0 = dpw_row_min_pk(part[0] | (part[1] << 16)), w1 = dpw_row_min_pk(part[2] | (part[3] << 16));
                if ((lane & 15) == 0) {                // load i holds rows 16 w + 4 i + (lane >> 4)
                    const uint32_t t0 = 16 * wv + (lane >> 4);
                    fresh_y[t0] = w0 & 0xffffu;
                    fresh_y[t0 + 4] = w0 >> 16;
                    fresh_y[t0 + 8] = w1 & 0xffffu;
                    fresh_y[t0 + 12] = w1 >> 16;
                }
            }
            // ---- phase 0 (b): the older sources
            {
                const uint32_t Q = (probe & 1) ? 0u : min(WS - DPW_B, L);
                for (uint32_t q0 = 0; q0 < Q; q0 += DPW_NT) {          // uniform
                    if (q0 + 64 * wv >= Q) break;                      // none of this wave's sources left (uniform in the wave)
                    const uint32_t q = q0 + tid;
                    const uint32_t a0 = DPW_B + 1 + q;
                    const uint32_t idx = (jb - DPW_B - 1 - q) & MASK;
                    const uint32_t v = ringv[idx], e = ringe[idx];
                    bool ok = q < Q && jb >= DPW_B + q && v < 0x8000u;   // prefix length x = jb - 128 - q >= 0
                    const uint32_t lo = e > a0 ? e - a0 : 0;
                    ok = ok && lo < DPW_B;
                    const int rr = (int)v - (int)a0;                       // the value dominates up to step rr
                    const bool always = ok && lo == 0, late = ok && lo != 0;
                    // the value's part
                    dpw_table_min(pfx, (uint32_t)(DPW_B - 1 - min(max(rr, 0), (int)DPW_B - 1)), v, always && rr >= 0);
                    dpw_table_min(csfx, lo, v, late && rr >= (int)DPW_B - 1);
                    // the age's part, from step max(lo, rr + 1) on
                    const uint32_t s = (uint32_t)max((int)lo, rr + 1);
                    dpw_table_min(slope, s, a0, ok && (int)s < (int)DPW_B && rr + 1 < (int)DPW_B);
                    // the value dominates on [lo, rr] only, lo > 0: listed
                    const bool part = late && rr >= (int)lo && rr < (int)DPW_B - 1;
                    const unsigned long long lm = __ballot(part);
                    if (lm) {
                        const uint32_t cnt = (uint32_t)__popcll(lm);
                        uint32_t base = 0;
                        if (lane == 0) base = atomicAdd(&lcount[0], cnt);
                        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                        const uint32_t pos = base + (uint32_t)__popcll(lm & ((1ull << lane) - 1ull));
                        if (part) {
                            if (pos < DPW_LIST) {
                                dpw_u16 *g = lst + 24 * (pos >> 3) + (pos & 7);
                                g[0] = (uint16_t)v; g[8] = (uint16_t)(0x10000u - lo); g[16] = (uint16_t)rr;
                            } else
                                for (uint32_t t = lo; t <= (uint32_t)rr; t++) atomicMin(&glb[t], v);
                        }
                    }
                }
            }
            dpw_lds_barrier();
            // ---- phase 1: the listed sources against the targets (pairs of sources in the 16-bit halves of a word: a step
            // outside [lo, rr] wraps to a value above every window); waves 0..2 scan a table each
            {
                const uint32_t cnt = min(lcount[0], DPW_LIST), ng = (probe & 4) ? 0u : (cnt + 7) >> 3;
                if (ng) {
                    dp_u16x2 acc = dp_pair(DPW_INF);
                    const dp_u16x2 t2 = dp_pair(tt);
                    for (uint32_t g = sl; g < ng; g += 4) {
                        const uint4 *gp = reinterpret_cast<const uint4 *>(lst + 24 * g);
                        uint4 v4 = gp[0];
                        const uint4 n4 = gp[1], r4 = gp[2];
                        const uint32_t left = cnt - 8 * g;                  // sources of this group that exist (1 .. 8, or more)
                        if (left < 8) {
                            if (left < 2) v4.x |= 0xffff0000u;
                            if (left < 3) v4.y = 0xffffffffu; else if (left < 4) v4.y |= 0xffff0000u;
                            if (left < 5) v4.z = 0xffffffffu; else if (left < 6) v4.z |= 0xffff0000u;
                            if (left < 7) v4.w = 0xffffffffu; else v4.w |= 0xffff0000u;
                        }
#define DPW_LST(V, N, R) acc = __builtin_elementwise_min(acc, __builtin_elementwise_max(__builtin_elementwise_max(dp_bits(V), dp_bits(N) + t2), dp_bits(R) - t2))
                        DPW_LST(v4.x, n4.x, r4.x);
                        DPW_LST(v4.y, n4.y, r4.y);
                        DPW_LST(v4.z, n4.z, r4.z);
                        DPW_LST(v4.w, n4.w, r4.w);
#undef DPW_LST
                    }
                    uint32_t m = min((uint32_t)acc.x, (uint32_t)acc.y);
                    m = min(m, (uint32_t)__builtin_amdgcn_update_dpp((int)m, (int)m, 0xB1, 0xF, 0xF, false));
                    m = min(m, (uint32_t)__builtin_amdgcn_update_dpp((int)m, (int)m, 0x4E, 0xF, 0xF, false));
                    if (sl == 0) glb[tt] = min(glb[tt], m);
                }
                if (wv < 3 && !(probe & 8)) {
                    // elements 2 lane, 2 lane + 1 of table wv: prefix minima
                    const uint2 ev = reinterpret_cast<const uint2 *>(pfx + 128 * wv)[lane];
                    const uint32_t inc = dpw_wave_prefix_min(min(ev.x, ev.y));
                    const uint32_t exc = (uint32_t)__builtin_amdgcn_update_dpp((int)DPW_INF, (int)inc, 0x138, 0xF, 0xF, false);   // wave_shr:1
                    const uint32_t p0 = min(ev.x, exc), p1 = min(ev.y, p0);
                    reinterpret_cast<uint2 *>(spfx + 128 * wv)[lane] = make_uint2(p0, p1);
                }
            }
            if (!(probe & 128)) dpw_lds_barrier();
            // ---- phase 2: out[t] = min over the inner cut points c that reach t (and c = t) of G[c], with the matrix part; the
            // rings move on; the tables are made ready for the next block
            {
                uint32_t bits = (probe & 16) ? 0u : crb;
                if ((tt >> 5) == sl && !(probe & 32)) bits |= 1u << (tt & 31);
                uint32_t best = DPW_INF;
                while (bits) {
                    const uint32_t c = 32 * sl + (uint32_t)__builtin_ctz(bits);
                    bits &= bits - 1;
                    const uint32_t sp = sslope[c];
                    best = min(best, min(min(spfx[DPW_B - 1 - c], scsfx[c]), min(glb[c], sp >= DPW_INF ? DPW_INF : sp + c)));
                }
                best = min(best, (uint32_t)__builtin_amdgcn_update_dpp((int)best, (int)best, 0xB1, 0xF, 0xF, false));
                best = min(best, (uint32_t)__builtin_amdgcn_update_dpp((int)best, (int)best, 0x4E, 0xF, 0xF, false));
                if (sl == 0 && !(probe & 32)) {
                    const uint32_t v = min(min(best, fresh_y[tt]), DPW_INF);
                    const uint32_t j = jb + 1 + tt;
                    ringv[(j - 1) & MASK] = (uint16_t)v;
                    ringe[(j - 1) & MASK] = (uint16_t)cex;
                    if (tt == DPW_B - 1) lcount[1] = v;
                }
                if (sl == 1) { pfx[tt] = DPW_INF; slope[tt] = DPW_INF; csfx[tt] = DPW_INF; gl[128 * ((b + 1) & 1) + tt] = DPW_INF; }
                if (tid == 2) lcount[0] = 0;
            }
            if (!(probe & 128)) dpw_lds_barrier();
            L = lcount[1];
        }
    }
}

__global__ void k_dpw_bt(const uint32_t *__restrict__ mml, const uint16_t *__restrict__ ext16, uint32_t n, uint32_t window, uint32_t *__restrict__ bt,
                         unsigned long long *__restrict__ flag, uint32_t first_valid)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > n) return;
    if (j == 0) { bt[0] = 0; return; }
    if (j < first_valid) { bt[j] = DP_NONE; return; }
    const uint32_t L = mml[j];
    uint32_t best_age = 0;                    // youngest count_solutions-kind candidate (age <= its value == L)
    bool s_kind = false;                      // candidate of age L with a smaller value: the S / backtrack_S branch
    const uint32_t amax = min(min(window, L), j);   // both kinds have age <= L
    for (uint32_t a = 1; a <= amax; a++) {
        const uint32_t x = j - a;
        if (ext16[x] > a) continue;           // block [x, j) not valid yet: f[x]+1 > j
        const uint32_t v = mml[x];
        if (v > n) continue;                  // a column before first_valid: no solution ends there (fbg.cpp:1973)
        if (a > v) { if (a == L) s_kind = true; }
        else if (v == L && best_age == 0) best_age = a;
    }
    if (s_kind) bt[j] = j - L;
    else if (best_age) bt[j] = j - best_age;
    else { bt[j] = 0; flag[4] = 1; }
}

// ---- parallel backtrack (fbg.cpp:2026-2039) by binary lifting --------------------------------------
// backtrack[] is a forest towards column 0.  up[k][j] = the column reached from j after 2^k hops (0 absorbs),
// dep[j] = number of hops from j to 0.  The boundary list of the reference, [.., backtrack-1, .., n], has
// dep[n] entries; entry t is found independently by jumping dep[n]-1-t hops from n.
__global__ void k_bt_lift0(const uint32_t *__restrict__ bt, uint32_t n, uint32_t *__restrict__ up0,
                           uint32_t *__restrict__ dep)
{
    // index n+1 is a dead end that absorbs every pointer that is not a backward pointer (columns without a
    // value, fbg.cpp:2004-2009 with backtrack_S still -1, or the non-elastic "no valid range" marker)
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > n + 1) return;
    if (j == n + 1) { up0[j] = j; dep[j] = 0; return; }
    const uint32_t b = j == 0 ? 0u : bt[j];
    up0[j] = (b >= j && j != 0) ? n + 1 : b;
    dep[j] = j == 0 ? 0u : 1u;
}

__global__ void k_bt_lift(const uint32_t *__restrict__ up_prev, const uint32_t *__restrict__ dep_prev, uint32_t n,
                          uint32_t *__restrict__ up_next, uint32_t *__restrict__ dep_next)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > n + 1) return;
    const uint32_t a = up_prev[j];
    up_next[j] = up_prev[a];
    dep_next[j] = dep_prev[j] + dep_prev[a];      // hops counted so far: exact once the chain reaches 0
}

__global__ void k_bt_emit(const uint32_t *__restrict__ up, const uint32_t *__restrict__ dep_final, uint32_t n,
                          uint32_t levels, uint64_t *__restrict__ boundaries, unsigned long long *__restrict__ result)
{
    // the chain from n is valid iff it ends in column 0 and not in the dead end
    uint32_t e = n;
    for (uint32_t k = 0; k < levels; k++) e = up[(size_t)k * (n + 2) + e];
    const bool ok = e == 0;
    const uint32_t cnt = dep_final[n];
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) { result[0] = ok ? cnt : 0; result[1] = ok ? 0 : 1; }
    if (!ok || t >= cnt) return;
    uint32_t hops = cnt - 1 - t, j = n;
    for (uint32_t k = 0; k < levels && hops; k++, hops >>= 1)
        if (hops & 1u) j = up[(size_t)k * (n + 2) + j];
    boundaries[t] = t == cnt - 1 ? (uint64_t)n : (uint64_t)j - 1;
}

__global__ void k_reverse_widen(const uint32_t *__restrict__ rev, const unsigned long long *__restrict__ result,
                                uint64_t *__restrict__ boundaries)
{
    const unsigned long long cnt = result[0];
    const unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < cnt) boundaries[k] = rev[cnt - 1 - k];
}

// fbg.cpp:2026-2039.  status: 0 ok, 1 = backtrack left the array (reference: out-of-bounds read)
__global__ void k_dp_backtrack(const uint32_t *__restrict__ bt, uint32_t n, uint64_t *__restrict__ boundaries,
                               unsigned long long *__restrict__ result)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    uint64_t cnt = 1;
    uint32_t j = n;
    while (bt[j] != 0) {
        if (bt[j] > n || cnt > (uint64_t)n + 1) { result[0] = 0; result[1] = 1; return; }
        cnt++; j = bt[j];
    }
    uint64_t k = cnt - 1;
    j = n;
    boundaries[k--] = n;
    while (bt[j] != 0) { boundaries[k--] = (uint64_t)bt[j] - 1; j = bt[j]; }
    result[0] = cnt; result[1] = 0;
}

__global__ void k_widen(const uint32_t *__restrict__ src, uint64_t cnt, uint64_t *__restrict__ dst, int none_is_minus1)
{
    uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= cnt) return;
    uint32_t v = src[k];
    dst[k] = (none_is_minus1 && v == DP_NONE) ? ~0ull : (uint64_t)v;
}

static int fbg_backtrack_lifted(fbg_ctx *ctx, const uint32_t *bt, uint32_t n, uint64_t *d_boundaries,
                                unsigned long long *sc)
{
    hipStream_t st = ctx->stream;
    uint32_t levels = 1;
    while ((1ull << (levels - 1)) <= (uint64_t)n) levels++;   // the last table spans 2^(levels-1) > n hops
    const size_t stride = (size_t)n + 2;
    FBG_TRY(fbg_reserve(ctx, ctx->bt_up, (size_t)levels * stride * 4));
    FBG_TRY(fbg_reserve(ctx, ctx->bt_dep, 2 * stride * 4));
    uint32_t *up = ctx->bt_up.as<uint32_t>(), *dep0 = ctx->bt_dep.as<uint32_t>(), *dep1 = dep0 + stride;
    const unsigned grid = fbg_blocks(stride, 256);
    hipLaunchKernelGGL(k_bt_lift0, dim3(grid), dim3(256), 0, st, bt, n, up, dep0);
    uint32_t *dp = dep0, *dn = dep1;
    for (uint32_t k = 1; k < levels; k++) {
        hipLaunchKernelGGL(k_bt_lift, dim3(grid), dim3(256), 0, st, up + (size_t)(k - 1) * stride, dp, n,
                           up + (size_t)k * stride, dn);
        uint32_t *t = dp; dp = dn; dn = t;
    }
    hipLaunchKernelGGL(k_bt_emit, dim3(grid), dim3(256), 0, st, up, dp, n, levels, d_boundaries, sc);
    FBG_HIP_TRY(ctx, hipGetLastError());
    return FBG_OK;
}

int fbg_dp_minmax(fbg_ctx *ctx, const uint64_t *d_f, uint64_t n, uint64_t *d_boundaries, uint64_t *count_out,
                  uint64_t *d_mml, uint64_t *d_bt)
{
    hipStream_t st = ctx->stream;
    FBG_TRY(fbg_stage_begin(ctx, FBG_STAGE_DP));
    const size_t w = (n + 2) * 4;
    FBG_TRY(fbg_reserve(ctx, ctx->dp_a, w));   // e
    FBG_TRY(fbg_reserve(ctx, ctx->dp_b, w));   // hist -> bstart
    FBG_TRY(fbg_reserve(ctx, ctx->dp_c, w));   // cursor / thead
    FBG_TRY(fbg_reserve(ctx, ctx->dp_d, w));   // items
    FBG_TRY(fbg_reserve(ctx, ctx->dp_e, w));   // count
    FBG_TRY(fbg_reserve(ctx, ctx->dp_f, w));   // bcount
    FBG_TRY(fbg_reserve(ctx, ctx->dp_g, w));   // mml
    FBG_TRY(fbg_reserve(ctx, ctx->dp_h, w));   // bt
    FBG_TRY(fbg_reserve(ctx, ctx->list, w));   // tnext
    FBG_TRY(fbg_reserve(ctx, ctx->scalars, 256 * sizeof(unsigned long long)));
    uint32_t *e = ctx->dp_a.as<uint32_t>(), *bstart = ctx->dp_b.as<uint32_t>(), *cur = ctx->dp_c.as<uint32_t>(),
             *items = ctx->dp_d.as<uint32_t>(), *count = ctx->dp_e.as<uint32_t>(), *bcount = ctx->dp_f.as<uint32_t>(),
             *mml = ctx->dp_g.as<uint32_t>(), *bt = ctx->dp_h.as<uint32_t>(), *tnext = ctx->list.as<uint32_t>();
    unsigned long long *sc = ctx->scalars.as<unsigned long long>() + 16;
    FBG_HIP_TRY(ctx, hipMemsetAsync(sc, 0, 8 * sizeof(unsigned long long), st));
    hipLaunchKernelGGL(k_dp_keys, dim3(fbg_blocks(n, 256, 2048)), dim3(256), 0, st, d_f, n, e, sc + 2);
    // the bucket order of fbg.cpp:1941-1953 (counting sort of x by f[x]+1): only the wave-parallel and the literal sweep
    // read it -- built when one of them is about to run
    bool buckets_ready = false;
    auto build_buckets = [&]() -> int {
        if (buckets_ready) return FBG_OK;
        FBG_HIP_TRY(ctx, hipMemsetAsync(bstart, 0, w, st));
        hipLaunchKernelGGL(k_dp_hist, dim3(fbg_blocks(n, 256)), dim3(256), 0, st, e, n, bstart);
        size_t bytes = 0;
        hipError_t er = rocprim::exclusive_scan(nullptr, bytes, bstart, bstart, 0u, (size_t)(n + 2), rocprim::plus<uint32_t>(), st);
        if (er != hipSuccess) return fbg_fail(ctx, FBG_ERR_HIP, "rocprim scan size query failed");
        // ctx->tmp may hold the block matrices of an abandoned attempt: they are dead by now
        FBG_TRY(fbg_reserve(ctx, ctx->tmp, bytes));
        size_t have = ctx->tmp.cap;
        er = rocprim::exclusive_scan(ctx->tmp.p, have, bstart, bstart, 0u, (size_t)(n + 2), rocprim::plus<uint32_t>(), st);
        if (er != hipSuccess) return fbg_fail(ctx, FBG_ERR_HIP, "rocprim exclusive_scan: %s", hipGetErrorString(er));
        FBG_HIP_TRY(ctx, hipMemcpyAsync(cur, bstart, w, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(k_dp_scatter, dim3(fbg_blocks(n, 256)), dim3(256), 0, st, e, n, cur, items);
        FBG_HIP_TRY(ctx, hipMemsetAsync(count, 0, w, st));
        FBG_HIP_TRY(ctx, hipMemsetAsync(bcount, 0, w, st));
        FBG_HIP_TRY(ctx, hipMemsetAsync(mml, 0, w, st));
        FBG_HIP_TRY(ctx, hipMemsetAsync(bt, 0, w, st));
        FBG_HIP_TRY(ctx, hipMemsetAsync(cur, 0xff, w, st));   // thead = DP_NONE
        buckets_ready = true;
        return FBG_OK;
    };
    // which sweep: the wave-parallel one needs f[0] == 0 and a bounded block length
    unsigned long long hk[5];
    uint64_t f0 = 1;
    FBG_HIP_TRY(ctx, hipMemcpyAsync(hk, sc, sizeof(hk), hipMemcpyDeviceToHost, st));
    FBG_HIP_TRY(ctx, hipMemcpyAsync(&f0, d_f, sizeof(f0), hipMemcpyDeviceToHost, st));
    FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
    if (hk[2] != 0) return fbg_fail(ctx, FBG_ERR_INVALID, "f[] has %llu entries outside [x, n]", hk[2]);
    const unsigned long long max_ext = hk[3];
    int R = 0;
    // f[0] > 0 (only without the elastic tricks): nothing ends before step f[0] + 1, where the sweep stands exactly as
    // it does after step 1 otherwise (S = f[0] + 1, I = f[0], no counts: the invariant of the derivation above holds
    // from there on); the steps before it keep the running S of fbg.cpp:1967 and are filled in closed form.  The wave
    // sweep (k_dp_wave) starts from column 0 being a candidate at once and keeps its f[0] = 0 condition.
    const uint32_t first_valid = (uint32_t)std::min<uint64_t>(f0 + 1, n + 1);
    if (f0 < n && !ctx->opt.dp_literal) {
        const unsigned long long bound = 2 * max_ext + 2;   // minmaxlength[j] <= 2*max_ext + 1
        R = bound <= 64 ? 1 : bound <= 128 ? 2 : bound <= 254 ? 4 : (f0 == 0 && bound <= 512) ? 8 : (f0 == 0 && bound <= 1024) ? 16 : 0;
    }
    bool literal = R == 0;
    bool tiled = false;
    // extensions of hundreds of columns: the matrix chain with 16-bit entries (k_dpw_*), smallest window first
    const bool wide_ok = f0 < n && !ctx->opt.dp_literal && !ctx->opt.dp_wave && max_ext + 2 > 256 && max_ext + 2 <= 16384 && n >= 2 * DPW_B;
    auto try_wide = [&](bool &done) -> int {
        const uint32_t nblocks = (uint32_t)((n + DPW_B - 1) / DPW_B);
        uint16_t *ext16 = ctx->dp_e.as<uint16_t>();
        hipLaunchKernelGGL(k_dpw_prep, dim3(fbg_blocks(n + 1, 256)), dim3(256), 0, st, e, (uint32_t)n, ext16);
        uint32_t WS = 1024;
        while (WS < max_ext + 2) WS *= 2;
        if (!ctx->opt.dpw_matrix) {
            const uint32_t probe = 0;                   // bits that leave parts of k_dpw_chain2 out: timing only, wrong results (scripts/gpu_dpw_probe.py)
            // round 4: a 128 x 128 matrix and 128 x 128 bits per block, whatever the window (k_dpw_blockY), the older sources in
            // closed form (k_dpw_chain2); 272 bytes per column
            const size_t ybytes = (size_t)nblocks * DPW_B * DPW_B * 2, rbytes = (size_t)nblocks * DPW_B * 16;
            FBG_TRY(fbg_reserve(ctx, ctx->tmp, ybytes + rbytes));
            uint16_t *My = ctx->tmp.as<uint16_t>();
            unsigned long long *Rb = reinterpret_cast<unsigned long long *>(ctx->tmp.as<uint8_t>() + ybytes);
            hipLaunchKernelGGL(k_dpw_blockY, dim3(nblocks), dim3(256), 0, st, ext16, (uint32_t)n, My, Rb);
            for (; WS <= 16384 && !done; WS *= 2) {
                FBG_HIP_TRY(ctx, hipMemsetAsync(sc + 4, 0, sizeof(unsigned long long), st));
#define FBG_DPW2(W)                                                                                                                   \
    do {                                                                                                                              \
        const size_t lds = DPW_CHAIN2_LDS(W);                                                                                         \
        if (lds > 48 * 1024)                                                                                                          \
            FBG_HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_dpw_chain2<W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((k_dpw_chain2<W>), dim3(1), dim3(DPW_NT + 128), lds, st, My, Rb, ext16, (uint32_t)n, nblocks, mml, sc, first_valid, probe); \
    } while (0)
                switch (WS) {
                case 1024: FBG_DPW2(1024); break;
                case 2048: FBG_DPW2(2048); break;
                case 4096: FBG_DPW2(4096); break;
                case 8192: FBG_DPW2(8192); break;
                default: FBG_DPW2(16384); break;
                }
#undef FBG_DPW2
                hipLaunchKernelGGL(k_dpw_bt, dim3(fbg_blocks(n + 1, 256)), dim3(256), 0, st, mml, ext16, (uint32_t)n, WS, bt, sc, first_valid);
                FBG_HIP_TRY(ctx, hipMemcpyAsync(hk, sc, sizeof(hk), hipMemcpyDeviceToHost, st));
                FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
                done = hk[4] == 0;
                if (done) { int k = 3; for (uint32_t w2 = 1024; w2 < WS; w2 *= 2) k++; ctx->dp_kind = k; }
            }
            if (!done) FBG_HIP_TRY(ctx, hipMemsetAsync(sc + 4, 0, sizeof(unsigned long long), st));
            return FBG_OK;
        }
        for (; WS <= 16384 && !done; WS *= 2) {
            const size_t mbytes = (size_t)nblocks * DPW_B * WS * 2;
            if (mbytes > (64ull << 30)) break;
            if (ctx->tmp.cap < mbytes) {         // the matrices are an option, not a need: without room for them the literal sweep runs
                size_t free_b = 0, total_b = 0;
                FBG_HIP_TRY(ctx, hipMemGetInfo(&free_b, &total_b));
                if (free_b + ctx->tmp.cap < mbytes + (1ull << 30)) break;
            }
            FBG_TRY(fbg_reserve(ctx, ctx->tmp, mbytes));
            uint16_t *Mw = ctx->tmp.as<uint16_t>();
            FBG_HIP_TRY(ctx, hipMemsetAsync(sc + 4, 0, sizeof(unsigned long long), st));
#define FBG_DPW(W)                                                                                                                    \
    do {                                                                                                                              \
        hipLaunchKernelGGL((k_dpw_blockM<W>), dim3(nblocks, W / 256), dim3(256), 0, st, ext16, (uint32_t)n, Mw);                       \
        hipLaunchKernelGGL((k_dpw_chain<W>), dim3(1), dim3(1024), 0, st, Mw, (uint32_t)n, nblocks, mml, sc, first_valid);              \
    } while (0)
            switch (WS) {
            case 1024: FBG_DPW(1024); break;
            case 2048: FBG_DPW(2048); break;
            case 4096: FBG_DPW(4096); break;
            case 8192: FBG_DPW(8192); break;
            default: FBG_DPW(16384); break;
            }
#undef FBG_DPW
            hipLaunchKernelGGL(k_dpw_bt, dim3(fbg_blocks(n + 1, 256)), dim3(256), 0, st, mml, ext16, (uint32_t)n, WS, bt, sc, first_valid);
            FBG_HIP_TRY(ctx, hipMemcpyAsync(hk, sc, sizeof(hk), hipMemcpyDeviceToHost, st));
            FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
            done = hk[4] == 0;
            if (done) { int k = 3; for (uint32_t w2 = 1024; w2 < WS; w2 *= 2) k++; ctx->dp_kind = k; }     // 3: 1024, 4: 2048, ... 7: 16384
        }
        if (!done) FBG_HIP_TRY(ctx, hipMemsetAsync(sc + 4, 0, sizeof(unsigned long long), st));
        return FBG_OK;
    };
    if (literal && wide_ok) {
        FBG_TRY(try_wide(tiled));
        if (tiled) literal = false;
    }
    if (!literal && !tiled) {
        bool settled = false;
        if (!ctx->opt.dp_wave || f0 != 0) {
            // sweeps that work straight from f (no bucket order needed).  The window that is provably enough
            // (64 R >= 2 max_ext + 2) is rarely needed: start with the smallest one that holds every extension and
            // widen when the sweep reports a value at its limit
            uint8_t *ext7 = ctx->dp_e.as<uint8_t>(), *clen = ctx->dp_f.as<uint8_t>();
            // Exactness of a window: columns whose minimal extension does not fit it are no candidates inside it, and a
            // block longer than the window is not looked at -- both cost more than the window, so as long as every
            // minmaxlength found stays BELOW the window (the flag k_dp_expand raises otherwise) no such block could have
            // improved on it and the values are the reference's.  (Induction over j: the first column where the windowed
            // value exceeds the true one would need a true optimum through a dead candidate or a long block, i.e. a true
            // value above the window, yet the windowed value is below it and never smaller than the true one.)
            int Rt = 1;
            while (64ull * Rt < max_ext + 2 && Rt < 4) Rt *= 2;
            if (ctx->opt.dp_safe_window) Rt = R;
            for (; Rt <= 4 && Rt <= R && !settled; Rt *= 2) {
                const bool tile = Rt == 1 && ctx->opt.dp_tile && f0 == 0;
                FBG_HIP_TRY(ctx, hipMemsetAsync(sc + 4, 0, sizeof(unsigned long long), st));
                hipLaunchKernelGGL(k_dp_prep, dim3(fbg_blocks(n + 1, 256)), dim3(256), 0, st, e, (uint32_t)n,
                                   tile ? 127u : 255u, ext7, clen);
                if (tile) {
                    hipLaunchKernelGGL(k_dp_tile, dim3(1), dim3(64), 0, st, ext7, clen, (uint32_t)n, mml, bt, sc);
                } else {
                    const uint32_t WN = 64u * Rt;
                    const uint32_t nblocks = (uint32_t)((n + WN - 1) / WN);
                    FBG_TRY(fbg_reserve(ctx, ctx->tmp, (size_t)nblocks * WN * WN));
                    uint8_t *Wt = ctx->tmp.as<uint8_t>();
                    const size_t lds = (size_t)WN * DPB_PADDED(WN) + 2 * WN + 2 * WN * (WN / 64), lds2 = 2 * (size_t)WN * WN;
                    const unsigned grid = fbg_blocks(nblocks, 1, 256 * 16);
                    const uint32_t Fg = 16;                              // blocks per group of the chain (a chain step costs ~0.5 us)
                    const uint32_t ngroups = (nblocks + Fg - 1) / Fg;
                    uint8_t *Sg = reinterpret_cast<uint8_t *>(cur);      // (n+2)*4 bytes >= ngroups * WN: free on this path
#define FBG_DP_PIPE(RR)                                                                                                   \
    do {                                                                                                                 \
        if (lds > 64 * 1024)                                                                                             \
            FBG_HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_dp_blockW<RR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        if (lds2 > 64 * 1024)                                                                                            \
            FBG_HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_dp_compose<RR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2)); \
        hipLaunchKernelGGL((k_dp_blockW<RR>), dim3(grid), dim3(64 * RR), lds, st, ext7, (const uint8_t *)nullptr, (uint32_t)n, nblocks, Wt);             \
        hipLaunchKernelGGL((k_dp_compose<RR>), dim3(ngroups), dim3(256), lds2, st, Wt, nblocks, Fg);                     \
        if (ctx->opt.dp_chain1) hipLaunchKernelGGL((k_dp_chain<RR>), dim3(1), dim3(64), 0, st, Wt, (uint32_t)n, nblocks, Fg, Sg, mml, sc);   \
        else hipLaunchKernelGGL((k_dp_chain6<RR>), dim3(1), dim3(384), 0, st, Wt, (uint32_t)n, nblocks, Fg, Sg, mml);        \
        hipLaunchKernelGGL((k_dp_expand<RR>), dim3(grid), dim3(64), 0, st, Wt, Sg, (uint32_t)n, nblocks, Fg, mml, sc, first_valid);    \
    } while (0)
                    if (Rt == 1) FBG_DP_PIPE(1);
                    else if (Rt == 2) FBG_DP_PIPE(2);
                    else FBG_DP_PIPE(4);
#undef FBG_DP_PIPE
                    hipLaunchKernelGGL(k_dp_bt, dim3(fbg_blocks(n + 1, 256)), dim3(256), 0, st, mml, ext7, (uint32_t)n, WN, bt, sc, first_valid);
                }
                FBG_HIP_TRY(ctx, hipMemcpyAsync(hk, sc, sizeof(hk), hipMemcpyDeviceToHost, st));
                FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
                settled = hk[4] == 0;
            }
            tiled = settled;
            if (settled) ctx->dp_kind = 1;
            if (!settled) FBG_HIP_TRY(ctx, hipMemsetAsync(sc + 4, 0, sizeof(unsigned long long), st));
            if (!settled && wide_ok) {
                FBG_TRY(try_wide(settled));
                tiled = settled;
            }
        }
        if (!settled) FBG_TRY(build_buckets());        // also zeroes what the abandoned attempts left in count / bcount / mml / bt
        if (!settled && f0 == 0 && (R > 4 || ctx->opt.dp_wave)) {
            switch (R) {
            case 1: hipLaunchKernelGGL((k_dp_wave<1>), dim3(1), dim3(64), 0, st, bstart, items, (uint32_t)n, mml, bt, sc); break;
            case 2: hipLaunchKernelGGL((k_dp_wave<2>), dim3(1), dim3(64), 0, st, bstart, items, (uint32_t)n, mml, bt, sc); break;
            case 4: hipLaunchKernelGGL((k_dp_wave<4>), dim3(1), dim3(64), 0, st, bstart, items, (uint32_t)n, mml, bt, sc); break;
            case 8: hipLaunchKernelGGL((k_dp_wave<8>), dim3(1), dim3(64), 0, st, bstart, items, (uint32_t)n, mml, bt, sc); break;
            default: hipLaunchKernelGGL((k_dp_wave<16>), dim3(1), dim3(64), 0, st, bstart, items, (uint32_t)n, mml, bt, sc); break;
            }
            FBG_HIP_TRY(ctx, hipMemcpyAsync(hk, sc, sizeof(hk), hipMemcpyDeviceToHost, st));
            FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
            settled = hk[4] == 0;
            if (settled) ctx->dp_kind = 2;
        }
        if (!settled) literal = true;   // guards tripped: the literal sweep
    }
    if (literal) {
        ctx->dp_kind = 0;
        FBG_TRY(build_buckets());
        FBG_HIP_TRY(ctx, hipMemsetAsync(count, 0, w, st));
        FBG_HIP_TRY(ctx, hipMemsetAsync(bcount, 0, w, st));
        FBG_HIP_TRY(ctx, hipMemsetAsync(mml, 0, w, st));
        FBG_HIP_TRY(ctx, hipMemsetAsync(bt, 0, w, st));
        hipLaunchKernelGGL(k_dp_minmax, dim3(1), dim3(64), 0, st, bstart, items, (uint32_t)n, count, bcount, cur, tnext, mml, bt);
        hipLaunchKernelGGL(k_dp_backtrack, dim3(1), dim3(64), 0, st, bt, (uint32_t)n, d_boundaries, sc);
    } else if (tiled) {
        FBG_TRY(fbg_backtrack_lifted(ctx, bt, (uint32_t)n, d_boundaries, sc));
    } else {
        hipLaunchKernelGGL(k_dp_backtrack, dim3(1), dim3(64), 0, st, bt, (uint32_t)n, d_boundaries, sc);
    }
    if (d_mml) hipLaunchKernelGGL(k_widen, dim3(fbg_blocks(n + 1, 256)), dim3(256), 0, st, mml, n + 1, d_mml, 0);
    if (d_bt) hipLaunchKernelGGL(k_widen, dim3(fbg_blocks(n + 1, 256)), dim3(256), 0, st, bt, n + 1, d_bt, 1);
    FBG_HIP_TRY(ctx, hipGetLastError());
    FBG_TRY(fbg_stage_end(ctx, FBG_STAGE_DP, 6));
    unsigned long long h[3];
    FBG_HIP_TRY(ctx, hipMemcpyAsync(h, sc, sizeof(h), hipMemcpyDeviceToHost, st));
    FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
    *count_out = h[0];
    if (h[1] != 0) return fbg_fail(ctx, FBG_ERR_NO_SEGMENTATION, "No valid segmentation found!");
    return FBG_OK;
}

// ---- non-elastic recurrence (fbg.cpp:616-664), one lane ----------------------------------------

__global__ void k_dp_repeatfree(const uint64_t *__restrict__ v, uint32_t n, uint32_t *__restrict__ s,
                                uint32_t *__restrict__ prev, uint64_t *__restrict__ boundaries,
                                unsigned long long *__restrict__ result)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    for (uint32_t j = 0; j < n; j++) {
        uint32_t sj = j + 2, pj = j + 1;                                          // 623-624
        const uint64_t vj = v[j];
        if (vj <= j) {
            uint32_t jp = (uint32_t)vj;
            for (;;) {
                if (jp != 0 && s[jp - 1] == jp + 1) { jp--; continue; }          // 629-632
                const uint32_t a = jp == 0 ? 0u : s[jp - 1], b = j - jp + 1;
                const uint32_t cand = max(a, b);
                if (sj > cand) { sj = cand; pj = jp; }                            // 633-636
                if (sj == j - jp + 1) break;
                if (jp == 0) break;
                jp--;
            }
        }
        s[j] = sj; prev[j] = pj;
    }
    if (s[n - 1] == n + 1) { result[0] = 0; result[1] = 1; return; }              // 648-652
    uint64_t cnt = 1;
    uint32_t j = n - 1;
    while (prev[j] != 0) { cnt++; j = prev[j] - 1; }                              // 657-660
    uint64_t k = cnt - 1;
    j = n - 1;
    boundaries[k--] = j;
    while (prev[j] != 0) { boundaries[k--] = (uint64_t)prev[j] - 1; j = prev[j] - 1; }
    result[0] = cnt; result[1] = 0;
}

// ---- non-elastic recurrence on the block-matrix machinery ---------------------------------------------
// In prefix lengths (j' = j+1, x' = jp): s'[j'] = min over x' <= v[j'-1] of max(s'[x'], j' - x') for steps
// with a valid block (v[j'-1] <= j'-1), no value otherwise (fbg.cpp:622-643).  A candidate is valid at a
// step iff its age j'-x' is at least j' - v[j'-1]: a per-step threshold instead of a per-candidate one.
__global__ void k_ne_prep(const uint64_t *__restrict__ v, uint32_t n, uint8_t *__restrict__ ext1, uint8_t *__restrict__ amin,
                          unsigned long long *__restrict__ sc)
{
    const uint32_t jp = blockIdx.x * blockDim.x + threadIdx.x;   // prefix length 0..n
    if (jp > n) return;
    ext1[jp] = 1;
    uint32_t am = 255;
    if (jp >= 1) {
        const uint64_t vj = v[jp - 1];
        if (vj <= jp - 1) am = (uint32_t)min((uint64_t)255, (uint64_t)jp - vj);
    }
    amin[jp] = (uint8_t)am;
    // longest minimal block, to size the window
    unsigned long long e = am == 255 ? 0 : am;
    for (int d = 32; d >= 1; d >>= 1) e = max(e, (unsigned long long)__shfl_down(e, d, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(sc + 3, e);
}

// s[j], prev[j] (fbg.cpp:623-636) from the prefix values: first (largest jp) candidate reaching the minimum
__global__ void k_ne_finish(const uint32_t *__restrict__ sp, const uint8_t *__restrict__ amin, uint32_t n, uint32_t window,
                            uint32_t *__restrict__ s, uint32_t *__restrict__ prev, uint32_t *__restrict__ btp,
                            unsigned long long *__restrict__ flag)
{
    const uint32_t jp = blockIdx.x * blockDim.x + threadIdx.x;   // prefix length 1..n  <->  column j = jp-1
    if (jp == 0) { if (blockIdx.x == 0) btp[0] = 0; return; }
    if (jp > n) return;
    const uint32_t L = sp[jp], am = amin[jp], j = jp - 1;
    if (am == 255) { s[j] = j + 2; prev[j] = j + 1; btp[jp] = DP_NONE; return; }       // no valid block ends here
    if (L >= DPB_INF) {   // no value inside the window: undecidable here, let the literal kernel do this input
        flag[4] = 1; s[j] = j + 2; prev[j] = j + 1; btp[jp] = DP_NONE; return;
    }
    uint32_t best = 0;
    const uint32_t amax = min(window, jp);
    for (uint32_t a = am; a <= amax; a++) {
        const uint32_t v = sp[jp - a];
        if (v < DPB_INF && max(v, a) == L) { best = a; break; }
    }
    if (!best) { flag[4] = 1; best = am; }
    s[j] = L;
    prev[j] = jp - best;
    btp[jp] = jp - best;
}

__global__ void k_ne_fix_last(uint64_t *__restrict__ boundaries, const unsigned long long *__restrict__ result)
{
    // the shared backtrack emits n as last boundary (elastic convention); segment() ends at n-1 (fbg.cpp:655-656)
    if (blockIdx.x == 0 && threadIdx.x == 0 && result[0] > 0) boundaries[result[0] - 1] -= 1;
}

int fbg_dp_repeatfree(fbg_ctx *ctx, const uint64_t *d_v, uint64_t n, uint64_t *d_s, uint64_t *d_prev,
                      uint64_t *d_boundaries, uint64_t *count_out)
{
    hipStream_t st = ctx->stream;
    FBG_TRY(fbg_stage_begin(ctx, FBG_STAGE_DP));
    FBG_TRY(fbg_reserve(ctx, ctx->dp_g, (n + 2) * 4));
    FBG_TRY(fbg_reserve(ctx, ctx->dp_h, (n + 2) * 4));
    FBG_TRY(fbg_reserve(ctx, ctx->scalars, 256 * sizeof(unsigned long long)));
    uint32_t *s = ctx->dp_g.as<uint32_t>(), *prev = ctx->dp_h.as<uint32_t>();
    unsigned long long *sc = ctx->scalars.as<unsigned long long>() + 16;
    FBG_HIP_TRY(ctx, hipMemsetAsync(sc, 0, 16 * sizeof(unsigned long long), st));
    bool literal = ctx->opt.dp_literal != 0;
    if (!literal) {
        const size_t w = (n + 2) * 4;
        FBG_TRY(fbg_reserve(ctx, ctx->dp_a, w));
        FBG_TRY(fbg_reserve(ctx, ctx->dp_b, w));
        FBG_TRY(fbg_reserve(ctx, ctx->dp_c, w));
        FBG_TRY(fbg_reserve(ctx, ctx->dp_e, w));
        FBG_TRY(fbg_reserve(ctx, ctx->dp_f, w));
        FBG_TRY(fbg_reserve(ctx, ctx->list, w));
        uint8_t *ext1 = ctx->dp_e.as<uint8_t>(), *amin = ctx->dp_f.as<uint8_t>();
        uint32_t *sp = ctx->dp_a.as<uint32_t>(), *btp = ctx->dp_b.as<uint32_t>(), *rev = ctx->list.as<uint32_t>();
        uint8_t *Sg = ctx->dp_c.as<uint8_t>();
        hipLaunchKernelGGL(k_ne_prep, dim3(fbg_blocks(n + 1, 256)), dim3(256), 0, st, d_v, (uint32_t)n, ext1, amin, sc);
        unsigned long long hk[5];
        FBG_HIP_TRY(ctx, hipMemcpyAsync(hk, sc, sizeof(hk), hipMemcpyDeviceToHost, st));
        FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
        const unsigned long long bound = 2 * hk[3] + 2;
        const int R = bound <= 64 ? 1 : bound <= 128 ? 2 : bound <= 254 ? 4 : 0;
        if (R == 0) {
            literal = true;
        } else {
            const uint32_t WN = 64u * R;
            const uint32_t nblocks = (uint32_t)((n + WN - 1) / WN), Fg = 4, ngroups = (nblocks + Fg - 1) / Fg;
            FBG_TRY(fbg_reserve(ctx, ctx->tmp, (size_t)nblocks * WN * WN));
            uint8_t *Wt = ctx->tmp.as<uint8_t>();
            const size_t lds = (size_t)WN * DPB_PADDED(WN) + 2 * WN + 2 * WN * (WN / 64), lds2 = 2 * (size_t)WN * WN;
            const unsigned grid = fbg_blocks(nblocks, 1, 256 * 16);
#define FBG_NE_PIPE(RR)                                                                                                   \
    do {                                                                                                                 \
        if (lds > 64 * 1024)                                                                                             \
            FBG_HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_dp_blockW<RR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        if (lds2 > 64 * 1024)                                                                                            \
            FBG_HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_dp_compose<RR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2)); \
        hipLaunchKernelGGL((k_dp_blockW<RR>), dim3(grid), dim3(64 * RR), lds, st, ext1, amin, (uint32_t)n, nblocks, Wt);  \
        hipLaunchKernelGGL((k_dp_compose<RR>), dim3(ngroups), dim3(256), lds2, st, Wt, nblocks, Fg);                     \
        if (ctx->opt.dp_chain1) hipLaunchKernelGGL((k_dp_chain<RR>), dim3(1), dim3(64), 0, st, Wt, (uint32_t)n, nblocks, Fg, Sg, sp, sc);   \
        else hipLaunchKernelGGL((k_dp_chain6<RR>), dim3(1), dim3(384), 0, st, Wt, (uint32_t)n, nblocks, Fg, Sg, sp);        \
        hipLaunchKernelGGL((k_dp_expand<RR>), dim3(grid), dim3(64), 0, st, Wt, Sg, (uint32_t)n, nblocks, Fg, sp, sc + 8, 0u); \
    } while (0)
            if (R == 1) FBG_NE_PIPE(1);
            else if (R == 2) FBG_NE_PIPE(2);
            else FBG_NE_PIPE(4);
#undef FBG_NE_PIPE
            hipLaunchKernelGGL(k_ne_finish, dim3(fbg_blocks(n + 1, 256)), dim3(256), 0, st, sp, amin, (uint32_t)n, WN, s, prev,
                               btp, sc);
            FBG_HIP_TRY(ctx, hipMemcpyAsync(hk, sc, sizeof(hk), hipMemcpyDeviceToHost, st));
            FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
            if (hk[4] != 0) {
                literal = true;   // window too small for this input: statement-by-statement kernel
                FBG_HIP_TRY(ctx, hipMemsetAsync(sc, 0, 8 * sizeof(unsigned long long), st));
            } else {
                // no proper segmentation <=> the last prefix has no value (fbg.cpp:648-652)
                (void)rev;
                FBG_TRY(fbg_backtrack_lifted(ctx, btp, (uint32_t)n, d_boundaries, sc));
                hipLaunchKernelGGL(k_ne_fix_last, dim3(1), dim3(64), 0, st, d_boundaries, sc);
            }
        }
    }
    if (literal)
        hipLaunchKernelGGL(k_dp_repeatfree, dim3(1), dim3(64), 0, st, d_v, (uint32_t)n, s, prev, d_boundaries, sc);
    if (d_s) hipLaunchKernelGGL(k_widen, dim3(fbg_blocks(n, 256)), dim3(256), 0, st, s, n, d_s, 0);
    if (d_prev) hipLaunchKernelGGL(k_widen, dim3(fbg_blocks(n, 256)), dim3(256), 0, st, prev, n, d_prev, 0);
    FBG_HIP_TRY(ctx, hipGetLastError());
    FBG_TRY(fbg_stage_end(ctx, FBG_STAGE_DP, 3));
    unsigned long long h[2];
    FBG_HIP_TRY(ctx, hipMemcpyAsync(h, sc, sizeof(h), hipMemcpyDeviceToHost, st));
    FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
    *count_out = h[0];
    if (h[1] != 0) return fbg_fail(ctx, FBG_ERR_NO_SEGMENTATION, "No proper segmentation exists.");
    return FBG_OK;
}

// ---- non-elastic mode with gaps: segment2elasticValid (fbg.cpp:738-866) ---------------------------------------
// v[] from the f of the elastic scan without tricks: block [jp..j] passes the interval-union test of fbg.cpp:786-808
// iff F[jp] <= j (all rows have a symbol in it and every occurrence of every row's string starts at some row's first
// symbol at or after jp -- what compute_f evaluates with all rows coloured).  The reference's left-moving pointer
// (763-822) never binds, because the largest passing jp cannot decrease as j grows: v[j] = max{jp : F[jp] <= j}.

__global__ void k_gv_scatter(const uint64_t *__restrict__ f, uint32_t n, uint32_t *__restrict__ best)
{
    const uint32_t jp = blockIdx.x * blockDim.x + threadIdx.x;
    if (jp >= n) return;
    const uint64_t F = f[jp];
    if (F < n) atomicMax(best + F, jp + 1);          // 0 = no block ends here
}

__global__ void k_gv_finish(const uint32_t *__restrict__ best, uint32_t n, uint64_t *__restrict__ v)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t b = best[j];                      // running maximum by now
    v[j] = b ? (uint64_t)b - 1 : (uint64_t)j + 1;   // 764, 806
}

int fbg_gapped_v_from_f(fbg_ctx *ctx, const uint64_t *d_f, uint64_t n, uint64_t *d_v)
{
    hipStream_t st = ctx->stream;
    FBG_TRY(fbg_reserve(ctx, ctx->dp_a, (n + 2) * 4));
    uint32_t *best = ctx->dp_a.as<uint32_t>();
    FBG_HIP_TRY(ctx, hipMemsetAsync(best, 0, (n + 2) * 4, st));
    hipLaunchKernelGGL(k_gv_scatter, dim3(fbg_blocks(n, 256)), dim3(256), 0, st, d_f, (uint32_t)n, best);
    size_t bytes = 0;
    hipError_t er = rocprim::inclusive_scan(nullptr, bytes, best, best, (size_t)n, rocprim::maximum<uint32_t>(), st);
    if (er != hipSuccess) return fbg_fail(ctx, FBG_ERR_HIP, "rocprim scan size query failed");
    FBG_TRY(fbg_reserve(ctx, ctx->tmp, bytes));
    size_t have = ctx->tmp.cap;
    er = rocprim::inclusive_scan(ctx->tmp.p, have, best, best, (size_t)n, rocprim::maximum<uint32_t>(), st);
    if (er != hipSuccess) return fbg_fail(ctx, FBG_ERR_HIP, "rocprim inclusive_scan: %s", hipGetErrorString(er));
    hipLaunchKernelGGL(k_gv_finish, dim3(fbg_blocks(n, 256)), dim3(256), 0, st, best, (uint32_t)n, d_v);
    FBG_HIP_TRY(ctx, hipGetLastError());
    return FBG_OK;
}

// The recurrence of fbg.cpp:830-846 is a chain: column j takes either a new block [v[j]..j] on top of s[v[j]-1] or the
// solution of column j-1 with its last block extended, so one wave walks the columns -- but it walks from event to event,
// not from column to column.  Between two events (a new block taken at 838-841 or 834-837, or a column without a block,
// 832) the state is the last event's (a, P) with the last block growing: s[j] = max(a, j - P + 1), prev[j] = P.  A tile of
// 64 columns sits in the lanes; every lane evaluates its column under the assumption that nothing happens between the
// last event and itself (its lookback s[v-1] is then either a finished value or the same closed form), a ballot finds
// the first lane whose test fires -- for that lane the assumption is true -- and the walk jumps there.  Lookbacks in
// front of the tile come from a ring of the last GD_RING values of s in LDS (or, beyond it, from memory), fetched by
// all lanes at once before the walk.  INV = n+1 is the reference's initial value of both arrays; with prev[j-1] = n+1
// the reference's j - prev[j-1] + 1 wraps around to something larger than any score (838, 843), spelled GD_HUGE here.
#define GD_RING 8192u
#define GD_HUGE 0xffffffffu
#define GD_TILES 8u          // tiles per prefetch group

__global__ __launch_bounds__(64) void k_dp_gapped(const uint64_t *__restrict__ v, uint32_t n, uint32_t *__restrict__ s,
                                                 uint32_t *__restrict__ prev, uint32_t *__restrict__ btp)
{
    __shared__ uint32_t ring[GD_RING];
    const uint32_t lane = threadIdx.x, INV = n + 1;
    uint32_t sa = INV, sp = INV;                               // the last event: s and prev it left behind
    auto fetch = [&](uint32_t t0) -> uint32_t {                // v of column t0 + lane; INV = no block ends here (832)
        const uint32_t j = t0 + lane;
        if (j >= n || j == 0) return INV;                      // the loop starts at 1 (830)
        const uint64_t x = v[j];
        return x > j ? INV : (uint32_t)x;
    };
    if (lane == 0) btp[0] = 0;
    uint32_t cur[GD_TILES], nxt[GD_TILES];
#pragma unroll
    for (uint32_t q = 0; q < GD_TILES; q++) nxt[q] = fetch(q * 64);
    for (uint32_t g0 = 0; g0 < n; g0 += 64 * GD_TILES) {
        // the prefetched columns are settled here, once per group: left to the compiler, the wait for them lands in front
        // of every tile's walk, and a wait for loads is a wait for the previous tile's stores as well
#pragma unroll
        for (uint32_t q = 0; q < GD_TILES; q++) { cur[q] = nxt[q]; asm volatile("" : "+v"(cur[q])); }
        if (g0 + 64 * GD_TILES < n) {
#pragma unroll
            for (uint32_t q = 0; q < GD_TILES; q++) nxt[q] = fetch(g0 + 64 * GD_TILES + q * 64);
        }
#pragma unroll
        for (uint32_t q = 0; q < GD_TILES; q++) {
            const uint32_t t0 = g0 + q * 64;
            if (t0 >= n) break;
            const uint32_t vj = cur[q], j = t0 + lane;
            const bool skip = vj == INV, first = vj == 0;
            const uint32_t look = vj - 1;                      // column whose s the new block builds on
            // s[v-1] for lookbacks that land in front of this tile
            uint32_t aext = INV;
            bool far = false;
            if (!skip && !first && look < t0) {
                if (t0 - look <= GD_RING) aext = ring[look % GD_RING]; else far = true;
            }
            if (__ballot(far)) {
                __threadfence();
                if (far) aext = __hip_atomic_load(s + look, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0) here, in the rare branch, not in the walk below
            }
            const uint32_t len_new = j - vj + 1;
            uint32_t s_vec = INV, p_vec = INV;
            const uint32_t cnt = min(64u, n - t0);
            uint32_t k0 = 0;                                   // lanes below k0 are finished
            while (k0 < cnt) {
                // the state of column j-1 if nothing happens between the last event and j
                const uint32_t b = sp == INV ? GD_HUGE : max(sa, j - sp + 1);
                // every lane takes part in the exchange (a lane switched off would hand out nothing)
                const uint32_t fin = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((look - t0) & 63u) << 2), (int)s_vec);
                const uint32_t A = look < t0 ? aext : look - t0 < k0 ? fin : max(sa, look - sp + 1);
                const uint32_t a = max(A, len_new);
                const bool fire = skip || first || a < b;                                   // 832, 834, 838
                const unsigned long long mask = __ballot(fire && lane >= k0 && lane < cnt);
                const uint32_t ks = mask ? (uint32_t)__builtin_ctzll(mask) : cnt;
                if (lane >= k0 && lane < ks) { s_vec = b; p_vec = sp; }                     // 842-845
                if (ks >= cnt) break;
                const uint32_t ns = skip ? INV : first ? j + 1 : a;                         // 836, 839
                const uint32_t np = skip ? INV : vj;                                        // 837, 840
                if (lane == ks) { s_vec = ns; p_vec = np; }
                sa = (uint32_t)__builtin_amdgcn_readlane((int)ns, (int)ks);
                sp = (uint32_t)__builtin_amdgcn_readlane((int)np, (int)ks);
                k0 = ks + 1;
            }
            if (j < n) {
                s[j] = s_vec; prev[j] = p_vec;
                btp[j + 1] = p_vec;                // in prefix lengths: the block that ends prefix j+1 starts after prefix prev[j]
                ring[j % GD_RING] = s_vec;
            }
            __syncthreads();
        }
    }
}

// "No valid segmentation found!" is decided on s[n-1] (fbg.cpp:850-854)
__global__ void k_gd_verdict(const uint32_t *__restrict__ s, uint32_t n, uint64_t *__restrict__ boundaries,
                             unsigned long long *__restrict__ result)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    if (s[n - 1] == n + 1) { result[0] = 0; result[1] = 1; return; }
    if (result[0] > 0) boundaries[result[0] - 1] -= 1;          // the shared backtrack ends at n; this one at n-1 (857-858)
}

int fbg_dp_gapped(fbg_ctx *ctx, const uint64_t *d_v, uint64_t n, uint64_t *d_s, uint64_t *d_prev,
                  uint64_t *d_boundaries, uint64_t *count_out)
{
    hipStream_t st = ctx->stream;
    FBG_TRY(fbg_stage_begin(ctx, FBG_STAGE_DP));
    const size_t w = (n + 2) * 4;
    FBG_TRY(fbg_reserve(ctx, ctx->dp_g, w));
    FBG_TRY(fbg_reserve(ctx, ctx->dp_h, w));
    FBG_TRY(fbg_reserve(ctx, ctx->dp_b, w));
    FBG_TRY(fbg_reserve(ctx, ctx->scalars, 256 * sizeof(unsigned long long)));
    uint32_t *s = ctx->dp_g.as<uint32_t>(), *prev = ctx->dp_h.as<uint32_t>(), *btp = ctx->dp_b.as<uint32_t>();
    unsigned long long *sc = ctx->scalars.as<unsigned long long>() + 16;
    FBG_HIP_TRY(ctx, hipMemsetAsync(sc, 0, 16 * sizeof(unsigned long long), st));
    hipLaunchKernelGGL(k_dp_gapped, dim3(1), dim3(64), 0, st, d_v, (uint32_t)n, s, prev, btp);
    FBG_TRY(fbg_backtrack_lifted(ctx, btp, (uint32_t)n, d_boundaries, sc));
    hipLaunchKernelGGL(k_gd_verdict, dim3(1), dim3(64), 0, st, s, (uint32_t)n, d_boundaries, sc);
    if (d_s) hipLaunchKernelGGL(k_widen, dim3(fbg_blocks(n, 256)), dim3(256), 0, st, s, n, d_s, 0);
    if (d_prev) hipLaunchKernelGGL(k_widen, dim3(fbg_blocks(n, 256)), dim3(256), 0, st, prev, n, d_prev, 0);
    FBG_HIP_TRY(ctx, hipGetLastError());
    FBG_TRY(fbg_stage_end(ctx, FBG_STAGE_DP, 3));
    unsigned long long h[2];
    FBG_HIP_TRY(ctx, hipMemcpyAsync(h, sc, sizeof(h), hipMemcpyDeviceToHost, st));
    FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
    *count_out = h[1] ? 0 : h[0];
    if (h[1] != 0) return fbg_fail(ctx, FBG_ERR_NO_SEGMENTATION, "No valid segmentation found!");
    return FBG_OK;
}
