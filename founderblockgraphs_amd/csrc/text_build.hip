// text_build.hip -- MSA -> indexed text and per-row tables.
//
// Restates on the device what the reference does on the host before its scan:
//   * load_cst writes every row without '-' followed by '#' (fbg.cpp:372-386) and sdsl::construct
//     appends one 0 byte (fbg.cpp:428);
//   * segment_elastic_minmaxlength builds per-row rank/select over non-gap cells and over
//     ignore characters (fbg.cpp:1845-1917).
// Here those become plain arrays: pos[i], tot[i], the text pointer of every cell (prow, only for
// MSAs with gaps), the MSA column of every text position (colT, ditto) and the first ignore
// column at or after every cell (igrow, only with --ignore-chars).
// One workgroup sweeps one row in chunks with a running carry, so every load and store is a
// contiguous stream along the row.
#include "fbg_internal.h"

#define TB_THREADS 256
#define TB_ITEMS 8
#define TB_CHUNK (TB_THREADS * TB_ITEMS)

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

// exclusive scan of one value per thread over a 256-thread block; *total = block sum
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *total, uint32_t *lds4)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = wave_incl_scan(v);
    if (lane == 63) lds4[w] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < TB_THREADS / 64; k++) {
        uint32_t s = lds4[k];
        if (k < w) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// tot[i] += number of non-gap cells of segment blockIdx.x of row blockIdx.y; scalars[0] += gaps,
// scalars[1] += ignore cells; hist[c] += cells holding byte c (the symbol histogram of the text follows from it:
// fbg_build_text).  The histogram is kept per workgroup in LDS; equal bytes within a wave are counted with one
// ballot and one update (DNA: 4-5 rounds per 64 bytes instead of 64 colliding atomics).
#define RC_SEG 65536
#define RC_UNROLL 8
__global__ __launch_bounds__(TB_THREADS) void k_row_count(const uint8_t *__restrict__ msa, uint64_t n,
                                                          const uint8_t *__restrict__ is_ignore,
                                                          uint32_t *__restrict__ tot,
                                                          unsigned long long *__restrict__ scalars,
                                                          unsigned long long *__restrict__ hist,
                                                          uint32_t *__restrict__ segcnt, uint8_t *__restrict__ T_copy, uint32_t row0 = 0)
{
    // T_copy (optional; only when every row takes the 8-byte path: n and the MSA's address multiples of 8): the text of a
    // gap-free MSA -- row i at T + i * (n + 1), a '#' behind it -- is written along the way, from the very words the
    // counting loads (8-byte stores at addresses shifted by i bytes: the hardware takes them unaligned).  Should the MSA
    // turn out to have gaps the caller writes the text again, properly (k_write_text).
    typedef uint64_t u64_unaligned __attribute__((aligned(1)));
    __shared__ uint32_t red[2][TB_THREADS / 64];
    __shared__ uint32_t sh[256];
    const uint64_t i = (uint64_t)blockIdx.y + row0;           // (row0: the rows of one chunk of a streamed upload, fbg_build_text)
    const uint8_t *row = msa + i * n;
    const uint64_t x_lo = (uint64_t)blockIdx.x * RC_SEG, x_hi = min(n, x_lo + RC_SEG);
    const int lane = threadIdx.x & 63;
    sh[threadIdx.x] = 0;
    __syncthreads();
    uint32_t nongap = 0, ign = 0;
    auto count = [&](uint32_t c, bool valid) {                         // one byte per lane
        nongap += valid && c != '-';
        if (is_ignore && valid) ign += is_ignore[c];
        unsigned long long rest = __ballot(valid);
        while (rest) {
            const int l = __ffsll((long long)rest) - 1;
            const uint32_t v = __shfl(c, l, 64);
            const unsigned long long same = __ballot(valid && c == v);
            if (lane == l) atomicAdd(&sh[v], (uint32_t)__popcll(same));
            rest &= ~same;
        }
    };
    if ((((uintptr_t)row + x_lo) & 7) == 0 && !is_ignore) {
        // 8 bytes per lane and load.  The segment's alphabet is guessed from its first 64 bytes (up to RC_CAND symbols,
        // '-' always among them); the bytes of a word equal to a candidate are counted with a handful of word
        // operations per candidate into per-thread counters; only bytes outside the guess go through count()
        constexpr int RC_CAND = 6;
        uint32_t cand[RC_CAND];
        {
            const uint64_t x = x_lo + lane;
            uint32_t c = x < x_hi ? (uint32_t)row[x] : 0x100u;
            cand[0] = '-';
            unsigned long long rest = __ballot(c < 0x100u && c != '-');
#pragma unroll
            for (int k = 1; k < RC_CAND; k++) {
                cand[k] = 0x100u;                                       // no byte equals 0x100: an unused candidate
                if (rest) {
                    const int l = __ffsll((long long)rest) - 1;
                    cand[k] = __shfl(c, l, 64);
                    rest &= ~__ballot(c == cand[k]);
                }
            }
        }
        uint32_t pc[RC_CAND];
#pragma unroll
        for (int k = 0; k < RC_CAND; k++) pc[k] = 0;
        const uint64_t *words = reinterpret_cast<const uint64_t *>(row + x_lo);
        const uint64_t nbytes = x_hi > x_lo ? x_hi - x_lo : 0, nwords = nbytes / 8;
        for (uint64_t q = threadIdx.x; q < RC_SEG / 8; q += (uint64_t)TB_THREADS * RC_UNROLL) {     // uniform trip count
            uint64_t wv[RC_UNROLL];
#pragma unroll
            for (int u = 0; u < RC_UNROLL; u++) {
                const uint64_t qq = q + (uint64_t)u * TB_THREADS;
                wv[u] = qq < nwords ? words[qq] : 0ull;
            }
            if (T_copy) {
#pragma unroll
                for (int u = 0; u < RC_UNROLL; u++) {
                    const uint64_t qq = q + (uint64_t)u * TB_THREADS;
                    if (qq < nwords) *reinterpret_cast<u64_unaligned *>(T_copy + i * (n + 1) + x_lo + 8 * qq) = wv[u];
                }
            }
#pragma unroll
            for (int u = 0; u < RC_UNROLL; u++) {
                const bool valid = q + (uint64_t)u * TB_THREADS < nwords;
                uint64_t hit = 0;                                       // 0x80 in every byte that equals some candidate
#pragma unroll
                for (int k = 0; k < RC_CAND; k++) {
                    const uint64_t d = wv[u] ^ (0x0101010101010101ull * (cand[k] & 0xffu));
                    uint64_t z = ~(((d & 0x7f7f7f7f7f7f7f7full) + 0x7f7f7f7f7f7f7f7full) | d) & 0x8080808080808080ull;   // bytes of d that are 0
                    if (cand[k] > 0xffu || !valid) z = 0;
                    pc[k] += (uint32_t)__popcll(z);
                    hit |= z;
                }
                const bool odd = valid && hit != 0x8080808080808080ull;  // a byte outside the guess
                if (__ballot(odd)) {
#pragma unroll
                    for (int bb = 0; bb < 8; bb++)
                        count((uint32_t)(wv[u] >> (8 * bb)) & 0xffu, odd && !((hit >> (8 * bb + 7)) & 1ull));
                }
            }
        }
#pragma unroll
        for (int k = 0; k < RC_CAND; k++) {
            uint32_t v = pc[k];
            for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d, 64);
            if (lane == 0 && v && cand[k] <= 0xffu) atomicAdd(&sh[cand[k]], v);
            if (k > 0) nongap += lane == 0 ? v : 0u;                    // cand[0] is the gap symbol
        }
        const uint64_t tail = x_lo + nwords * 8 + threadIdx.x;         // up to seven bytes left (first wave only)
        if (threadIdx.x < 64) count(tail < x_hi ? (uint32_t)row[tail] : 0u, tail < x_hi);
    } else if ((((uintptr_t)row + x_lo) & 3) == 0) {
        // 4 bytes per lane and load (rows of MSAs whose width is a multiple of 4 all start aligned)
        const uint32_t *words = reinterpret_cast<const uint32_t *>(row + x_lo);
        const uint64_t nbytes = x_hi > x_lo ? x_hi - x_lo : 0, nwords = nbytes / 4;
        for (uint64_t q = threadIdx.x; q < RC_SEG / 4; q += (uint64_t)TB_THREADS * RC_UNROLL) {     // uniform trip count
            uint32_t wv[RC_UNROLL];
#pragma unroll
            for (int u = 0; u < RC_UNROLL; u++) {
                const uint64_t qq = q + (uint64_t)u * TB_THREADS;
                wv[u] = qq < nwords ? words[qq] : 0u;
            }
#pragma unroll
            for (int u = 0; u < RC_UNROLL; u++) {
                const bool valid = q + (uint64_t)u * TB_THREADS < nwords;
#pragma unroll
                for (int bb = 0; bb < 4; bb++) count((wv[u] >> (8 * bb)) & 0xffu, valid);
            }
        }
        const uint64_t tail = x_lo + nwords * 4 + threadIdx.x;         // up to three bytes left (first wave only)
        if (threadIdx.x < 64) count(tail < x_hi ? (uint32_t)row[tail] : 0u, tail < x_hi);
    } else {
        for (uint64_t x = x_lo + threadIdx.x; x < x_lo + RC_SEG; x += (uint64_t)TB_THREADS * RC_UNROLL) {   // uniform trip count
            uint32_t c[RC_UNROLL];
#pragma unroll
            for (int u = 0; u < RC_UNROLL; u++) {
                const uint64_t xx = x + (uint64_t)u * TB_THREADS;
                c[u] = xx < x_hi ? (uint32_t)row[xx] : 0x100u;         // 0x100: past the end
            }
#pragma unroll
            for (int u = 0; u < RC_UNROLL; u++) count(c[u] & 0xffu, c[u] < 0x100u);
        }
    }
    for (int d = 32; d >= 1; d >>= 1) {
        nongap += __shfl_down(nongap, d, 64);
        ign += __shfl_down(ign, d, 64);
    }
    if (lane == 0) { red[0][threadIdx.x >> 6] = nongap; red[1][threadIdx.x >> 6] = ign; }
    __syncthreads();
    if (sh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)sh[threadIdx.x]);
    if (T_copy && threadIdx.x == 1 && x_hi == n) T_copy[i * (n + 1) + n] = '#';
    if (threadIdx.x == 0) {
        uint32_t a = 0, b = 0;
        for (int k = 0; k < TB_THREADS / 64; k++) { a += red[0][k]; b += red[1][k]; }
        atomicAdd(&tot[i], a);
        segcnt[i * gridDim.x + blockIdx.x] = a;                         // non-gap cells of this segment: k_seg_offsets
        atomicAdd(&scalars[0], (unsigned long long)((x_hi > x_lo ? x_hi - x_lo : 0) - a));
        if (b) atomicAdd(&scalars[1], (unsigned long long)b);
    }
}

// pos[i] = sum_{k<i} (tot[k] + 1); scalars[2] = N = sum + m + 1
__global__ void k_row_offsets(const uint32_t *__restrict__ tot, uint64_t m, uint32_t *__restrict__ pos,
                              unsigned long long *__restrict__ scalars)
{
    // one wave: every lane sums a stretch of rows, the stretches are scanned across the lanes
    if (blockIdx.x != 0 || threadIdx.x >= 64) return;
    const uint64_t per = (m + 63) / 64, lo = min(m, (uint64_t)threadIdx.x * per), hi = min(m, lo + per);
    unsigned long long mine = 0;
    for (uint64_t i = lo; i < hi; i++) mine += (unsigned long long)tot[i] + 1;
    unsigned long long inc = mine;
    for (int d = 1; d < 64; d <<= 1) { const unsigned long long o = __shfl_up(inc, d, 64); if ((int)threadIdx.x >= d) inc += o; }
    unsigned long long s = inc - mine;
    for (uint64_t i = lo; i < hi; i++) { pos[i] = (uint32_t)s; s += (unsigned long long)tot[i] + 1; }
    if (threadIdx.x == 63) scalars[2] = inc + 1;
}

// segoff[i][s] = non-gap cells of row i before segment s (rows with gaps are written segment by segment)
__global__ void k_seg_offsets(const uint32_t *__restrict__ segcnt, uint64_t m, uint32_t nseg, uint32_t *__restrict__ segoff)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    uint32_t run = 0;
    for (uint32_t q = 0; q < nseg; q++) { segoff[i * nseg + q] = run; run += segcnt[i * nseg + q]; }
}

template <bool GAPPED, bool REVERSED>
__global__ __launch_bounds__(TB_THREADS) void k_write_text(const uint8_t *__restrict__ msa, uint64_t n,
                                                           const uint32_t *__restrict__ pos,
                                                           const uint32_t *__restrict__ tot,
                                                           uint8_t *__restrict__ T, uint32_t *__restrict__ prow,
                                                           uint32_t *__restrict__ colT, const uint32_t *__restrict__ segoff,
                                                           uint16_t *__restrict__ winrow)
{
    // blockIdx.y = row; blockIdx.x = segment of RC_SEG columns (gapped rows: the segment's first text position comes
    // from segoff) or the whole row (gridDim.x = 1, segoff = nullptr)
    __shared__ uint32_t lds4[TB_THREADS / 64];
    // rows with gaps (round 4): the chunk's symbols and their columns are packed in LDS first and go out from there, consecutive
    // threads to consecutive text positions -- written straight from the threads that read them, 8 cells each, every store
    // instruction of a wave touched 64 places 8 positions apart (text and colT: 1.68 ms at C5 for 3 GB)
    __shared__ uint32_t s_col[GAPPED ? TB_CHUNK : 1];
    __shared__ uint8_t s_sym[GAPPED ? TB_CHUNK : 4];
    const uint64_t i = blockIdx.y;
    const uint8_t *row = msa + i * n;
    const uint32_t p0 = pos[i];
    const uint64_t x_lo = segoff ? (uint64_t)blockIdx.x * RC_SEG : 0, x_hi = segoff ? min(n, x_lo + RC_SEG) : n;
    uint32_t carry = segoff ? segoff[i * gridDim.x + blockIdx.x] : 0u;
    const bool aligned8 = (reinterpret_cast<uintptr_t>(row + x_lo) & 7) == 0;     // the thread's 8 cells are one aligned word
    for (uint64_t base = x_lo; base < x_hi; base += TB_CHUNK) {
        const uint64_t x0 = base + (uint64_t)threadIdx.x * TB_ITEMS;
        uint8_t c[TB_ITEMS];
        uint32_t cnt = 0;
        if (TB_ITEMS == 8 && aligned8 && x0 + 8 <= x_hi) {
            const uint64_t w8 = *reinterpret_cast<const uint64_t *>(row + x0);
#pragma unroll
            for (int k = 0; k < TB_ITEMS; k++) { c[k] = (uint8_t)(w8 >> (8 * k)); cnt += c[k] != '-'; }
        } else {
#pragma unroll
            for (int k = 0; k < TB_ITEMS; k++) {
                c[k] = x0 + k < x_hi ? row[x0 + k] : (uint8_t)'-';
                cnt += c[k] != '-';
            }
        }
        if (GAPPED) {
            uint32_t total;
            const uint32_t off0 = block_excl_scan(cnt, &total, lds4);          // (two barriers inside: the LDS of the chunk before is read)
            uint32_t off = off0;
#pragma unroll
            for (int k = 0; k < TB_ITEMS; k++) {
                if (x0 + k < x_hi) {
                    if (prow) prow[i * n + x0 + k] = p0 + carry + off;      // pos_i + rank_i(x)
                    if (c[k] != '-') { s_sym[off] = c[k]; s_col[off] = (uint32_t)(x0 + k); off++; }
                }
            }
            __syncthreads();
            const uint32_t q0 = p0 + carry;
            for (uint32_t j = threadIdx.x; j < total; j += TB_THREADS) {
                const uint32_t q = q0 + j;
                T[q] = s_sym[j];
                colT[q] = s_col[j];
                if (winrow && (q & 127u) == 0) winrow[q >> 7] = (uint16_t)i;   // the row of every window of 128 positions (gapped_rank.hip)
            }
            carry += total;
        } else {
#pragma unroll
            for (int k = 0; k < TB_ITEMS; k++)
                if (x0 + k < n) T[p0 + (REVERSED ? (n - 1 - (x0 + k)) : (x0 + k))] = c[k];
        }
    }
    if (threadIdx.x == 0 && x_hi == n) {
        T[p0 + tot[i]] = '#';
        if (GAPPED) colT[p0 + tot[i]] = (uint32_t)n;
        if (GAPPED && i + 1 == gridDim.y) colT[p0 + tot[i] + 1] = (uint32_t)n;   // the sentinel behind the last row has no column either
    }
}

// Gap-free rows, left to right: the text is the MSA with a '#' after every row.  Row i moves from msa + i*n to
// T + i*(n+1): source and destination are aligned differently in every row, so each destination word is put
// together from the two aligned source words that hold its bytes (v_alignbyte); all global accesses are 4 bytes
// wide and coalesced.  blockIdx.y = row, blockIdx.x = segment of CR_SEG destination bytes.
#define CR_SEG 32768
__global__ __launch_bounds__(TB_THREADS) void k_copy_rows(const uint8_t *__restrict__ msa, uint64_t n, uint64_t m,
                                                          uint8_t *__restrict__ T)
{
    const uint64_t i = blockIdx.y;
    const uint8_t *src = msa + i * n;
    uint8_t *dst = T + i * (n + 1);
    const uint64_t x_lo = (uint64_t)blockIdx.x * CR_SEG, x_hi = min(n, x_lo + CR_SEG);
    if (x_lo >= x_hi) return;
    // destination words fully inside [x_lo, x_hi): from the first aligned address on
    const uint64_t head = (4 - ((uintptr_t)(dst + x_lo) & 3)) & 3;      // bytes before the first aligned word
    const uint64_t xa = min(x_hi, x_lo + head);
    const uint64_t nwords = (x_hi - xa) / 4;
    const uint64_t xt = xa + nwords * 4;                               // tail bytes [xt, x_hi)
    if (threadIdx.x < 8) {
        const uint64_t x = threadIdx.x < 4 ? x_lo + threadIdx.x : xt + (threadIdx.x - 4);
        const uint64_t end = threadIdx.x < 4 ? xa : x_hi;
        if (x < end) dst[x] = src[x];
    }
    if (threadIdx.x == 8 && x_hi == n) dst[n] = '#';
    const uintptr_t sa = (uintptr_t)(src + xa);
    const uint32_t *sw = reinterpret_cast<const uint32_t *>(sa & ~(uintptr_t)3);
    const uint32_t shift = (uint32_t)(sa & 3);
    const uint32_t *last = reinterpret_cast<const uint32_t *>(((uintptr_t)(msa + m * n) - 1) & ~(uintptr_t)3);   // last readable word
    uint32_t *dw = reinterpret_cast<uint32_t *>(dst + xa);
    for (uint64_t j = threadIdx.x; j < nwords; j += TB_THREADS) {
        const uint32_t lo = sw[j];
        const uint32_t hi = (shift && sw + j + 1 <= last) ? sw[j + 1] : 0u;
        dw[j] = __builtin_amdgcn_alignbyte(hi, lo, shift);
    }
}

// first ignore column of every segment of RC_SEG columns (n: none): the carries of k_ignore_sweep
__global__ __launch_bounds__(TB_THREADS) void k_ignore_segmin(const uint8_t *__restrict__ msa, uint64_t n,
                                                              const uint8_t *__restrict__ is_ignore, uint32_t *__restrict__ segmin)
{
    __shared__ uint32_t wmin[TB_THREADS / 64];
    const uint64_t i = blockIdx.y;
    const uint8_t *row = msa + i * n;
    const uint64_t x_lo = (uint64_t)blockIdx.x * RC_SEG, x_hi = min(n, x_lo + RC_SEG);
    uint32_t first = (uint32_t)n;
    for (uint64_t x = x_lo + threadIdx.x; x < x_hi; x += TB_THREADS)
        if (is_ignore[row[x]]) { first = (uint32_t)x; break; }          // ascending per thread: its first hit is its smallest
    for (int d = 32; d >= 1; d >>= 1) first = min(first, (uint32_t)__shfl_down(first, d, 64));
    if ((threadIdx.x & 63) == 0) wmin[threadIdx.x >> 6] = first;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < TB_THREADS / 64; k++) first = min(first, wmin[k]);
        segmin[i * gridDim.x + blockIdx.x] = first;
    }
}

// igrow[i*n + x] = smallest column c >= x with is_ignore[msa[i][c]], or n (fbg.cpp:1669-1670).  blockIdx.y = row,
// blockIdx.x = segment of RC_SEG columns, swept right to left; what lies to the right of the segment comes from segmin
__global__ __launch_bounds__(TB_THREADS) void k_ignore_sweep(const uint8_t *__restrict__ msa, uint64_t n,
                                                             const uint8_t *__restrict__ is_ignore,
                                                             const uint32_t *__restrict__ segmin,
                                                             uint32_t *__restrict__ igrow)
{
    __shared__ uint32_t wmin[TB_THREADS / 64];
    const uint64_t i = blockIdx.y;
    const uint8_t *row = msa + i * n;
    const uint64_t x_lo = (uint64_t)blockIdx.x * RC_SEG, x_hi = min(n, x_lo + RC_SEG);
    uint32_t carry = (uint32_t)n;  // first ignore column to the right of the current chunk
    for (uint32_t q = blockIdx.x + 1; q < gridDim.x; q++) {
        const uint32_t v = segmin[i * gridDim.x + q];
        if (v < (uint32_t)n) { carry = v; break; }
    }
    const uint64_t nchunks = (x_hi - x_lo + TB_CHUNK - 1) / TB_CHUNK;
    for (uint64_t ch = nchunks; ch-- > 0;) {
        const uint64_t x0 = x_lo + ch * TB_CHUNK + (uint64_t)threadIdx.x * TB_ITEMS;
        uint32_t loc[TB_ITEMS];
        uint32_t first = 0xffffffffu;  // first ignore column among this thread's items
#pragma unroll
        for (int k = TB_ITEMS - 1; k >= 0; k--) {
            if (x0 + k < x_hi && is_ignore[row[x0 + k]]) first = (uint32_t)(x0 + k);
            loc[k] = first;
        }
        // suffix-min across threads: value for thread t = min(first of threads > t)
        uint32_t v = first;
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t o = __shfl_down(v, d, 64);
            if (lane + d < 64) v = min(v, o);
        }
        if (lane == 0) wmin[w] = v;
        __syncthreads();
        uint32_t after = carry;  // min over later waves
        for (int k = TB_THREADS / 64 - 1; k > w; k--) after = min(after, wmin[k]);
        uint32_t chunk_min = carry;
        for (int k = 0; k < TB_THREADS / 64; k++) chunk_min = min(chunk_min, wmin[k]);
        // exclusive suffix-min inside the wave
        uint32_t ex = __shfl_down(v, 1, 64);
        if (lane == 63) ex = 0xffffffffu;
        uint32_t right = min(ex, after);
#pragma unroll
        for (int k = 0; k < TB_ITEMS; k++)
            if (x0 + k < x_hi) igrow[i * n + x0 + k] = min(loc[k], right);
        __syncthreads();
        carry = chunk_min;
    }
}

int fbg_build_text(fbg_ctx *ctx, const uint8_t *ignore, uint64_t ignore_len)
{
    const uint64_t m = ctx->m, n = ctx->n;
    hipStream_t st = ctx->stream;
    FBG_TRY(fbg_stage_begin(ctx, FBG_STAGE_TEXT));
    int launches = 0;
    FBG_TRY(fbg_reserve(ctx, ctx->pos, m * 4));
    FBG_TRY(fbg_reserve(ctx, ctx->tot, m * 4));
    FBG_TRY(fbg_reserve(ctx, ctx->scalars, 256 * sizeof(unsigned long long)));
    FBG_TRY(fbg_reserve(ctx, ctx->small, 8192));   // [0,256) ignore table, [2048,2304) code table, [4096,6144) symbol histogram
    FBG_HIP_TRY(ctx, hipMemsetAsync(ctx->scalars.p, 0, 64 * sizeof(unsigned long long), st));

    ctx->have_ignore = ignore_len > 0;
    ctx->cells_built = false;
    memset(ctx->ignore_tab, 0, sizeof(ctx->ignore_tab));
    if (ctx->have_ignore) {
        for (uint64_t k = 0; k < ignore_len; k++) ctx->ignore_tab[ignore[k]] = 1;   // fbg.cpp:1853,1868
        FBG_HIP_TRY(ctx, hipMemcpyAsync(ctx->small.as<uint8_t>(), ctx->ignore_tab, 256, hipMemcpyHostToDevice, st));
    }
    unsigned long long *sc = ctx->scalars.as<unsigned long long>();
    unsigned long long *d_hist = reinterpret_cast<unsigned long long *>(ctx->small.as<uint8_t>() + 4096);
    FBG_HIP_TRY(ctx, hipMemsetAsync(d_hist, 0, 2048, st));
    FBG_HIP_TRY(ctx, hipMemsetAsync(ctx->tot.p, 0, m * 4, st));
    const unsigned nseg = (unsigned)((n + RC_SEG - 1) / RC_SEG);
    FBG_TRY(fbg_reserve(ctx, ctx->segtab, (size_t)m * nseg * 4 * 3));   // per (row, segment): non-gap cells, their prefix, first ignore column
    uint32_t *segcnt = ctx->segtab.as<uint32_t>(), *segoff = segcnt + (size_t)m * nseg;
    // (the kernel can count the ignore cells too; nobody asks for that number, and without it rows take the word-wide path)
    // Optimistic text: most MSAs that come this way have no gaps, and then the text is the MSA with a '#' per row --
    // written by the counting pass itself from the words it loads (one read of the MSA instead of two)
    const bool fused = !ctx->reversed && n % 8 == 0 && ((uintptr_t)ctx->d_msa & 7) == 0 && m * (n + 1) + 1 < (1ull << 40);
    if (fused) FBG_TRY(fbg_reserve(ctx, ctx->text, m * (n + 1) + 1 + 64));
    const uint8_t *up = ctx->up_host;                           // fbg_elastic_f: the MSA is still in (pinned) host memory
    ctx->up_host = nullptr;
    ctx->pre_pass1 = false;
    if (up && fused && m >= 16) {
        // Streamed upload: the rows arrive in chunks on a copy stream of their own, and what needs nothing but the rows that are
        // there runs behind every chunk -- the counting pass that also writes the text, and (speculatively: the alphabet and
        // "no gaps" are taken from the first chunk and checked at the end, fbg_msd_pre_*) pass 1 of the MSD sort over the text
        // positions that are complete.  The 1 GB upload of BASELINE config 3 takes 18 ms; 4 ms of the build hide behind it.
        constexpr int CHUNKS = 8;
        if (!ctx->up_stream) {
            FBG_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->up_stream, hipStreamNonBlocking));
            for (auto &e : ctx->up_ev) FBG_HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        uint8_t *d_msa = ctx->msa_own.as<uint8_t>();
        FBG_HIP_TRY(ctx, hipEventRecord(ctx->up_ev[CHUNKS], st));                 // (the copies start after what the stream holds: memsets above)
        FBG_HIP_TRY(ctx, hipStreamWaitEvent(ctx->up_stream, ctx->up_ev[CHUNKS], 0));
        uint64_t r0s[CHUNKS + 1];
        for (int c = 0; c <= CHUNKS; c++) r0s[c] = m * c / CHUNKS;
        for (int c = 0; c < CHUNKS; c++) {
            FBG_HIP_TRY(ctx, hipMemcpyAsync(d_msa + r0s[c] * n, up + r0s[c] * n, (r0s[c + 1] - r0s[c]) * n, hipMemcpyHostToDevice, ctx->up_stream));
            FBG_HIP_TRY(ctx, hipEventRecord(ctx->up_ev[c], ctx->up_stream));
        }
        bool pre = false;
        for (int c = 0; c < CHUNKS; c++) {
            FBG_HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->up_ev[c], 0));
            hipLaunchKernelGGL(k_row_count, dim3(nseg, (unsigned)(r0s[c + 1] - r0s[c])), dim3(TB_THREADS), 0, st, ctx->d_msa,
                               n, (const uint8_t *)nullptr, ctx->tot.as<uint32_t>(), sc, d_hist, segcnt, ctx->text.as<uint8_t>(), (uint32_t)r0s[c]);
            if (c == 0) {
                unsigned long long h0[1];
                FBG_HIP_TRY(ctx, hipMemcpyAsync(h0, sc, sizeof(h0), hipMemcpyDeviceToHost, st));
                FBG_HIP_TRY(ctx, hipMemcpyAsync(ctx->byte_hist, d_hist, 256 * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
                FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
                if (h0[0] == 0) {
                    // the whole MSA as the first chunk promises it: no gaps, these symbols in these proportions
                    for (int b = 0; b < 256; b++) ctx->byte_hist[b] = ctx->byte_hist[b] * m / r0s[1];
                    ctx->byte_hist['-'] = 0; ctx->byte_hist['#'] += m; ctx->byte_hist[0] += 1;
                    ctx->N = m * (n + 1) + 1;
                    ctx->gapfree = true;
                    FBG_HIP_TRY(ctx, hipMemsetAsync(ctx->text.as<uint8_t>() + ctx->N - 1, 0, 65, st));
                    int ok = 0;
                    FBG_TRY(fbg_msd_pre_begin(ctx, &ok));
                    pre = ok != 0;
                }
            }
            if (pre) FBG_TRY(fbg_msd_pre_pass1(ctx, c + 1 < CHUNKS ? r0s[c + 1] * (n + 1) : ctx->N));
        }
        ctx->pre_pass1 = pre;
        launches += CHUNKS - 1;
    } else {
        if (up) FBG_TRY(fbg_upload(ctx, ctx->msa_own.p, up, m * n));
        hipLaunchKernelGGL(k_row_count, dim3(nseg, (unsigned)m), dim3(TB_THREADS), 0, st, ctx->d_msa,
                           n, (const uint8_t *)nullptr, ctx->tot.as<uint32_t>(), sc, d_hist, segcnt, fused ? ctx->text.as<uint8_t>() : (uint8_t *)nullptr);
    }
    hipLaunchKernelGGL(k_row_offsets, dim3(1), dim3(64), 0, st, ctx->tot.as<uint32_t>(), m,
                       ctx->pos.as<uint32_t>(), sc);
    launches += 2;
    unsigned long long h[3];
    FBG_HIP_TRY(ctx, hipMemcpyAsync(h, sc, sizeof(h), hipMemcpyDeviceToHost, st));
    FBG_HIP_TRY(ctx, hipMemcpyAsync(ctx->byte_hist, d_hist, 256 * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    FBG_HIP_TRY(ctx, hipStreamSynchronize(st));
    // symbol histogram of the text: the cells without the gaps, a '#' per row, the sentinel
    ctx->byte_hist['-'] = 0;
    ctx->byte_hist['#'] += m;
    ctx->byte_hist[0] += 1;
    ctx->gapfree = h[0] == 0;
    ctx->N = h[2];
    if (ctx->pre_pass1) {
        // what the first chunk promised: no gaps, and exactly the symbols the keys were set up for
        bool same = ctx->gapfree && ctx->N == m * (n + 1) + 1;
        for (int b = 0; b < 256 && same; b++) same = (ctx->byte_hist[b] != 0) == ((ctx->pre_symbols[b >> 6] >> (b & 63)) & 1ull);
        ctx->pre_pass1 = same;
    }
    if (ctx->N >= (1ull << 32) && !(ctx->allow_wide && ctx->gapfree && !ctx->reversed && !ctx->have_ignore && ctx->N < (1ull << 40)))
        return fbg_fail(ctx, FBG_ERR_TOO_LARGE, "text length %llu needs >32-bit positions (only the partitioned index of a "
                        "gap-free MSA, elastic scan, goes beyond)", h[2]);
    if (ctx->reversed && !ctx->gapfree)
        return fbg_fail(ctx, FBG_ERR_INVALID,
                        "the non-elastic scan needs gap-free rows (the reference drops rows with gaps when "
                        "--gap-limit=1, fbg.cpp:176-177); this MSA has %llu gap cells", h[0]);
    ctx->mp = (uint32_t)((m + 63) & ~63ull);

    const size_t tbytes = ctx->N + 64;
    FBG_TRY(fbg_reserve(ctx, ctx->text, tbytes));
    // sentinel (the 0 byte sdsl::construct appends) + zero padding for 8-byte compares
    FBG_HIP_TRY(ctx, hipMemsetAsync(ctx->text.as<uint8_t>() + ctx->N - 1, 0, 65, st));
    uint8_t *T = ctx->text.as<uint8_t>();
    const uint32_t *pos = ctx->pos.as<uint32_t>(), *tot = ctx->tot.as<uint32_t>();
    if (!ctx->gapfree) {
        // (the text pointer of every cell, prow, is written on demand: fbg_build_cell_tables)
        FBG_TRY(fbg_reserve(ctx, ctx->colT, ctx->N * 4));
        FBG_TRY(fbg_reserve(ctx, ctx->gwin_rows, ((ctx->N >> 7) + 2) * 2));
        hipLaunchKernelGGL(k_seg_offsets, dim3(fbg_blocks(m, 64)), dim3(64), 0, st, segcnt, m, nseg, segoff);
        hipLaunchKernelGGL((k_write_text<true, false>), dim3(nseg, (unsigned)m), dim3(TB_THREADS), 0, st, ctx->d_msa, n,
                           pos, tot, T, (uint32_t *)nullptr, ctx->colT.as<uint32_t>(), (const uint32_t *)segoff,
                           m < 65535 ? ctx->gwin_rows.as<uint16_t>() : (uint16_t *)nullptr);
        launches++;
    } else if (ctx->reversed) {
        hipLaunchKernelGGL((k_write_text<false, true>), dim3(1, (unsigned)m), dim3(TB_THREADS), 0, st, ctx->d_msa, n,
                           pos, tot, T, (uint32_t *)nullptr, (uint32_t *)nullptr, (const uint32_t *)nullptr, (uint16_t *)nullptr);
    } else if (!fused) {
        hipLaunchKernelGGL(k_copy_rows, dim3((unsigned)((n + CR_SEG - 1) / CR_SEG), (unsigned)m), dim3(TB_THREADS), 0, st,
                           ctx->d_msa, n, m, T);
    }
    launches++;
    FBG_HIP_TRY(ctx, hipGetLastError());
    return fbg_stage_end(ctx, FBG_STAGE_TEXT, launches);
}

// The per-cell tables only the record path's column scan reads (scan.hip): prow, the text pointer of every cell of an
// MSA with gaps (fbg.cpp:1596-1600,1687-1691), and igrow, the first ignore column at or after every cell.  4 bytes per
// cell each, so they are written only when that path is taken (the text and colT are written again along the way).
int fbg_build_cell_tables(fbg_ctx *ctx)
{
    if (ctx->cells_built) return FBG_OK;
    const uint64_t m = ctx->m, n = ctx->n;
    hipStream_t st = ctx->stream;
    const unsigned nseg = (unsigned)((n + RC_SEG - 1) / RC_SEG);
    uint32_t *segcnt = ctx->segtab.as<uint32_t>(), *segoff = segcnt + (size_t)m * nseg, *segmin = segoff + (size_t)m * nseg;
    if (!ctx->gapfree) {
        FBG_TRY(fbg_reserve(ctx, ctx->prow, m * n * 4));
        hipLaunchKernelGGL((k_write_text<true, false>), dim3(nseg, (unsigned)m), dim3(TB_THREADS), 0, st, ctx->d_msa, n,
                           ctx->pos.as<uint32_t>(), ctx->tot.as<uint32_t>(), ctx->text.as<uint8_t>(), ctx->prow.as<uint32_t>(),
                           ctx->colT.as<uint32_t>(), (const uint32_t *)segoff, (uint16_t *)nullptr);
    }
    if (ctx->have_ignore) {
        const uint8_t *d_is_ignore = ctx->small.as<uint8_t>();
        FBG_TRY(fbg_reserve(ctx, ctx->igrow, m * n * 4));
        hipLaunchKernelGGL(k_ignore_segmin, dim3(nseg, (unsigned)m), dim3(TB_THREADS), 0, st, ctx->d_msa, n, d_is_ignore, segmin);
        hipLaunchKernelGGL(k_ignore_sweep, dim3(nseg, (unsigned)m), dim3(TB_THREADS), 0, st, ctx->d_msa, n, d_is_ignore,
                           (const uint32_t *)segmin, ctx->igrow.as<uint32_t>());
    }
    FBG_HIP_TRY(ctx, hipGetLastError());
    ctx->cells_built = true;
    return FBG_OK;
}

// ---- synthetic MSA generator of SURVEY.md section 8(d) ---------------------------------------

__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ void k_synth(uint8_t *__restrict__ out, uint64_t m, uint64_t n, uint64_t seed, uint64_t seed2,
                        uint64_t gap_thr, uint32_t gap_run, uint64_t seed3, uint64_t n_thr)
{
    const uint64_t total = m * n;
    for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < total;
         c += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t j = c % n;
        uint8_t ch = "ACGT"[splitmix64(seed + c) >> 62];
        bool gap = false;
        if (gap_run > 0) {
            // (i,j) is a gap iff some cell (i,j-d), 0 <= d < gap_run, starts a run
            for (uint32_t d = 0; d < gap_run && d <= j; d++)
                if (splitmix64(seed2 + c - d) < gap_thr) { gap = true; break; }
        }
        if (gap) ch = '-';
        else if (n_thr > 0 && splitmix64(seed3 + c) < n_thr) ch = 'N';
        out[c] = ch;
    }
}

extern "C" int fbg_msa_synthetic(fbg_ctx *ctx, uint8_t *d_msa, uint64_t m, uint64_t n, uint64_t seed,
                                 uint64_t seed2, uint64_t gap_start_threshold, uint32_t gap_run_len,
                                 uint64_t seed3, uint64_t n_threshold)
{
    if (!ctx || !d_msa || m == 0 || n == 0) return FBG_ERR_INVALID;
    FBG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_synth, dim3(fbg_blocks(m * n, 256 * 8, 8192)), dim3(256), 0, ctx->stream, d_msa, m, n,
                       seed, seed2, gap_start_threshold, gap_run_len, seed3, n_threshold);
    FBG_HIP_TRY(ctx, hipGetLastError());
    return FBG_OK;
}
