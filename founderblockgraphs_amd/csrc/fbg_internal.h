// fbg_internal.h -- shared declarations of the MI355X segmentation engine (libfbg_hip.so).
// gfx950 only: 64-wide wavefronts are assumed throughout.
#pragma once
#include <cstring>
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include "../../include/fbg_hip.h"

#define FBG_WAVE 64

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// Pinned bounce buffers for large host <-> device transfers of pageable memory (ctx.hip: fbg_upload / fbg_download)
struct Stager {
    void *base = nullptr;
    size_t slot_bytes = 0;
    int nslots = 0;
    hipEvent_t *ev = nullptr;
    hipStream_t stream = nullptr;
};

struct StageTimer {
    hipEvent_t start = nullptr, stop = nullptr;
    bool recorded = false;
    int launches = 0;
};

// Behaviour switches of one context (fbg_set_option; include/fbg_hip.h lists the keys).  The environment is never
// consulted by the compute code: fbg_ctx_create copies FBG_<KEY> variables into a new context only when
// FBG_DEBUG_ENV=1 is set (ctx.hip, the library's one getenv site).
struct FbgOptions {
    int64_t no_ranked = 0, no_packed = 0, force_wide = 0, full_keys = 0, no_msd_sort = 0, msd_min = -1, bp_min = -1,
            record_scatter = 0, lcp_text = 0, no_aux_stream = 0, rank_no_threshold = 0, dp_literal = 0, dp_wave = 0,
            dp_safe_window = 0, dp_tile = 0, pure_scan = 0, gapped_rank = 0, part_tricks_off = 0, msd_sample_bins = 0, msd_min_force = 0, msd_probe = 0, msd_xcd = -1, rank_no_lean = 0, no_stream_upload = 0,
            span_scan = 0, span_key_flags = 0, span_slow_split = 0, poison = 0, dpw_matrix = 0, dp_chain1 = 0;
};

struct fbg_ctx {
    int device = 0;
    FbgOptions opt;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // second stream for work that only raises column maxima (k_tie_simple) beside the candidate kernels
    hipStream_t aux = nullptr;
    hipEvent_t aux_fork = nullptr, aux_join = nullptr;
    bool aux_pending = false;
    std::string err;
    uint64_t held_bytes = 0;
    Stager stager;

    // current MSA (row-major m x n bytes, device)
    const uint8_t *d_msa = nullptr;
    DevBuf msa_own;
    uint64_t m = 0, n = 0;

    // a streamed upload (fbg_elastic_f from pinned host memory; text_build.hip): the MSA still on the host while the index build
    // starts, pass 1 of the MSD sort done ahead on the alphabet the first chunk of rows promised (msd_sort.hip fbg_msd_pre_*)
    const uint8_t *up_host = nullptr;
    hipStream_t up_stream = nullptr;
    hipEvent_t up_ev[9] = {};
    bool pre_pass1 = false;
    uint64_t pre_symbols[4] = {0, 0, 0, 0};   // the symbols (bytes of the text) the speculative keys were set up for
    alignas(16) unsigned char pre_state[768]; // KeyGeom + the sort's arguments between fbg_msd_pre_begin and fbg_msd_sort
    uint64_t pre_tiles = 0;                    // tiles of pass 1 launched so far
    int pass1_ahead = 0;                       // the last MSD sort found its pass 1 done (fbg_get_option "pass1_ahead")

    // index state
    bool index_valid = false;
    int reversed = 0;
    bool gapfree = true;
    bool have_ignore = false;
    uint64_t N = 0;            // text length incl. sentinel
    uint64_t byte_hist[256] = {0}; // symbol histogram of the current text (fbg_build_text; read by fbg_key_setup)
    bool allow_wide = false;       // the caller can work with text positions beyond 32 bits (partitioned index only)
    uint32_t mp = 0;           // rows padded to a multiple of 64 (column-tile pitch)
    DevBuf text;               // N + 64 bytes, zero padded
    DevBuf pos, tot;           // u32[m]
    DevBuf segtab;             // u32[3][m][segments]: per 65536-column segment of a row: non-gap cells, prefix, first ignore column
    DevBuf prow;               // u32[m*n]: text pointer of cell (i,x), row-major (gapped MSAs only)
    DevBuf igrow;              // u32[m*n]: first ignore-char column >= x, row-major (ignore chars only)
    DevBuf rec;                // uint4[N] by text position: {rank, lcp-prev | hint<<31, lcp-next | hint<<31, 0}
    uint32_t *sa_ptr = nullptr; // u32[N] suffix array (lives in the sort's value buffer)
    bool lcp_from_keys = false; // neighbour LCPs came from the sorted keys (few ties) or from text compares
    DevBuf colT;               // u32[N]: MSA column of each text position (gapped MSAs only)
    DevBuf xlist;              // u32[n+1]: columns whose coloured ranks may contain consecutive integers
    // rank-order scan (rank_scan.hip): valid when `ranked`
    bool ranked = false;
    DevBuf gmax, excol, xslot; // u32[n+1]: column maxima of g, exception flags, exception slots
    DevBuf xbits;              // bitmap of the exception columns
    DevBuf exc_scratch;        // per-workgroup column state of k_scan_exceptions_big (MSAs of more than 4096 rows)
    DevBuf exc;                // uint4[n_exc * m]: (rank, lcp_prev, lcp_next) of the rows of the exception columns
    uint32_t n_exc = 0;
    // rank-order scan of MSAs with gaps / ignore characters (gapped_rank.hip): valid when `granked`
    bool granked = false;
    int grs_tricks_off = 0;        // the setting of the elastic tricks gmax was computed for
    bool grs_skip = false;         // rebuilding the record way after the scan ran out of room
    uint32_t grs_ign_lo = 0, grs_ign_hi = 0;
    DevBuf gwin;                   // 16 bytes per 128 text positions: column, row, gap runs
    DevBuf gbits;                  // 1 bit per text position: not the column after its predecessor's
    DevBuf gwin_rows;              // u16 per window of 128 positions: the row of its first position (written with the text)
    const uint64_t *grs_ebits = nullptr;   // gbits while the pack kernels are to fold it into bit 31 of the sort's values
    bool grs_flagged = false;      // the sorted values carry that bit
    bool pairs_similar = false;    // the sample of the (key, position) sort says: rows that resemble each other (ties everywhere)
    bool grs_ties_done = false;    // the tie groups of the kept slots are in text order
    bool gpart = false;            // one key-range partition of such an index (fbg_part_*)
    bool grs_part_failed = false;  // its exact redo of a few columns ran out of room
    uint32_t grs_t = 1;            // the threshold of the last scan (1: none), the columns it redid exactly
    uint64_t grs_redone = 0;
    // group-level scan on column spans for similar rows with gaps / ignore characters (span_scan.hip): `granked` and `spanned`
    bool spanned = false;
    DevBuf sp_cells;               // u32[N]: cell | flags of every text position, the payload of the sort
    DevBuf sp_flagT;               // u8[N]: the flags alone where the cells take all 32 bits (sp_key_flags)
    bool sp_key_flags = false;     // span scan: the flags W / I are the key words' two lowest bits (2^30 cells and more): the sort ahead
    uint64_t alloc_calls = 0, alloc_us = 0;   // device allocations of the context so far and the host time they took (free + malloc)
    int sp_decline = 0;            // why the group-level scan handed the slots back (0: it did not; include/fbg_hip.h "span_decline")
    bool sp_key_flags_sorted = false;   // ... the sorted slots at hand
    DevBuf sp_cwin;                // 20 bytes per 128 cells of a row: text position of a cell
    DevBuf sp_tiles, sp_gstart, sp_gcol, sp_gflags, sp_rstart, sp_rid, sp_gplo, sp_gphi, sp_gval, sp_odd, sp_irr, sp_chain, sp_slow, sp_mins;
    uint32_t sp_chain_n = 0, sp_slow_n = 0;
    uint64_t sp_G = 0, sp_R = 0, sp_n_irr = 0, sp_work = 0;
    uint32_t sp_n_odd[4] = {0, 0, 0, 0}, sp_odd_cap = 0;
    const uint32_t *sort_payload = nullptr;   // while set: the pack kernels of the (key, value) sorts write payload[p] instead of p | flag
    int64_t msd_decline = -1;    // the last fbg_msd_sort: 0 sorted; 1 not tried (geometry), 2 / 4 a stretch of pass 1, 8 the arena of pass 2, 16 a sub-bucket too large
    int dp_kind = -1;            // which sweep produced the last fbg_dp_minmax result (fbg_get_option "dp_kind")
    bool cells_built = false;      // prow / igrow hold the current MSA (built on demand: only the record path reads them)
    uint8_t ignore_tab[256] = {0};
    uint64_t *rk_keys = nullptr; // sorted slots: keys (pairs layout, positions in sa_ptr) or key << rk_pb | position (packed)
    int rk_b = 0, rk_key_bits = 0, rk_K = 0;
    int rk_layout = 0, rk_pb = 0;   // FBG_SLOTS_*
    // partitioned index (partition.hip): this GPU holds the SA slots of key range `part` of `nparts`
    bool part_active = false;
    int part = 0, nparts = 1;
    uint64_t part_count = 0;   // owned slots (keys / vals arrays: FBG_PART_HALO + part_count + FBG_PART_HALO)
    uint64_t part_T = 0;       // candidates found by phase 1
    uint32_t part_gmin = 0;    // largest threshold any partition scanned with (0: none, nothing to verify)

    // scratch
    DevBuf keysA, keysB, valsA, valsB, grp, flags, list, tie_list, big_groups, tmp, small, scalars;
    DevBuf kargs;              // arguments a kernel reads from memory (k_rank_scan_lean) ...
    alignas(16) unsigned char kargs_host[512];   // ... and the host copy they are sent from
    DevBuf msd_w, msd_v;       // sub-bucket stretches of the MSD sort of 12-byte slots (msd_sort_pairs.hip)
    DevBuf dp_a, dp_b, dp_c, dp_d, dp_e, dp_f, dp_g, dp_h, io_a, io_b, io_c, io_d;
    DevBuf bt_up, bt_dep;      // binary-lifting tables of the parallel backtrack
    DevBuf ps_a, ps_b, ps_c, ps_d, ps_e, ps_f, ps_g, ps_h;   // group / run tables of the scan for similar rows (pure_scan.hip)

    StageTimer timers[FBG_STAGE_COUNT];
};

// ---- error plumbing ----------------------------------------------------------------------
int fbg_fail(fbg_ctx *ctx, int code, const char *fmt, ...);
#define FBG_HIP_TRY(ctx, expr)                                                              \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fbg_fail((ctx), e_ == hipErrorOutOfMemory ? FBG_ERR_OOM : FBG_ERR_HIP,   \
                            "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                            __LINE__);                                                      \
    } while (0)
#define FBG_TRY(expr)                \
    do {                             \
        int rc_ = (expr);            \
        if (rc_ != FBG_OK) return rc_; \
    } while (0)

int fbg_reserve(fbg_ctx *ctx, DevBuf &b, size_t bytes);   // grow-only device allocation
// blocking host <-> device copies; pageable memory of 32 MB and more goes through pinned bounce buffers filled by
// several host threads (the runtime's own staging of pageable memory is a single-threaded memcpy)
int fbg_upload(fbg_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int fbg_download(fbg_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
void fbg_release(fbg_ctx *ctx, DevBuf &b);
int fbg_stage_begin(fbg_ctx *ctx, int stage);
int fbg_stage_end(fbg_ctx *ctx, int stage, int launches);

#define FBG_SLOTS_PAIRS 0    // keys[k] = key, vals[k] = position (32 bits)
#define FBG_SLOTS_PACKED 1   // keys[k] = key << pb | position
#define FBG_SLOTS_WIDE 2     // keys[k] = key << pb | position >> 32, vals[k] = low 32 bits of the position
// Key geometry of the round-0 sort.  compact: separators ('#', sentinel) share code 0 with the smallest symbol and
// blank the rest of the key (rank_scan.hip undoes the ambiguity with the row arithmetic of gap-free MSAs);
// packed: one 64-bit word per suffix, key << pb | position, sorted by its key bits only.
struct KeyGeom {
    int b = 0, K = 0, key_bits = 0;
    bool compact = false, packed = false;
    bool wide = false;            // (key word, low position word) pairs, key << pb | position >> 32: texts of 2^32 symbols and more
    int pb = 0;
    const uint8_t *d_code = nullptr;
};

// ---- stages (each in its own translation unit) ----------------------------------------------
int fbg_build_text(fbg_ctx *ctx, const uint8_t *ignore, uint64_t ignore_len);  // text_build.hip
int fbg_suffix_sort(fbg_ctx *ctx);                                            // suffix_sort.hip
int fbg_neighbour_lcp(fbg_ctx *ctx);                                          // lcp.hip
int fbg_rank_scan_try(fbg_ctx *ctx, uint64_t *keys, uint32_t *vals, const KeyGeom &g, int *done);       // rank_scan.hip
int fbg_pure_scan_try(fbg_ctx *ctx, uint64_t *keys, uint32_t *vals, const KeyGeom &g, int *done);       // pure_scan.hip
int fbg_rank_finish(fbg_ctx *ctx, uint64_t x0, uint64_t x1, int mode, int disable_tricks, uint64_t *d_out);
int fbg_rank_materialize(fbg_ctx *ctx, uint32_t *d_sa, uint32_t *d_isa, uint32_t *d_pl, uint32_t *d_pr);
#define FBG_STAGE_RANKSCAN FBG_STAGE_TILE
int fbg_grs_prepare(fbg_ctx *ctx, int *launches);
int fbg_grs_part_classify(fbg_ctx *ctx, uint64_t *keys, uint32_t *vals, uint64_t count, const KeyGeom &g, int eligible, uint8_t *d_blob, int *ok);
int fbg_grs_part_scan(fbg_ctx *ctx, const uint8_t *d_blobs, uint32_t *d_gmax, int *ok);
int fbg_grs_part_unfilled(fbg_ctx *ctx, uint64_t *unfilled);
int fbg_grs_part_rescan(fbg_ctx *ctx);
int fbg_grs_strip(fbg_ctx *ctx, uint32_t *vals);
int fbg_grs_try(fbg_ctx *ctx, uint64_t *keys, uint32_t *vals, const KeyGeom &g, int *done);             // gapped_rank.hip
int fbg_grs_finish(fbg_ctx *ctx, uint64_t x0, uint64_t x1, int disable_tricks, uint64_t *d_out, int *ok);
int fbg_grs_materialize(fbg_ctx *ctx, uint32_t *d_sa, uint32_t *d_isa, uint32_t *d_pl, uint32_t *d_pr);
bool fbg_span_eligible(fbg_ctx *ctx, const KeyGeom &g);
bool fbg_span_key_flags(fbg_ctx *ctx, KeyGeom &g);                                                 // span_scan.hip
int fbg_span_prepare(fbg_ctx *ctx, const KeyGeom &g, int *launches);
int fbg_span_try(fbg_ctx *ctx, uint64_t *keys, uint32_t *vals, const KeyGeom &g, int *done);
int fbg_span_rescan(fbg_ctx *ctx, int disable_tricks, int *ok);
int fbg_build_cell_tables(fbg_ctx *ctx);                                                                // text_build.hip
int fbg_key_setup(fbg_ctx *ctx, bool compact, KeyGeom *g, int *launches);                                // suffix_sort.hip
int fbg_rank_part_classify(fbg_ctx *ctx, uint64_t *keys, uint32_t *vals, uint64_t count, const KeyGeom &g, int pre_ok,
                           uint8_t *d_blob, int *ok);                         // rank_scan.hip
int fbg_rank_part_runs(fbg_ctx *ctx, const uint8_t *d_blobs, uint32_t *d_gmax, int *ok);
int fbg_rank_part_unfilled(fbg_ctx *ctx, uint64_t *unfilled);
int fbg_rank_part_rescan(fbg_ctx *ctx);
int fbg_part_sort(fbg_ctx *ctx, int part, int nparts, uint8_t *d_blob, int *ok);
int fbg_msd_sort(fbg_ctx *ctx, const KeyGeom &g, uint64_t **sorted, int *ok, int *launches);         // msd_sort.hip
int fbg_msd_pre_begin(fbg_ctx *ctx, int *ok);                     // pass 1 ahead of the rest, on a text that is still arriving
int fbg_msd_pre_pass1(fbg_ctx *ctx, uint64_t avail);              // ... over the tiles whose positions (and 64 beyond) are below avail
bool fbg_msd_pre_geom(fbg_ctx *ctx, KeyGeom *g);                  // the key geometry it used, if pass 1 was done ahead
int fbg_msd_sort_part(fbg_ctx *ctx, const KeyGeom &g, uint64_t lo, uint64_t hi, int nohi, int nparts, uint64_t out_offset,
                      uint64_t *count, int *ok, int *launches);                                       // msd_sort_pairs.hip                        // suffix_sort.hip
int fbg_sample_sort_pairs(fbg_ctx *ctx, const KeyGeom &g, int *ok, int *launches);                      // msd_sort_pairs.hip
int fbg_scan_columns(fbg_ctx *ctx, uint64_t x0, uint64_t x1, int mode, int disable_tricks,
                     uint64_t *d_out);                                        // scan.hip
int fbg_dp_minmax(fbg_ctx *ctx, const uint64_t *d_f, uint64_t n, uint64_t *d_boundaries,
                  uint64_t *count_out, uint64_t *d_mml, uint64_t *d_bt);      // dp.hip
int fbg_dp_repeatfree(fbg_ctx *ctx, const uint64_t *d_v, uint64_t n, uint64_t *d_s,
                      uint64_t *d_prev, uint64_t *d_boundaries, uint64_t *count_out);
int fbg_gapped_v_from_f(fbg_ctx *ctx, const uint64_t *d_f, uint64_t n, uint64_t *d_v);
int fbg_dp_gapped(fbg_ctx *ctx, const uint64_t *d_v, uint64_t n, uint64_t *d_s, uint64_t *d_prev,
                  uint64_t *d_boundaries, uint64_t *count_out);

#define FBG_SCAN_F 0
#define FBG_SCAN_V 1

static inline unsigned fbg_blocks(uint64_t work, unsigned per_block, unsigned cap = 0x7fffffffu)
{
    uint64_t b = (work + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (unsigned)b;
}
