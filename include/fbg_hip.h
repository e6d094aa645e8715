/*
 * fbg_hip.h -- C ABI of libfbg_hip.so, the MI355X (gfx950) segmentation engine.
 *
 * The reference (algbio/founderblockgraphs, founderblockgraph.cpp = "fbg.cpp") has no FFI
 * or plugin seam; the hot path sits behind two in-process C++ calls made from main():
 *
 *   segment_elastic_minmaxlength(MSA, cst, ignorechars, out_indices, disable_efg_tricks, f)
 *                                             fbg.cpp:1836-1844, called at fbg.cpp:3393
 *   segment(MSA, cst, out_labels, out_edges)  fbg.cpp:526-531,   called at fbg.cpp:3437
 *
 * together with the index they consume (load_cst, fbg.cpp:361-436).  The entry points
 * below are what a maintainer binds instead of those calls (INTEGRATION.md shows the
 * patch).  Everything is plain C: pointers + sizes, int status codes, caller-owned
 * buffers, one opaque context, calls on one context serialised by the caller.  All
 * column/row/boundary values are uint64_t because the reference's size_type is
 * (fbg.cpp:47).
 *
 * Limits of this build: one context indexes texts of N = (#non-gap cells) + m + 1 < 2^32 symbols (32-bit suffix
 * ranks); longer texts (below 2^40) go through a group (fbg_group_*, partitioned index); n < 2^31.  Violations return
 * FBG_ERR_TOO_LARGE.
 * There is no CPU fallback: without a HIP device every compute call fails with
 * FBG_ERR_NO_DEVICE / FBG_ERR_HIP.
 */
#ifndef FBG_HIP_H
#define FBG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fbg_ctx fbg_ctx;

enum {
    FBG_OK = 0,
    FBG_ERR_INVALID = 1,         /* bad argument / call order */
    FBG_ERR_NO_SEGMENTATION = 2, /* "No valid segmentation found!" fbg.cpp:1932-1937,
                                    "No proper segmentation exists." fbg.cpp:648-652 */
    FBG_ERR_OOM = 3,
    FBG_ERR_HIP = 4,             /* a HIP runtime call or kernel failed; see fbg_last_error */
    FBG_ERR_TOO_LARGE = 5,
    FBG_ERR_NO_DEVICE = 6,
    FBG_ERR_HASH_COLLISION = 7   /* fbg_block_graph only: use the caller's own label numbering instead */
};

#define FBG_MAX_ROWS 32768  /* fbg_block_graph only: a workgroup groups the labels of a block -- in LDS up to 4096 rows, in a
                               stretch of device memory of its own beyond (exercised at 5000 rows; at this limit 512 workgroups
                               hold 0.4 GB of such stretches and sort 32768 pairs each per block in device memory).  More rows:
                               FBG_ERR_TOO_LARGE, number the labels on the host.  The segmentation itself has no row limit. */

/* stage ids for fbg_stage_ms() */
enum {
    FBG_STAGE_TEXT = 0,      /* MSA -> gap-stripped text + per-row tables (fbg.cpp:372-386,1845-1917) */
    FBG_STAGE_SUFFIX_SORT,   /* suffix array + inverse (sdsl::construct, fbg.cpp:428) */
    FBG_STAGE_LCP,           /* per-position LCP with SA predecessor / successor */
    FBG_STAGE_TILE,          /* rank-order extension scan (gap-free MSAs; runs inside fbg_index_build; its time
                                is also contained in FBG_STAGE_SUFFIX_SORT) */
    FBG_STAGE_SCAN,          /* compute_f / v[] column scan (fbg.cpp:1579-1695, 552-611) */
    FBG_STAGE_DP,            /* bucket pass + DP sweep + backtrack (fbg.cpp:1940-2039, 616-664) */
    FBG_STAGE_RANK_KERNEL,   /* the k_rank_scan / k_rank_scan_lean launch alone (1 launch), for roofline accounting */
    FBG_STAGE_SORT_PASS1,    /* the three kernels of the MSD sort alone, one launch each (k_msd_pack_split, k_msd_split, */
    FBG_STAGE_SORT_PASS2,    /* k_msd_finish), for roofline accounting; 0 launches when another sort ran                */
    FBG_STAGE_SORT_PASS3,
    FBG_STAGE_COUNT
};

/* ---- context ------------------------------------------------------------------------ */
int fbg_ctx_create(int device, fbg_ctx **out);
void fbg_ctx_destroy(fbg_ctx *ctx);
/* Message of the last failing call on this context ("" if none). ctx may be NULL after a
 * failed fbg_ctx_create, in which case a process-wide message is returned. */
const char *fbg_last_error(const fbg_ctx *ctx);
/*
 * Behaviour switches of one context (integers; 0 = default unless stated).  They select between code paths that all
 * give the same results -- tests use them to reach every path; the environment has no influence on the library
 * unless FBG_DEBUG_ENV=1 is set, in which case FBG_<KEY IN CAPITALS>=<integer> presets new contexts.
 *   no_ranked          gap-free MSAs take the record path (text-order scan) instead of the rank-order scan
 *   no_packed, force_wide, full_keys   slot layout / key length of the round-0 sort
 *   no_msd_sort        rocPRIM radix sort instead of the three-pass MSD sort
 *   msd_min, bp_min    text lengths from which the MSD sort / the records-by-position passes are used (-1 = 2^24)
 *   record_scatter     records reach text order by a scatter instead of the by-position passes
 *   lcp_text           neighbour LCPs by text comparison even when the keys would do
 *   pure_scan          1: gap-free MSAs always take the group-level scan for similar rows (pure_scan.hip), -1: never
 *   no_aux_stream      k_tie_simple on the main stream
 *   rank_no_threshold  no extension threshold in the rank-order scan
 *   dp_literal, dp_wave, dp_tile, dp_safe_window   which sweep kernel runs the min-max-length / non-elastic DP
 *   gapped_rank        -1: MSAs with gaps / ignore characters always take the record path (no scan in suffix order)
 *   part_tricks_off    1: the partitioned index of an MSA with gaps / ignore characters is scanned without the elastic tricks
 *   msd_min_force      1: the sample sort of (key, position) pairs also for rows that resemble each other (tests)
 *   msd_sample_bins    1: the finish of the sample sort bins by sampled keys instead of symbol ranks (the earlier method, kept for tests)
 *   gapped_rank also takes 2 (no flag bits in the sort's values), 3 (flag bits, no threshold), 4 (threshold forced: tests)
 *   msd_probe          1: the finish of the sample sort (k_pp_finish) also runs in timing variants (copy only, single phases) before
 *                      the real launch -- for a kernel trace read in launch order (scripts/gpu_trace_order.sh); results
 *                      unchanged.  (The three-pass MSD sort had such variants in round 3: profiles/r03_msd_probe_order.txt)
 *   no_stream_upload   1: fbg_elastic_f copies the whole MSA to the device before the index build starts, also from memory of
 *                      fbg_host_alloc (default: the rows go up in eight chunks while the text is written and pass 1 of the MSD sort
 *                      runs on the rows that are there; results unchanged)
 *   rank_no_lean       1: the rank-order scan with k_rank_scan also where its lean form (k_rank_scan_lean: packed slots, threshold
 *                      above the key length) applies; results unchanged
 *   msd_xcd            which passes of the MSD sort place their writes by XCD (-1 = 3): bit 0 pass 2 (the tiles of a bucket
 *                      go to the workgroups of one XCD), bit 1 pass 1 (a stretch per bucket and XCD); 0: neither (the
 *                      layout of rounds 1-3); results unchanged
 *   span_scan          MSAs with gaps / ignore characters whose rows resemble each other take the group-level scan on
 *                      column spans (span_scan.hip); 1: every such MSA takes it, -1: none, 2: as 0, and the sorted slots
 *                      are checked to be the cells in key order (debugging), 3: as 1, with the groups of more than 1024
 *                      members worked off by chains along the later keys' groups (the earlier method, kept for tests)
 *   span_key_flags     1: that scan's slot layout for MSAs of 2^30 cells and more (the two flags of a slot in the key word,
 *                      the cell alone in the value) whatever the size (tests); results unchanged
 *   poison             1 .. 255: every device buffer the context allocates from now on is filled with that byte first
 *                      (debugging aid: reads of memory nobody wrote show up in a fresh process too); results unchanged
 *   dpw_matrix         1: the sweep over windows of 1024 .. 16384 columns with one 128 x window matrix per block (the method of
 *                      rounds 2-3: k_dpw_blockM / k_dpw_chain) instead of the matrix of the 128 columns before a block and the
 *                      closed form for the older ones (k_dpw_blockY / k_dpw_chain2); results unchanged
 *   dp_chain1          1: the walk over the groups of byte matrices on one wave (k_dp_chain, rounds 1-3) instead of six
 *                      (k_dp_chain6); results unchanged
 *   span_slow_split    workgroups that share the odd members of one large group whose pairs are all compared (0 = 32);
 *                      results unchanged
 * fbg_get_option also answers "index_kind" (read-only): -1 no index, 0 per-position records, 1 rank-order scan of a
 * gap-free MSA, 2 scan in suffix order of an MSA with gaps / ignore characters (slot by slot, or -- "span_scan_used" = 1 --
 * group by group), 3 one partition of a partitioned index.  "span_groups", "span_odd_groups", "span_irregular",
 * "span_scan_work" (read-only): the group-level scan's table sizes and the text comparisons it expected;
 * "alloc_calls", "alloc_us" (read-only): device buffers the context has allocated or enlarged so far, and the host time
 * that took in microseconds (hipFree + hipMalloc).
 * "span_decline" (read-only): why the group-level scan handed the last build to the record path: 0 it did not, 1 / 3 / 5 /
 * 6 / 8 / 9 a list or table of its kernels was full, 2 more than 32 text comparisons per suffix ahead, 4 / 7 too many groups
 * that need every pair of members compared, 10 an ignore character it cannot express;
 * "span_key_flags_used" (read-only): 1 when the slots at hand have that scan's layout for 2^30 cells and more.
 * "dp_kind" (read-only): the sweep that produced the last fbg_minmax_dp result: -1 none yet, 0 statement by statement,
 * 1 matrix chain with byte entries (windows up to 256 columns), 2 wave-parallel sweep, 3 .. 7 matrix chain with 16-bit
 * entries over windows of 1024 / 2048 / 4096 / 8192 / 16384 columns.
 * "msd_decline" (read-only): the three-pass MSD sort of the last index build: -1 not reached, 0 it sorted the slots, 1 the
 * geometry did not suit it (rocPRIM sorted), else the capacity that did not hold (2 / 4 a stretch of pass 1, 8 the arena
 * of pass 2, 16 a sub-bucket beyond the largest finish).  "pass1_ahead" (read-only): 1 when that sort found its pass 1 done
 * during a streamed upload (fbg_elastic_f from memory of fbg_host_alloc).
 * Unknown key: FBG_ERR_INVALID.
 */
int fbg_set_option(fbg_ctx *ctx, const char *key, int64_t value);
int fbg_get_option(const fbg_ctx *ctx, const char *key, int64_t *value);
/* Run all work of this context on an existing hipStream_t (NULL = the context's own). */
int fbg_set_stream(fbg_ctx *ctx, void *hip_stream);
/* Device time of a stage during the most recent call that ran it, in ms (HIP events on the
 * context's stream); launches = number of kernel launches the figure covers. */
int fbg_stage_ms(fbg_ctx *ctx, int stage, float *ms, int *launches);
/* Bytes of device memory currently held by the context's workspaces. */
uint64_t fbg_device_bytes(const fbg_ctx *ctx);
/* Free the suffix-sort / DP scratch buffers (the index itself stays valid).  Called automatically
 * after the suffix sort when the text is longer than 1.5e9 symbols, so that a 4e9-symbol index
 * fits the 288 GB of one MI355X. */
int fbg_release_scratch(fbg_ctx *ctx);

/* ---- host-buffer entry points (what main() of the C++ host calls) -------------------- */

/*
 * Replaces load_cst + compute_f (fbg.cpp:361-436, 1845-1923).  msa: m rows of n bytes,
 * row-major, '-' = gap, any other byte is a symbol.  f[0..n) is MAX-MERGED into, exactly as
 * the reference does (fbg.cpp:1681; main() zero-fills it, fbg.cpp:3388).
 * Returns FBG_ERR_NO_SEGMENTATION iff disable_tricks && f[0] == n (fbg.cpp:1932-1937); f is
 * still written in that case.
 */
int fbg_elastic_f(fbg_ctx *ctx, const uint8_t *msa, uint64_t m, uint64_t n,
                  const uint8_t *ignore_chars, uint64_t ignore_len, int disable_tricks,
                  uint64_t *f);

/*
 * Replaces the sort + DP + backtrack of segment_elastic_minmaxlength (fbg.cpp:1940-2039).
 * boundaries_out: room for n+1 values; *count_out receives the number of blocks.  The last
 * boundary is n (fbg.cpp:2027-2028).  minmaxlength_out / backtrack_out (n+1 values each) may
 * be NULL.  FBG_ERR_NO_SEGMENTATION where the reference would index backtrack[] out of range
 * (only reachable with --disable-elastic-tricks).
 * Device scratch: 64-128 bytes per column; with extensions f[x] - x beyond 254 columns up to 2 * 4096 bytes per column
 * (at most 16 GB) for the 16-bit matrices -- taken only when the device has the room, the statement-by-statement sweep
 * runs otherwise; both give the reference's arrays.
 */
int fbg_minmax_dp(fbg_ctx *ctx, const uint64_t *f, uint64_t n, uint64_t *boundaries_out,
                  uint64_t *count_out, uint64_t *minmaxlength_out, uint64_t *backtrack_out);

/*
 * Replaces load_cst + the v[j] scan of segment() (fbg.cpp:552-611).  Rows must be gap-free
 * (the reference only reaches segment() with --gap-limit=1, which drops rows with gaps,
 * fbg.cpp:176-177, 3436-3437); a gap returns FBG_ERR_INVALID.  v[0..n) is overwritten.
 */
int fbg_repeatfree_v(fbg_ctx *ctx, const uint8_t *msa, uint64_t m, uint64_t n, uint64_t *v);

/*
 * Replaces the s[]/prev[] DP and backtrack of segment() (fbg.cpp:616-664).  s_out, prev_out:
 * n values each (may be NULL); boundaries_out: room for n values, last one is n-1.
 * FBG_ERR_NO_SEGMENTATION iff s[n-1] == n+1 (fbg.cpp:648-652); s/prev are still written.
 */
int fbg_repeatfree_dp(fbg_ctx *ctx, const uint64_t *v, uint64_t n, uint64_t *s_out,
                      uint64_t *prev_out, uint64_t *boundaries_out, uint64_t *count_out);

/*
 * Replaces load_cst + the v[j] scan of segment2elasticValid (fbg.cpp:763-822), the non-elastic
 * mode for --gap-limit != 1: rows may hold gaps.  v[j] = largest jp such that the gap-stripped
 * strings of block [jp..j] occur nowhere but at the m aligned places, j+1 if there is none
 * (fbg.cpp:764, 805-807).  v[0..n) is overwritten.
 */
int fbg_gapped_v(fbg_ctx *ctx, const uint8_t *msa, uint64_t m, uint64_t n, uint64_t *v);

/*
 * Replaces the s[]/prev[] recurrence and backtrack of segment2elasticValid (fbg.cpp:827-866),
 * a heuristic (new block at v[j], or the previous column's solution with its last block
 * extended), reproduced with the reference's unsigned wrap-around; s[0] = prev[0] = n+1 always
 * (the loop starts at 1).  s_out, prev_out: n values each (may be NULL); boundaries_out: room
 * for n values, last one is n-1.  FBG_ERR_NO_SEGMENTATION iff s[n-1] == n+1 (fbg.cpp:850-854);
 * s/prev are still written.
 */
int fbg_gapped_dp(fbg_ctx *ctx, const uint64_t *v, uint64_t n, uint64_t *s_out,
                  uint64_t *prev_out, uint64_t *boundaries_out, uint64_t *count_out);

/* ---- device-resident staged API (bench, multi-GPU column shards) --------------------- */

/* Borrow an MSA already resident in device memory (row-major m x n bytes). */
int fbg_msa_set_device(fbg_ctx *ctx, const uint8_t *d_msa, uint64_t m, uint64_t n);
/* Copy a host MSA into a context-owned device buffer. */
int fbg_msa_load_host(fbg_ctx *ctx, const uint8_t *msa, uint64_t m, uint64_t n);
/*
 * Fill a device buffer with the synthetic MSA of SURVEY.md section 8(d):
 *   cell(i,j) = "ACGT"[splitmix64(seed + i*n + j) >> 62];
 *   gap_run_len > 0: cell (i,j) starts a run of gap_run_len '-' iff
 *                    splitmix64(seed2 + i*n + j) < gap_start_threshold;
 *   n_threshold > 0: a non-gap cell becomes 'N' iff splitmix64(seed3 + i*n + j) < n_threshold.
 */
int fbg_msa_synthetic(fbg_ctx *ctx, uint8_t *d_msa, uint64_t m, uint64_t n, uint64_t seed,
                      uint64_t seed2, uint64_t gap_start_threshold, uint32_t gap_run_len,
                      uint64_t seed3, uint64_t n_threshold);
/*
 * Build the index of the current MSA: text, suffix array, inverse, neighbour LCPs, and the
 * column-tiled tables the scan reads.  reversed = 0 for the elastic scan, 1 for the
 * non-elastic v[] scan (rows written back to front).  ignore_chars only matter for
 * reversed = 0.  Every rank of a multi-GPU job builds the same index (replicas).
 */
int fbg_index_build(fbg_ctx *ctx, int reversed, const uint8_t *ignore_chars, uint64_t ignore_len);
/*
 * Partitioned index for multi-GPU jobs (gap-free MSAs without ignore characters; the only path that accepts texts of
 * 2^32 symbols and more, elastic scan): every rank holds the same
 * MSA, but sorts and scans only the suffixes of key range `part` of `nparts`, so index memory and sort time
 * divide by the number of GPUs.  The caller moves two small things between the ranks:
 *
 *   fbg_part_index_build   text, splitters (identical on every rank), local sort, local classification;
 *                          d_blob (FBG_PART_HALO_BYTES, device) receives this partition's edge slots
 *       -> all-gather the blobs of all ranks, in rank order, into d_blobs (nparts * FBG_PART_HALO_BYTES)
 *   fbg_part_scan          halos in, runs walked; d_gmax (n + 1 words, device) receives the per-column maxima
 *                          of this partition's suffixes, word n its verdict
 *       -> all-reduce(MAX) d_gmax over the ranks
 *   fbg_part_finish        takes the reduced maxima; afterwards fbg_scan_f / fbg_scan_v work as after
 *                          fbg_index_build.  *ok = 2 (the same on every rank): the scan used a threshold that a few
 *                          columns did not clear -- call fbg_part_rescan (maxima without threshold into d_gmax),
 *                          all-reduce(MAX) once more and call fbg_part_finish again.
 *
 * *ok = 0 (from any of the three; the same value on every rank after the collective that follows) means the
 * input does not suit this path (similar rows, long runs): fall back to fbg_index_build on every rank.
 * Replaces the same reference code as fbg_index_build + fbg_scan_f (fbg.cpp:428, 1610-1694).
 */
#define FBG_PART_HALO 64
#define FBG_PART_HALO_BYTES (2 * FBG_PART_HALO * 12 + 16)
int fbg_part_index_build(fbg_ctx *ctx, int reversed, int part, int nparts, void *d_blob, int *ok);
/* The same for MSAs with gaps and / or ignore characters (texts below 2^32 symbols, elastic scan): the partitions are
 * scanned in suffix order like the whole index of such an MSA (gapped_rank.hip).  The scan of fbg_part_scan is for one
 * setting of the elastic tricks -- option part_tricks_off, read at fbg_part_index_build -- and fbg_scan_f must ask for
 * that one.  ignore_chars = NULL, ignore_len = 0: fbg_part_index_build. */
int fbg_part_index_build_ignore(fbg_ctx *ctx, int reversed, int part, int nparts, const uint8_t *ignore_chars, uint64_t ignore_len,
                                void *d_blob, int *ok);
int fbg_part_scan(fbg_ctx *ctx, const void *d_blobs, uint32_t *d_gmax, int *ok);
int fbg_part_finish(fbg_ctx *ctx, const uint32_t *d_gmax, int *ok);
int fbg_part_rescan(fbg_ctx *ctx, uint32_t *d_gmax);
/* Columns [x0, x1) of f, max-merged into d_f[x0..x1) (d_f has n entries, device memory). */
int fbg_scan_f(fbg_ctx *ctx, uint64_t x0, uint64_t x1, int disable_tricks, uint64_t *d_f);
/* Columns [x0, x1) of v, written to d_v[x0..x1). Requires fbg_index_build(reversed = 1). */
int fbg_scan_v(fbg_ctx *ctx, uint64_t x0, uint64_t x1, uint64_t *d_v);
/* DP + backtrack on device arrays; boundaries land in d_boundaries (n+1 values), count on host.
 * d_mml / d_bt (n+1 values each) may be NULL. */
int fbg_minmax_dp_device(fbg_ctx *ctx, const uint64_t *d_f, uint64_t n, uint64_t *d_boundaries,
                         uint64_t *count_out, uint64_t *d_mml, uint64_t *d_bt);
int fbg_repeatfree_dp_device(fbg_ctx *ctx, const uint64_t *d_v, uint64_t n, uint64_t *d_s,
                             uint64_t *d_prev, uint64_t *d_boundaries, uint64_t *count_out);
/* v[] of segment2elasticValid (fbg.cpp:763-822) for all n columns into d_v (device).  Requires
 * fbg_index_build(reversed = 0) or a finished partitioned index, without ignore characters. */
int fbg_scan_gapped_v(fbg_ctx *ctx, uint64_t *d_v);
int fbg_gapped_dp_device(fbg_ctx *ctx, const uint64_t *d_v, uint64_t n, uint64_t *d_s,
                         uint64_t *d_prev, uint64_t *d_boundaries, uint64_t *count_out);
/*
 * Nodes and edges of the elastic founder graph for a segmentation of the current MSA (the one last given to
 * fbg_elastic_f / fbg_msa_load_host / fbg_msa_set_device): what output_efg computes by hashing the gap-stripped
 * label of every (row, block) twice (fbg.cpp:1210-1260).  boundaries[nb] as produced by fbg_minmax_dp (block j covers
 * columns boundaries[j-1]+1 .. boundaries[j], the last entry is n).  All outputs are host buffers:
 *   node_of[j*m + i]     node of row i in block j in the reference's numbering (blocks in order, within a block by
 *                        first appearance in row order, fbg.cpp:1232-1246); 0xffffffff = the row has only gaps there
 *   first_node[nb + 1]   first node of every block; first_node[j+1] - first_node[j] = entry j of the B line
 *   rep_row[j*m + k]     the row whose label defines node first_node[j] + k (its S line)
 *   edge_count[nb], edges[j*m + e]   the L lines into block j: src << 32 | dst, ascending (std::set order,
 *                        fbg.cpp:1230,1253), e < edge_count[j]; edge_count[0] = 0
 * Exact: rows grouped by a 128-bit hash are compared byte by byte with their group's first row; should two different
 * labels ever collide the call returns FBG_ERR_HASH_COLLISION and the caller numbers the labels itself.
 */
int fbg_block_graph(fbg_ctx *ctx, const uint64_t *boundaries, uint64_t nb, uint32_t *node_of, uint64_t *first_node,
                    uint32_t *rep_row, uint64_t *edge_count, uint64_t *edges);
/* Copy out index arrays for tests: any pointer may be NULL. Host buffers of N entries
 * (N from fbg_text_length). SA / ISA / LCP-with-predecessor / LCP-with-successor by text position. */
uint64_t fbg_text_length(const fbg_ctx *ctx);
int fbg_index_download(fbg_ctx *ctx, uint8_t *text, uint32_t *sa, uint32_t *isa,
                       uint32_t *lcp_prev, uint32_t *lcp_next);
/* Block until all work queued on the context's stream has finished. */
int fbg_sync(fbg_ctx *ctx);

/* ---- several GPUs as one engine (fbg_ctx_create(ndev, dev_ids); 8e) --------------------------------
 *
 * Replaces the std::thread fan-out of segment_elastic_minmaxlength_multithread (fbg.cpp:2180-2289): a group holds one
 * context per entry of dev_ids and drives them from one host thread each.  ndev <= 0 or dev_ids == NULL: all visible
 * devices (the first ndev of them).  An id may repeat: several contexts on one device -- the way to exercise the
 * multi-device code on one GPU and to work a text too long for one index off in partitions.  Plans, tried in this
 * order (fbg_group_plan_used tells which one ran):
 *   FBG_PLAN_PARTITIONED  key-range partitioned index (fbg_part_*): an all-gather of FBG_PART_HALO_BYTES per
 *                         partition and an all-reduce(MAX) of n + 1 words; gap-free MSAs without ignore characters,
 *                         texts of any length below 2^40 symbols (the only way beyond 2^32); partitions may outnumber
 *                         the members (option "partitions", default: as many as keep a partition below ~1e9
 *                         suffixes and within device memory), each member then works several off in turn
 *   FBG_PLAN_COLUMNS      replicated index, member r scans columns [r * ceil(n / W), ...) -- compute_f_range's
 *                         partition, fbg.cpp:2278-2284 -- one all-gather of f
 *   FBG_PLAN_ROW_PAIRS    texts of 2^32 symbols and more that the partitioned index declines (elastic f only): one
 *                         all-reduce(MAX) of f
 * The exchanges are RCCL collectives (ncclAllGather / ncclAllReduce, one communicator per member) when every member
 * has its own device, device-to-device copies otherwise or with option "exchange" = 1.  The result lands in member
 * 0's device memory and, for the host-buffer entry points, in the caller's array; the sweep and fbg_block_graph are
 * then run on fbg_group_member(g, 0).  Group options: "partitions", "plan" (FBG_PLAN_*), "exchange" (0 auto,
 * 1 copies, 2 RCCL); any other key goes to every member (fbg_set_option).
 */
typedef struct fbg_group fbg_group;
enum { FBG_PLAN_AUTO = 0, FBG_PLAN_PARTITIONED = 1, FBG_PLAN_COLUMNS = 2, FBG_PLAN_ROW_PAIRS = 3 };
int fbg_group_create(int ndev, const int *dev_ids, fbg_group **out);
void fbg_group_destroy(fbg_group *g);
const char *fbg_group_last_error(const fbg_group *g);     /* g may be NULL after a failed fbg_group_create */
int fbg_group_size(const fbg_group *g);
fbg_ctx *fbg_group_member(fbg_group *g, int i);
int fbg_group_set_option(fbg_group *g, const char *key, int64_t value);
int fbg_group_plan_used(const fbg_group *g, int *partitions);
/* the same contracts as fbg_elastic_f / fbg_repeatfree_v / fbg_gapped_v */
int fbg_group_elastic_f(fbg_group *g, const uint8_t *msa, uint64_t m, uint64_t n, const uint8_t *ignore_chars,
                        uint64_t ignore_len, int disable_tricks, uint64_t *f);
int fbg_group_repeatfree_v(fbg_group *g, const uint8_t *msa, uint64_t m, uint64_t n, uint64_t *v);
int fbg_group_gapped_v(fbg_group *g, const uint8_t *msa, uint64_t m, uint64_t n, uint64_t *v);
/* staged: the MSA onto every member's device (once per device), then f of all columns into member 0's memory;
 * *d_f (n values, device memory of member 0) stays valid until the next call on the group */
int fbg_group_msa_load_host(fbg_group *g, const uint8_t *msa, uint64_t m, uint64_t n);
int fbg_group_msa_synthetic(fbg_group *g, uint64_t m, uint64_t n, uint64_t seed, uint64_t seed2, uint64_t gap_start_threshold,
                            uint32_t gap_run_len, uint64_t seed3, uint64_t n_threshold);
int fbg_group_scan_f(fbg_group *g, const uint8_t *ignore_chars, uint64_t ignore_len, int disable_tricks, uint64_t **d_f);

/* Pinned host memory: an MSA (or an output array) allocated here moves over PCIe by DMA straight from / into the
 * caller's pages; any other host memory is accepted too and goes through the library's own pinned bounce buffers. */
void *fbg_host_alloc(uint64_t bytes);
void fbg_host_free(void *p);

#ifdef __cplusplus
}
#endif
#endif /* FBG_HIP_H */
