// ctx.hip -- context, error plumbing, workspaces and the C-ABI entry points of include/fbg_hip.h.
#include "fbg_internal.h"
#include <chrono>
#include <stdarg.h>
#include <stdio.h>
#include <ctype.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

static std::string g_global_err;

int fbg_fail(fbg_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf; else g_global_err = buf;
    return code;
}

int fbg_reserve(fbg_ctx *ctx, DevBuf &b, size_t bytes)
{
    if (bytes == 0) bytes = 256;
    if (b.cap >= bytes) return FBG_OK;
    const auto t_begin = std::chrono::steady_clock::now();
    if (b.p) {
        FBG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        FBG_HIP_TRY(ctx, hipFree(b.p));
        ctx->held_bytes -= b.cap;
        b.p = nullptr; b.cap = 0;
    }
    size_t want = (bytes + 255) & ~(size_t)255;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        b.p = nullptr;
        return fbg_fail(ctx, FBG_ERR_OOM, "hipMalloc(%zu bytes) failed: %s (context already holds %llu bytes)",
                        want, hipGetErrorString(e), (unsigned long long)ctx->held_bytes);
    }
    b.cap = want;
    ctx->held_bytes += want;
    ctx->alloc_calls++;
    ctx->alloc_us += (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_begin).count();
    // (debugging aid: fresh buffers full of a byte pattern instead of whatever the allocator hands out -- zeros, in a new process)
    if (ctx->opt.poison) FBG_HIP_TRY(ctx, hipMemsetAsync(b.p, (int)(ctx->opt.poison & 255), want, ctx->stream));
    return FBG_OK;
}

void fbg_release(fbg_ctx *ctx, DevBuf &b)
{
    if (b.p) {
        (void)hipFree(b.p);
        ctx->held_bytes -= b.cap;
        b.p = nullptr; b.cap = 0;
    }
}

// ---- host <-> device transfers -----------------------------------------------------------------------------------
// hipMemcpy of pageable memory is staged by the runtime through one thread's memcpy: 1 GB of MSA arrives at a
// fraction of the PCIe rate.  Here up to 8 host threads copy chunks into a ring of pinned slots and queue the DMA
// of each chunk behind its own copy; a slot is reused once the DMA that last read it has finished (its event).
#define FBG_STAGE_SLOT (8u << 20)
#define FBG_STAGE_SLOTS 16
#define FBG_STAGE_MIN (32u << 20)

static bool host_pointer_is_pinned(const void *p)
{
    hipPointerAttribute_t at;
    const hipError_t e = hipPointerGetAttributes(&at, p);
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }     // plain malloc'ed memory: not known to the runtime
    return at.type == hipMemoryTypeHost;
}

static int stager_ready(fbg_ctx *ctx)
{
    Stager &s = ctx->stager;
    if (s.base) return FBG_OK;
    FBG_HIP_TRY(ctx, hipHostMalloc(&s.base, (size_t)FBG_STAGE_SLOT * FBG_STAGE_SLOTS, hipHostMallocDefault));
    s.slot_bytes = FBG_STAGE_SLOT; s.nslots = FBG_STAGE_SLOTS;
    s.ev = new hipEvent_t[s.nslots];
    for (int i = 0; i < s.nslots; i++) FBG_HIP_TRY(ctx, hipEventCreateWithFlags(&s.ev[i], hipEventDisableTiming));
    FBG_HIP_TRY(ctx, hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
    return FBG_OK;
}

static int staged_copy(fbg_ctx *ctx, void *dst, const void *src, size_t bytes, bool to_device)
{
    FBG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    FBG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));              // whatever produced / will consume the device side
    const void *host = to_device ? src : dst;
    if (bytes < FBG_STAGE_MIN || host_pointer_is_pinned(host)) {
        FBG_HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, ctx->stream));
        FBG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return FBG_OK;
    }
    FBG_TRY(stager_ready(ctx));
    Stager &s = ctx->stager;
    const size_t nchunks = (bytes + s.slot_bytes - 1) / s.slot_bytes;
    unsigned hw = std::thread::hardware_concurrency();
    int T = (int)std::min<size_t>(std::min<unsigned>(std::max(2u, hw / 2), 8u), nchunks);
    if (T > s.nslots / 2) T = s.nslots / 2;
    // Thread t owns the pinned slots t and t + T and takes the chunks t, t + T, t + 2T, ...: a slot is only ever
    // touched by one thread, so "the slot's previous use is over" is that thread's own event, recorded by itself.
    std::atomic<int> failed{0};
    auto worker = [&](int t) {
        if (hipSetDevice(ctx->device) != hipSuccess) { failed = 1; return; }
        const int slots[2] = {t, t + T};
        uint8_t *pins[2] = {static_cast<uint8_t *>(s.base) + (size_t)slots[0] * s.slot_bytes,
                            static_cast<uint8_t *>(s.base) + (size_t)slots[1] * s.slot_bytes};
        if (to_device) {
            size_t k = 0;
            for (size_t c = (size_t)t; c < nchunks && !failed; c += (size_t)T, k++) {
                const int w = (int)(k & 1);
                const size_t off = c * s.slot_bytes, len = std::min(s.slot_bytes, bytes - off);
                if (k >= 2 && hipEventSynchronize(s.ev[slots[w]]) != hipSuccess) { failed = 1; return; }
                memcpy(pins[w], static_cast<const uint8_t *>(src) + off, len);
                if (hipMemcpyAsync(static_cast<uint8_t *>(dst) + off, pins[w], len, hipMemcpyHostToDevice, s.stream) != hipSuccess ||
                    hipEventRecord(s.ev[slots[w]], s.stream) != hipSuccess) { failed = 1; return; }
            }
        } else {
            // device -> slot queued one chunk ahead; wait for a chunk's own event, then slot -> caller's memory
            auto issue = [&](size_t c, int w) {
                const size_t off = c * s.slot_bytes, len = std::min(s.slot_bytes, bytes - off);
                return hipMemcpyAsync(pins[w], static_cast<const uint8_t *>(src) + off, len, hipMemcpyDeviceToHost, s.stream) == hipSuccess &&
                       hipEventRecord(s.ev[slots[w]], s.stream) == hipSuccess;
            };
            if ((size_t)t < nchunks && !issue((size_t)t, 0)) { failed = 1; return; }
            size_t k = 0;
            for (size_t c = (size_t)t; c < nchunks && !failed; c += (size_t)T, k++) {
                const int w = (int)(k & 1);
                if (c + (size_t)T < nchunks && !issue(c + (size_t)T, w ^ 1)) { failed = 1; return; }
                if (hipEventSynchronize(s.ev[slots[w]]) != hipSuccess) { failed = 1; return; }
                const size_t off = c * s.slot_bytes, len = std::min(s.slot_bytes, bytes - off);
                memcpy(static_cast<uint8_t *>(dst) + off, pins[w], len);
            }
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; t++) th.emplace_back(worker, t);
    worker(0);
    for (auto &t : th) t.join();
    FBG_HIP_TRY(ctx, hipStreamSynchronize(s.stream));
    if (failed) return fbg_fail(ctx, FBG_ERR_HIP, "staged %s copy of %zu bytes failed: %s", to_device ? "host-to-device" : "device-to-host",
                                bytes, hipGetErrorString(hipGetLastError()));
    return FBG_OK;
}

int fbg_upload(fbg_ctx *ctx, void *d_dst, const void *h_src, size_t bytes) { return staged_copy(ctx, d_dst, h_src, bytes, true); }
int fbg_download(fbg_ctx *ctx, void *h_dst, const void *d_src, size_t bytes) { return staged_copy(ctx, h_dst, d_src, bytes, false); }

int fbg_stage_begin(fbg_ctx *ctx, int stage)
{
    StageTimer &t = ctx->timers[stage];
    if (!t.start) {
        FBG_HIP_TRY(ctx, hipEventCreate(&t.start));
        FBG_HIP_TRY(ctx, hipEventCreate(&t.stop));
    }
    t.recorded = false;
    FBG_HIP_TRY(ctx, hipEventRecord(t.start, ctx->stream));
    return FBG_OK;
}

int fbg_stage_end(fbg_ctx *ctx, int stage, int launches)
{
    StageTimer &t = ctx->timers[stage];
    FBG_HIP_TRY(ctx, hipEventRecord(t.stop, ctx->stream));
    t.recorded = true;
    t.launches = launches;
    return FBG_OK;
}

// option table: key -> member of FbgOptions
struct OptKey { const char *name; int64_t FbgOptions::*field; };
static const OptKey g_opt_keys[] = {
    {"no_ranked", &FbgOptions::no_ranked}, {"no_packed", &FbgOptions::no_packed}, {"force_wide", &FbgOptions::force_wide},
    {"full_keys", &FbgOptions::full_keys}, {"no_msd_sort", &FbgOptions::no_msd_sort}, {"msd_min", &FbgOptions::msd_min},
    {"bp_min", &FbgOptions::bp_min}, {"record_scatter", &FbgOptions::record_scatter}, {"lcp_text", &FbgOptions::lcp_text},
    {"no_aux_stream", &FbgOptions::no_aux_stream}, {"rank_no_threshold", &FbgOptions::rank_no_threshold},
    {"dp_literal", &FbgOptions::dp_literal}, {"dp_wave", &FbgOptions::dp_wave}, {"dp_safe_window", &FbgOptions::dp_safe_window},
    {"dp_tile", &FbgOptions::dp_tile}, {"pure_scan", &FbgOptions::pure_scan}, {"gapped_rank", &FbgOptions::gapped_rank}, {"part_tricks_off", &FbgOptions::part_tricks_off}, {"msd_sample_bins", &FbgOptions::msd_sample_bins}, {"msd_min_force", &FbgOptions::msd_min_force}, {"msd_probe", &FbgOptions::msd_probe}, {"msd_xcd", &FbgOptions::msd_xcd}, {"rank_no_lean", &FbgOptions::rank_no_lean}, {"no_stream_upload", &FbgOptions::no_stream_upload},
    {"span_scan", &FbgOptions::span_scan}, {"span_key_flags", &FbgOptions::span_key_flags}, {"span_slow_split", &FbgOptions::span_slow_split}, {"poison", &FbgOptions::poison}, {"dpw_matrix", &FbgOptions::dpw_matrix}, {"dp_chain1", &FbgOptions::dp_chain1},
};

// The one place the library reads the environment: FBG_DEBUG_ENV=1 lets FBG_<KEY>=<integer> preset the options of
// every context created afterwards (debugging a deployed binary); without it no variable has any effect.
static void options_from_env(FbgOptions &o)
{
    auto env = [](const std::string &name) -> const char * { return getenv(name.c_str()); };
    const char *dbg = env("FBG_DEBUG_ENV");
    if (!dbg || strcmp(dbg, "1") != 0) return;
    for (const OptKey &k : g_opt_keys) {
        std::string name = "FBG_";
        for (const char *c = k.name; *c; c++) name += (char)toupper((unsigned char)*c);
        if (const char *v = env(name)) o.*(k.field) = *v ? strtoll(v, nullptr, 10) : 1;
    }
}

extern "C" {

int fbg_set_option(fbg_ctx *ctx, const char *key, int64_t value)
{
    if (!ctx || !key) return FBG_ERR_INVALID;
    for (const OptKey &k : g_opt_keys)
        if (strcmp(k.name, key) == 0) { ctx->opt.*(k.field) = value; return FBG_OK; }
    return fbg_fail(ctx, FBG_ERR_INVALID, "fbg_set_option: unknown key '%s'", key);
}

int fbg_get_option(const fbg_ctx *ctx, const char *key, int64_t *value)
{
    if (!ctx || !key || !value) return FBG_ERR_INVALID;
    for (const OptKey &k : g_opt_keys)
        if (strcmp(k.name, key) == 0) { *value = ctx->opt.*(k.field); return FBG_OK; }
    if (strcmp(key, "grs_threshold") == 0) { *value = ctx->grs_t; return FBG_OK; }
    if (strcmp(key, "grs_redone") == 0) { *value = (int64_t)ctx->grs_redone; return FBG_OK; }
    if (strcmp(key, "dp_kind") == 0) { *value = ctx->dp_kind; return FBG_OK; }
    if (strcmp(key, "msd_decline") == 0) { *value = ctx->msd_decline; return FBG_OK; }
    if (strcmp(key, "pass1_ahead") == 0) { *value = ctx->pass1_ahead; return FBG_OK; }
    if (strcmp(key, "alloc_calls") == 0) { *value = (int64_t)ctx->alloc_calls; return FBG_OK; }
    if (strcmp(key, "alloc_us") == 0) { *value = (int64_t)ctx->alloc_us; return FBG_OK; }
    if (strcmp(key, "span_decline") == 0) { *value = ctx->sp_decline; return FBG_OK; }
    if (strcmp(key, "span_key_flags_used") == 0) { *value = (ctx->index_valid && ctx->granked && ctx->spanned && ctx->sp_key_flags_sorted) ? 1 : 0; return FBG_OK; }
    if (strcmp(key, "span_scan_used") == 0) { *value = (ctx->index_valid && ctx->granked && ctx->spanned) ? 1 : 0; return FBG_OK; }
    if (strcmp(key, "span_scan_work") == 0) { *value = (int64_t)ctx->sp_work; return FBG_OK; }
    if (strcmp(key, "span_groups") == 0) { *value = (int64_t)ctx->sp_G; return FBG_OK; }
    if (strcmp(key, "span_odd_groups") == 0) { *value = (int64_t)ctx->sp_n_odd[0] + ctx->sp_n_odd[1] + ctx->sp_n_odd[2] + ctx->sp_n_odd[3]; return FBG_OK; }
    if (strcmp(key, "span_irregular") == 0) { *value = (int64_t)ctx->sp_n_irr; return FBG_OK; }
    if (strcmp(key, "span_chain") == 0) { *value = (int64_t)ctx->sp_chain_n; return FBG_OK; }
    if (strcmp(key, "span_slow_groups") == 0) { *value = (int64_t)ctx->sp_slow_n; return FBG_OK; }
    if (strncmp(key, "span_dbg", 8) == 0 && key[8] >= '0' && key[8] <= '7' && !key[9]) {   // debug builds (SP_PHASE_TIMERS): cycles per phase of k_sp_odd_pairs
        unsigned long long v = 0;
        if (hipStreamSynchronize(ctx->stream) != hipSuccess || ctx->scalars.cap < 256 * 8 ||
            hipMemcpy(&v, (const unsigned long long *)ctx->scalars.p + 208 + 24 + (key[8] - '0'), 8, hipMemcpyDeviceToHost) != hipSuccess) return FBG_ERR_HIP;
        *value = (int64_t)v;
        return FBG_OK;
    }
    if (strcmp(key, "index_kind") == 0) {      // read-only: which form the current index has
        *value = !ctx->index_valid ? -1 : ctx->part_active ? 3 : ctx->granked ? 2 : ctx->ranked ? 1 : 0;
        return FBG_OK;
    }
    return FBG_ERR_INVALID;
}

void *fbg_host_alloc(uint64_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
void fbg_host_free(void *p) { if (p) (void)hipHostFree(p); }

int fbg_ctx_create(int device, fbg_ctx **out)
{
    if (!out) return FBG_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fbg_fail(nullptr, FBG_ERR_NO_DEVICE,
                        "no HIP device visible (%s); libfbg_hip has no CPU fallback",
                        e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (device < 0 || device >= count)
        return fbg_fail(nullptr, FBG_ERR_INVALID, "device %d out of range (0..%d)", device, count - 1);
    e = hipSetDevice(device);
    if (e != hipSuccess) return fbg_fail(nullptr, FBG_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(e));
    fbg_ctx *ctx = new fbg_ctx();
    ctx->device = device;
    options_from_env(ctx->opt);
    e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete ctx;
        return fbg_fail(nullptr, FBG_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    ctx->own_stream = true;
    *out = ctx;
    return FBG_OK;
}

void fbg_ctx_destroy(fbg_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    DevBuf *bufs[] = {&ctx->msa_own, &ctx->text, &ctx->pos, &ctx->tot, &ctx->segtab, &ctx->exc_scratch, &ctx->prow, &ctx->igrow, &ctx->rec,
                      &ctx->xlist, &ctx->gmax, &ctx->excol, &ctx->xslot, &ctx->xbits, &ctx->exc, &ctx->colT, &ctx->keysA, &ctx->keysB, &ctx->valsA, &ctx->valsB, &ctx->grp, &ctx->flags,
                      &ctx->list, &ctx->tie_list, &ctx->big_groups, &ctx->kargs, &ctx->msd_w, &ctx->msd_v, &ctx->tmp, &ctx->small, &ctx->scalars, &ctx->dp_a, &ctx->dp_b, &ctx->dp_c,
                      &ctx->dp_d, &ctx->dp_e, &ctx->dp_f, &ctx->dp_g, &ctx->dp_h, &ctx->io_a, &ctx->io_b,
                      &ctx->io_c, &ctx->io_d, &ctx->bt_up, &ctx->bt_dep, &ctx->ps_a, &ctx->ps_b, &ctx->ps_c, &ctx->ps_d,
                      &ctx->ps_e, &ctx->ps_f, &ctx->ps_g, &ctx->ps_h, &ctx->gwin, &ctx->gbits, &ctx->gwin_rows,
                      &ctx->sp_cells, &ctx->sp_flagT, &ctx->sp_cwin, &ctx->sp_tiles, &ctx->sp_gstart, &ctx->sp_gcol, &ctx->sp_gflags, &ctx->sp_rstart, &ctx->sp_rid,
                      &ctx->sp_gplo, &ctx->sp_gphi, &ctx->sp_gval, &ctx->sp_odd, &ctx->sp_irr, &ctx->sp_chain, &ctx->sp_slow, &ctx->sp_mins};
    for (DevBuf *b : bufs) fbg_release(ctx, *b);
    if (ctx->up_stream) {
        (void)hipStreamSynchronize(ctx->up_stream);
        for (auto &e : ctx->up_ev) if (e) (void)hipEventDestroy(e);
        (void)hipStreamDestroy(ctx->up_stream);
    }
    for (auto &t : ctx->timers) {
        if (t.start) (void)hipEventDestroy(t.start);
        if (t.stop) (void)hipEventDestroy(t.stop);
    }
    if (ctx->stager.base) {
        for (int i = 0; i < ctx->stager.nslots; i++) (void)hipEventDestroy(ctx->stager.ev[i]);
        delete[] ctx->stager.ev;
        (void)hipStreamDestroy(ctx->stager.stream);
        (void)hipHostFree(ctx->stager.base);
    }
    if (ctx->aux_fork) (void)hipEventDestroy(ctx->aux_fork);
    if (ctx->aux_join) (void)hipEventDestroy(ctx->aux_join);
    if (ctx->aux) (void)hipStreamDestroy(ctx->aux);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *fbg_last_error(const fbg_ctx *ctx) { return ctx ? ctx->err.c_str() : g_global_err.c_str(); }

int fbg_set_stream(fbg_ctx *ctx, void *hip_stream)
{
    if (!ctx) return FBG_ERR_INVALID;
    FBG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream) { (void)hipStreamDestroy(ctx->stream); ctx->own_stream = false; }
    if (hip_stream) {
        ctx->stream = (hipStream_t)hip_stream;
    } else {
        FBG_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        ctx->own_stream = true;
    }
    return FBG_OK;
}

int fbg_stage_ms(fbg_ctx *ctx, int stage, float *ms, int *launches)
{
    if (!ctx || stage < 0 || stage >= FBG_STAGE_COUNT) return FBG_ERR_INVALID;
    StageTimer &t = ctx->timers[stage];
    if (ms) *ms = 0.f;
    if (launches) *launches = 0;
    if (!t.recorded) return FBG_OK;
    FBG_HIP_TRY(ctx, hipEventSynchronize(t.stop));
    float v = 0.f;
    FBG_HIP_TRY(ctx, hipEventElapsedTime(&v, t.start, t.stop));
    if (ms) *ms = v;
    if (launches) *launches = t.launches;
    return FBG_OK;
}

uint64_t fbg_device_bytes(const fbg_ctx *ctx) { return ctx ? ctx->held_bytes : 0; }

int fbg_release_scratch(fbg_ctx *ctx)
{
    if (!ctx) return FBG_ERR_INVALID;
    FBG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    FBG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    // valsB stays: it is the suffix array; so does whichever key buffer holds the sorted slots of a rank-order index
    const bool sorted_in_A = (ctx->ranked || ctx->granked) && ctx->rk_keys == ctx->keysA.as<uint64_t>();
    DevBuf *bufs[] = {sorted_in_A ? &ctx->keysB : &ctx->keysA, &ctx->valsA, &ctx->grp, &ctx->flags, &ctx->list, &ctx->tie_list, &ctx->msd_w, &ctx->msd_v,
                      &ctx->tmp, &ctx->dp_a, &ctx->dp_b, &ctx->dp_c, &ctx->dp_d, &ctx->dp_e, &ctx->dp_f,
                      &ctx->dp_g, &ctx->dp_h, &ctx->ps_a, &ctx->ps_b, &ctx->ps_c, &ctx->ps_d, &ctx->ps_e, &ctx->ps_f,
                      &ctx->ps_g, &ctx->ps_h, &ctx->sp_cells, &ctx->sp_flagT, &ctx->sp_tiles};
    for (DevBuf *b : bufs) fbg_release(ctx, *b);
    return FBG_OK;
}

int fbg_sync(fbg_ctx *ctx)
{
    if (!ctx) return FBG_ERR_INVALID;
    FBG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return FBG_OK;
}

static int check_dims(fbg_ctx *ctx, uint64_t m, uint64_t n)
{
    if (!ctx) return FBG_ERR_INVALID;
    if (m == 0 || n == 0) return fbg_fail(ctx, FBG_ERR_INVALID, "empty MSA (m=%llu, n=%llu)",
                                          (unsigned long long)m, (unsigned long long)n);
    // whether a text beyond 32-bit positions can be indexed depends on the path taken: fbg_build_text decides
    if (n >= (1ull << 31) || m * (n + 1) + 1 >= (1ull << 40))
        return fbg_fail(ctx, FBG_ERR_TOO_LARGE, "MSA of %llu x %llu cells is beyond this engine (n < 2^31, text < 2^40)",
                        (unsigned long long)m, (unsigned long long)n);
    return FBG_OK;
}

int fbg_msa_set_device(fbg_ctx *ctx, const uint8_t *d_msa, uint64_t m, uint64_t n)
{
    FBG_TRY(check_dims(ctx, m, n));
    if (!d_msa) return fbg_fail(ctx, FBG_ERR_INVALID, "null MSA pointer");
    ctx->d_msa = d_msa; ctx->m = m; ctx->n = n; ctx->index_valid = false;
    return FBG_OK;
}

int fbg_msa_load_host(fbg_ctx *ctx, const uint8_t *msa, uint64_t m, uint64_t n)
{
    FBG_TRY(check_dims(ctx, m, n));
    if (!msa) return fbg_fail(ctx, FBG_ERR_INVALID, "null MSA pointer");
    FBG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    FBG_TRY(fbg_reserve(ctx, ctx->msa_own, m * n));
    FBG_TRY(fbg_upload(ctx, ctx->msa_own.p, msa, m * n));
    ctx->d_msa = ctx->msa_own.as<uint8_t>(); ctx->m = m; ctx->n = n; ctx->index_valid = false;
    return FBG_OK;
}

int fbg_index_build(fbg_ctx *ctx, int reversed, const uint8_t *ignore_chars, uint64_t ignore_len)
{
    if (!ctx) return FBG_ERR_INVALID;
    if (!ctx->d_msa) return fbg_fail(ctx, FBG_ERR_INVALID, "fbg_index_build: no MSA set");
    FBG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    for (int st : {FBG_STAGE_SORT_PASS1, FBG_STAGE_SORT_PASS2, FBG_STAGE_SORT_PASS3, FBG_STAGE_RANK_KERNEL}) {   // (stages only some paths reach)
        ctx->timers[st].recorded = false;
        ctx->timers[st].launches = 0;
    }
    ctx->index_valid = false;
    ctx->granked = false;
    ctx->spanned = false;
    ctx->gpart = false;
    ctx->reversed = reversed ? 1 : 0;
    FBG_TRY(fbg_build_text(ctx, reversed ? nullptr : ignore_chars, reversed ? 0 : ignore_len));
    FBG_TRY(fbg_suffix_sort(ctx));
    FBG_TRY(fbg_neighbour_lcp(ctx));
    if (ctx->N > 1500000000ull) {
        // keep the footprint of huge indexes down -- where the device is filling up: giving 50 GB back and taking them again costs
        // the next build seconds of hipFree / hipMalloc (4.2 s measured for 2 * 10^9 symbols, ten times the build itself)
        size_t free_b = 0, total_b = 0;
        FBG_HIP_TRY(ctx, hipMemGetInfo(&free_b, &total_b));
        if (free_b < total_b / 2) FBG_TRY(fbg_release_scratch(ctx));
    }
    ctx->index_valid = true;
    return FBG_OK;
}

int fbg_part_index_build(fbg_ctx *ctx, int reversed, int part, int nparts, void *d_blob, int *ok)
{
    return fbg_part_index_build_ignore(ctx, reversed, part, nparts, nullptr, 0, d_blob, ok);
}

int fbg_part_index_build_ignore(fbg_ctx *ctx, int reversed, int part, int nparts, const uint8_t *ignore_chars, uint64_t ignore_len, void *d_blob, int *ok)
{
    if (!ctx) return FBG_ERR_INVALID;
    if (!ctx->d_msa) return fbg_fail(ctx, FBG_ERR_INVALID, "fbg_part_index_build: no MSA set");
    if (!d_blob || !ok || nparts < 1 || part < 0 || part >= nparts || (ignore_len && !ignore_chars))
        return fbg_fail(ctx, FBG_ERR_INVALID, "fbg_part_index_build: bad arguments");
    FBG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->index_valid = false;
    ctx->ranked = false;
    ctx->granked = false;
    ctx->spanned = false;
    ctx->reversed = reversed ? 1 : 0;
    ctx->gpart = false;
    ctx->allow_wide = true;                        // the partitions together may hold a text of 2^32 symbols and more
    const int rc = fbg_build_text(ctx, reversed ? nullptr : ignore_chars, reversed ? 0 : ignore_len);
    ctx->allow_wide = false;
    FBG_TRY(rc);
    return fbg_part_sort(ctx, part, nparts, static_cast<uint8_t *>(d_blob), ok);
}

int fbg_part_scan(fbg_ctx *ctx, const void *d_blobs, uint32_t *d_gmax, int *ok)
{
    if (!ctx) return FBG_ERR_INVALID;
    if (!ctx->part_active || ctx->index_valid)
        return fbg_fail(ctx, FBG_ERR_INVALID, "fbg_part_scan needs fbg_part_index_build first");
    if (!d_blobs || !d_gmax || !ok) return fbg_fail(ctx, FBG_ERR_INVALID, "fbg_part_scan: bad arguments");
    FBG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->gpart) return fbg_grs_part_scan(ctx, static_cast<const uint8_t *>(d_blobs), d_gmax, ok);
    return fbg_rank_part_runs(ctx, static_cast<const uint8_t *>(d_blobs), d_gmax, ok);
}

int fbg_part_finish(fbg_ctx *ctx, const uint32_t *d_gmax, int *ok)
{
    if (!ctx) return FBG_ERR_INVALID;
    if (!ctx->part_active || ctx->index_valid)
        return fbg_fail(ctx, FBG_ERR_INVALID, "fbg_part_finish needs fbg_part_scan first");
    if (!d_gmax || !ok) return fbg_fail(ctx, FBG_ERR_INVALID, "fbg_part_finish: bad arguments");
    FBG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    uint32_t verdict = 1;
    FBG_HIP_TRY(ctx, hipMemcpyAsync(ctx->gmax.p, d_gmax, (ctx->n + 1) * 4, hipMemcpyDeviceToDevice, ctx->stream));
    FBG_HIP_TRY(ctx, hipMemcpyAsync(&verdict, d_gmax + ctx->n, 4, hipMemcpyDeviceToHost, ctx->stream));
    FBG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *ok = verdict == 0;
    if (*ok) {
        // a column whose maximum is below the threshold some partition scanned with may have lost a larger value
        uint64_t unfilled = 0;
        if (ctx->gpart) FBG_TRY(fbg_grs_part_unfilled(ctx, &unfilled));
        else FBG_TRY(fbg_rank_part_unfilled(ctx, &unfilled));
        if (unfilled) { *ok = 2; return FBG_OK; }
        ctx->n_exc = 0; ctx->index_valid = true;
        if (ctx->gpart) ctx->granked = true; else ctx->ranked = true;
    }
    return FBG_OK;
}

int fbg_part_rescan(fbg_ctx *ctx, uint32_t *d_gmax)
{
    if (!ctx) return FBG_ERR_INVALID;
    if (!ctx->part_active || ctx->index_valid)
        return fbg_fail(ctx, FBG_ERR_INVALID, "fbg_part_rescan follows a fbg_part_finish that returned *ok = 2");
    if (!d_gmax) return fbg_fail(ctx, FBG_ERR_INVALID, "fbg_part_rescan: bad arguments");
    FBG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    uint32_t verdict = 0;
    if (ctx->gpart) { FBG_TRY(fbg_grs_part_rescan(ctx)); verdict = ctx->grs_part_failed ? 1u : 0u; }
    else FBG_TRY(fbg_rank_part_rescan(ctx));
    FBG_HIP_TRY(ctx, hipMemcpyAsync(d_gmax, ctx->gmax.p, (ctx->n + 1) * 4, hipMemcpyDeviceToDevice, ctx->stream));
    FBG_HIP_TRY(ctx, hipMemcpyAsync(d_gmax + ctx->n, &verdict, 4, hipMemcpyHostToDevice, ctx->stream));
    FBG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return FBG_OK;
}

int fbg_scan_f(fbg_ctx *ctx, uint64_t x0, uint64_t x1, int disable_tricks, uint64_t *d_f)
{
    if (!ctx) return FBG_ERR_INVALID;
    if (!ctx->index_valid || ctx->reversed)
        return fbg_fail(ctx, FBG_ERR_INVALID, "fbg_scan_f needs fbg_index_build(reversed=0) first");
    if (x0 > x1 || x1 > ctx->n || !d_f) return fbg_fail(ctx, FBG_ERR_INVALID, "fbg_scan_f: bad column range");
    FBG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return fbg_scan_columns(ctx, x0, x1, FBG_SCAN_F, disable_tricks, d_f);
}

int fbg_scan_v(fbg_ctx *ctx, uint64_t x0, uint64_t x1, uint64_t *d_v)
{
    if (!ctx) return FBG_ERR_INVALID;
    if (!ctx->index_valid || !ctx->reversed)
        return fbg_fail(ctx, FBG_ERR_INVALID, "fbg_scan_v needs fbg_index_build(reversed=1) first");
    if (x0 > x1 || x1 > ctx->n || !d_v) return fbg_fail(ctx, FBG_ERR_INVALID, "fbg_scan_v: bad column range");
    FBG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return fbg_scan_columns(ctx, x0, x1, FBG_SCAN_V, 1, d_v);
}

int fbg_minmax_dp_device(fbg_ctx *ctx, const uint64_t *d_f, uint64_t n, uint64_t *d_boundaries,
                         uint64_t *count_out, uint64_t *d_mml, uint64_t *d_bt)
{
    if (!ctx) return FBG_ERR_INVALID;
    if (!d_f || !d_boundaries || !count_out || n == 0 || n >= (1ull << 31))
        return fbg_fail(ctx, FBG_ERR_INVALID, "fbg_minmax_dp_device: bad arguments");
    FBG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return fbg_dp_minmax(ctx, d_f, n, d_boundaries, count_out, d_mml, d_bt);
}

int fbg_repeatfree_dp_device(fbg_ctx *ctx, const uint64_t *d_v, uint64_t n, uint64_t *d_s,
                             uint64_t *d_prev, uint64_t *d_boundaries, uint64_t *count_out)
{
    if (!ctx) return FBG_ERR_INVALID;
    if (!d_v || !d_boundaries || !count_out || n == 0 || n >= (1ull << 31))
        return fbg_fail(ctx, FBG_ERR_INVALID, "fbg_repeatfree_dp_device: bad arguments");
    FBG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return fbg_dp_repeatfree(ctx, d_v, n, d_s, d_prev, d_boundaries, count_out);
}

int fbg_scan_gapped_v(fbg_ctx *ctx, uint64_t *d_v)
{
    if (!ctx) return FBG_ERR_INVALID;
    if (!ctx->index_valid || ctx->reversed)
        return fbg_fail(ctx, FBG_ERR_INVALID, "fbg_scan_gapped_v needs fbg_index_build(reversed=0) first");
    if (!d_v) return fbg_fail(ctx, FBG_ERR_INVALID, "fbg_scan_gapped_v: null d_v");
    FBG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint64_t n = ctx->n;
    FBG_TRY(fbg_reserve(ctx, ctx->io_d, n * sizeof(uint64_t)));
    uint64_t *d_f = ctx->io_d.as<uint64_t>();
    FBG_HIP_TRY(ctx, hipMemsetAsync(d_f, 0, n * sizeof(uint64_t), ctx->stream));
    FBG_TRY(fbg_scan_columns(ctx, 0, n, FBG_SCAN_F, 1, d_f));        // f without the elastic tricks
    return fbg_gapped_v_from_f(ctx, d_f, n, d_v);
}

int fbg_gapped_dp_device(fbg_ctx *ctx, const uint64_t *d_v, uint64_t n, uint64_t *d_s,
                         uint64_t *d_prev, uint64_t *d_boundaries, uint64_t *count_out)
{
    if (!ctx) return FBG_ERR_INVALID;
    if (!d_v || !d_boundaries || !count_out || n == 0 || n >= (1ull << 31))
        return fbg_fail(ctx, FBG_ERR_INVALID, "fbg_gapped_dp_device: bad arguments");
    FBG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return fbg_dp_gapped(ctx, d_v, n, d_s, d_prev, d_boundaries, count_out);
}

// ---- host-buffer entry points ---------------------------------------------------------------

int fbg_elastic_f(fbg_ctx *ctx, const uint8_t *msa, uint64_t m, uint64_t n, const uint8_t *ignore_chars,
                  uint64_t ignore_len, int disable_tricks, uint64_t *f)
{
    if (!ctx) return FBG_ERR_INVALID;
    if (!f) return fbg_fail(ctx, FBG_ERR_INVALID, "fbg_elastic_f: null f");
    if (msa && ignore_len == 0 && !ctx->opt.no_stream_upload && m * n >= FBG_STAGE_MIN && host_pointer_is_pinned(msa) && check_dims(ctx, m, n) == FBG_OK) {
        // the rows go up in chunks while the index build starts on those that are there (text_build.hip)
        FBG_HIP_TRY(ctx, hipSetDevice(ctx->device));
        FBG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        FBG_TRY(fbg_reserve(ctx, ctx->msa_own, m * n));
        ctx->d_msa = ctx->msa_own.as<uint8_t>(); ctx->m = m; ctx->n = n; ctx->index_valid = false;
        ctx->up_host = msa;
        const int rc = fbg_index_build(ctx, 0, ignore_chars, ignore_len);
        ctx->up_host = nullptr;
        if (ctx->up_stream) (void)hipStreamSynchronize(ctx->up_stream);   // (the caller's memory is free again whatever happened)
        if (rc != FBG_OK) return rc;
    } else {
        FBG_TRY(fbg_msa_load_host(ctx, msa, m, n));
        FBG_TRY(fbg_index_build(ctx, 0, ignore_chars, ignore_len));
    }
    FBG_TRY(fbg_reserve(ctx, ctx->io_a, n * sizeof(uint64_t)));
    FBG_HIP_TRY(ctx, hipMemcpyAsync(ctx->io_a.p, f, n * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    FBG_TRY(fbg_scan_f(ctx, 0, n, disable_tricks, ctx->io_a.as<uint64_t>()));
    FBG_HIP_TRY(ctx, hipMemcpyAsync(f, ctx->io_a.p, n * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    FBG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (disable_tricks && f[0] == n)
        return fbg_fail(ctx, FBG_ERR_NO_SEGMENTATION, "No valid segmentation found!");
    return FBG_OK;
}

int fbg_minmax_dp(fbg_ctx *ctx, const uint64_t *f, uint64_t n, uint64_t *boundaries_out, uint64_t *count_out,
                  uint64_t *minmaxlength_out, uint64_t *backtrack_out)
{
    if (!ctx) return FBG_ERR_INVALID;
    if (!f || !boundaries_out || !count_out || n == 0 || n >= (1ull << 31))
        return fbg_fail(ctx, FBG_ERR_INVALID, "fbg_minmax_dp: bad arguments");
    FBG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    size_t w = (n + 1) * sizeof(uint64_t);
    FBG_TRY(fbg_reserve(ctx, ctx->io_a, w));
    FBG_TRY(fbg_reserve(ctx, ctx->io_b, w));
    FBG_TRY(fbg_reserve(ctx, ctx->io_c, w));
    FBG_TRY(fbg_reserve(ctx, ctx->io_d, w));
    FBG_HIP_TRY(ctx, hipMemcpyAsync(ctx->io_a.p, f, n * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    int rc = fbg_dp_minmax(ctx, ctx->io_a.as<uint64_t>(), n, ctx->io_b.as<uint64_t>(), count_out,
                           ctx->io_c.as<uint64_t>(), ctx->io_d.as<uint64_t>());
    if (rc != FBG_OK && rc != FBG_ERR_NO_SEGMENTATION) return rc;
    if (minmaxlength_out)
        FBG_HIP_TRY(ctx, hipMemcpyAsync(minmaxlength_out, ctx->io_c.p, w, hipMemcpyDeviceToHost, ctx->stream));
    if (backtrack_out)
        FBG_HIP_TRY(ctx, hipMemcpyAsync(backtrack_out, ctx->io_d.p, w, hipMemcpyDeviceToHost, ctx->stream));
    if (rc == FBG_OK)
        FBG_HIP_TRY(ctx, hipMemcpyAsync(boundaries_out, ctx->io_b.p, *count_out * sizeof(uint64_t),
                                        hipMemcpyDeviceToHost, ctx->stream));
    FBG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return rc;
}

int fbg_repeatfree_v(fbg_ctx *ctx, const uint8_t *msa, uint64_t m, uint64_t n, uint64_t *v)
{
    if (!ctx) return FBG_ERR_INVALID;
    if (!v) return fbg_fail(ctx, FBG_ERR_INVALID, "fbg_repeatfree_v: null v");
    FBG_TRY(fbg_msa_load_host(ctx, msa, m, n));
    FBG_TRY(fbg_index_build(ctx, 1, nullptr, 0));
    FBG_TRY(fbg_reserve(ctx, ctx->io_a, n * sizeof(uint64_t)));
    FBG_TRY(fbg_scan_v(ctx, 0, n, ctx->io_a.as<uint64_t>()));
    FBG_HIP_TRY(ctx, hipMemcpyAsync(v, ctx->io_a.p, n * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    FBG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return FBG_OK;
}

int fbg_repeatfree_dp(fbg_ctx *ctx, const uint64_t *v, uint64_t n, uint64_t *s_out, uint64_t *prev_out,
                      uint64_t *boundaries_out, uint64_t *count_out)
{
    if (!ctx) return FBG_ERR_INVALID;
    if (!v || !boundaries_out || !count_out || n == 0 || n >= (1ull << 31))
        return fbg_fail(ctx, FBG_ERR_INVALID, "fbg_repeatfree_dp: bad arguments");
    FBG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    size_t w = n * sizeof(uint64_t);
    FBG_TRY(fbg_reserve(ctx, ctx->io_a, w));
    FBG_TRY(fbg_reserve(ctx, ctx->io_b, w));
    FBG_TRY(fbg_reserve(ctx, ctx->io_c, w));
    FBG_TRY(fbg_reserve(ctx, ctx->io_d, w));
    FBG_HIP_TRY(ctx, hipMemcpyAsync(ctx->io_a.p, v, w, hipMemcpyHostToDevice, ctx->stream));
    int rc = fbg_dp_repeatfree(ctx, ctx->io_a.as<uint64_t>(), n, ctx->io_c.as<uint64_t>(),
                               ctx->io_d.as<uint64_t>(), ctx->io_b.as<uint64_t>(), count_out);
    if (rc != FBG_OK && rc != FBG_ERR_NO_SEGMENTATION) return rc;
    if (s_out) FBG_HIP_TRY(ctx, hipMemcpyAsync(s_out, ctx->io_c.p, w, hipMemcpyDeviceToHost, ctx->stream));
    if (prev_out) FBG_HIP_TRY(ctx, hipMemcpyAsync(prev_out, ctx->io_d.p, w, hipMemcpyDeviceToHost, ctx->stream));
    if (rc == FBG_OK)
        FBG_HIP_TRY(ctx, hipMemcpyAsync(boundaries_out, ctx->io_b.p, *count_out * sizeof(uint64_t),
                                        hipMemcpyDeviceToHost, ctx->stream));
    FBG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return rc;
}

int fbg_gapped_v(fbg_ctx *ctx, const uint8_t *msa, uint64_t m, uint64_t n, uint64_t *v)
{
    if (!ctx) return FBG_ERR_INVALID;
    if (!v) return fbg_fail(ctx, FBG_ERR_INVALID, "fbg_gapped_v: null v");
    FBG_TRY(fbg_msa_load_host(ctx, msa, m, n));
    FBG_TRY(fbg_index_build(ctx, 0, nullptr, 0));
    FBG_TRY(fbg_reserve(ctx, ctx->io_a, n * sizeof(uint64_t)));
    FBG_TRY(fbg_scan_gapped_v(ctx, ctx->io_a.as<uint64_t>()));
    FBG_HIP_TRY(ctx, hipMemcpyAsync(v, ctx->io_a.p, n * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    FBG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return FBG_OK;
}

int fbg_gapped_dp(fbg_ctx *ctx, const uint64_t *v, uint64_t n, uint64_t *s_out, uint64_t *prev_out,
                  uint64_t *boundaries_out, uint64_t *count_out)
{
    if (!ctx) return FBG_ERR_INVALID;
    if (!v || !boundaries_out || !count_out || n == 0 || n >= (1ull << 31))
        return fbg_fail(ctx, FBG_ERR_INVALID, "fbg_gapped_dp: bad arguments");
    FBG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    size_t w = n * sizeof(uint64_t);
    FBG_TRY(fbg_reserve(ctx, ctx->io_a, w));
    FBG_TRY(fbg_reserve(ctx, ctx->io_b, w));
    FBG_TRY(fbg_reserve(ctx, ctx->io_c, w));
    FBG_TRY(fbg_reserve(ctx, ctx->io_d, w));
    FBG_HIP_TRY(ctx, hipMemcpyAsync(ctx->io_a.p, v, w, hipMemcpyHostToDevice, ctx->stream));
    int rc = fbg_dp_gapped(ctx, ctx->io_a.as<uint64_t>(), n, ctx->io_c.as<uint64_t>(),
                           ctx->io_d.as<uint64_t>(), ctx->io_b.as<uint64_t>(), count_out);
    if (rc != FBG_OK && rc != FBG_ERR_NO_SEGMENTATION) return rc;
    if (s_out) FBG_HIP_TRY(ctx, hipMemcpyAsync(s_out, ctx->io_c.p, w, hipMemcpyDeviceToHost, ctx->stream));
    if (prev_out) FBG_HIP_TRY(ctx, hipMemcpyAsync(prev_out, ctx->io_d.p, w, hipMemcpyDeviceToHost, ctx->stream));
    if (rc == FBG_OK)
        FBG_HIP_TRY(ctx, hipMemcpyAsync(boundaries_out, ctx->io_b.p, *count_out * sizeof(uint64_t),
                                        hipMemcpyDeviceToHost, ctx->stream));
    FBG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return rc;
}

uint64_t fbg_text_length(const fbg_ctx *ctx) { return ctx && ctx->index_valid ? ctx->N : 0; }

int fbg_index_download(fbg_ctx *ctx, uint8_t *text, uint32_t *sa, uint32_t *isa, uint32_t *lcp_prev,
                       uint32_t *lcp_next)
{
    if (!ctx) return FBG_ERR_INVALID;
    if (!ctx->index_valid) return fbg_fail(ctx, FBG_ERR_INVALID, "fbg_index_download: no index");
    if (ctx->part_active) return fbg_fail(ctx, FBG_ERR_INVALID, "fbg_index_download: the index is partitioned over several GPUs");
    if (ctx->N >= (1ull << 32)) return fbg_fail(ctx, FBG_ERR_TOO_LARGE, "fbg_index_download: 32-bit arrays");
    if (ctx->granked && ctx->spanned && (sa || isa || lcp_prev || lcp_next))
        return fbg_fail(ctx, FBG_ERR_INVALID, "fbg_index_download: the group-level scan (span_scan.hip) leaves the suffixes with equal keys unordered; "
                                              "set option span_scan = -1 for an index with a suffix array");
    size_t N = ctx->N;
    // test / debugging API: plain blocking copies
    FBG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (text) FBG_HIP_TRY(ctx, hipMemcpy(text, ctx->text.p, N, hipMemcpyDeviceToHost));
    if (sa && !ctx->ranked && !ctx->granked) FBG_HIP_TRY(ctx, hipMemcpy(sa, ctx->sa_ptr, N * 4, hipMemcpyDeviceToHost));
    if ((sa || isa || lcp_prev || lcp_next) && (ctx->ranked || ctx->granked)) {
        // rank-order index: no per-position records exist; derive the arrays from the sorted keys once
        FBG_TRY(fbg_reserve(ctx, ctx->io_a, N * 4));
        FBG_TRY(fbg_reserve(ctx, ctx->io_b, N * 4));
        FBG_TRY(fbg_reserve(ctx, ctx->io_c, N * 4));
        FBG_TRY(fbg_reserve(ctx, ctx->io_d, N * 4));
        if (ctx->granked)
            FBG_TRY(fbg_grs_materialize(ctx, ctx->io_d.as<uint32_t>(), ctx->io_a.as<uint32_t>(), ctx->io_b.as<uint32_t>(), ctx->io_c.as<uint32_t>()));
        else
            FBG_TRY(fbg_rank_materialize(ctx, ctx->io_d.as<uint32_t>(), ctx->io_a.as<uint32_t>(), ctx->io_b.as<uint32_t>(),
                                         ctx->io_c.as<uint32_t>()));
        FBG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (sa) FBG_HIP_TRY(ctx, hipMemcpy(sa, ctx->io_d.p, N * 4, hipMemcpyDeviceToHost));
        if (isa) FBG_HIP_TRY(ctx, hipMemcpy(isa, ctx->io_a.p, N * 4, hipMemcpyDeviceToHost));
        if (lcp_prev) FBG_HIP_TRY(ctx, hipMemcpy(lcp_prev, ctx->io_b.p, N * 4, hipMemcpyDeviceToHost));
        if (lcp_next) FBG_HIP_TRY(ctx, hipMemcpy(lcp_next, ctx->io_c.p, N * 4, hipMemcpyDeviceToHost));
    } else if (isa || lcp_prev || lcp_next) {
        // records {rank, lcp_prev|hint, lcp_next|hint, -}: pull one word column at a time (strided 2-D copy)
        uint32_t *dst[3] = {isa, lcp_prev, lcp_next};
        for (int w = 0; w < 3; w++) {
            if (!dst[w]) continue;
            FBG_HIP_TRY(ctx, hipMemcpy2DAsync(dst[w], 4, (const char *)ctx->rec.p + 4 * w, 16, 4, N,
                                              hipMemcpyDeviceToHost, ctx->stream));
        }
        FBG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        for (int w = 1; w < 3; w++)
            if (dst[w]) for (size_t k = 0; k < N; k++) dst[w][k] &= 0x7fffffffu;   // strip the run hints
    }
    FBG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return FBG_OK;
}

} // extern "C"
