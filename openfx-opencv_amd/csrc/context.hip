// context.hip -- device context, scratch management and error reporting of libofxcv_hip.so.
#include <cmath>
#include <cstdlib>
#include <new>

#include <algorithm>
#include "common.h"

int ofxcv_fail(ofxcv_ctx *ctx, int status, const char *fmt, ...) {
    if (ctx) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
        va_end(ap);
    }
    return status;
}

// The lock around the runtime operations that were seen to crash against each other on ROCm 7.2 (stream capture, graph
// instantiation / launch / destruction, stream creation, device allocations and frees, host registration, context teardown).
// Several of them touch process-global runtime state (memory-object maps, capture bookkeeping), so the lock is PROCESS-WIDE.
// OFXCV_LOCK_PER_DEVICE=1 selects one lock per PHYSICAL device (it used to be indexed by the logical device, so with
// OFXCV_VIRTUAL_DEVICES several "per-device" locks covered one GPU and admitted exactly the concurrent hipGraphLaunch pairs the lock exists
// to prevent -- the round-5 soak that did not return ran in that configuration).  The per-device form has still not run on two physical
// devices (one-GPU boxes only); since round 6 a Farneback call launches eagerly and takes no lock at all, so what is left under the lock is
// rare (allocations when a scratch grows, stream creation, cross-context event operations of a few microseconds).
std::shared_mutex &ofxcv_capture_mutex(int hip_device) {
    static std::shared_mutex m[64];
    static const bool per_device = [] {
        const char *e = std::getenv("OFXCV_LOCK_PER_DEVICE");
        return e && e[0] == '1';
    }();
    return m[per_device ? ((unsigned)hip_device & 63u) : 0u];
}
// (debug, tools/soak_hang_hunt.sh: OFXCV_LOCK_BY_LOGICAL=1 restores the round-5 indexing -- one lock per LOGICAL device -- to reproduce that round's hang)
int ofxcv_lock_index(const ofxcv_ctx *ctx) {
    static const bool by_logical = [] {
        const char *e = std::getenv("OFXCV_LOCK_BY_LOGICAL");
        return e && e[0] == '1';
    }();
    return by_logical ? ctx->device : ctx->hip_device;
}

int ofxcv_ctx_quiesce(ofxcv_ctx *ctx) {
    hipStream_t streams[] = {ctx->compute, ctx->copy, ctx->prep, ctx->coarse, ctx->last_stream};
    for (hipStream_t st : streams)
        if (st) OFXCV_HIP_CHECK(ctx, hipStreamSynchronize(st));
    return OFXCV_OK;
}

int ofxcv_reserve(ofxcv_ctx *ctx, DevBuf &b, size_t bytes) {
    if (bytes <= b.bytes) return OFXCV_OK;
    std::unique_lock<std::shared_mutex> lock(ofxcv_capture_mutex(ofxcv_lock_index(ctx)));
    if (b.ptr) {
        int rc = ofxcv_ctx_quiesce(ctx);
        if (rc) return rc;
        OFXCV_HIP_CHECK(ctx, hipFree(b.raw ? b.raw : b.ptr));
        b.ptr = b.raw = nullptr;
        b.bytes = 0;
    }
    // measurement aid (tools/col_time.py): environment OFXCV_SCRATCH_SKEW=bytes places every scratch buffer that far (a multiple of 256) behind the
    // address hipMalloc returned -- does a kernel's time depend on where its fields lie in the channel interleave?
    size_t skew = 0;
    if (const char *e = std::getenv("OFXCV_SCRATCH_SKEW")) skew = (size_t)std::strtoull(e, nullptr, 0) & ~(size_t)255;
    OFXCV_HIP_CHECK(ctx, hipMalloc(&b.raw, bytes + skew));
    b.ptr = (char *)b.raw + skew;
    b.bytes = bytes;
    return OFXCV_OK;
}

int ofxcv_upload_rows(ofxcv_ctx *ctx, void *d_dst, size_t row, const void *h_src, ptrdiff_t src_row_bytes, int rows, hipStream_t s) {
    if (src_row_bytes >= (ptrdiff_t)row) {
        OFXCV_HIP_CHECK(ctx, hipMemcpy2DAsync(d_dst, row, h_src, (size_t)src_row_bytes, row, rows, hipMemcpyHostToDevice, s));
        return OFXCV_OK;
    }
    std::vector<char> tmp(row * rows);
    for (int y = 0; y < rows; y++) std::memcpy(tmp.data() + (size_t)y * row, (const char *)h_src + (ptrdiff_t)y * src_row_bytes, row);
    OFXCV_HIP_CHECK(ctx, hipMemcpyAsync(d_dst, tmp.data(), row * rows, hipMemcpyHostToDevice, s));
    OFXCV_HIP_CHECK(ctx, hipStreamSynchronize(s));  // `tmp` dies with this scope
    return OFXCV_OK;
}

int ofxcv_download_rows(ofxcv_ctx *ctx, void *h_dst, ptrdiff_t dst_row_bytes, const void *d_src, size_t row, int rows, hipStream_t s) {
    if (dst_row_bytes >= (ptrdiff_t)row) {
        OFXCV_HIP_CHECK(ctx, hipMemcpy2DAsync(h_dst, (size_t)dst_row_bytes, d_src, row, row, rows, hipMemcpyDeviceToHost, s));
        return OFXCV_OK;
    }
    std::vector<char> tmp(row * rows);
    OFXCV_HIP_CHECK(ctx, hipMemcpyAsync(tmp.data(), d_src, row * rows, hipMemcpyDeviceToHost, s));
    OFXCV_HIP_CHECK(ctx, hipStreamSynchronize(s));
    for (int y = 0; y < rows; y++) std::memcpy((char *)h_dst + (ptrdiff_t)y * dst_row_bytes, tmp.data() + (size_t)y * row, row);
    return OFXCV_OK;
}

int ofxcv_prof_mark(ofxcv_ctx *ctx, hipStream_t s) {
    hipEvent_t e;
    OFXCV_HIP_CHECK(ctx, hipEventCreate(&e));
    ctx->prof_ev.push_back(e);
    OFXCV_HIP_CHECK(ctx, hipEventRecord(e, s));
    return OFXCV_OK;
}

int ofxcv_prof_drain(ofxcv_ctx *ctx) {
    for (size_t i = 0; i + 1 < ctx->prof_ev.size(); i += 2) {
        float ms = 0;
        OFXCV_HIP_CHECK(ctx, hipEventSynchronize(ctx->prof_ev[i + 1]));
        OFXCV_HIP_CHECK(ctx, hipEventElapsedTime(&ms, ctx->prof_ev[i], ctx->prof_ev[i + 1]));
        ctx->prof_ms += ms;
        ctx->prof_launches++;
    }
    for (hipEvent_t e : ctx->prof_ev) (void)hipEventDestroy(e);
    ctx->prof_ev.clear();
    return OFXCV_OK;
}

int ofxcv_farneback_streams(ofxcv_ctx *ctx) {
    if (ctx->prep) return OFXCV_OK;
    std::unique_lock<std::shared_mutex> lock(ofxcv_capture_mutex(ofxcv_lock_index(ctx)));  // stream creation: see ofxcv_ctx_create
    int lo = 0, hi = 0;  // (numerically lower = higher priority)
    OFXCV_HIP_CHECK(ctx, hipDeviceGetStreamPriorityRange(&lo, &hi));
    if (ctx->fb_priority >= 1) OFXCV_HIP_CHECK(ctx, hipStreamCreateWithPriority(&ctx->prep, hipStreamNonBlocking, hi));
    else OFXCV_HIP_CHECK(ctx, hipStreamCreateWithFlags(&ctx->prep, hipStreamNonBlocking));
    if (ctx->fb_priority >= 2) {
        OFXCV_HIP_CHECK(ctx, hipStreamCreateWithPriority(&ctx->coarse, hipStreamNonBlocking, hi));
        OFXCV_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->ev_coarse, hipEventDisableTiming));
    }
    OFXCV_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    for (hipEvent_t &e : ctx->ev_level) OFXCV_HIP_CHECK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return OFXCV_OK;
}

int ofxcv_col_abort_check(ofxcv_ctx *ctx) {
    if (!ctx->fb_col_abort) return OFXCV_OK;
    volatile unsigned *w = ctx->fb_col_abort;
    if (!*w) return OFXCV_OK;
    *w = 0;  // reported once: later calls on this context are judged on their own
    ctx->fb_col_aborts_seen++;
    return ofxcv_fail(ctx, OFXCV_ERR_HIP, "calc_optical_flow_farneback: a bounded wait inside iterate_col_kernel ran out; the flows of that call are not valid");
}

int ofxcv_cv_round(double v) { return (int)std::lrint(v); }

extern "C" {

// OFXCV_VIRTUAL_DEVICES=N (> 0): N LOGICAL devices over the physical ones (logical d -> physical d % count).  Everything that is per device in
// a host process -- the render threads' device choice, the per-device runtime lock (OFXCV_LOCK_PER_DEVICE=1), the per-device caches of named
// frames -- then runs as on an N-GPU node on a box with one GPU (tests, tools/bench_host_threads.py --devices N).
static int ofxcv_physical_devices() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
int ofxcv_device_count(void) {
    const int n = ofxcv_physical_devices();
    if (n <= 0) return 0;
    const char *e = std::getenv("OFXCV_VIRTUAL_DEVICES");
    const int v = e ? atoi(e) : 0;
    return v > 0 ? std::min(v, 64) : n;
}

const char *ofxcv_status_string(int status) {
    switch (status) {
        case OFXCV_OK: return "ok";
        case OFXCV_ERR_INVALID: return "invalid argument";
        case OFXCV_ERR_HIP: return "HIP runtime error";
        case OFXCV_ERR_MEMORY: return "out of memory";
        case OFXCV_ERR_UNSUPPORTED: return "unsupported parameters";
        case OFXCV_ERR_NO_DEVICE: return "no usable gfx950 device";
        default: return "unknown status";
    }
}

int ofxcv_ctx_create(int device, ofxcv_ctx **out) {
    if (!out) return OFXCV_ERR_INVALID;
    *out = nullptr;
    const int n = ofxcv_physical_devices();
    if (n <= 0 || device < 0 || device >= ofxcv_device_count()) return OFXCV_ERR_NO_DEVICE;
    ofxcv_ctx *ctx = new (std::nothrow) ofxcv_ctx();
    if (!ctx) return OFXCV_ERR_MEMORY;
    ctx->device = device;
    ctx->hip_device = device % n;
    if (const char *e = getenv("OFXCV_STREAM_PRIORITY")) ctx->fb_priority = atoi(e);
    int rc = OFXCV_OK;
    auto init = [&]() -> int {
        // stream creation changes the runtime's stream list, which another thread's hipGraphLaunch walks: under the runtime lock
        std::unique_lock<std::shared_mutex> lock(ofxcv_capture_mutex(ofxcv_lock_index(ctx)));
        OFXCV_HIP_CHECK(ctx, hipSetDevice(ctx->hip_device));
        OFXCV_HIP_CHECK(ctx, hipDeviceGetAttribute(&ctx->num_cus, hipDeviceAttributeMultiprocessorCount, ctx->hip_device));
        OFXCV_HIP_CHECK(ctx, hipDeviceGetAttribute(&ctx->max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, ctx->hip_device));
        {
            hipDeviceProp_t prop;
            OFXCV_HIP_CHECK(ctx, hipGetDeviceProperties(&prop, ctx->hip_device));
            ctx->is_gfx950 = std::strncmp(prop.gcnArchName, "gfx950", 6) == 0;
        }
        OFXCV_HIP_CHECK(ctx, hipStreamCreateWithFlags(&ctx->compute, hipStreamNonBlocking));
        OFXCV_HIP_CHECK(ctx, hipStreamCreateWithFlags(&ctx->copy, hipStreamNonBlocking));
        for (int i = 0; i < 3; i++) OFXCV_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->ev_h2d[i], hipEventDisableTiming));
        OFXCV_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->ev_done, hipEventDisableTiming));
        return OFXCV_OK;
    };
    rc = init();
    if (rc != OFXCV_OK) {
        fprintf(stderr, "ofxcv_ctx_create: %s\n", ctx->err);
        ofxcv_ctx_destroy(ctx);
        return rc;
    }
    *out = ctx;
    return OFXCV_OK;
}

void ofxcv_ctx_destroy(ofxcv_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->hip_device);
    std::unique_lock<std::shared_mutex> lock(ofxcv_capture_mutex(ofxcv_lock_index(ctx)));
    (void)ofxcv_ctx_quiesce(ctx);
    for (hipEvent_t e : ctx->prof_ev) (void)hipEventDestroy(e);
    for (FbGraph &g : ctx->fb_graphs)
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    for (hipEvent_t e : ctx->ev_level)
        if (e) (void)hipEventDestroy(e);
    if (ctx->prep) (void)hipStreamDestroy(ctx->prep);
    if (ctx->coarse) (void)hipStreamDestroy(ctx->coarse);
    if (ctx->ev_coarse) (void)hipEventDestroy(ctx->ev_coarse);
    DevBuf *bufs[] = {&ctx->fb_planes, &ctx->fb_tmp, &ctx->fb_flow, &ctx->fb_coef, &ctx->fb_vsum, &ctx->fb_col_flag, &ctx->d_stage, &ctx->ip_tmp, &ctx->ip_maps, &ctx->ip_img, &ctx->ip_work, &ctx->ip_flag, &ctx->ip_trace, &ctx->ip_sched2, &ctx->ip_tmap, &ctx->ip_omap, &ctx->seg_work};
    for (DevBuf *b : bufs)
        if (b->ptr) (void)hipFree(b->raw ? b->raw : b->ptr);
    if (ctx->ip_host_state && ctx->ip_host_state_free) ctx->ip_host_state_free(ctx->ip_host_state);
    if (ctx->d_srgb_lut) (void)hipFree(ctx->d_srgb_lut);
    if (ctx->fb_col_abort) (void)hipHostFree(ctx->fb_col_abort);
    if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
    if (ctx->ip_pinned) (void)hipHostFree(ctx->ip_pinned);
    for (int i = 0; i < 3; i++)
        if (ctx->ev_h2d[i]) (void)hipEventDestroy(ctx->ev_h2d[i]);
    if (ctx->ev_done) (void)hipEventDestroy(ctx->ev_done);
    if (ctx->compute) (void)hipStreamDestroy(ctx->compute);
    if (ctx->copy) (void)hipStreamDestroy(ctx->copy);
    delete ctx;
}

long ofxcv_host_zero_copy_calls(const ofxcv_ctx *ctx) { return ctx ? ctx->host_zero_copy_calls : -1; }

long ofxcv_host_direct_calls(const ofxcv_ctx *ctx) { return ctx ? ctx->host_direct_calls : -1; }

long ofxcv_inpaint_fallback_count(const ofxcv_ctx *ctx) { return ctx ? ctx->ip_fallbacks : -1; }

const char *ofxcv_last_error(const ofxcv_ctx *ctx) { return ctx ? ctx->err : "null context"; }

int ofxcv_ctx_device(const ofxcv_ctx *ctx) { return ctx ? ctx->device : -1; }

// measurement: nanoseconds this context's Farneback calls have held the runtime lock exclusively, and how many times
int ofxcv_lock_hold(const ofxcv_ctx *ctx, long *ns, long *holds) {
    if (!ctx || !ns || !holds) return OFXCV_ERR_INVALID;
    *ns = ctx->lock_hold_ns;
    *holds = ctx->lock_holds;
    return OFXCV_OK;
}

void *ofxcv_ctx_stream(const ofxcv_ctx *ctx) { return ctx ? (void *)ctx->compute : nullptr; }

int ofxcv_ctx_set_option(ofxcv_ctx *ctx, const char *name, int value) {
    if (!ctx || !name) return OFXCV_ERR_INVALID;
    // captured launch sequences bake the kernel choice in: drop them whenever an option changes
    std::unique_lock<std::shared_mutex> lock(ofxcv_capture_mutex(ofxcv_lock_index(ctx)));
    for (FbGraph &g : ctx->fb_graphs)
        if (g.exec) {
            (void)hipGraphExecDestroy(g.exec);
            g.exec = nullptr;
        }
    int luma = ctx->lut_luma601 ? 601 : 709, graph = ctx->fb_no_graph ? 0 : 1;
    // every option: name, target, accepted range (include/ofxcv_hip.h documents them; 25 in all)
    struct { const char *n; int *v; int lo, hi; } knobs[] = {
        // what the results are
        {"farneback.opencv_rounding", &ctx->fb_opencv_rounding, 0, 2}, {"farneback.gaussian_kernel_generation", &ctx->fb_gauss_generation, 3, 4},
        {"farneback.filter_contraction", &ctx->fb_filter_contraction, 0, 1}, {"farneback.resize_generation", &ctx->fb_resize_generation, 0, 2},
        {"lut.luma", &luma, 601, 709},
        // how a Farneback call is planned and launched (never changes a result)
        {"farneback.graph", &graph, 0, 1}, {"farneback.col", &ctx->fb_col, 0, 1}, {"farneback.col_min", &ctx->fb_col_min, 1, 1 << 30},
        {"farneback.col_ring", &ctx->fb_col_ring, 0, 1}, {"farneback.col_spin", &ctx->fb_col_spin, 1, 1 << 30}, {"farneback.col_trace", &ctx->fb_col_trace, 0, 1},
        {"farneback.batch_mb", &ctx->fb_batch_mb, 1, 1 << 20}, {"farneback.halo_geom", &ctx->fb_halo_geom, 0, (72 << 8) | 0x7f},
        {"farneback.fused_pyramid", &ctx->fb_pyr_mode, 0, 3},
        // host images
        {"host.register", &ctx->host_register, 0, 2}, {"host.split", &ctx->host_split, 0, 2}, {"host.cache_mb", &ctx->host_cache_mb, 0, 1 << 18},
        {"host.coalesce", &ctx->host_coalesce, 0, 2}, {"host.coalesce_max", &ctx->host_coalesce_max, 0, OFXCV_FARNEBACK_MAX_BATCH},
        {"host.coalesce_min", &ctx->host_coalesce_min, 1, 64},
        // inpaint
        {"inpaint.tiles", &ctx->ip_tiles, 0, 1}, {"inpaint.max_tiles", &ctx->ip_max_tiles, 0, 240}, {"inpaint.spin_limit", &ctx->ip_spin_limit, -1, 1 << 30},
        {"inpaint.portion", &ctx->ip_portion, 0, 1 << 30}, {"inpaint.parallel_march", &ctx->ip_parallel_march, 0, 1 << 30}};
    for (auto &k : knobs)
        if (!std::strcmp(name, k.n)) {
            if (value < k.lo || value > k.hi || (k.v == &luma && value != 601 && value != 709))
                return ofxcv_fail(ctx, OFXCV_ERR_INVALID, "option '%s': %d outside %d..%d", name, value, k.lo, k.hi);
            *k.v = value;
            ctx->lut_luma601 = luma == 601;
            ctx->fb_no_graph = graph == 0;
            return OFXCV_OK;
        }
    return ofxcv_fail(ctx, OFXCV_ERR_INVALID, "unknown option '%s'", name);
}

int ofxcv_ctx_get_option(const ofxcv_ctx *ctx, const char *name, int *value) {
    if (!ctx || !name || !value) return OFXCV_ERR_INVALID;
    if (!std::strcmp(name, "farneback.opencv_rounding")) *value = ctx->fb_opencv_rounding;
    else if (!std::strcmp(name, "farneback.gaussian_kernel_generation")) *value = ctx->fb_gauss_generation;
    else if (!std::strcmp(name, "farneback.resize_generation")) *value = ctx->fb_resize_generation;
    else if (!std::strcmp(name, "farneback.filter_contraction")) *value = ctx->fb_filter_contraction;
    else if (!std::strcmp(name, "lut.luma")) *value = ctx->lut_luma601 ? 601 : 709;
    else if (!std::strcmp(name, "farneback.graph")) *value = ctx->fb_no_graph ? 0 : 1;
    else if (!std::strcmp(name, "host.register")) *value = ctx->host_register;
    else if (!std::strcmp(name, "host.split")) *value = ctx->host_split;
    else if (!std::strcmp(name, "host.cache_mb")) *value = ctx->host_cache_mb;
    else if (!std::strcmp(name, "host.split_calls")) *value = (int)ctx->host_split_calls;
    else if (!std::strcmp(name, "host.coalesce")) *value = ctx->host_coalesce;
    else if (!std::strcmp(name, "farneback.batch_mb")) *value = ctx->fb_batch_mb;
    else if (!std::strcmp(name, "farneback.col")) *value = ctx->fb_col;
    else if (!std::strcmp(name, "farneback.col_min")) *value = ctx->fb_col_min;
    else if (!std::strcmp(name, "farneback.col_aborts")) {
        // the abort word of the column-owning kernel (waits for the context's streams first); reading it through this getter does not clear it
        *value = 0;
        if (ctx->fb_col_abort) {
            if (hipSetDevice(ctx->hip_device) != hipSuccess || ofxcv_ctx_quiesce(const_cast<ofxcv_ctx *>(ctx)) != OFXCV_OK) return OFXCV_ERR_HIP;
            *value = (int)(*(volatile unsigned *)ctx->fb_col_abort | (ctx->fb_col_aborts_seen ? 1u : 0u));
        }
    }
    else return OFXCV_ERR_INVALID;
    return OFXCV_OK;
}

int ofxcv_profile_enable(ofxcv_ctx *ctx, int enable) {
    if (!ctx) return OFXCV_ERR_INVALID;
    ctx->prof_on = enable < 0 ? 0 : (enable > 2 ? 1 : enable);
    return OFXCV_OK;
}

int ofxcv_profile_read(ofxcv_ctx *ctx, double *total_ms, long *launches, int reset) {
    if (!ctx || !total_ms || !launches) return OFXCV_ERR_INVALID;
    int rc = ofxcv_prof_drain(ctx);
    if (rc) return rc;
    *total_ms = ctx->prof_ms;
    *launches = ctx->prof_launches;
    if (reset) {
        ctx->prof_ms = 0;
        ctx->prof_launches = 0;
    }
    return OFXCV_OK;
}

int ofxcv_ctx_synchronize(ofxcv_ctx *ctx, void *stream) {
    if (!ctx) return OFXCV_ERR_INVALID;
    OFXCV_HIP_CHECK(ctx, hipStreamSynchronize(ofxcv_stream(ctx, stream)));
    // the abort word of iterate_col_kernel: a bounded wait that ran out leaves wrong flows behind -- fail loudly here, at the library's own
    // synchronisation point (the waits terminate by construction, this is for a faulting device); a read of pinned memory, no copy
    int rc = ofxcv_col_abort_check(ctx);
    if (rc) return rc;
    return OFXCV_OK;
}

}  // extern "C"
