// common.h -- internal definitions shared by the HIP translation units of libofxcv_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <mutex>
#include <shared_mutex>
#include <vector>

#include "ofxcv_hip.h"

// One growable HBM scratch allocation (never shrinks; freed with the context).
struct DevBuf {
    void *ptr = nullptr;
    size_t bytes = 0;
    void *raw = nullptr;  // what hipMalloc returned (ptr = raw + the placement skew of ofxcv_reserve)
};

// cache of captured Farneback launch sequences (see ofxcv_calc_optical_flow_farneback)
#define OFXCV_FB_MAX_LEVELS 10
#define OFXCV_FB_MAX_BATCH OFXCV_FARNEBACK_MAX_BATCH  // frame pairs per batched call (pointer tables travel as kernel arguments)
constexpr int kFbGraphSlots = 12;  // (the batch context of the host path's submission queue replays calls of 1 .. 16 pairs)
struct FbGraphKey {  // compared with memcmp: zero-filled before it is set
    int n, width, height, levels, winsize, iterations, poly_n, flags;
    double pyr_scale, poly_sigma;
    const void *planes, *tmp, *cflow, *vsum;  // scratch addresses baked into the graph
    const void *prev[OFXCV_FB_MAX_BATCH], *next[OFXCV_FB_MAX_BATCH], *flow[OFXCV_FB_MAX_BATCH];
    size_t prev_step[OFXCV_FB_MAX_BATCH], next_step[OFXCV_FB_MAX_BATCH], flow_step[OFXCV_FB_MAX_BATCH];
    const void *rgba[OFXCV_FB_MAX_BATCH];  // F7 fused into the call (ofxcv_calc_optical_flow_farneback_batch_rgba)
    ptrdiff_t rgba_step[OFXCV_FB_MAX_BATCH];
    unsigned rgba_mu[OFXCV_FB_MAX_BATCH], rgba_mv[OFXCV_FB_MAX_BATCH];
    double rsx, rsy;
};
struct FbGraph {
    FbGraphKey key;
    hipGraphExec_t exec = nullptr;
    unsigned long used = 0;  // stamp of its last replay: the least recently used entry goes
};

struct ofxcv_ctx {
    int device = 0;      // LOGICAL device: index of the per-device caches and queues
    int hip_device = 0;  // the HIP device behind it (OFXCV_VIRTUAL_DEVICES=N maps N logical devices onto the physical ones: device % physical count); index of the runtime lock with OFXCV_LOCK_PER_DEVICE
    long lock_hold_ns = 0, lock_holds = 0;  // time this context's Farneback calls spent holding the runtime lock exclusively (graph capture / launch)
    hipStream_t compute = nullptr;  // default stream for kernels when the caller passes NULL
    hipStream_t copy = nullptr;     // H2D / D2H staging stream of the host-buffer entry points
    hipEvent_t ev_h2d[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_done = nullptr;
    hipStream_t prep = nullptr;     // Farneback: pyramid + polynomial expansion of all levels, ahead of the level walk
    hipStream_t coarse = nullptr;   // Farneback: the coarse (latency-bound) pyramid levels of the walk, at high priority (option below)
    hipEvent_t ev_fork = nullptr, ev_level[OFXCV_FB_MAX_LEVELS + 1] = {}, ev_coarse = nullptr;
    int fb_priority = 0;            // environment OFXCV_STREAM_PRIORITY at context creation: 0 (default) all streams alike; 1 preparation stream
                                    // at high priority; 2 also the coarse levels of the walk, on a high-priority stream of their own.
                                    // Measured (profiles/r03_scheduling.txt): +1-3 % with several calls in flight, but single calls on a
                                    // context with a high-priority stream were seen to run 3x slower (236 instead of 725 pairs/s): off.
    FbGraph fb_graphs[kFbGraphSlots];
    unsigned long fb_graph_clock = 0;
    // option "farneback.graph": 0 (default since round 6) the ~110 launches of a call are enqueued one by one -- 0.3 ms of host time for a call of 8 pairs,
    // the first kernels run while the rest is enqueued, and NO runtime lock is held; 1 the call is captured once per (pointers, geometry, parameters) and
    // replayed with one hipGraphLaunch under the runtime lock (rounds 1-5; measured equal within the box-to-box spread: tools/graph_vs_eager.py)
    bool fb_no_graph = true;
    bool fb_one_stream = false;     // pyramid + polynomial expansion on the call's own stream (set when a graph capture could not be completed)
    int fb_pyr_mode = 1;            // option "farneback.fused_pyramid": 1 (default) LDS-fused / direct pyramid kernels; 0 the two-pass kernels; test hooks: 2 the coarse
                                    // levels by the general byte-wise tile kernel (cross-check of pyr_fused_al_kernel), 3 the 3-tap levels without the wavefront-row form
    int fb_gauss_generation = 3;    // option "farneback.gaussian_kernel_generation": getGaussianKernel of OpenCV 2.4 / 3.x (3) or 4.x (4)
    int fb_filter_contraction = 0;  // option "farneback.filter_contraction": 1 = the separable filters of the pyramid and resize's vertical lerp as fused multiply-adds (OpenCV 4.x AVX2 / NEON paths); 0 = scalar order (2.4 / 3.x)
    int fb_resize_generation = 0;   // option "farneback.resize_generation": association of cv::resize's exact-2x INTER_AREA rewrite (farneback.hip: resize_combine)
    int num_cus = 256;
    int max_lds = 160 * 1024;       // LDS a workgroup may use on this device (hipDeviceAttributeMaxSharedMemoryPerBlock)
    bool is_gfx950 = true;          // LDS-DMA of 16 bytes per lane (buffer_load_dwordx4 ... lds) exists on gfx950 only: the R1 ring of iterate_col_kernel
    hipStream_t last_stream = nullptr;  // last caller-supplied stream (ofxcv_stream)
    char err[512] = {0};

    // F0: 65536-entry 8.8 fixed-point sRGB table (openfx-supportext ofxsLut.h semantics)
    uint16_t *d_srgb_lut = nullptr;

    // Farneback scratch (sized for the largest frame seen so far)
    DevBuf fb_planes;  // R0, R1, M0, M1: 4 fields x 5 planes
    DevBuf fb_tmp;     // blurred half-resolution rows (pyramid) + pyramid image I
    DevBuf fb_flow;    // two ping-pong coarse flow fields
    DevBuf fb_coef;    // polyexp / blur coefficient tables
    DevBuf fb_vsum;    // f64 column sums of the OpenCV-rounding validation mode
    int fb_opencv_rounding = 1;  // 1 (default) OpenCV's running-sum order, strip-parallel; 0 direct window sums (fast opt-in); 2 OpenCV's order as a serial column scan
    int fb_halo_geom = 0;        // option "farneback.halo_geom": test hook of the overlapped-strip form's geometry (encoding: farneback.hip halo_geom); 0 = by level size
    int lut_luma601 = 0;         // option "lut.luma" 709 (default) | 601: luma weights of the gray conversion (supportext's are not verifiable here)
    int fb_col = 1;              // option "farneback.col": column-owning form (iterate_col_kernel: two steps of a level per launch) on the levels whose launches fill the chip
    int fb_col_min = 128;        // option "farneback.col_min": workgroups (tile columns x pairs) below which no launch takes that form; from there on a cost model decides per level how many pairs do (col_pairs in farneback.hip); values below the default force the form (tests)
    int fb_col_spin = 1 << 22;   // option "farneback.col_spin": polls of one LDS wait before the kernel raises the abort word
    int fb_col_trace = 0;        // option "farneback.col_trace": the (iterate, iterate) launches run the instantiation that stamps the shader clock per phase (ofxcv_debug_col_trace)
    int fb_reuse_prep = 0;       // environment OFXCV_DEBUG_REUSE_PREP=1, read per call (measurement probe, not an option): skip the pyramid images / polynomial expansions, the scratch still holds those of the same frames
    int fb_col_ring = 1;         // option "farneback.col_ring": the step pairs of the column-owning form that open with an iteration gather R1 from a ring of rows in LDS filled by LDS-DMA (1, default); 0 = every gather from memory (cross-check, A/B)
    DevBuf fb_col_flag;          // the trace area of iterate_col_kernel (farneback.col_trace)
    unsigned *fb_col_abort = nullptr;  // the abort word of iterate_col_kernel: 64 bytes of pinned, host-coherent memory the kernel stores to when a bounded
                                       // LDS wait runs out; read (and cleared) by the host at every synchronisation point of the library without a copy
    long fb_col_aborts_seen = 0;       // calls that were reported as failed because of it
    int fb_batch_mb = 160;       // option "farneback.batch_mb": a pyramid level is walked with as many pairs per launch as keep its
                                 // working set (80 B/px per pair) under this many MiB (Infinity Cache: 256 MiB), at least one

    // inpaint scratch
    DevBuf ip_tmp;   // undilated mask
    DevBuf ip_maps;  // distance / order maps and the per-level pixel lists of the colour fill
    DevBuf ip_img;   // device copies of the host images (render_host)
    DevBuf ip_work;  // 4-byte-per-pixel working images of the colour fill
    DevBuf ip_flag;  // error flag of the dataflow fill (a poll gave up)
    DevBuf ip_trace; // OFXCV_FILL_TRACE: phase stamps of the fill (measurement hook)
    DevBuf ip_tmap, ip_omap;  // persistent padded distance / order maps (defaults everywhere between calls)
    int ip_map_w = 0, ip_map_h = 0;
    void *ip_host_state = nullptr;           // host state of the front march (inpaint.hip: March)
    void (*ip_host_state_free)(void *) = nullptr;
    void *ip_pinned = nullptr;   // pinned mirror of the per-pixel upload arrays of the pipelined fill
    size_t ip_pinned_bytes = 0;
    int ip_portion = 0;      // option "inpaint.portion": fill-order pixels per portion of the pipelined fill (0 = default)
    DevBuf ip_sched2; // level schedule of the fall-back fill
    int ip_max_tiles = 0;    // option "inpaint.max_tiles": workgroups (tiles) per fill launch; 0 = the chip's share of this call (192 / concurrent fills, at least 48)
    int ip_tiles = 1;        // option "inpaint.tiles": tile schedule of the dataflow fill (a workgroup per occupied tile of a portion, hand-offs inside a tile through LDS); 0 = component schedule (every hand-off through the L2)
    int ip_spin_limit = -1;  // option "inpaint.spin_limit" (tests force the fall-back with 0)
    int ip_parallel_march = 0;  // option "inpaint.parallel_march": 0 (default) serial front march, pipelined with the fill; 1 the hole's 4-connected
                                // components marched side by side on host threads and merged into the exact fill order (from 8192 hole pixels; n > 1:
                                // from n).  Measured at 1920x1080 (profiles/r03_inpaint_march.txt): the march itself 6.2 -> 2.9 + 2.6 ms, but the call
                                // 11.7 -> 13.6 ms: the GPU fill (a dependency chain of ~8 ms) is the long pole and now starts 2.9 ms later.
    long ip_fallbacks = 0;   // fills that were repeated with the barrier-scheduled kernel
    DevBuf seg_work; // mean-shift pyramid (source + result per level) and mask

    // measurement hook: event pairs around the dominant kernel (see ofxcv_profile_enable)
    int prof_on = 0;        // 0 off, 1 event pairs around the dominant kernel, 2 around the carry pre-pass of the OpenCV-order mode
    bool prof_now = false;  // set by the level walk for the launches that are to be bracketed
    std::vector<hipEvent_t> prof_ev;  // start/stop pairs, recorded but not yet read
    double prof_ms = 0;
    long prof_launches = 0;

    int host_register = 1;         // option "host.register": 0 stage through the pinned ring; 1 (default) copies straight from / into the host's
                                   // pageable images; 2 the host's images registered for the duration of the call (zero copy)
    int host_split = 2;            // option "host.split": the two flows of an output frame as two single-pair calls, the first one while the
                                   // third frame is still on the wire (1), as one batched call after the third upload (0), or 1 when this
                                   // is the only host-image call in flight in the process and 0 otherwise (2, default)
    long host_split_calls = 0;
    int host_coalesce = 1;         // option "host.coalesce": host-image calls that find another one in flight on their device hand their frame pairs to the
                                   // device's submission queue, which runs everything queued as ONE batched Farneback call (1, default); 0 never; 2 always
                                   // (also a lone call: tests)
    int host_coalesce_max = 0;     // option "host.coalesce_max": frame pairs per coalesced call (2 .. OFXCV_FARNEBACK_MAX_BATCH); 0 (default) = one round of the chip
                                   // in the column-owning form of level 0 (8 pairs at 1920x1080, 4 at 3840x2160)
    int host_coalesce_min = 4;     // option "host.coalesce_min": host-image calls in flight on the device (this one included) from which a call goes to the queue
    long host_coalesced_calls = 0, host_coalesced_pairs = 0, host_coalesced_batches = 0;  // calls of this context served by the queue, their pairs, the pairs of the batches they rode in
    int fb_reserve_pairs = 0;      // the Farneback scratch is sized for at least this many pairs (the batch context of the submission queue: no re-allocation as batches grow)
    int host_cache_mb = 512;       // option "host.cache_mb": budget of the device's cache of named frames' gray images (0 = off)
    long host_cache_hits = 0, host_cache_misses = 0;  // named frames of this context's calls found on the device / uploaded and kept
    long host_direct_calls = 0;
    long host_zero_copy_calls = 0, host_staged_calls = 0;

    // host-path staging
    DevBuf d_stage;            // device side: 2 f32 frames, 2 gray frames, flow, rgba
    void *h_pinned = nullptr;  // pinned host ring
    size_t h_pinned_bytes = 0;
};

int ofxcv_fail(ofxcv_ctx *ctx, int status, const char *fmt, ...);
int ofxcv_reserve(ofxcv_ctx *ctx, DevBuf &b, size_t bytes);

#define OFXCV_HIP_CHECK(ctx, expr)                                                                    \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess)                                                                         \
            return ofxcv_fail((ctx), _e == hipErrorOutOfMemory ? OFXCV_ERR_MEMORY : OFXCV_ERR_HIP,    \
                              "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

#define OFXCV_LAUNCH_CHECK(ctx, name)                                                                 \
    do {                                                                                              \
        hipError_t _e = hipGetLastError();                                                            \
        if (_e != hipSuccess)                                                                         \
            return ofxcv_fail((ctx), OFXCV_ERR_HIP, "launch of %s failed: %s", name, hipGetErrorString(_e)); \
    } while (0)

static inline hipStream_t ofxcv_stream(ofxcv_ctx *ctx, void *stream) {
    if (stream) ctx->last_stream = reinterpret_cast<hipStream_t>(stream);  // may still be using the scratch buffers
    return stream ? reinterpret_cast<hipStream_t>(stream) : ctx->compute;
}

// Process-wide lock for the HIP runtime operations that are not safe against each other across host threads on ROCm 7.2:
// stream capture + graph instantiation, stream creation, device allocations / frees / host registration, graph-exec
// destruction, context teardown AND hipGraphLaunch.  History: a capture in flight was invalidated by another thread's allocation (round 1);
// hipGraphExecDestroy tears down the exec's internal streams while another thread's hipGraphLaunch walks the runtime's
// stream list (segfault in hip::Graph::UpdateStreams, about 1 in 10 runs of four concurrent render threads); and two
// threads inside hipGraphLaunch at the same time crash in the same function (1 in 30 runs), as does a launch racing a
// stream creation (4 in 150).  Kernel launches, copies and
// synchronisation stay concurrent; a graph launch costs the host 0.4 ms, far below the GPU time of the call it starts.
// (A shared_mutex because readers may come back if a later runtime makes concurrent launches safe.)
// One lock per device: the structures that raced are per-device (the stream list a graph launch walks), and a host
// process that drives several GPUs from its render threads must not serialise all of them on one lock.
int ofxcv_lock_index(const ofxcv_ctx *ctx);  // ctx->hip_device (debug environment OFXCV_LOCK_BY_LOGICAL=1: ctx->device, the round-5 indexing)
std::shared_mutex &ofxcv_capture_mutex(int hip_device);  // (the PHYSICAL device: logical devices that share a GPU share its lock)
// Waits for everything this context has in flight (its own streams and the last caller-supplied one); never a
// device-wide synchronisation, which would stall -- and invalidate the captures of -- other contexts' threads.
int ofxcv_ctx_quiesce(ofxcv_ctx *ctx);

static inline int ofxcv_div_up(int a, int b) { return (a + b - 1) / b; }

// Host image rows <-> a contiguous device image.  OFX row strides may be negative (bottom-up images) or carry
// padding: a non-negative stride goes through hipMemcpy2DAsync, anything else through a contiguous host copy.
// The download variant has finished (stream synchronised) when it returns only in the second case.
int ofxcv_upload_rows(ofxcv_ctx *ctx, void *d_dst, size_t row, const void *h_src, ptrdiff_t src_row_bytes, int rows, hipStream_t s);
int ofxcv_download_rows(ofxcv_ctx *ctx, void *h_dst, ptrdiff_t dst_row_bytes, const void *d_src, size_t row, int rows, hipStream_t s);

// after a synchronisation of the context's streams: OFXCV_ERR_HIP (once; the word is cleared) if a bounded wait of iterate_col_kernel ran out since the last check
int ofxcv_col_abort_check(ofxcv_ctx *ctx);
int ofxcv_farneback_streams(ofxcv_ctx *ctx);  // lazily creates the preparation stream and the per-level events

// measurement hook helpers (context.hip)
int ofxcv_prof_mark(ofxcv_ctx *ctx, hipStream_t s);  // records one event of a start/stop pair
// F7 as a launch of its own on stream s (lut.hip); capture-safe
int ofxcv_launch_flow_to_rgba(ofxcv_ctx *ctx, hipStream_t s, const float *d_flow, size_t flow_step, int width, int height, float *d_dst,
                              ptrdiff_t dst_row_bytes, unsigned chan_u_mask, unsigned chan_v_mask, double render_scale_x, double render_scale_y);
int ofxcv_prof_drain(ofxcv_ctx *ctx);

// host-side cvRound (round half to even) / cvFloor, shared by geometry helpers
int ofxcv_cv_round(double v);
