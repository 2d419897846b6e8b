// vectorgen.hip -- VectorGeneratorPlugin::calcOpticalFlow (VectorGenerator/VectorGenerator.cpp:353-520,
// Farneback branch) for host-resident OFX images.
//
// The reference marshals each OFX image into a cv::Mat through CVImageWrapper / OFX::ImageMemory
// (OpenCV/GenericOpenCVPlugin.cpp:58-165, 223-265).  Three ways here (context option "host.register"):
//  * direct (1, default): asynchronous copies straight from the host's pageable frames on the copy stream while the compute
//    stream converts the frames that have arrived; the forward and the backward flow of an output frame run as ONE batched
//    Farneback call; with all four destination channels mapped a kernel composes the RGBA image in HBM and one copy takes
//    it into the host's image.  On the MI355X box the runtime moves pageable memory at the rate of pinned memory, and calls
//    of several render threads overlap (profiles/r03_host_overlap.txt, r03_host_path_threads.txt: 410 / 782 / 730 pairs/s
//    for 1 / 2 / 4 calling threads).
//  * registered buffers (2): the host's own buffers are registered with the driver for the duration of the call
//    (hipHostRegister, no cache -- a registration does not survive the host freeing and re-allocating the addresses), the
//    copy engine reads the f32 frames in place, and one kernel stores the four flow channels of both directions straight
//    into the host's destination image.  Same speed as the direct form for one calling thread, but registering and
//    unregistering stalls other threads' GPU work (407 / 444 / 434 pairs/s).
//  * pinned ring (0; and whatever the others cannot address: bottom-up / oddly strided images): the frames are copied in
//    row blocks into a pinned ring and sent to HBM with hipMemcpyAsync.
// With a partial channel map only the flows (8 B/px) come back and the mapped channels of the destination are filled on
// the host (unmapped channels stay untouched, :507-516).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <condition_variable>
#include <functional>
#include <memory>
#include <string>
#include <thread>

#include "common.h"

namespace {

// A few helper threads for the two host-side copies of this path (frames into the pinned ring, flow into the
// destination image): one core moves ~25 GB/s, the PCIe link takes 55.  One job at a time; a caller that finds the pool
// busy (several render threads at once already use several cores) simply does its copy itself.  The pool is created on
// first use and never torn down (no static destructor joins threads while a host unloads the plugin).
class HostPool {
    std::mutex m_, user_;
    std::condition_variable cv_job_, cv_done_;
    const std::function<void(int)> *job_ = nullptr;
    std::atomic<int> next_{0};
    int ntasks_ = 0, active_ = 0, nworkers_ = 0;
    unsigned long gen_ = 0;

    void worker() {
        unsigned long seen = 0;
        for (;;) {
            const std::function<void(int)> *job;
            int n;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_job_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                job = job_;
                n = ntasks_;
            }
            for (int i; (i = next_.fetch_add(1)) < n;) (*job)(i);
            {
                std::lock_guard<std::mutex> lk(m_);
                if (--active_ == 0) cv_done_.notify_one();
            }
        }
    }
    HostPool() {
        const unsigned hc = std::thread::hardware_concurrency();
        nworkers_ = hc >= 8 ? 3 : (hc >= 4 ? 1 : 0);
        for (int i = 0; i < nworkers_; i++) std::thread([this] { worker(); }).detach();
    }

public:
    static HostPool &get() {
        static HostPool *p = new HostPool();  // intentionally leaked
        return *p;
    }
    void run(int n, const std::function<void(int)> &fn) {
        std::unique_lock<std::mutex> user(user_, std::try_to_lock);
        if (!user.owns_lock() || nworkers_ == 0 || n <= 1) {
            for (int i = 0; i < n; i++) fn(i);
            return;
        }
        {
            std::lock_guard<std::mutex> lk(m_);
            job_ = &fn;
            ntasks_ = n;
            next_.store(0);
            active_ = nworkers_;
            ++gen_;
        }
        cv_job_.notify_all();
        for (int i; (i = next_.fetch_add(1)) < n;) fn(i);
        std::unique_lock<std::mutex> lk(m_);
        cv_done_.wait(lk, [&] { return active_ == 0; });
        job_ = nullptr;
    }
};

__global__ __launch_bounds__(256) void scale_flow_kernel(float2 *__restrict__ flow, size_t n, double rsx, double rsy) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float2 f = flow[i];
    f.x = (float)(f.x / rsx);
    f.y = (float)(f.y / rsy);
    flow[i] = f;
}

int reserve_pinned(ofxcv_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->h_pinned_bytes) return OFXCV_OK;
    std::unique_lock<std::shared_mutex> lock(ofxcv_capture_mutex(ofxcv_lock_index(ctx)));
    if (ctx->h_pinned) {
        int rc = ofxcv_ctx_quiesce(ctx);
        if (rc) return rc;
        OFXCV_HIP_CHECK(ctx, hipHostFree(ctx->h_pinned));
        ctx->h_pinned = nullptr;
        ctx->h_pinned_bytes = 0;
    }
    OFXCV_HIP_CHECK(ctx, hipHostMalloc(&ctx->h_pinned, bytes, hipHostMallocDefault));
    ctx->h_pinned_bytes = bytes;
    return OFXCV_OK;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

// ---- converted frames kept in HBM across calls ----
// Consecutive output frames of a sequence share two of their three source frames (frame t needs t-1, t, t+1), and what the
// flow needs of a source frame is its 8-bit gray image: 1/16 of the f32 RGBA pixels the host hands over.  A caller that can
// NAME the pixels of a frame (OFX hosts do: kOfxImagePropUniqueIdentifier changes whenever an image's pixels change) passes
// that name with the frame (ofxcv_vectorgen_flows_host_keyed); the gray image of a named frame stays on the device -- one
// cache per device, shared by the contexts of all render threads, `host.cache_mb` of the 288 GB (512 MB = 240 frames at
// 1920x1080) -- and a frame that is found there is neither uploaded nor converted again: per output frame of a sequence one
// upload instead of three.  Entries are pinned while a call uses them and evicted least-recently-used; an entry is
// published to other threads only after the work that fills it has been enqueued and its event recorded (they wait for the
// event on their own stream).  The cache trusts the names: a caller that reuses a name for different pixels gets the old
// frame -- exactly the contract of the OFX property.
namespace {
struct GrayEntry {
    std::string key;
    int w = 0, h = 0, ncomp = 0;
    int luma601 = 0;  // the conversion's luma weights (option "lut.luma"): part of what the gray bytes are
    size_t bytes = 0;
    uint8_t *ptr = nullptr;
    hipEvent_t ready = nullptr;
    int pins = 0;
    bool recorded = false, failed = false;
    unsigned long stamp = 0;
};
class GrayCache {
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<GrayEntry *> entries_;
    size_t bytes_ = 0, budget_ = 0;
    unsigned long clock_ = 0;

    void drop(size_t i, ofxcv_ctx *ctx) {  // mu_ held; entry unpinned: nothing of it is in flight (its users synchronised before unpinning)
        GrayEntry *e = entries_[i];
        {
            std::unique_lock<std::shared_mutex> lock(ofxcv_capture_mutex(ofxcv_lock_index(ctx)));
            if (e->ptr) (void)hipFree(e->ptr);
            if (e->ready) (void)hipEventDestroy(e->ready);
            (void)hipGetLastError();
        }
        bytes_ -= e->bytes;
        delete e;
        entries_.erase(entries_.begin() + (ptrdiff_t)i);
    }

public:
    static GrayCache &of(int device) {
        static GrayCache *c = new GrayCache[64];  // intentionally leaked (no static destructor while a host unloads the plugin)
        return c[device & 63];
    }
    // The entry of (key, geometry), pinned; *fill = the caller has to produce it (and call published()).  nullptr: not cacheable
    // right now (budget used up by pinned entries, allocation failure, a producer that failed) -- the caller takes the plain path.
    GrayEntry *acquire(ofxcv_ctx *ctx, const char *key, int w, int h, int ncomp, size_t bytes, size_t budget, bool *fill, bool *pending) {
        *fill = *pending = false;
        std::unique_lock<std::mutex> lk(mu_);
        for (GrayEntry *e : entries_)
            if (e->w == w && e->h == h && e->ncomp == ncomp && e->luma601 == ctx->lut_luma601 && !e->failed && e->key == key) {
                // Another thread may still be enqueueing the work that fills it (*pending).  Not waited for here: a caller takes all its
                // frames first, and two callers that each fill what the other waits for would wait for ever -- wait_published() is
                // called after the caller has published its own entries.
                *pending = !e->recorded;
                e->pins++;
                e->stamp = ++clock_;
                return e;
            }
        // failed entries nobody uses any more, then least-recently-used ones, make room
        for (size_t i = entries_.size(); i-- > 0;)
            if (entries_[i]->failed && entries_[i]->pins == 0) drop(i, ctx);
        // The budget is the cache's, not the caller's: the value the latest caller brought (contexts of one host share an option default; a
        // host that lowers it means it for the device).  Evictions go on until the new entry fits -- also after an evicted entry's buffer
        // has been taken over (the take-over used to end the loop without looking at the budget again).
        budget_ = budget;
        GrayEntry *e = nullptr;
        while (bytes_ + (e ? 0 : bytes) > budget_ - (e ? std::min(budget_, bytes) : 0)) {
            size_t lru = entries_.size();
            for (size_t i = 0; i < entries_.size(); i++)
                if (entries_[i]->pins == 0 && (lru == entries_.size() || entries_[i]->stamp < entries_[lru]->stamp)) lru = i;
            if (lru == entries_.size()) {
                if (e) break;  // (nothing else can go: the taken-over buffer is used all the same -- it was inside the budget a moment ago)
                return nullptr;
            }
            if (!e && entries_[lru]->bytes == bytes) {
                // the usual case (a sequence has one frame size): the evicted entry's buffer and event are taken over as they are --
                // no hipFree (it waits for the whole device) and no hipMalloc on the path of a render call
                e = entries_[lru];
                entries_.erase(entries_.begin() + (ptrdiff_t)lru);
                bytes_ -= bytes;
                continue;
            }
            drop(lru, ctx);
        }
        if (!e) {
            e = new GrayEntry();
            std::unique_lock<std::shared_mutex> lock(ofxcv_capture_mutex(ofxcv_lock_index(ctx)));
            if (hipMalloc((void **)&e->ptr, bytes) != hipSuccess || hipEventCreateWithFlags(&e->ready, hipEventDisableTiming) != hipSuccess) {
                (void)hipGetLastError();
                if (e->ptr) (void)hipFree(e->ptr);
                delete e;
                return nullptr;
            }
        }
        e->recorded = e->failed = false;
        e->key = key;
        e->w = w; e->h = h; e->ncomp = ncomp;
        e->luma601 = ctx->lut_luma601;
        e->bytes = bytes;
        e->pins = 1;
        e->stamp = ++clock_;
        bytes_ += bytes;
        entries_.push_back(e);
        *fill = true;
        return e;
    }
    void published(GrayEntry *e, bool ok) {  // the producer has recorded e->ready behind the work that fills the entry (or given up)
        {
            std::lock_guard<std::mutex> lk(mu_);
            e->recorded = ok;
            e->failed = !ok;
        }
        cv_.notify_all();
    }
    bool wait_published(GrayEntry *e) {  // false: its producer gave up
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return e->recorded || e->failed; });
        return e->recorded;
    }
    void release(GrayEntry *e) {
        std::lock_guard<std::mutex> lk(mu_);
        e->pins--;
    }
    void stats(size_t &bytes, int &n) {
        std::lock_guard<std::mutex> lk(mu_);
        bytes = bytes_;
        n = (int)entries_.size();
    }
    // every unpinned entry (tests; a host that wants the memory back)
    void clear(ofxcv_ctx *ctx) {
        std::lock_guard<std::mutex> lk(mu_);
        for (size_t i = entries_.size(); i-- > 0;)
            if (entries_[i]->pins == 0) drop(i, ctx);
    }
};
}  // namespace

// ---- registered host buffers ----
// Registered for the duration of ONE call: a registration pins the pages behind an address range at that moment; a host
// that frees a frame buffer and gets the same addresses back from its allocator would leave a cached registration
// pointing at the old pages (measured: a pointer-keyed cache produced wrong frames with numpy-allocated buffers).
// Register / unregister take the exclusive runtime lock like every other memory operation.
struct HostRegistrations {
    ofxcv_ctx *ctx;
    void *p[4];
    int n = 0;
    explicit HostRegistrations(ofxcv_ctx *c) : ctx(c) {}
    bool add(const void *ptr, size_t bytes) {
        std::unique_lock<std::shared_mutex> lock(ofxcv_capture_mutex(ofxcv_lock_index(ctx)));
        if (hipHostRegister(const_cast<void *>(ptr), bytes, hipHostRegisterDefault) != hipSuccess) {
            (void)hipGetLastError();  // e.g. already registered by the host application itself (or by another render thread
            return false;             // reading the same source frame): this call stages through the ring
        }
        p[n++] = const_cast<void *>(ptr);
        return true;
    }
    // Every way out of the call -- error returns included -- passes here: whatever the call still has in flight (DMA out of
    // the registered frames, the kernel that stores into the registered destination) is waited for BEFORE the ranges are
    // unregistered, so a failed call returns its error instead of leaving the GPU writing to unpinned pages.
    ~HostRegistrations() {
        if (!n) return;
        (void)hipStreamSynchronize(ctx->copy);
        (void)hipStreamSynchronize(ctx->compute);
        release_all();
    }
    // the source frames are only read by the copy engine: once the copy stream has drained they can be unregistered while
    // the kernels still run (0.1-0.2 ms each, off the end of the call)
    void release_sources_after_uploads() {
        if (hipStreamSynchronize(ctx->copy) != hipSuccess) return;
        release_all();
    }
    void release_all() {
        if (!n) return;
        std::unique_lock<std::shared_mutex> lock(ofxcv_capture_mutex(ofxcv_lock_index(ctx)));
        for (int i = 0; i < n; i++) (void)hipHostUnregister(p[i]);
        (void)hipGetLastError();
        n = 0;
    }
};

// the four RGBA channels of the destination from the two flow fields, one 16-byte store per pixel:
// channel c <- flow[k[c]].{x|y} / render scale
struct ChanMap {
    int k[4], comp[4];
};
__global__ __launch_bounds__(256) void flows_to_rgba_kernel(const float2 *__restrict__ f0, const float2 *__restrict__ f1, int width, int height,
                                                            float *__restrict__ dst, ptrdiff_t dst_row_bytes, ChanMap m, double rsx, double rsy) {
    const int y = blockIdx.y, x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= width) return;
    const size_t i = (size_t)y * width + x;
    float2 a = f0[i], b = f1 ? f1[i] : make_float2(0.f, 0.f);
    a.x = (float)(a.x / rsx); a.y = (float)(a.y / rsy);
    b.x = (float)(b.x / rsx); b.y = (float)(b.y / rsy);
    float v[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const float2 f = m.k[c] == 1 ? b : a;
        v[c] = m.comp[c] ? f.y : f.x;
    }
    *(float4 *)((char *)dst + (ptrdiff_t)y * dst_row_bytes + (size_t)x * 16) = make_float4(v[0], v[1], v[2], v[3]);
}

// ---- concurrent host-image calls of one device, coalesced into ONE batched Farneback call ----
// VectorGenerator is eRenderFullySafe with host frame threading off (VectorGenerator.cpp:108, GenericOpenCVPlugin.cpp:350-357): a host renders
// several output frames at once, each render() on its own thread with its own context.  Left alone, every thread submits its own call of two
// pairs, and from five threads on the GPU runs many 2-pair calls beside each other (1 000 - 1 120 pairs/s at 1920x1080) where calls of 8 pairs
// reach 1 700: level 0 only takes the column-owning form (two iterations per launch) from 5 pairs on.  So a call that finds host.coalesce_min (4)
// host-image calls in flight on its device, itself included, does not start a Farneback call of its own: it uploads and converts its frames as
// ever, records an event behind the last conversion and hands its pairs to the device's SUBMISSION QUEUE.  The first caller that finds no
// coalesced call running becomes the leader: it takes everything queued with its geometry and parameters, up to one round of the chip (its own
// pairs at once when nothing else is queued), makes the batch context's stream wait for the riders' events and runs ONE batched Farneback call
// there -- launched kernel by kernel, reading the riders' gray frames and writing their flow fields and RGBA images (F7 inside the call) in
// place -- then makes every rider's stream wait for the call's event and releases the riders, who enqueue their own downloads behind it.  The
// leader keeps the slot until the call has completed: what arrives meanwhile rides in the next one.  All ordering is on the device; a host
// thread only ever waits for its own stream (and the leader for its call).  Results are those of the callers' own calls bit for bit (a batch is
// bit-identical to its single calls).
namespace {
// measurement aid (environment OFXCV_HOST_TRACE=1; tools/host_queue_trace.py): per host-image call the times of its phases in microseconds since the
// first traced call -- entry, frames enqueued, frames complete (= queued), its batched call started / finished, the caller woke up, image downloaded --
// the pairs of the call it rode in and whether this thread led it
struct HostTrace {
    static bool on() {
        static const bool v = [] { const char *e = std::getenv("OFXCV_HOST_TRACE"); return e && e[0] == '1'; }();
        return v;
    }
    static std::mutex &mu() { static std::mutex m; return m; }
    static std::vector<double> &log() { static std::vector<double> *v = new std::vector<double>(); return *v; }
    static double now() {
        static const auto t0 = std::chrono::steady_clock::now();
        return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    }
};
struct FlowSig {  // what two requests must share to ride in one call
    int w, h, levels, iterations, poly_n, rounding, gauss_gen, contraction, resize_gen;
    double poly_sigma, rsx, rsy;  // (the render scale: one per batched call -- the F7 pixels are written inside it)
    bool operator==(const FlowSig &o) const {
        return w == o.w && h == o.h && levels == o.levels && iterations == o.iterations && poly_n == o.poly_n && rounding == o.rounding &&
               gauss_gen == o.gauss_gen && contraction == o.contraction && resize_gen == o.resize_gen && poly_sigma == o.poly_sigma && rsx == o.rsx && rsy == o.rsy;
    }
};
struct BatchStatus {  // shared by the requests of one batched call: set by the leader once the call has completed
    std::atomic<int> aborted{0};
};
struct FlowReq {
    FlowSig sig;
    int n_other = 0;
    const uint8_t *gray[3] = {nullptr, nullptr, nullptr};  // device, rows gray_pitch apart; complete once `ready` has fired
    hipEvent_t ready = nullptr;   // recorded on the caller's compute stream behind its last conversion
    hipStream_t stream = nullptr; // the caller's compute stream: the leader makes it wait for the batched call
    float *d_flow[2] = {nullptr, nullptr};  // where the call writes the flows (the caller's own buffers)
    float *d_rgba = nullptr;  // all four channels mapped: the call also writes the image here (rows width * 16 bytes), channel -> (direction, coordinate) in cm
    ChanMap cm;
    bool no_graph = true;  // (the oldest request of a call decides; never changes a result)
    int state = 0;  // 0 queued, 1 riding in a call that is being enqueued, 2 enqueued: the caller's stream waits for it
    int rc = OFXCV_OK, batch_pairs = 0;
    std::shared_ptr<BatchStatus> status;
    const volatile unsigned *abort_word = nullptr;  // the batch context's (iterate_col_kernel's bounded waits)
    double t_start = 0, t_end = 0, t_launched = 0;  // its batched call as the leader saw it (HostTrace): begun, everything enqueued, complete
    bool led = false;
    char err[256] = {0};
};
class FlowQueue {
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<FlowReq *> q_;
    int running_ = 0;
    long batches_ = 0, batch_pairs_ = 0, batch_max_ = 0;  // coalesced calls formed on this device, their pairs, the largest
    static constexpr int kMaxDepth = 4;
    ofxcv_ctx *bctx_[kMaxDepth] = {};
    bool slot_busy_[kMaxDepth] = {};

    static void fail_all(const std::vector<FlowReq *> &batch, int rc, const char *text) {
        for (FlowReq *r : batch) {
            r->rc = rc;
            std::snprintf(r->err, sizeof(r->err), "%s", text);
        }
    }
    // The leader's work, queue unlocked: nobody else touches this slot's batch context meanwhile.  Everything is ENQUEUED here, nothing waited
    // for: the batch stream waits for the callers' frames (their `ready` events), the callers' streams wait for the batched call (its `done`
    // event).  Event operations across contexts take the runtime lock like everywhere else in the library (they must not run beside another
    // thread's stream capture on ROCm 7.2).
    ofxcv_ctx *enqueue(int device, int slot, const std::vector<FlowReq *> &batch, int reserve_pairs) {
        ofxcv_ctx *&b = bctx_[slot];
        if (!b) {
            const int rc = ofxcv_ctx_create(device, &b);
            if (rc) {
                fail_all(batch, rc, "host-image queue: the batch context could not be created");
                return nullptr;
            }
            b->host_coalesce = 0;
        }
        const FlowSig &sg = batch[0]->sig;
        auto run_on = [&]() -> int {
            OFXCV_HIP_CHECK(b, hipSetDevice(b->hip_device));
            const struct { const char *name; int *cur; int want; } opts[] = {{"farneback.opencv_rounding", &b->fb_opencv_rounding, sg.rounding},
                                                                           {"farneback.gaussian_kernel_generation", &b->fb_gauss_generation, sg.gauss_gen},
                                                                           {"farneback.filter_contraction", &b->fb_filter_contraction, sg.contraction},
                                                                           {"farneback.resize_generation", &b->fb_resize_generation, sg.resize_gen}};
            for (const auto &o : opts)
                if (*o.cur != o.want) {
                    const int rc = ofxcv_ctx_set_option(b, o.name, o.want);  // (drops the captured launch sequences)
                    if (rc) return rc;
                }
            b->fb_no_graph = batch[0]->no_graph;
            b->fb_reserve_pairs = std::max(b->fb_reserve_pairs, reserve_pairs);  // (the scratch is sized once for the largest call)
            int np = 0;
            for (const FlowReq *r : batch) np += r->n_other;
            // The call reads the riders' gray frames and writes their flow fields -- and, where all four channels are mapped, their RGBA images (F7 inside
            // the last level-0 launch) -- IN PLACE: launched kernel by kernel, it has no captured pointers to keep fixed, so nothing is gathered or copied.
            const size_t gray_pitch = align_up((size_t)sg.w, 256);
            const uint8_t *prevs[OFXCV_FARNEBACK_MAX_BATCH], *nexts[OFXCV_FARNEBACK_MAX_BATCH];
            float *flows[OFXCV_FARNEBACK_MAX_BATCH], *rgbas[OFXCV_FARNEBACK_MAX_BATCH];
            size_t gsteps[OFXCV_FARNEBACK_MAX_BATCH], fsteps[OFXCV_FARNEBACK_MAX_BATCH];
            ptrdiff_t rsteps[OFXCV_FARNEBACK_MAX_BATCH];
            unsigned mus[OFXCV_FARNEBACK_MAX_BATCH], mvs[OFXCV_FARNEBACK_MAX_BATCH];
            int p = 0;
            for (const FlowReq *r : batch)
                for (int k = 0; k < r->n_other; k++, p++) {
                    prevs[p] = r->gray[0];
                    nexts[p] = r->gray[k + 1];
                    flows[p] = r->d_flow[k];
                    gsteps[p] = gray_pitch;
                    fsteps[p] = (size_t)sg.w * 8;
                    rgbas[p] = r->d_rgba;
                    rsteps[p] = (ptrdiff_t)sg.w * 16;
                    mus[p] = mvs[p] = 0;
                    for (int c = 0; c < 4 && r->d_rgba; c++)  // the channels this direction owns (resolved by the caller: v over u, the later direction over the earlier)
                        if (r->cm.k[c] == k) (r->cm.comp[c] ? mvs[p] : mus[p]) |= 1u << c;
                }
            {
                std::unique_lock<std::shared_mutex> lock(ofxcv_capture_mutex(ofxcv_lock_index(b)));
                for (const FlowReq *r : batch) OFXCV_HIP_CHECK(b, hipStreamWaitEvent(b->compute, r->ready, 0));
            }
            // VectorGenerator.cpp:391,395,403: pyr_scale 0.5, winsize 3, flags 0; :494-519 the write-back
            int rc = ofxcv_calc_optical_flow_farneback_batch_rgba(b, np, prevs, gsteps, nexts, gsteps, flows, fsteps, sg.w, sg.h, 0.5, sg.levels, 3, sg.iterations, sg.poly_n,
                                                                  sg.poly_sigma, 0, rgbas, rsteps, mus, mvs, sg.rsx, sg.rsy, b->compute);
            if (rc) return rc;
            {
                std::unique_lock<std::shared_mutex> lock(ofxcv_capture_mutex(ofxcv_lock_index(b)));
                OFXCV_HIP_CHECK(b, hipEventRecord(b->ev_done, b->compute));
                for (const FlowReq *r : batch) OFXCV_HIP_CHECK(b, hipStreamWaitEvent(r->stream, b->ev_done, 0));
            }
            auto st = std::make_shared<BatchStatus>();
            for (FlowReq *r : batch) {
                r->batch_pairs = np;
                r->status = st;
                r->abort_word = b->fb_col_abort;
            }
            return OFXCV_OK;
        };
        const int rc = run_on();
        if (rc) {
            (void)hipStreamSynchronize(b->compute);  // whatever was enqueued before the failure: nothing of it may outlive the callers' buffers
            fail_all(batch, rc, b->err);
            return nullptr;
        }
        return b;
    }

public:
    static FlowQueue &of(int device) {
        static FlowQueue *q = new FlowQueue[64];  // intentionally leaked, like the cache above
        return q[device & 63];
    }
    // Queues the request and returns when the batched call it rides in has been enqueued (the caller's stream then waits for that call) -- or,
    // for the leader, when that call has completed: a device runs `depth` coalesced calls at a time, and what arrives while they run rides in the next.
    int submit(int device, FlowReq &r, int max_pairs, int depth) {
        max_pairs = std::max(2, std::min(max_pairs, OFXCV_FARNEBACK_MAX_BATCH));
        depth = std::max(1, std::min(depth, (int)kMaxDepth));
        std::unique_lock<std::mutex> lk(mu_);
        q_.push_back(&r);
        for (;;) {
            if (r.state == 2) return r.rc;
            if (r.state == 0 && running_ < depth) {
                // lead: the oldest request decides geometry and parameters; every queued request that shares them rides along, in order of arrival
                std::vector<FlowReq *> batch;
                const FlowSig sg = q_.front()->sig;
                int np = 0;
                for (size_t i = 0; i < q_.size();) {
                    FlowReq *c = q_[i];
                    if (c->sig == sg && np + c->n_other <= max_pairs) {
                        np += c->n_other;
                        c->state = 1;
                        batch.push_back(c);
                        q_.erase(q_.begin() + (ptrdiff_t)i);
                    } else
                        i++;
                }
                batches_++;
                batch_pairs_ += np;
                batch_max_ = std::max(batch_max_, (long)np);
                int slot = 0;
                while (slot_busy_[slot]) slot++;  // (running_ < depth <= kMaxDepth: one is free)
                slot_busy_[slot] = true;
                running_++;
                lk.unlock();
                const double t_start = HostTrace::on() ? HostTrace::now() : 0;
                ofxcv_ctx *b = enqueue(device, slot, batch, max_pairs);
                const double t_launched = HostTrace::on() ? HostTrace::now() : 0;
                std::shared_ptr<BatchStatus> st = batch[0]->status;
                lk.lock();
                for (FlowReq *c : batch) {  // the riders go on: their streams wait for the call, they enqueue their downloads behind it
                    c->state = 2;
                    c->t_start = t_start;
                    c->t_launched = t_launched;
                }
                r.led = true;
                cv_.notify_all();
                lk.unlock();
                if (b) {  // the leader keeps the slot until the call has completed
                    (void)hipEventSynchronize(b->ev_done);
                    if (ofxcv_col_abort_check(b) != OFXCV_OK && st) st->aborted.store(1);
                }
                r.t_end = HostTrace::on() ? HostTrace::now() : 0;
                lk.lock();
                running_--;
                slot_busy_[slot] = false;
                cv_.notify_all();
                continue;
            }
            cv_.wait(lk);
        }
    }
    // the idle batch contexts (streams, scratch, captured launch sequences) go; the next coalesced call re-creates them
    void shutdown() {
        std::lock_guard<std::mutex> lk(mu_);
        if (HostTrace::on() && batches_)
            std::fprintf(stderr, "ofxcv: submission queue: %ld coalesced calls, %.2f pairs on average, largest %ld\n", batches_, (double)batch_pairs_ / batches_, batch_max_);
        for (int i = 0; i < kMaxDepth; i++)
            if (bctx_[i] && !slot_busy_[i]) {
                ofxcv_ctx_destroy(bctx_[i]);
                bctx_[i] = nullptr;
            }
    }
};
}  // namespace

constexpr int kNotRegistered = 12345;  // internal: the registered-buffer path does not apply, stage through the pinned ring

static int flows_host_registered(ofxcv_ctx *ctx, const float *h_ref, ptrdiff_t ref_row_bytes, int n_other, const float *const h_other[2],
                                 const ptrdiff_t other_row_bytes[2], int ncomp, int width, int height, float *h_dst, ptrdiff_t dst_row_bytes,
                                 const unsigned chan_u_mask[2], const unsigned chan_v_mask[2], double render_scale_x, double render_scale_y,
                                 int levels, int iterations, int poly_n, double poly_sigma) {
    const int nf = 1 + n_other;
    const size_t row = (size_t)width * ncomp * sizeof(float), drow = (size_t)width * 16;
    const float *src[3] = {h_ref, h_other[0], n_other > 1 ? h_other[1] : nullptr};
    const ptrdiff_t src_rb[3] = {ref_row_bytes, other_row_bytes[0], n_other > 1 ? other_row_bytes[1] : 0};
    for (int f = 0; f < nf; f++)
        if (src_rb[f] < (ptrdiff_t)row) return kNotRegistered;  // bottom-up images: ring
    // channel -> (flow, coordinate): per direction v wins over u, a later direction overwrites an earlier one (:507-516)
    ChanMap cm = {{-1, -1, -1, -1}, {0, 0, 0, 0}};
    for (int k = 0; k < n_other; k++) {
        const unsigned mu = chan_u_mask[k] & 15u, mv = chan_v_mask[k] & 15u;
        for (int c = 0; c < 4; c++) {
            if (mv & (1u << c)) { cm.k[c] = k; cm.comp[c] = 1; }
            else if (mu & (1u << c)) { cm.k[c] = k; cm.comp[c] = 0; }
        }
    }
    // Only whole pixels are stored into host memory by the kernel (the default mapping: forward.u/v, backward.u/v): 4-byte
    // partial stores over PCIe were measured at a third of the rate (2.1 ms instead of 0.63 ms per 1080p frame), slower
    // than the ring's flow download + host scatter.
    if (cm.k[0] < 0 || cm.k[1] < 0 || cm.k[2] < 0 || cm.k[3] < 0 || dst_row_bytes < (ptrdiff_t)drow || ((uintptr_t)h_dst & 15) || (dst_row_bytes & 15))
        return kNotRegistered;
    HostRegistrations regs(ctx);  // waits for the call's streams, then unregisters, on every return below
    const size_t frame = align_up(row * height, 256), gray_pitch = align_up((size_t)width, 256), gray = gray_pitch * height,
                 flow_bytes = align_up((size_t)width * height * 8, 256);
    int rc = ofxcv_reserve(ctx, ctx->d_stage, nf * (frame + gray) + n_other * flow_bytes);
    if (rc) return rc;
    char *dp = (char *)ctx->d_stage.ptr;
    uint8_t *d_gray[3];
    float *d_flow[2] = {nullptr, nullptr};
    for (int f = 0; f < nf; f++) d_gray[f] = (uint8_t *)(dp + nf * frame + f * gray);
    for (int k = 0; k < n_other; k++) d_flow[k] = (float *)(dp + nf * (frame + gray) + k * flow_bytes);

    // uploads straight from the host's buffers on the copy stream; the compute stream converts frame f as soon as it has
    // arrived, so the next frame is on the wire meanwhile.  A frame is registered right before its upload is issued, so the
    // registration of frame f+1 (0.2 ms of host time) overlaps the DMA of frame f; the destination is registered last,
    // while the GPU computes.
    for (int f = 0; f < nf; f++) {
        if (!regs.add(src[f], (size_t)(height - 1) * src_rb[f] + row)) return kNotRegistered;
        if ((size_t)src_rb[f] == row)  // contiguous rows: one linear DMA
            OFXCV_HIP_CHECK(ctx, hipMemcpyAsync(dp + f * frame, src[f], row * height, hipMemcpyHostToDevice, ctx->copy));
        else
            OFXCV_HIP_CHECK(ctx, hipMemcpy2DAsync(dp + f * frame, row, src[f], (size_t)src_rb[f], row, height, hipMemcpyHostToDevice, ctx->copy));
        OFXCV_HIP_CHECK(ctx, hipEventRecord(ctx->ev_h2d[f], ctx->copy));
    }
    for (int f = 0; f < nf; f++) {
        OFXCV_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->compute, ctx->ev_h2d[f], 0));
        rc = ofxcv_to_byte_grayscale(ctx, (const float *)(dp + f * frame), (ptrdiff_t)row, ncomp, width, height, d_gray[f], (ptrdiff_t)gray_pitch, ctx->compute);
        if (rc) return rc;
    }
    // The two flows of an output frame are independent pairs with the same first frame: one batched call (every launch of
    // the level walk carries both).  VectorGenerator.cpp:391,395,403: pyr_scale 0.5, winsize 3, flags 0
    {
        const uint8_t *prevs[2] = {d_gray[0], d_gray[0]}, *nexts[2] = {d_gray[1], n_other > 1 ? d_gray[2] : nullptr};
        const size_t gsteps[2] = {gray_pitch, gray_pitch}, fsteps[2] = {(size_t)width * 8, (size_t)width * 8};
        rc = ofxcv_calc_optical_flow_farneback_batch(ctx, n_other, prevs, gsteps, nexts, gsteps, d_flow, fsteps, width, height, 0.5, levels, 3,
                                                     iterations, poly_n, poly_sigma, 0, ctx->compute);
        if (rc) return rc;
    }
    regs.release_sources_after_uploads();
    void *d_dst = nullptr;
    if (!regs.add(h_dst, (size_t)(height - 1) * dst_row_bytes + drow) || hipHostGetDevicePointer(&d_dst, h_dst, 0) != hipSuccess || !d_dst) {
        (void)hipGetLastError();
        return kNotRegistered;  // the destination cannot be addressed by the kernel: the ring path serves the call
    }
    hipLaunchKernelGGL(flows_to_rgba_kernel, dim3(ofxcv_div_up(width, 256), height), dim3(256), 0, ctx->compute, (const float2 *)d_flow[0],
                       (const float2 *)d_flow[1], width, height, (float *)d_dst, dst_row_bytes, cm, render_scale_x, render_scale_y);
    OFXCV_LAUNCH_CHECK(ctx, "flows_to_rgba_kernel");
    OFXCV_HIP_CHECK(ctx, hipStreamSynchronize(ctx->copy));
    OFXCV_HIP_CHECK(ctx, hipStreamSynchronize(ctx->compute));
    ctx->host_zero_copy_calls++;
    return ofxcv_col_abort_check(ctx);  // `regs` unregisters the host ranges here: nothing of this call is in flight any more
}

// One reference frame against one or two other frames (forward: t+1, backward: t-1 -- the two directions a default
// VectorGenerator output frame needs, VectorGenerator.cpp:739-779).  The reference frame is uploaded and converted once, the
// flows of both directions are ONE batched Farneback call, and one pass writes the destination.
//
// Uploads.  Default: hipMemcpyAsync straight from the host's own (pageable) frames on the copy stream.  On this platform
// the runtime moves pageable memory at the rate of pinned memory (measured on the MI355X box, tools/host_pieces.py and
// tools/host_overlap.py: 47 GB/s against 56 GB/s pinned for one thread; 3 uploads + 1 download of 33 MB frames from 1 / 2 / 4
// threads: 409 / 535 / 519 per second pageable, 392 / 479 / 538 pinned), without the per-call hipHostRegister of the
// registered form (which stalls other threads' GPU work) and without the staging memcpy of the ring (30 GB/s per host
// thread, the largest term of a staged call).  Bottom-up or oddly strided images, and option "host.register" 0, stage
// through the pinned ring instead.
// Write-back.  With all four destination channels mapped (the default output frame) a kernel composes the RGBA image in
// HBM and one copy takes it into the host's image; otherwise only the flows come back (8 B/px each) and the mapped
// channels are filled on the host (unmapped channels stay untouched, :507-516).
static int flows_host(ofxcv_ctx *ctx, const float *h_ref, ptrdiff_t ref_row_bytes, int n_other, const float *const h_other[2],
                      const ptrdiff_t other_row_bytes[2], int ncomp, int width, int height, float *h_dst, ptrdiff_t dst_row_bytes,
                      const unsigned chan_u_mask[2], const unsigned chan_v_mask[2], double render_scale_x, double render_scale_y,
                      int levels, int iterations, int poly_n, double poly_sigma, const char *const keys[3] = nullptr) {
    OFXCV_HIP_CHECK(ctx, hipSetDevice(ctx->hip_device));
    const int nf = 1 + n_other;
    const size_t row = (size_t)width * ncomp * sizeof(float), drow = (size_t)width * 16;
    double tr[16] = {HostTrace::on() ? HostTrace::now() : 0};
    struct TraceOut {
        double *t;
        ~TraceOut() {
            if (!HostTrace::on()) return;
            t[6] = HostTrace::now();
            std::lock_guard<std::mutex> lk(HostTrace::mu());
            HostTrace::log().insert(HostTrace::log().end(), t, t + 16);
        }
    } trace_out{tr};
    if (ctx->host_register == 2) {  // opt-in: the host's buffers registered for the call, the kernel stores into the host image
        int rc0 = flows_host_registered(ctx, h_ref, ref_row_bytes, n_other, h_other, other_row_bytes, ncomp, width, height, h_dst, dst_row_bytes,
                                        chan_u_mask, chan_v_mask, render_scale_x, render_scale_y, levels, iterations, poly_n, poly_sigma);
        if (rc0 != kNotRegistered) return rc0;
    }
    const float *src[3] = {h_ref, h_other[0], n_other > 1 ? h_other[1] : nullptr};
    const ptrdiff_t src_rb[3] = {ref_row_bytes, other_row_bytes[0], n_other > 1 ? other_row_bytes[1] : 0};
    bool topdown = true;
    for (int f = 0; f < nf; f++) topdown = topdown && src_rb[f] >= (ptrdiff_t)row;
    const bool direct_up = ctx->host_register != 0 && topdown;
    // channel -> (flow, coordinate): per direction v wins over u, a later direction overwrites an earlier one (:507-516)
    ChanMap cm = {{-1, -1, -1, -1}, {0, 0, 0, 0}};
    for (int k = 0; k < n_other; k++) {
        const unsigned mu = chan_u_mask[k] & 15u, mv = chan_v_mask[k] & 15u;
        for (int c = 0; c < 4; c++) {
            if (mv & (1u << c)) { cm.k[c] = k; cm.comp[c] = 1; }
            else if (mu & (1u << c)) { cm.k[c] = k; cm.comp[c] = 0; }
        }
    }
    const bool whole_pixels = cm.k[0] >= 0 && cm.k[1] >= 0 && cm.k[2] >= 0 && cm.k[3] >= 0 && dst_row_bytes >= (ptrdiff_t)drow;
    const bool direct_down = ctx->host_register != 0 && whole_pixels;

    const size_t frame = align_up(row * height, 256), gray_pitch = align_up((size_t)width, 256), gray = gray_pitch * height,
                 flow_bytes = align_up((size_t)width * height * 8, 256), rgba_bytes = align_up(drow * height, 256);
    int rc = ofxcv_reserve(ctx, ctx->d_stage, nf * (frame + gray) + n_other * flow_bytes + (direct_down ? rgba_bytes : 0));
    if (rc) return rc;
    const size_t pin_frames = direct_up ? 0 : nf * frame, pin_flows = direct_down ? 0 : n_other * flow_bytes;
    if (pin_frames + pin_flows) {
        rc = reserve_pinned(ctx, pin_frames + pin_flows);
        if (rc) return rc;
    }
    char *hp = (char *)ctx->h_pinned, *dp = (char *)ctx->d_stage.ptr;
    char *d_frame[3], *h_frame[3];
    uint8_t *d_gray[3];
    float *d_flow[2] = {nullptr, nullptr}, *h_flow[2] = {nullptr, nullptr};
    for (int f = 0; f < nf; f++) {
        d_frame[f] = dp + f * frame;
        h_frame[f] = direct_up ? nullptr : hp + f * frame;
        d_gray[f] = (uint8_t *)(dp + nf * frame + f * gray);
    }
    for (int k = 0; k < n_other; k++) {
        d_flow[k] = (float *)(dp + nf * (frame + gray) + k * flow_bytes);
        h_flow[k] = direct_down ? nullptr : (float *)(hp + pin_frames + k * flow_bytes);
    }
    float *d_rgba = direct_down ? (float *)(dp + nf * (frame + gray) + n_other * flow_bytes) : nullptr;
    // Frames the caller named (keys): their gray images live in the device's cache.  Declared before `drain`: the entries are
    // unpinned only after this call's streams have drained.
    struct Pins {
        GrayCache *cache;
        GrayEntry *e[3] = {nullptr, nullptr, nullptr};
        bool fill[3] = {false, false, false}, done[3] = {false, false, false}, pending[3] = {false, false, false};
        ~Pins() {
            for (int f = 0; f < 3; f++)
                if (e[f]) {
                    if (fill[f] && !done[f]) cache->published(e[f], false);  // a way out before the entry was filled: nobody may use it
                    cache->release(e[f]);
                }
        }
    } pins{&GrayCache::of(ctx->device)};
    if (keys && ctx->host_cache_mb > 0)
        for (int f = 0; f < nf; f++)
            if (keys[f] && keys[f][0]) {
                bool dup = false;  // the same frame twice in one call (a still): the second one takes the plain path
                for (int g = 0; g < f; g++) dup = dup || (pins.e[g] && pins.e[g]->key == keys[f]);
                if (!dup) pins.e[f] = pins.cache->acquire(ctx, keys[f], width, height, ncomp, gray, (size_t)ctx->host_cache_mb << 20, &pins.fill[f], &pins.pending[f]);
            }
    // whatever happens below, nothing of this call may still be reading the host's frames or writing its image on return
    struct Drain {
        ofxcv_ctx *c;
        ~Drain() {
            (void)hipStreamSynchronize(c->copy);
            (void)hipStreamSynchronize(c->compute);
        }
    } drain{ctx};

    // Two directions, one calling thread: a synchronous call cannot overlap its uploads with anybody else's kernels, but it can
    // with its own -- the forward pair only needs the first two frames, so it runs as a single-pair call while the third frame
    // is still on the wire (0.7 ms of a 1080p frame), and the backward pair after it.  Two single-pair calls take ~0.3 ms more GPU
    // time than one batched call of two, which is why several render threads at once (they keep link and GPU busy between them)
    // stay with the batched form.  Same results either way (a batch is bit-identical to its single calls).
    // VectorGenerator.cpp:391,395,403: pyr_scale 0.5, winsize 3, flags 0
    const size_t gsteps[2] = {gray_pitch, gray_pitch}, fsteps[2] = {(size_t)width * 8, (size_t)width * 8};
    struct InFlight {  // host-image calls in flight on this (logical) device
        static std::atomic<int> &n(int device) { static std::atomic<int> v[64]; return v[device & 63]; }
        std::atomic<int> &c;
        int mine;
        explicit InFlight(int device) : c(n(device)), mine(c.fetch_add(1) + 1) {}
        ~InFlight() { c.fetch_sub(1); }
    } in_flight(ctx->device);
    bool avail[3] = {false, false, false}, pair_done[2] = {false, false};
    int to_upload = 0;
    for (int f = 0; f < nf; f++) to_upload += !(pins.e[f] && !pins.fill[f]);
    // (with every frame already on the device there is no upload to hide a pair behind: one batched call)
    const bool split = n_other == 2 && to_upload > 0 && (ctx->host_split == 1 || (ctx->host_split == 2 && in_flight.mine == 1));
    auto enqueue_ready_pairs = [&]() -> int {  // split form: direction k as soon as the reference frame and frame k+1 are there
        for (int k = 0; split && k < n_other; k++)
            if (!pair_done[k] && avail[0] && avail[k + 1]) {
                const uint8_t *prevs[1] = {d_gray[0]}, *nexts[1] = {d_gray[k + 1]};
                int r = ofxcv_calc_optical_flow_farneback_batch(ctx, 1, prevs, gsteps, nexts, gsteps, &d_flow[k], fsteps, width, height, 0.5, levels, 3,
                                                                iterations, poly_n, poly_sigma, 0, ctx->compute);
                if (r) return r;
                pair_done[k] = true;
            }
        return OFXCV_OK;
    };
    // A frame that is on the device already: nothing to upload, nothing to convert -- its gray image (1 B/px) is copied into this
    // call's own slot, so that the Farneback call sees the same pointers as ever (its launch graph is cached by pointer) and the
    // entry is only read here.  The event belongs to another context's stream: waited for under the runtime lock, like
    // everything that must not run beside another thread's stream capture on ROCm 7.2 (outside it the wait failed with
    // "dependency created on uncaptured work in another stream" while another thread was capturing).
    auto take_cached = [&](int f) -> int {
        {
            std::unique_lock<std::shared_mutex> lock(ofxcv_capture_mutex(ofxcv_lock_index(ctx)));
            OFXCV_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->compute, pins.e[f]->ready, 0));
        }
        OFXCV_HIP_CHECK(ctx, hipMemcpyAsync(d_gray[f], pins.e[f]->ptr, gray, hipMemcpyDeviceToDevice, ctx->compute));
        avail[f] = true;
        ctx->host_cache_hits++;
        return OFXCV_OK;
    };
    // Cached frames first: whatever only needs them is enqueued BEFORE the first copy call (a copy from pageable memory returns
    // when the runtime has staged the frame).  Frames another thread is filling right now come last.
    tr[9] = HostTrace::on() ? HostTrace::now() : 0;   // scratch reserved, named frames looked up
    for (int f = 0; f < nf; f++)
        if (pins.e[f] && !pins.fill[f] && !pins.pending[f] && (rc = take_cached(f))) return rc;
    if ((rc = enqueue_ready_pairs())) return rc;
    tr[10] = HostTrace::on() ? HostTrace::now() : 0;  // cached frames taken
    // uploads on the copy stream; the compute stream converts frame f as soon as it has arrived
    const int rows_per_chunk = std::max(1, (int)((size_t)(4u << 20) / row));  // ring: ~4 MiB per DMA
    auto upload = [&](int f) -> int {
        if (direct_up) {
            if ((size_t)src_rb[f] == row)  // contiguous rows: one linear copy
                OFXCV_HIP_CHECK(ctx, hipMemcpyAsync(d_frame[f], src[f], row * height, hipMemcpyHostToDevice, ctx->copy));
            else
                OFXCV_HIP_CHECK(ctx, hipMemcpy2DAsync(d_frame[f], row, src[f], (size_t)src_rb[f], row, height, hipMemcpyHostToDevice, ctx->copy));
        } else {
            for (int y0 = 0; y0 < height; y0 += rows_per_chunk) {
                int y1 = std::min(height, y0 + rows_per_chunk);
                const int nblk = std::min(4, y1 - y0);
                HostPool::get().run(nblk, [&](int b) {
                    const int ya = y0 + (int)((long)(y1 - y0) * b / nblk), yb = y0 + (int)((long)(y1 - y0) * (b + 1) / nblk);
                    for (int y = ya; y < yb; y++) std::memcpy(h_frame[f] + (size_t)y * row, (const char *)src[f] + (ptrdiff_t)y * src_rb[f], row);
                });
                OFXCV_HIP_CHECK(ctx, hipMemcpyAsync(d_frame[f] + (size_t)y0 * row, h_frame[f] + (size_t)y0 * row, (size_t)(y1 - y0) * row,
                                                    hipMemcpyHostToDevice, ctx->copy));
            }
        }
        // (the kernels of the frames before it are enqueued before the next upload starts, and run during it)
        OFXCV_HIP_CHECK(ctx, hipEventRecord(ctx->ev_h2d[f], ctx->copy));
        OFXCV_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->compute, ctx->ev_h2d[f], 0));
        rc = ofxcv_to_byte_grayscale(ctx, (const float *)d_frame[f], (ptrdiff_t)row, ncomp, width, height, d_gray[f], (ptrdiff_t)gray_pitch, ctx->compute);
        if (rc) return rc;
        if (pins.e[f]) {  // a named frame: its gray image goes into the cache entry as well
            OFXCV_HIP_CHECK(ctx, hipMemcpyAsync(pins.e[f]->ptr, d_gray[f], gray, hipMemcpyDeviceToDevice, ctx->compute));
            {
                std::unique_lock<std::shared_mutex> lock(ofxcv_capture_mutex(ofxcv_lock_index(ctx)));
                OFXCV_HIP_CHECK(ctx, hipEventRecord(pins.e[f]->ready, ctx->compute));
            }
            pins.cache->published(pins.e[f], true);
            pins.done[f] = true;
            ctx->host_cache_misses++;
        }
        avail[f] = true;
        return enqueue_ready_pairs();
    };
    for (int f = 0; f < nf; f++)
        if (!avail[f] && !pins.pending[f]) {
            tr[13] += 1;
            if ((rc = upload(f))) return rc;
        }
    tr[11] = HostTrace::on() ? HostTrace::now() : 0;  // own uploads + conversions enqueued
    // Frames some other thread was filling when this call looked them up.  Its own entries are published by now, so waiting is
    // safe; should the other thread have given up, the frame is uploaded after all (unnamed).
    for (int f = 0; f < nf; f++)
        if (!avail[f]) {
            tr[14] += 1;
            if (pins.cache->wait_published(pins.e[f])) {
                if ((rc = take_cached(f)) || (rc = enqueue_ready_pairs())) return rc;
            } else {
                pins.cache->release(pins.e[f]);
                pins.e[f] = nullptr;
                if ((rc = upload(f))) return rc;
            }
        }
    if (direct_up) ctx->host_direct_calls++;
    else ctx->host_staged_calls++;
    if (split) ctx->host_split_calls++;
    // The two flows of an output frame are independent pairs with the same first frame: one batched call (every launch of
    // the level walk carries both).
    bool composed = false;
    // Another host-image call is in flight on this device (several render threads): the pairs go to the device's submission queue and ride in ONE
    // batched Farneback call with whatever the other threads have queued (FlowQueue above); a lone call runs its own.
    const bool coalesce = !split && ctx->host_coalesce && (ctx->host_coalesce == 2 || in_flight.c.load() >= ctx->host_coalesce_min);
    tr[1] = HostTrace::on() ? HostTrace::now() : 0;
    std::shared_ptr<BatchStatus> batch_status;
    const volatile unsigned *batch_abort_word = nullptr;
    // after this call's streams have drained (so the batched call it rode in has completed): did a bounded wait of iterate_col_kernel run out there?
    auto coalesced_abort = [&]() -> int {
        if ((batch_status && batch_status->aborted.load()) || (batch_abort_word && *batch_abort_word))
            return ofxcv_fail(ctx, OFXCV_ERR_HIP, "calc_optical_flow_farneback: a bounded wait inside iterate_col_kernel ran out in the coalesced call; its flows are not valid");
        return OFXCV_OK;
    };
    if (coalesce) {
        {
            std::unique_lock<std::shared_mutex> lock(ofxcv_capture_mutex(ofxcv_lock_index(ctx)));
            OFXCV_HIP_CHECK(ctx, hipEventRecord(ctx->ev_done, ctx->compute));  // behind the last conversion: the frames are complete when it fires
        }
        tr[2] = HostTrace::on() ? HostTrace::now() : 0;
        FlowReq rq;
        rq.sig = {width, height, levels, iterations, poly_n, ctx->fb_opencv_rounding, ctx->fb_gauss_generation, ctx->fb_filter_contraction, ctx->fb_resize_generation, poly_sigma, render_scale_x, render_scale_y};
        rq.n_other = n_other;
        rq.no_graph = true;  // (the queue's call is launched kernel by kernel: no runtime lock held, the first kernels run while the rest is enqueued)
        rq.ready = ctx->ev_done;
        rq.stream = ctx->compute;
        for (int f = 0; f < nf; f++) rq.gray[f] = d_gray[f];
        rq.d_flow[0] = d_flow[0];
        rq.d_flow[1] = d_flow[1];
        if (direct_down) {
            rq.d_rgba = d_rgba;
            rq.cm = cm;
        }
        // pairs per coalesced call: what fills ONE round of the chip in the column-owning form of level 0 (a workgroup per 60-pixel tile column and pair,
        // one workgroup per CU: 8 pairs at 1920x1080, 4 at 3840x2160) -- measured: 12 queued pairs run faster as 8 + (4 + newcomers) than as 12
        int max_pairs = ctx->host_coalesce_max;
        if (max_pairs <= 0) max_pairs = std::max(2, std::min(OFXCV_FARNEBACK_MAX_BATCH, (ctx->num_cus / std::max(1, ofxcv_div_up(width, 60))) & ~1));
        rc = FlowQueue::of(ctx->device).submit(ctx->device, rq, max_pairs, 1);
        if (rc) return ofxcv_fail(ctx, rc, "%s", rq.err);
        OFXCV_HIP_CHECK(ctx, hipSetDevice(ctx->hip_device));
        ctx->host_coalesced_calls++;
        ctx->host_coalesced_pairs += n_other;
        ctx->host_coalesced_batches += rq.batch_pairs;
        composed = direct_down;
        batch_status = rq.status;
        batch_abort_word = rq.abort_word;
        tr[3] = rq.t_start; tr[4] = rq.t_end; tr[5] = HostTrace::on() ? HostTrace::now() : 0; tr[7] = rq.batch_pairs; tr[8] = rq.led ? 1 : 0; tr[15] = rq.t_launched;
    } else if (!split) {
        const uint8_t *prevs[2] = {d_gray[0], d_gray[0]}, *nexts[2] = {d_gray[1], n_other > 1 ? d_gray[2] : nullptr};
        rc = ofxcv_calc_optical_flow_farneback_batch(ctx, n_other, prevs, gsteps, nexts, gsteps, d_flow, fsteps, width, height, 0.5, levels, 3,
                                                     iterations, poly_n, poly_sigma, 0, ctx->compute);
        if (rc) return rc;
    }
    if (direct_down) {
        if (!composed) {
            hipLaunchKernelGGL(flows_to_rgba_kernel, dim3(ofxcv_div_up(width, 256), height), dim3(256), 0, ctx->compute, (const float2 *)d_flow[0],
                               (const float2 *)d_flow[1], width, height, d_rgba, (ptrdiff_t)drow, cm, render_scale_x, render_scale_y);
            OFXCV_LAUNCH_CHECK(ctx, "flows_to_rgba_kernel");
        }
        if ((size_t)dst_row_bytes == drow)
            OFXCV_HIP_CHECK(ctx, hipMemcpyAsync(h_dst, d_rgba, drow * height, hipMemcpyDeviceToHost, ctx->compute));
        else
            OFXCV_HIP_CHECK(ctx, hipMemcpy2DAsync(h_dst, (size_t)dst_row_bytes, d_rgba, drow, drow, height, hipMemcpyDeviceToHost, ctx->compute));
        OFXCV_HIP_CHECK(ctx, hipStreamSynchronize(ctx->compute));
        if ((rc = coalesced_abort())) return rc;
        return ofxcv_col_abort_check(ctx);  // (the host images of a call whose kernel gave up are not valid: fail, do not hand them over)
    }
    for (int k = 0; k < n_other; k++) {
        if (render_scale_x != 1.0 || render_scale_y != 1.0) {
            size_t n = (size_t)width * height;
            hipLaunchKernelGGL(scale_flow_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->compute, (float2 *)d_flow[k], n,
                               render_scale_x, render_scale_y);
            OFXCV_LAUNCH_CHECK(ctx, "scale_flow_kernel");
        }
        OFXCV_HIP_CHECK(ctx, hipMemcpyAsync(h_flow[k], d_flow[k], (size_t)width * height * 8, hipMemcpyDeviceToHost, ctx->compute));
    }
    OFXCV_HIP_CHECK(ctx, hipStreamSynchronize(ctx->compute));
    if ((rc = coalesced_abort())) return rc;
    {
        const int arc = ofxcv_col_abort_check(ctx);
        if (arc) return arc;
    }

    // write-back into the host-owned destination (:507-516): the channel -> (flow, coordinate) table was resolved above
    int nmap = 0, dst_c[4];
    const float *map_src[4];
    for (int c = 0; c < 4; c++)
        if (cm.k[c] >= 0) {
            dst_c[nmap] = c;
            map_src[nmap++] = h_flow[cm.k[c]] + cm.comp[c];
        }
    const int nblk = nmap ? std::min(8, height) : 0;
    HostPool::get().run(nblk, [&](int b) {
        for (int y = (int)((long)height * b / nblk), ye = (int)((long)height * (b + 1) / nblk); y < ye; y++) {
            float *d = (float *)((char *)h_dst + (ptrdiff_t)y * dst_row_bytes);
            const size_t ro = (size_t)y * width * 2;
            if (nmap == 2) {
                const float *s0 = map_src[0] + ro, *s1 = map_src[1] + ro;
                const int d0 = dst_c[0], d1 = dst_c[1];
                for (int x = 0; x < width; x++) {
                    d[x * 4 + d0] = s0[x * 2];
                    d[x * 4 + d1] = s1[x * 2];
                }
            } else if (nmap == 4) {
                const float *s0 = map_src[0] + ro, *s1 = map_src[1] + ro, *s2 = map_src[2] + ro, *s3 = map_src[3] + ro;
                for (int x = 0; x < width; x++) {
                    d[x * 4 + 0] = s0[x * 2];
                    d[x * 4 + 1] = s1[x * 2];
                    d[x * 4 + 2] = s2[x * 2];
                    d[x * 4 + 3] = s3[x * 2];
                }
            } else {
                for (int x = 0; x < width; x++)
                    for (int q = 0; q < nmap; q++) d[x * 4 + dst_c[q]] = map_src[q][ro + x * 2];
            }
        }
    });
    return OFXCV_OK;
}


extern "C" int ofxcv_vectorgen_flow_host(ofxcv_ctx *ctx, const float *h_ref, ptrdiff_t ref_row_bytes, const float *h_other,
                                         ptrdiff_t other_row_bytes, int ncomp, int width, int height, float *h_dst,
                                         ptrdiff_t dst_row_bytes, unsigned chan_u_mask, unsigned chan_v_mask,
                                         double render_scale_x, double render_scale_y, int levels, int iterations, int poly_n,
                                         double poly_sigma) {
    if (!ctx) return OFXCV_ERR_INVALID;
    if (!h_ref || !h_other || !h_dst || width <= 0 || height <= 0 || render_scale_x == 0 || render_scale_y == 0)
        return ofxcv_fail(ctx, OFXCV_ERR_INVALID, "vectorgen_flow_host: bad argument");
    if (ncomp != 3 && ncomp != 4) return ofxcv_fail(ctx, OFXCV_ERR_UNSUPPORTED, "vectorgen_flow_host: RGB or RGBA sources only");
    const float *others[2] = {h_other, nullptr};
    const ptrdiff_t rbs[2] = {other_row_bytes, 0};
    const unsigned mu[2] = {chan_u_mask, 0}, mv[2] = {chan_v_mask, 0};
    return flows_host(ctx, h_ref, ref_row_bytes, 1, others, rbs, ncomp, width, height, h_dst, dst_row_bytes, mu, mv, render_scale_x,
                      render_scale_y, levels, iterations, poly_n, poly_sigma);
}

extern "C" int ofxcv_vectorgen_flows_host(ofxcv_ctx *ctx, const float *h_ref, ptrdiff_t ref_row_bytes, const float *h_fwd,
                                          ptrdiff_t fwd_row_bytes, const float *h_bwd, ptrdiff_t bwd_row_bytes, int ncomp, int width,
                                          int height, float *h_dst, ptrdiff_t dst_row_bytes, unsigned fwd_u_mask, unsigned fwd_v_mask,
                                          unsigned bwd_u_mask, unsigned bwd_v_mask, double render_scale_x, double render_scale_y, int levels,
                                          int iterations, int poly_n, double poly_sigma) {
    if (!ctx) return OFXCV_ERR_INVALID;
    if (!h_ref || (!h_fwd && !h_bwd) || !h_dst || width <= 0 || height <= 0 || render_scale_x == 0 || render_scale_y == 0)
        return ofxcv_fail(ctx, OFXCV_ERR_INVALID, "vectorgen_flows_host: bad argument");
    if (ncomp != 3 && ncomp != 4) return ofxcv_fail(ctx, OFXCV_ERR_UNSUPPORTED, "vectorgen_flows_host: RGB or RGBA sources only");
    const float *others[2];
    ptrdiff_t rbs[2];
    unsigned mu[2], mv[2];
    int n = 0;
    if (h_fwd) { others[n] = h_fwd; rbs[n] = fwd_row_bytes; mu[n] = fwd_u_mask; mv[n++] = fwd_v_mask; }
    if (h_bwd) { others[n] = h_bwd; rbs[n] = bwd_row_bytes; mu[n] = bwd_u_mask; mv[n++] = bwd_v_mask; }
    return flows_host(ctx, h_ref, ref_row_bytes, n, others, rbs, ncomp, width, height, h_dst, dst_row_bytes, mu, mv, render_scale_x,
                      render_scale_y, levels, iterations, poly_n, poly_sigma);
}

extern "C" int ofxcv_vectorgen_flows_host_keyed(ofxcv_ctx *ctx, const float *h_ref, ptrdiff_t ref_row_bytes, const float *h_fwd,
                                                ptrdiff_t fwd_row_bytes, const float *h_bwd, ptrdiff_t bwd_row_bytes, int ncomp, int width,
                                                int height, float *h_dst, ptrdiff_t dst_row_bytes, unsigned fwd_u_mask, unsigned fwd_v_mask,
                                                unsigned bwd_u_mask, unsigned bwd_v_mask, double render_scale_x, double render_scale_y, int levels,
                                                int iterations, int poly_n, double poly_sigma, const char *ref_key, const char *fwd_key,
                                                const char *bwd_key) {
    if (!ctx) return OFXCV_ERR_INVALID;
    if (!h_ref || (!h_fwd && !h_bwd) || !h_dst || width <= 0 || height <= 0 || render_scale_x == 0 || render_scale_y == 0)
        return ofxcv_fail(ctx, OFXCV_ERR_INVALID, "vectorgen_flows_host_keyed: bad argument");
    if (ncomp != 3 && ncomp != 4) return ofxcv_fail(ctx, OFXCV_ERR_UNSUPPORTED, "vectorgen_flows_host_keyed: RGB or RGBA sources only");
    const float *others[2];
    ptrdiff_t rbs[2];
    unsigned mu[2], mv[2];
    const char *keys[3] = {ref_key, nullptr, nullptr};
    int n = 0;
    if (h_fwd) { others[n] = h_fwd; rbs[n] = fwd_row_bytes; mu[n] = fwd_u_mask; mv[n] = fwd_v_mask; keys[++n] = fwd_key; }
    if (h_bwd) { others[n] = h_bwd; rbs[n] = bwd_row_bytes; mu[n] = bwd_u_mask; mv[n] = bwd_v_mask; keys[++n] = bwd_key; }
    return flows_host(ctx, h_ref, ref_row_bytes, n, others, rbs, ncomp, width, height, h_dst, dst_row_bytes, mu, mv, render_scale_x,
                      render_scale_y, levels, iterations, poly_n, poly_sigma, keys);
}

extern "C" long ofxcv_host_cache_hits(const ofxcv_ctx *ctx) { return ctx ? ctx->host_cache_hits : 0; }
extern "C" long ofxcv_host_cache_misses(const ofxcv_ctx *ctx) { return ctx ? ctx->host_cache_misses : 0; }
extern "C" int ofxcv_host_cache_stats(ofxcv_ctx *ctx, size_t *bytes, int *frames) {
    if (!ctx) return OFXCV_ERR_INVALID;
    size_t b = 0;
    int n = 0;
    GrayCache::of(ctx->device).stats(b, n);
    if (bytes) *bytes = b;
    if (frames) *frames = n;
    return OFXCV_OK;
}
extern "C" int ofxcv_host_cache_clear(ofxcv_ctx *ctx) {
    if (!ctx) return OFXCV_ERR_INVALID;
    OFXCV_HIP_CHECK(ctx, hipSetDevice(ctx->hip_device));
    GrayCache::of(ctx->device).clear(ctx);
    FlowQueue::of(ctx->device).shutdown();
    return OFXCV_OK;
}
// measurement aid (not part of the public header): copies up to n doubles of the host-call trace (OFXCV_HOST_TRACE=1; 16 per call) and clears it
extern "C" int ofxcv_debug_host_trace(double *out, int n) {
    std::lock_guard<std::mutex> lk(HostTrace::mu());
    std::vector<double> &v = HostTrace::log();
    const int m = (int)std::min<size_t>(v.size(), (size_t)std::max(0, n));
    if (out) std::memcpy(out, v.data(), sizeof(double) * m);
    v.clear();
    return m;
}
extern "C" int ofxcv_host_coalesce_stats(const ofxcv_ctx *ctx, long *calls, long *pairs, long *batch_pairs) {
    if (!ctx) return OFXCV_ERR_INVALID;
    if (calls) *calls = ctx->host_coalesced_calls;
    if (pairs) *pairs = ctx->host_coalesced_pairs;
    if (batch_pairs) *batch_pairs = ctx->host_coalesced_batches;
    return OFXCV_OK;
}
