// telea_march.h -- the fast-marching front of cvInpaint (photo/src/inpaint.cpp: icvCalcFMM, icvTeleaInpaintFMM's front
// recurrence, CvPriorityQueueFloat) on the host: distance map T and fill order of the hole pixels.  Depends only on the
// hole mask, never on colours.  Plain C++ (no HIP): included by inpaint.hip and by the CPU test tests/march/test_march.cpp.
#pragma once
#include <algorithm>
#include <atomic>
#include <climits>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <queue>
#include <thread>
#include <vector>

namespace ofxcv_telea {

enum : uint8_t { KNOWN = 0, BAND = 1, INSIDE = 2, CHANGE = 3 };
constexpr int kNeverFilled = INT_MAX;  // hole pixel the march never reaches (image row / column 0)

// ------------------------------------------------------------------ I3/I4 front march (host)

// Pop order of CvPriorityQueueFloat: smallest T first, equal T in push order.  T is never negative while it is queued,
// so its bit pattern orders like its value: one 64-bit key (T bits << 32 | push number) replaces the two-field compare,
// and the pixel is looked up by push number.
//
// The queue itself is a bucket queue, not a binary heap (round 4; 111 -> 60 ns per hole pixel for the whole inward march on the
// build host, tools/march_bench.cpp).  A pixel is pushed when one of its four neighbours pops, with a distance of at least that
// neighbour's + 0.707 (FastMarching_solve with both arguments >= the popped distance; every other non-INSIDE neighbour is
// still queued, i.e. not smaller either), so with buckets of 1/16 a push practically always lands in a LATER bucket than the
// one being popped: it is appended there unsorted, and a bucket is sorted once, when the front reaches it, and then read off
// in order.  Nothing depends on that argument: a key that does belong to the current bucket or an earlier one is inserted
// into the sorted remainder where it belongs, so pop() returns the smallest queued key in every case -- the order of a heap.
struct FrontQueue {
    static constexpr int kPerUnit = 16, kBuckets = 16 * 2048;  // distances >= 2048 share the last bucket
    std::vector<std::vector<uint64_t>> later;  // later[b]: keys of bucket b > cb, in push order
    std::vector<uint64_t> cur;                 // keys of the buckets <= cb, ascending; cur[0 .. ci) have popped
    size_t ci = 0;
    int cb = -1, hi = -1;                      // current bucket, highest bucket ever used
    std::vector<uint64_t> px;                  // push number -> (i << 32 | j)
    // Distances >= 2048 share the LAST bucket: once the front is there every push belongs to the current bucket, and a sorted insert per push
    // would make a very large hole quadratic.  Those pushes go into a binary min-heap instead; pop() takes the smaller of
    // the heap's top and the sorted remainder -- still the smallest queued key.
    std::vector<uint64_t> ovf;
    void clear() {
        ovf.clear();
        for (int b = 0; b <= hi && b < (int)later.size(); b++) later[b].clear();
        cur.clear();
        px.clear();
        ci = 0;
        cb = hi = -1;
    }
    void push(int i, int j, float T) {
        uint32_t bits;
        std::memcpy(&bits, &T, 4);
        const uint64_t key = ((uint64_t)bits << 32) | (uint32_t)px.size();
        px.push_back(((uint64_t)(uint32_t)i << 32) | (uint32_t)j);
        const float s = T * (float)kPerUnit;
        // monotone in T (T >= 0 while queued); a NaN or negative T -- neither occurs on the reference's maps -- is given a bucket instead of an
        // undefined conversion (NaN: the last one, negative: the first)
        const int b = !(s < (float)(kBuckets - 1)) ? kBuckets - 1 : (s > 0.f ? (int)s : 0);
        if (b > cb) {
            if (later.empty()) later.resize(kBuckets);
            later[b].push_back(key);
            if (b > hi) hi = b;
        } else if (cb == kBuckets - 1) {
            ovf.push_back(key);
            std::push_heap(ovf.begin(), ovf.end(), std::greater<uint64_t>());
        } else {
            cur.insert(std::lower_bound(cur.begin() + (ptrdiff_t)ci, cur.end(), key), key);
        }
    }
    bool pop(int &i, int &j) {
        while (ci == cur.size() && ovf.empty()) {  // on to the next occupied bucket
            int b = cb + 1;
            while (b <= hi && later[b].empty()) b++;
            if (b > hi) return false;
            cur.clear();
            cur.swap(later[b]);
            std::sort(cur.begin(), cur.end());
            ci = 0;
            cb = b;
        }
        uint64_t key;
        if (!ovf.empty() && (ci == cur.size() || ovf.front() < cur[ci])) {
            std::pop_heap(ovf.begin(), ovf.end(), std::greater<uint64_t>());
            key = ovf.back();
            ovf.pop_back();
        } else {
            key = cur[ci++];
        }
        const uint64_t p = px[(uint32_t)key];
        i = (int)(p >> 32);
        j = (int)(uint32_t)p;
        return true;
    }
};

// photo/src/inpaint.cpp FastMarching_solve for the four quadrants (up|down x left|right) and their minimum; the four
// neighbours are loaded once
inline float front_value(int i, int j, const uint8_t *f, const float *t, int ec) {
    const int c = i * ec + j;
    const double tu = t[c - ec], td = t[c + ec], tl = t[c - 1], tr = t[c + 1];
    const bool ku = f[c - ec] != INSIDE, kd = f[c + ec] != INSIDE, kl = f[c - 1] != INSIDE, kr = f[c + 1] != INSIDE;
    auto solve = [](double a11, bool k1, double a22, bool k2) -> float {
        double sol;
        if (k1) {
            if (k2) {
                if (std::fabs(a11 - a22) >= 1.0) sol = 1 + std::min(a11, a22);
                else sol = (a11 + a22 + std::sqrt((double)(2 - (a11 - a22) * (a11 - a22)))) * 0.5;
            } else
                sol = 1 + a11;
        } else if (k2)
            sol = 1 + a22;
        else
            sol = 1 + std::min(a11, a22);
        return (float)sol;
    };
    const float a = solve(tu, ku, tl, kl), b = solve(td, kd, tl, kl), cc = solve(tu, ku, tr, kr), d = solve(td, kd, tr, kr);
    return std::min(std::min(a, b), std::min(cc, d));
}

// Host state of the front march.  The maps are as large as the padded frame but only ever touched at hole pixels, their
// band and the outward ring: they persist with the context and are reset sparsely (the lists of touched indices), so a call
// costs time proportional to the hole, not to the frame (22 MB of fills per 1080p call otherwise).
struct Par;
void par_free(Par *);
void par_wait(Par *);  // joins component marches that may still be running on the pool

struct March {
    int w = 0, h = 0, range = 1;
    std::vector<float> t;      // (h+2)*(w+2) final distance map (negative outside the hole within `range`); default 1e6
    std::vector<int> ord;      // (h+2)*(w+2): 0 = not a hole pixel, k >= 1 = filled k-th, kNeverFilled = hole not (yet) reached
    std::vector<uint8_t> mask, band, ring;  // flag maps of the reference's set-up; default 0
    std::vector<int> holes, seeds, ring_px; // indices set in mask / band / ring
    std::vector<int> pix;      // fill order: padded linear index of the k-th filled pixel
    std::vector<int> level;    // dependency level (>= 1) of the k-th filled pixel
    std::vector<int> lvl_pix;  // pixels (padded linear index) sorted by (level, order)
    std::vector<int> lvl_ord;  // their order numbers
    std::vector<int> lvl_off;  // CSR offsets per (component, level) segment
    std::vector<int> comp_off; // CSR offsets per component into lvl_off's segments
    // dataflow schedule (radius <= kMaxLdsRange): the pixels of each component in fill order, no levels
    std::vector<int> cmp_pix, cmp_ord, cmp_off;
    std::vector<int> touched;  // scratch of the ring march
    std::vector<int> edge;     // hole pixels with a non-hole 8-neighbour (scratch of march_begin)
    std::vector<int> cmp_wg;   // per workgroup: {first pixel, end, first wavefront slot, slots in total}
    FrontQueue heap;           // the inward front
    FrontQueue ringq;          // the outward front of the ring (Telea), emptied by march_begin itself
    int filled = 0;
    bool dirty = false;
    struct Par *par = nullptr;  // scratch of the component-parallel form of the inward march (march_parallel_run), kept between calls
    bool par_on = false;        // this call's order comes from the merged component streams
    std::vector<int> up_idx, up_ord, cell, stack;  // upload / scheduling scratch kept between calls
    std::vector<float> up_t, up_ft;
    std::vector<int> sc_r, sc_c, sc_rt, sc_ct, sc_tile;  // scratch of the tile schedule (inpaint.hip: build_tile_portion)

    ~March() { par_free(par); }
    March() = default;
    March(const March &) = delete;
    March &operator=(const March &) = delete;

    void clear_sparse() {  // back to the defaults at every index a call touched
        for (int p : holes) { t[p] = 1.0e6f; ord[p] = 0; mask[p] = 0; }
        for (int p : seeds) { t[p] = 1.0e6f; band[p] = 0; }
        for (int p : ring_px) { t[p] = 1.0e6f; ring[p] = 0; }
        holes.clear(); seeds.clear(); ring_px.clear(); pix.clear(); touched.clear();
        heap.clear();
        filled = 0;
        dirty = false;
    }
    void prepare(int w_, int h_, int range_) {
        par_wait(par);  // a call that ended early may have left component marches running: they write t / mask
        par_on = false;
        const size_t en = (size_t)(w_ + 2) * (h_ + 2);
        if (w_ != w || h_ != h || t.size() != en) {
            w = w_; h = h_;
            t.assign(en, 1.0e6f);
            ord.assign(en, 0);
            mask.assign(en + 8, 0);  // (+8: march_begin reads the map in 4-byte words)
            band.assign(en, 0); ring.assign(en, 0);
            holes.clear(); seeds.clear(); ring_px.clear(); pix.clear(); touched.clear();
            heap.clear(); filled = 0; dirty = false;
        } else if (dirty) {
            clear_sparse();
        }
        range = range_;
    }
};

// cvInpaint set-up + icvCalcFMM(negate) (photo/src/inpaint.cpp): hole list, band seeds, outward ring (Telea) -- everything up
// to the inward front, which march_advance() then moves on in portions.  Returns false when there is nothing to fill.
inline bool march_begin(const uint8_t *mask_in, bool outside_ring, March &m) {
    const int w = m.w, h = m.h, range = m.range, ec = w + 2, er = h + 2;
    m.dirty = true;
    uint8_t *mask = m.mask.data(), *band = m.band.data();
    std::vector<int> &holes = m.holes, &seeds = m.seeds;
    for (int i = 0; i < h; i++) {  // the hole pixels in row-major order; a frame is mostly zeros: eight bytes per test
        const uint8_t *row = mask_in + (size_t)i * w;
        const int base = (i + 1) * ec + 1;
        auto take = [&](int j) {
            if (row[j]) {
                mask[base + j] = INSIDE;
                holes.push_back(base + j);
            }
        };
        int j = 0;
        for (; j + 8 <= w; j += 8) {
            uint64_t v;
            std::memcpy(&v, row + j, 8);
            if (!v) continue;
            for (int k = 0; k < 8; k++) take(j + k);
        }
        for (; j < w; j++) take(j);
    }
    if (holes.empty()) return false;
    // band = dilate(mask, 3x3 cross) - mask, frame zeroed; seeds in row-major order.  Only hole pixels with a non-hole
    // 8-neighbour have anything to add (to the band here, to the ring below): the map holds only KNOWN (0) and INSIDE (2) at this
    // point, so the three 3-byte rows around p are all INSIDE exactly when their AND is -- one test per interior pixel.
    const int d4[4] = {-ec, -1, 1, ec}, di4[4] = {-1, 0, 0, 1}, dj4[4] = {0, -1, 1, 0};
    std::vector<int> &edge = m.edge;
    edge.clear();
    auto row3 = [&](int q) {
        uint32_t v;
        std::memcpy(&v, mask + q, 4);
        return v;
    };
    for (int p : holes) {
        if (((row3(p - ec - 1) & row3(p - 1) & row3(p + ec - 1)) & 0xFFFFFFu) == 0x010101u * (unsigned)INSIDE) continue;
        edge.push_back(p);
        const int pi = p / ec, pj = p - pi * ec;
        for (int q = 0; q < 4; q++) {
            const int n = p + d4[q];  // a hole pixel is never on the frame: its 4 neighbours are inside the map
            const int ni = pi + di4[q], nj = pj + dj4[q];
            if (!mask[n] && !band[n] && ni > 0 && nj > 0 && ni < er - 1 && nj < ec - 1) {
                band[n] = INSIDE;
                seeds.push_back(n);
            }
        }
    }
    std::sort(seeds.begin(), seeds.end());
    FrontQueue &outq = m.ringq;
    outq.clear();
    for (int n : seeds) {
        const int i = n / ec, j = n - i * ec;
        m.heap.push(i, j, 0);
        outq.push(i, j, 0);
        m.t[n] = 0;
    }
    int ii, jj;
    float *t = m.t.data();
    if (outside_ring) {  // CV_INPAINT_TELEA only; CV_INPAINT_NS leaves T = 1e6 off the band
        // ring = dilate(mask, (2r+1)^2 rect) - mask - band, frame zeroed.  A non-hole pixel within Chebyshev distance r of
        // the hole is within r of a hole pixel that has a non-hole 8-neighbour, so only those spread the ring.
        uint8_t *ring = m.ring.data();
        bool any_ring = false;
        for (int p : edge) {
            const int pi = p / ec, pj = p - pi * ec;
            any_ring = true;
            for (int a = std::max(pi - range, 1); a <= std::min(pi + range, er - 2); a++)
                for (int c = std::max(pj - range, 1); c <= std::min(pj + range, ec - 2); c++)
                    if (!mask[a * ec + c] && !band[a * ec + c] && !ring[a * ec + c]) {
                        ring[a * ec + c] = INSIDE;
                        m.ring_px.push_back(a * ec + c);
                    }
        }
        if (!any_ring) return false;  // Out->Init fails in the reference: cvInpaint returns without filling
        uint8_t *f = ring;
        while (outq.pop(ii, jj)) {
            f[ii * ec + jj] = CHANGE;
            const int ni[4] = {ii - 1, ii, ii + 1, ii}, nj[4] = {jj, jj - 1, jj, jj + 1};
            for (int q = 0; q < 4; q++) {
                int i = ni[q], j = nj[q];
                if (i <= 0 || j <= 0 || i > er || j > ec) continue;
                if (f[i * ec + j] != INSIDE) continue;
                float dist = front_value(i, j, f, t, ec);
                t[i * ec + j] = dist;
                f[i * ec + j] = BAND;
                outq.push(i, j, dist);
                m.touched.push_back(i * ec + j);
            }
        }
        for (int n : seeds)
            if (f[n] == CHANGE) t[n] = -t[n];
        for (int n : m.touched)
            if (f[n] == CHANGE) t[n] = -t[n];
        m.touched.clear();
        // the seeds were marked CHANGE in the ring map as they were popped: they are reset with the seeds
        for (int n : seeds) ring[n] = 0;
    }
    // inward front over the hole; the reference passes `mask` ({KNOWN, INSIDE}) as the flag map
    for (int p : holes) m.ord[p] = kNeverFilled;
    m.filled = 0;
    return true;
}

// the front recurrence of icvTeleaInpaintFMM: moves the inward front on until at least `want` more pixels have their
// distance and order number (or the front is exhausted); returns how many were added to m.pix
int par_advance(March &m, int want);
inline int march_advance(March &m, int want) {
    if (m.par_on) return par_advance(m, want);
    const int ec = m.w + 2, er = m.h + 2;
    uint8_t *f = m.mask.data();
    float *t = m.t.data();
    const size_t before = m.pix.size();
    int ii, jj;
    while ((int)(m.pix.size() - before) < want && m.heap.pop(ii, jj)) {
        f[ii * ec + jj] = KNOWN;
        const int ni[4] = {ii - 1, ii, ii + 1, ii}, nj[4] = {jj, jj - 1, jj, jj + 1};
        for (int q = 0; q < 4; q++) {
            int i = ni[q], j = nj[q];
            if (i <= 1 || j <= 1 || i > er - 1 || j > ec - 1) continue;
            if (f[i * ec + j] != INSIDE) continue;
            float dist = front_value(i, j, f, t, ec);
            t[i * ec + j] = dist;
            f[i * ec + j] = BAND;
            m.heap.push(i, j, dist);
            m.ord[i * ec + j] = ++m.filled;
            m.pix.push_back(i * ec + j);
        }
    }
    return (int)(m.pix.size() - before);
}

// ------------------------------------------------------------------ component-parallel inward march
//
// The inward front only ever reads the distance / flag of a pixel's four neighbours, and a hole pixel's non-hole
// neighbours are band pixels (T = 0, static): 4-connected components of the hole never see each other, so their fronts
// can be marched independently -- one component per host thread.  What is global is the ORDER: cvInpaint fills a pixel
// when it is pushed, and the queue pops the smallest T first, equal T in push order (CvPriorityQueueFloat).  A pushed
// element is identified by (rank of the pop that pushed it, which of that pixel's four neighbours it is): parents pop
// before children, so merging the components' pop sequences by the key (T, global pop rank of the parent, neighbour index)
// reproduces the sequential pop order exactly, and numbering the children of every pop in that order reproduces the fill
// order.  The band seeds (all T = 0, pushed in row-major order) pop before any hole pixel (a hole pixel's T is at least
// 0.707) and may have children in several components.
struct Par {
    struct Push {
        int parent;  // >= 0: local pop number of the hole pixel that pushed it; < 0: -1 - (global index of the band seed)
        int q;       // which neighbour of the parent (up, left, down, right)
        int pixel;   // padded linear index
        uint32_t tbits;  // its distance (bit pattern: T >= 0 orders like its bits)
    };
    struct Comp {
        std::vector<int> seeds;         // global seed indices, ascending
        std::vector<Push> pushes;       // in local push order (the children of one pop are contiguous, in q order)
        std::vector<int> pop_push;      // k-th local pop of a hole pixel -> index into pushes
        std::vector<int> child_begin;   // k-th local pop -> first of its children in pushes (size pops + 1)
        std::vector<int> seed_child_begin;  // i-th seed of this component -> first of its children (size seeds + 1)
        std::vector<int> grank;         // global pop rank of the k-th local pop (set by the merge)
        int next = 0;                   // merge cursor: next local pop
        int size = 0;                   // hole pixels of the component
    };
    // What a marching thread publishes to the merging thread while it is still at work (streamed merge): the number of local
    // pops whose children are complete, that the children of its seeds are complete, that it has finished.  The vectors of a
    // Comp are reserved to their final size before the march starts (no reallocation), so everything below a published
    // count may be read while the rest is still being appended.
    struct Sync {
        alignas(64) std::atomic<int> closed{0};
        std::atomic<int> seeds_done{0}, finished{0};
    };
    std::unique_ptr<Sync[]> sync;
    bool async = false;                 // component marches may still be running on the pool
    bool seeds_merged = false;
    std::vector<int> waiting;           // components whose next pop the merge has to wait for
    std::vector<Comp> comps;
    std::vector<int> label;             // padded map: component of a hole pixel (reset sparsely)
    std::vector<int> labelled;
    // merge state
    struct Head {
        uint32_t tbits;
        int gparent, q, comp;
        bool operator>(const Head &o) const {
            if (tbits != o.tbits) return tbits > o.tbits;
            if (gparent != o.gparent) return gparent > o.gparent;
            return q > o.q;
        }
    };
    std::priority_queue<Head, std::vector<Head>, std::greater<Head>> heads;
    std::vector<int> seed_order;        // (seed, q, comp, push) of every seed child, sorted by (seed, q): 4 ints each
    size_t seed_cursor = 0;
    int nseeds = 0, next_rank = 0;
    bool heads_ready = false;
};


// a few helper threads shared by all contexts (created on first use, never joined: no static destructor while a host unloads us)
class MarchPool {
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
    MarchPool() {
        const unsigned hc = std::thread::hardware_concurrency();
        nworkers_ = hc >= 16 ? 7 : (hc >= 8 ? 3 : (hc >= 4 ? 1 : 0));
        for (int i = 0; i < nworkers_; i++) std::thread([this] { worker(); }).detach();
    }

    std::unique_lock<std::mutex> async_user_;
    std::function<void(int)> held_;

public:
    // tasks 0..n-1 on the workers only; the caller goes on (to merge what they produce) and calls finish() afterwards.
    // false: the pool is in use or has no workers -- the caller runs the tasks itself.
    bool start(int n, std::function<void(int)> fn) {
        std::unique_lock<std::mutex> user(user_, std::try_to_lock);
        if (!user.owns_lock() || nworkers_ == 0 || n <= 0) return false;
        async_user_ = std::move(user);
        held_ = std::move(fn);
        {
            std::lock_guard<std::mutex> lk(m_);
            job_ = &held_;
            ntasks_ = n;
            next_.store(0);
            active_ = nworkers_;
            ++gen_;
        }
        cv_job_.notify_all();
        return true;
    }
    void finish() {
        {
            std::unique_lock<std::mutex> lk(m_);
            cv_done_.wait(lk, [&] { return active_ == 0; });
            job_ = nullptr;
        }
        async_user_.unlock();
        async_user_ = std::unique_lock<std::mutex>();
    }
    static MarchPool &get() {
        static MarchPool *p = new MarchPool();  // intentionally leaked
        return *p;
    }
    int workers() const { return nworkers_; }
    // tasks 0..n-1, taken dynamically; a second caller at the same time runs its tasks on its own thread
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

inline void par_wait(Par *p) {
    if (p && p->async) {
        MarchPool::get().finish();
        p->async = false;
    }
}
inline void par_free(Par *p) {
    par_wait(p);
    delete p;
}

// the inward march of one component (everything icvTeleaInpaintFMM's front does to this component's pixels)
inline void par_march_component(March &m, Par &P, int ci) {
    Par::Comp &C = P.comps[ci];
    Par::Sync &S = P.sync[ci];
    const int ec = m.w + 2, er = m.h + 2;
    uint8_t *f = m.mask.data();
    float *t = m.t.data();
    const int *lab = P.label.data();
    // local queue: key = T bits << 32 | local push number (same relative order as the global push numbers, see above)
    std::priority_queue<uint64_t, std::vector<uint64_t>, std::greater<uint64_t>> q;
    std::vector<int> who;  // local push number -> index into pushes, or -1 - (position in C.seeds)
    who.reserve(C.seeds.size() + C.size);
    C.pushes.reserve(C.size);
    C.pop_push.reserve(C.size);
    C.child_begin.reserve(C.size + 1);
    C.seed_child_begin.reserve(C.seeds.size() + 1);
    for (size_t i = 0; i < C.seeds.size(); i++) {
        q.push((uint64_t)who.size());  // T = 0
        who.push_back(-1 - (int)i);
    }
    while (!q.empty()) {
        const int wh = who[(uint32_t)q.top()];
        q.pop();
        int ii, jj, parent;
        if (wh < 0) {  // a band seed: flag stays KNOWN (it is a band pixel of the mask map)
            const int n = m.seeds[C.seeds[-1 - wh]];
            ii = n / ec;
            jj = n - ii * ec;
            parent = -1 - C.seeds[-1 - wh];
            C.seed_child_begin.push_back((int)C.pushes.size());
        } else {
            const int n = C.pushes[wh].pixel;
            ii = n / ec;
            jj = n - ii * ec;
            f[n] = KNOWN;
            if (C.pop_push.empty()) {
                C.seed_child_begin.push_back((int)C.pushes.size());  // the seeds (T = 0) have all popped: end of the last one's children
                S.seeds_done.store(1, std::memory_order_release);
            }
            parent = (int)C.pop_push.size();
            C.pop_push.push_back(wh);
            C.child_begin.push_back((int)C.pushes.size());
            S.closed.store(parent, std::memory_order_release);  // the pops before this one have all their children
        }
        const int ni[4] = {ii - 1, ii, ii + 1, ii}, nj[4] = {jj, jj - 1, jj, jj + 1};
        for (int k = 0; k < 4; k++) {
            const int i = ni[k], j = nj[k];
            if (i <= 1 || j <= 1 || i > er - 1 || j > ec - 1) continue;
            const int n = i * ec + j;
            if (lab[n] != ci || f[n] != INSIDE) continue;  // (a seed may border other components: their threads push -- and write -- those)
            const float dist = front_value(i, j, f, t, ec);
            t[n] = dist;
            f[n] = BAND;
            uint32_t bits;
            std::memcpy(&bits, &dist, 4);
            q.push(((uint64_t)bits << 32) | (uint32_t)who.size());
            who.push_back((int)C.pushes.size());
            C.pushes.push_back({parent, k, n, bits});
        }
    }
    C.child_begin.push_back((int)C.pushes.size());
    if (C.pop_push.empty()) C.seed_child_begin.push_back((int)C.pushes.size());
    S.closed.store((int)C.pop_push.size(), std::memory_order_release);
    S.seeds_done.store(1, std::memory_order_release);
    S.finished.store(1, std::memory_order_release);
}

// After march_begin(): labels the 4-connected components of the hole, marches them on the pool and prepares the merge;
// march_advance() then hands the merged fill order out in portions exactly like the serial form.  Returns false (and
// leaves the serial form in place) when the hole is too small or in one piece.
inline bool march_parallel_run(March &m, int min_pixels = 8192) {
    if ((int)m.holes.size() < min_pixels || MarchPool::get().workers() == 0) return false;
    const int ec = m.w + 2, er = m.h + 2;
    if (!m.par) m.par = new Par();
    Par &P = *m.par;
    const size_t en = (size_t)ec * er;
    if (P.label.size() != en) {
        P.label.assign(en, -1);
        P.labelled.clear();
    }
    for (int p : P.labelled) P.label[p] = -1;
    P.labelled.clear();
    P.comps.clear();
    // components by flood fill over the hole pixels (flag INSIDE in the mask map at this point)
    const uint8_t *f = m.mask.data();
    std::vector<int> &stack = m.stack;
    const int d4[4] = {-ec, -1, 1, ec};
    for (int p0 : m.holes) {
        if (P.label[p0] >= 0) continue;
        const int ci = (int)P.comps.size();
        P.comps.emplace_back();
        int cnt = 0;
        stack.assign(1, p0);
        P.label[p0] = ci;
        P.labelled.push_back(p0);
        while (!stack.empty()) {
            const int p = stack.back();
            stack.pop_back();
            cnt++;
            for (int k = 0; k < 4; k++) {
                const int n = p + d4[k];  // a hole pixel is never on the frame of the padded map
                if (f[n] == INSIDE && P.label[n] < 0) {
                    P.label[n] = ci;
                    P.labelled.push_back(n);
                    stack.push_back(n);
                }
            }
        }
        P.comps[ci].size = cnt;
    }
    if (P.comps.size() < 2) return false;  // one piece: nothing to run side by side
    // the seeds of every component: band pixels with a neighbour in it (global seed order = row-major = m.seeds, sorted)
    P.nseeds = (int)m.seeds.size();
    for (int s = 0; s < P.nseeds; s++) {
        const int n = m.seeds[s];
        int last[4], nl = 0;
        for (int k = 0; k < 4; k++) {
            const int c = f[n + d4[k]] == INSIDE ? P.label[n + d4[k]] : -1;  // (a band pixel is inside the map's frame)
            bool seen = c < 0;
            for (int u = 0; u < nl; u++) seen = seen || last[u] == c;
            if (!seen) {
                last[nl++] = c;
                P.comps[c].seeds.push_back(s);
            }
        }
    }
    // largest components first (dynamic scheduling over the pool).  The marches run on the pool's threads while this thread
    // merges what they have produced so far (par_advance): the first portion of the fill order leaves after a fraction of a
    // millisecond instead of after the largest component's whole march.
    std::vector<int> order(P.comps.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = (int)i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return P.comps[a].size > P.comps[b].size; });
    P.sync.reset(new Par::Sync[P.comps.size()]);
    for (Par::Comp &C : P.comps) C.grank.assign(C.size, 0);
    Par *pp = &P;
    March *mp = &m;
    P.async = MarchPool::get().start((int)order.size(), [mp, pp, order](int i) { par_march_component(*mp, *pp, order[i]); });
    if (!P.async)
        for (int ci : order) par_march_component(m, P, ci);
    P.seed_order.clear();
    P.seeds_merged = false;
    P.waiting.clear();
    P.seed_cursor = 0;
    P.next_rank = P.nseeds;
    P.heads = decltype(P.heads)();
    P.heads_ready = false;
    m.par_on = true;
    return true;
}

// merge, seed phase: the children of the seeds in (seed, neighbour) order (every component has popped its seeds by then)
inline void par_merge_seeds(Par &P) {
    for (size_t ci = 0; ci < P.comps.size(); ci++)
        while (!P.sync[ci].seeds_done.load(std::memory_order_acquire)) std::this_thread::yield();
    P.seed_order.clear();
    for (size_t ci = 0; ci < P.comps.size(); ci++) {
        const Par::Comp &C = P.comps[ci];
        for (size_t i = 0; i < C.seeds.size(); i++)
            for (int u = C.seed_child_begin[i]; u < C.seed_child_begin[i + 1]; u++) {
                const int rec[4] = {C.seeds[i], C.pushes[u].q, (int)ci, u};
                P.seed_order.insert(P.seed_order.end(), rec, rec + 4);
            }
    }
    {
        const size_t n = P.seed_order.size() / 4;
        std::vector<size_t> idx(n);
        for (size_t i = 0; i < n; i++) idx[i] = i;
        std::sort(idx.begin(), idx.end(), [&](size_t a, size_t b) {
            const int *x = &P.seed_order[a * 4], *y = &P.seed_order[b * 4];
            return x[0] != y[0] ? x[0] < y[0] : x[1] < y[1];
        });
        std::vector<int> sorted(P.seed_order.size());
        for (size_t i = 0; i < n; i++) std::memcpy(&sorted[i * 4], &P.seed_order[idx[i] * 4], 16);
        P.seed_order.swap(sorted);
    }
    P.seeds_merged = true;
}

// hands out the merged fill order: at least `want` more pixels (or all that are left); same contract as march_advance
inline int par_advance(March &m, int want) {
    Par &P = *m.par;
    // (the host order map is not written here: nothing reads it after the march -- the device map gets its order numbers from
    // the positions in m.pix -- and a random 4-byte store per pixel would double the cost of the merge)
    const size_t before = m.pix.size();
    auto emit = [&](int pixel) {
        ++m.filled;
        m.pix.push_back(pixel);
    };
    if (!P.seeds_merged) par_merge_seeds(P);
    const size_t nseed_children = P.seed_order.size() / 4;
    while (P.seed_cursor < nseed_children && (int)(m.pix.size() - before) < want) {
        const int *r = &P.seed_order[P.seed_cursor * 4];
        emit(P.comps[r[2]].pushes[r[3]].pixel);
        P.seed_cursor++;
    }
    if (P.seed_cursor < nseed_children) return (int)(m.pix.size() - before);
    auto head_of = [&](int ci) {
        Par::Comp &C = P.comps[ci];
        const Par::Push &pu = C.pushes[C.pop_push[C.next]];
        Par::Head h;
        h.tbits = pu.tbits;
        h.gparent = pu.parent < 0 ? -1 - pu.parent : C.grank[pu.parent];
        h.q = pu.q;
        h.comp = ci;
        return h;
    };
    // the next pop of component ci for the merge: waits until its thread has closed it (all its children pushed) or finished
    auto fetch = [&](int ci) {
        Par::Comp &C = P.comps[ci];
        Par::Sync &S = P.sync[ci];
        for (;;) {
            if (C.next < S.closed.load(std::memory_order_acquire)) {
                P.heads.push(head_of(ci));
                return;
            }
            if (S.finished.load(std::memory_order_acquire)) {
                if (C.next < S.closed.load(std::memory_order_acquire)) P.heads.push(head_of(ci));
                return;
            }
            std::this_thread::yield();
        }
    };
    if (!P.heads_ready) {
        for (size_t ci = 0; ci < P.comps.size(); ci++) P.waiting.push_back((int)ci);
        P.heads_ready = true;
    }
    while ((int)(m.pix.size() - before) < want) {
        for (int ci : P.waiting) fetch(ci);  // the smallest key can only be chosen among ALL components' next pops
        P.waiting.clear();
        if (P.heads.empty()) break;
        const int ci = P.heads.top().comp;
        P.heads.pop();
        Par::Comp &C = P.comps[ci];
        const int k = C.next++;
        C.grank[k] = P.next_rank++;
        for (int u = C.child_begin[k]; u < C.child_begin[k + 1]; u++) emit(C.pushes[u].pixel);
        P.waiting.push_back(ci);
    }
    if (P.heads.empty() && P.waiting.empty()) par_wait(&P);  // everything is merged: the marches have finished
    return (int)(m.pix.size() - before);
}

}  // namespace ofxcv_telea
