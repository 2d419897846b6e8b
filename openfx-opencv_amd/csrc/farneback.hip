// farneback.hip -- dense Farneback optical flow for gfx950 (MI355X).
//
// Replaces cv::calcOpticalFlowFarneback as called at VectorGenerator/VectorGenerator.cpp:403
// (pyr_scale 0.5, winsize 3, flags 0; levels / iterations / poly_n / poly_sigma are plugin
// parameters, :390-399).  The stages follow OpenCV's modules/video/src/optflowgf.cpp:
//   pyramid image   convertTo(32F) + GaussianBlur(full res) + resize(INTER_LINEAR)   [F1,F2]
//   polyexp         FarnebackPolyExp: separable (2n+1)^2 weighted quadratic fit      [F3]
//   update          FarnebackUpdateMatrices: warped gather of R1 + border scale      [F4]
//   blur+solve      FarnebackUpdateFlow_Blur: box window of M, 2x2 solve per pixel   [F5]
//   prolongation    resize(prevFlow, INTER_LINEAR) * 1/pyr_scale                     [F6]
//
// Device data layout: every 5-channel field (R0, R1, M) is stored as 5 planes of
// `pitch` x height floats (pitch = width rounded up to 64 floats = 256 B) so that a wavefront
// reading 64 consecutive pixels of one plane issues one fully coalesced 256-byte request.
// The flow field is 2-channel interleaved (float2 per pixel), as OpenCV returns it.
//
// Numerics: every stage evaluates the same IEEE operations in the same order as the CPU code it
// replaces (f32 products and sums where OpenCV uses float, f64 accumulators where it uses
// double); the file is compiled with -ffp-contract=off so no FMA contraction changes a rounding.
// The one deliberate difference: OpenCV's box window keeps running sums (and rounds each row
// difference to f32 before accumulating it); here each pixel sums its own 3x3 window in f64.
#include "fb.h"

using namespace ofxcv_fb;

namespace ofxcv_fb {

// ------------------------------------------------------------------ host-side geometry (optflowgf.cpp calc())

int num_levels(int w, int h, double pyr_scale, int levels) {
    const int min_size = 32;
    int k;
    double scale = 1;
    for (k = 0; k < levels; k++) {
        scale *= pyr_scale;
        if (w * scale < min_size || h * scale < min_size) break;
    }
    return k;
}

void level_geom(int w, int h, double pyr_scale, int k, int &lw, int &lh, double &sigma, int &ksize) {
    double scale = 1;
    for (int i = 0; i < k; i++) scale *= pyr_scale;
    sigma = (1. / scale - 1) * 0.5;
    ksize = ofxcv_cv_round(sigma * 5) | 1;
    if (ksize < 3) ksize = 3;
    lw = ofxcv_cv_round(w * scale);
    lh = ofxcv_cv_round(h * scale);
}



// f64 scratch of the OpenCV-order / Gaussian window kernels for one pair at geometry w x h: the larger of
//   serial column scan / Gaussian vertical pass   5 * pitch * h values
//   overlapped strips                             4 * (strips of >= 9 rows + 1) * 5 * pitch + the edge rows (the column-owning form uses the edge rows only)
size_t vsum_doubles(int w, int h) {
    const size_t pitch = (size_t)plane_pitch(w);
    const size_t a = 5 * pitch * h;
    const size_t d = 4 * (size_t)(ofxcv_div_up(h, 9) + 1) * 5 * pitch + 16 * pitch;  // overlapped strips: T, T' and the edge rows, two buffers
    return round_up(std::max(a, d), 32);
}

int make_layout(ofxcv_ctx *ctx, int n, int width, int height, double pyr_scale, int levels, Layout &L) {
    L.n = n;
    L.field0 = 5 * (size_t)plane_pitch(width) * height;
    L.rtotal = 0;
    for (int k = levels; k >= 0; k--) {
        int w, h, ksz;
        double sigma;
        level_geom(width, height, pyr_scale, k, w, h, sigma, ksz);
        if (ksz > kMaxGaussTaps) return ofxcv_fail(ctx, OFXCV_ERR_UNSUPPORTED, "pyramid blur of %d taps exceeds %d", ksz, kMaxGaussTaps);
        L.rtotal += 2 * 5 * (size_t)plane_pitch(w) * h;
    }
    L.planes = 2 * L.field0 + L.rtotal;
    L.t1 = round_up((size_t)2 * (width + 4) * height, 64);  // (twice a frame's rows at level 0: all 32 frames of a call at the levels the fall-back serves)
    L.img = round_up((size_t)width * height, 64);
    L.cflow = 0;
    if (levels > 0) {
        int lw, lh, ks;
        double sg;
        level_geom(width, height, pyr_scale, 1, lw, lh, sg, ks);
        L.cflow = round_up((size_t)lw * lh * 2, 64);
    }
    L.vsum = vsum_doubles(width, height);
    return OFXCV_OK;
}

}  // namespace ofxcv_fb

namespace {
FlowTab one_flow(float *p, size_t step) {
    FlowTab t = {};
    t.p[0] = p;
    t.step[0] = step;
    return t;
}
Layout one_pair_layout() { return Layout(); }

}  // namespace

extern "C" {

int ofxcv_farneback_plane_pitch(int width) { return plane_pitch(width); }

// measurement aid (not part of the public header): the shader-clock stamps the traced workgroup of the last iterate_col_kernel
// launch left (option "farneback.col_trace" 1); n 64-bit words
int ofxcv_debug_col_trace(ofxcv_ctx *ctx, unsigned long long *out, int n) {
    if (!ctx || !out || !ctx->fb_col_flag.ptr || (size_t)n * 8 + 256 > ctx->fb_col_flag.bytes) return OFXCV_ERR_INVALID;
    OFXCV_HIP_CHECK(ctx, hipSetDevice(ctx->hip_device));
    int rc = ofxcv_ctx_quiesce(ctx);
    if (rc) return rc;
    OFXCV_HIP_CHECK(ctx, hipMemcpy(out, (char *)ctx->fb_col_flag.ptr + 256, (size_t)n * 8, hipMemcpyDeviceToHost));
    return OFXCV_OK;
}

int ofxcv_farneback_num_levels(int width, int height, double pyr_scale, int levels) {
    return num_levels(width, height, pyr_scale, levels);
}

int ofxcv_farneback_level_geom(int width, int height, double pyr_scale, int k, int *lw, int *lh, double *sigma, int *ksize) {
    if (!lw || !lh || !sigma || !ksize || width <= 0 || height <= 0 || k < 0) return OFXCV_ERR_INVALID;
    level_geom(width, height, pyr_scale, k, *lw, *lh, *sigma, *ksize);
    return OFXCV_OK;
}

int ofxcv_farneback_pyr_image(ofxcv_ctx *ctx, const uint8_t *d_img, size_t step, int width, int height, int lw, int lh,
                              double sigma, int ksize, float *d_I, void *stream) {
    if (!ctx) return OFXCV_ERR_INVALID;
    OFXCV_HIP_CHECK(ctx, hipSetDevice(ctx->hip_device));  // a thread may hold contexts on several devices
    if (!d_img || !d_I || width <= 0 || height <= 0 || lw <= 0 || lh <= 0 || ksize < 1 || !(ksize & 1) || step < (size_t)width)
        return ofxcv_fail(ctx, OFXCV_ERR_INVALID, "farneback_pyr_image: bad argument");
    int rc = ofxcv_reserve(ctx, ctx->fb_tmp, sizeof(float) * ((size_t)(2 * lw + 2) * height));
    if (rc) return rc;
    ImgTab imgs = {};
    imgs.p[0] = d_img;
    imgs.step[0] = step;
    return launch_pyr_image(ctx, ofxcv_stream(ctx, stream), imgs, 1, width, height, lw, lh, sigma, ksize, (float *)ctx->fb_tmp.ptr,
                            (size_t)(2 * lw + 2) * height, d_I, 0);
}

int ofxcv_farneback_polyexp(ofxcv_ctx *ctx, const float *d_I, int width, int height, float *d_R, int poly_n, double poly_sigma,
                            void *stream) {
    if (!ctx) return OFXCV_ERR_INVALID;
    OFXCV_HIP_CHECK(ctx, hipSetDevice(ctx->hip_device));  // a thread may hold contexts on several devices
    if (!d_I || !d_R || width <= 0 || height <= 0) return ofxcv_fail(ctx, OFXCV_ERR_INVALID, "farneback_polyexp: bad argument");
    return launch_polyexp(ctx, ofxcv_stream(ctx, stream), d_I, width, height, d_R, poly_n, poly_sigma, 1, 0, 0, 0, false);
}

int ofxcv_farneback_update_matrices(ofxcv_ctx *ctx, const float *d_R0, const float *d_R1, const float *d_flow, size_t flow_step,
                                    int width, int height, float *d_M, void *stream) {
    if (!ctx) return OFXCV_ERR_INVALID;
    OFXCV_HIP_CHECK(ctx, hipSetDevice(ctx->hip_device));  // a thread may hold contexts on several devices
    if (!d_R0 || !d_R1 || !d_flow || !d_M || width <= 0 || height <= 0 || (flow_step & 7))
        return ofxcv_fail(ctx, OFXCV_ERR_INVALID, "farneback_update_matrices: bad argument");
    const Prolong no_pr = {0, 0, 1.0, 1.0, 1.0};
    return launch_update_matrices(ctx, ofxcv_stream(ctx, stream), 2, d_R0, d_R1, one_flow(const_cast<float *>(d_flow), flow_step), no_pr, width, height, d_M, 0, 1, false);
}

int ofxcv_farneback_update_flow_blur(ofxcv_ctx *ctx, const float *d_R0, const float *d_R1, const float *d_M_in, float *d_M_out,
                                     float *d_flow, size_t flow_step, int width, int height, int winsize, int update, void *stream) {
    if (!ctx) return OFXCV_ERR_INVALID;
    OFXCV_HIP_CHECK(ctx, hipSetDevice(ctx->hip_device));  // a thread may hold contexts on several devices
    if (!d_M_in || width <= 0 || height <= 0 || winsize < 1 || !(winsize & 1) || (d_flow && (flow_step & 7)) ||
        (update && (!d_R0 || !d_R1 || !d_M_out || d_M_out == d_M_in)) || (!update && !d_flow))
        return ofxcv_fail(ctx, OFXCV_ERR_INVALID, "farneback_update_flow_blur: bad argument");
    Layout L = one_pair_layout();
    L.vsum = vsum_doubles(width, height);
    if (ctx->fb_opencv_rounding) {
        int rc = ofxcv_reserve(ctx, ctx->fb_vsum, sizeof(double) * L.vsum);
        if (rc) return rc;
    }
    L.vsum_ptr = (double *)ctx->fb_vsum.ptr;
    return launch_iteration(ctx, ofxcv_stream(ctx, stream), d_R0, d_R1, d_M_in, d_M_out, one_flow(d_flow, flow_step), width, height, winsize,
                            update != 0, L, false);
}

// The launch sequence of one call (n frame pairs).  The pyramid images and polynomial expansions of ALL levels depend only
// on the input frames, so they run on the context's preparation stream (coarsest level first) while the main stream
// walks the levels; an event per level hands R0/R1 over.  The coarse levels of a single pair cannot fill the chip (a
// 240x135 level is 127 workgroups on 256 CUs): their launches are latency-bound, which is what a batch amortises --
// every launch of the walk carries all n pairs in its grid's z dimension.
static int enqueue_farneback(ofxcv_ctx *ctx, hipStream_t s, hipStream_t sp, const Layout &L, const ImgTab &imgs, const FlowTab &out,
                             int width, int height, double pyr_scale, int levels, int winsize,
                             int iterations, int poly_n, double poly_sigma, int flags, bool profile, const RgbaTab *rgba = nullptr) {
    int rc;
    const int n = L.n;
    // scratch carving (sizes were reserved by the caller); pointers are pair 0's
    float *base = (float *)ctx->fb_planes.ptr;
    float *Mbuf[2] = {base, base + L.field0};
    float *Rk = base + 2 * L.field0;  // R0/R1 of level levels, levels-1, ..., 0 packed one after the other
    float *T1 = (float *)ctx->fb_tmp.ptr;
    float *I = T1 + L.t1;
    float *cflow[2] = {(float *)ctx->fb_flow.ptr, (float *)ctx->fb_flow.ptr + L.cflow};

    // fork: the preparation stream starts once the inputs are ready on the main stream
    OFXCV_HIP_CHECK(ctx, hipEventRecord(ctx->ev_fork, s));
    OFXCV_HIP_CHECK(ctx, hipStreamWaitEvent(sp, ctx->ev_fork, 0));
    float *R[kMaxLevels + 1][2];
    {
        float *p = Rk;
        for (int k = levels; k >= 0; k--) {
            int w, h, ksz;
            double sigma;
            level_geom(width, height, pyr_scale, k, w, h, sigma, ksz);
            const size_t field = 5 * (size_t)plane_pitch(w) * h;
            R[k][0] = p;
            R[k][1] = p + field;
            p += 2 * field;
            // (measurement probe farneback.reuse_prep: the pyramid images and expansions a previous call on the SAME frames left in the scratch are used
            // as they are -- the upper bound of what keeping a named frame's expansions on the device could save; results are unchanged)
            if (!ctx->fb_reuse_prep) {
                rc = launch_pyr_image(ctx, sp, imgs, 2 * n, width, height, w, h, sigma, ksz, T1, L.t1, I, L.img);
                if (rc) return rc;
                rc = launch_polyexp(ctx, sp, I, w, h, R[k][0], poly_n, poly_sigma, 2 * n, L.img, L.planes, field, true);  // R1 = the odd frames, packed
                if (rc) return rc;
            }
            OFXCV_HIP_CHECK(ctx, hipEventRecord(ctx->ev_level[k], sp));
        }
    }
    // Launch groups.  A level whose working set (M ping + M pong + R0 + R1 = 80 B/px) for the whole batch stays inside the
    // Infinity Cache is walked with all pairs in every launch; a larger level is walked in groups of as many pairs as fit
    // (at least one), group after group, so that the fields an iteration re-reads are still on the die when it comes back
    // to them (measured at 1920x1080, level 0: 42.5 us per pair and iteration alone, 46.8 us in a batch of three).
    const size_t budget = (size_t)std::max(1, ctx->fb_batch_mb) << 20;
    const size_t pair_cflow = 2 * L.cflow;
    auto sub_tab = [&](const FlowTab &t, int z0, int gn) {
        FlowTab r = {};
        for (int z = 0; z < gn; z++) {
            r.p[z] = t.p[z0 + z];
            r.step[z] = t.step[z0 + z];
        }
        return r;
    };
    auto coarse_tab = [&](float *p0, size_t step) {
        FlowTab t = {};
        for (int z = 0; z < n; z++) {
            t.p[z] = p0 + (size_t)z * pair_cflow;
            t.step[z] = step;
        }
        return t;
    };
    const FlowTab no_flow = {};
    FlowTab prev_all = {};
    bool have_prev = false;
    int pw = 0, ph = 0;
    bool rgba_fused[kMaxBatch] = {};  // F7 of the pair was done by the last level-0 launch (overlapped-strip form); otherwise it follows as its own launch
    hipStream_t s_main = s;
    const bool use_coarse = ctx->coarse && levels > 0 && sp != s && !profile;
    for (int k = levels; k >= 0; k--) {
        int w, h, ksz;
        double sigma;
        level_geom(width, height, pyr_scale, k, w, h, sigma, ksz);
        // the coarse levels may run on a high-priority stream of their own: their short launches are then not queued behind
        // another call's full-size kernels (several calls in flight on one device)
        if (use_coarse && k == levels) OFXCV_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->coarse, ctx->ev_fork, 0));
        if (use_coarse && k == 0) {
            OFXCV_HIP_CHECK(ctx, hipEventRecord(ctx->ev_coarse, ctx->coarse));
            OFXCV_HIP_CHECK(ctx, hipStreamWaitEvent(s_main, ctx->ev_coarse, 0));
        }
        hipStream_t s = (use_coarse && k > 0) ? ctx->coarse : s_main;
        OFXCV_HIP_CHECK(ctx, hipStreamWaitEvent(s, ctx->ev_level[k], 0));  // join (level 0's wait closes the fork)
        const int pitch = plane_pitch(w);
        const size_t level_bytes = 4 * sizeof(float) * 5 * (size_t)pitch * h;
        const int fit = (int)std::min<size_t>((size_t)n, std::max<size_t>(1, budget / level_bytes));
        const bool gaussian = (flags & OFXCV_OPTFLOW_FARNEBACK_GAUSSIAN) != 0;
        // the column-owning form streams every pair of the call through one launch (a workgroup per tile column and pair; the
        // fields are read once per two iterations, so the Infinity-Cache grouping below has nothing to keep on the die)
        // (one exception: a single iteration on a caller-supplied flow at level 0 would pair "first M from the given flow" with "last: flow
        // out" in ONE launch over the SAME buffer -- a workgroup's halo lanes read columns its neighbour overwrites; found by tests/perf/fuzz_halo.py)
        const bool given_in_place = k == 0 && !have_prev && (flags & OFXCV_OPTFLOW_USE_INITIAL_FLOW) && iterations == 1;
        const int ncol = given_in_place ? 0 : col_pairs(ctx, w, h, n, ctx->fb_opencv_rounding == 1 && winsize == 3 && !gaussian);
        struct Group {
            int z0, gn;
            bool col;
        };
        Group plan[kMaxBatch + 1];
        int ngroups = 0;
        if (ncol) plan[ngroups++] = {0, ncol, true};
        if (ncol < n) {
            const int rest = n - ncol;
            const int per_group = ofxcv_div_up(rest, ofxcv_div_up(rest, std::min(fit, rest)));  // groups of equal size (4 pairs, 3 fit: 2 + 2, not 3 + 1)
            for (int z = ncol; z < n; z += per_group) plan[ngroups++] = {z, std::min(per_group, n - z), false};
        }
        const FlowTab out_all = k == 0 ? out : coarse_tab(cflow[k & 1], (size_t)w * 8);
        for (int gi = 0; gi < ngroups; gi++) {
            const int z0 = plan[gi].z0, gn = plan[gi].gn;
            const bool col = plan[gi].col;
            Layout G = L;  // this group's view of the scratch: its first pair is "pair 0" of every launch
            G.n = gn;
            G.vsum_ptr = (double *)ctx->fb_vsum.ptr + (size_t)z0 * L.vsum;
            const size_t po = (size_t)z0 * L.planes;
            float *M0 = Mbuf[0] + po, *M1 = Mbuf[1] + po;
            const float *R0 = R[k][0] + po, *R1 = R[k][1] + po;
            float *Mg[2] = {M0, M1};
                        const FlowTab out_tab = sub_tab(out_all, z0, gn);
            // OpenCV-order window (the library default): overlapped strips, or -- above -- column-owning workgroups
            const bool halo = ctx->fb_opencv_rounding == 1 && winsize == 3 && !gaussian;
            // halo: the level's first field, its strip sums and edge rows come from the iteration kernel's "first" forms in one launch
            const bool halo_first = halo;
            HaloScratch hs = {};
            if (halo) hs = halo_scratch(width, height, G);
            const Prolong no_pr = {0, 0, 1.0, 1.0, 1.0};
            if (col) {
                int fk = kHaloZero;
                FlowTab ftab = no_flow;
                Prolong prc = no_pr;
                if (!have_prev && (flags & OFXCV_OPTFLOW_USE_INITIAL_FLOW)) {
                    // the caller's flow, area-resized to the top level and scaled; at k == 0 it is the flow buffer itself
                    fk = kHaloGiven;
                    ftab = sub_tab(out, z0, gn);
                    if (k > 0) {
                        double scale = 1;
                        for (int i = 0; i < k; i++) scale *= pyr_scale;
                        ftab = sub_tab(coarse_tab(cflow[(k & 1) ^ 1], (size_t)w * 8), z0, gn);
                        for (int z = 0; z < gn; z++)
                            if ((rc = launch_initial_flow(ctx, s, (const float *)out.p[z0 + z], out.step[z0 + z], width, height, ftab.p[z], w, h, scale))) return rc;
                    }
                } else if (have_prev) {
                    fk = kHaloCoarse;
                    ftab = sub_tab(prev_all, z0, gn);
                    prc = {pw, ph, 1. / pyr_scale, (double)pw / w, (double)ph / h, ctx->fb_filter_contraction};
                }
                // steps of the level: first M, iterations - 1 x iterate, last -- two per launch
                const int nsteps = iterations + 1;
                int cur = 0;
                for (int st = 0; st < nsteps; st += 2) {
                    const int k1 = st == 0 ? fk : (st == nsteps - 1 ? kHaloLast : kHaloIter);
                    const int k2 = st + 1 >= nsteps ? kColNone : (st + 1 == nsteps - 1 ? kHaloLast : kHaloIter);
                    RgbaTab rg = {};
                    const bool sink = rgba && k == 0 && (k1 == kHaloLast || k2 == kHaloLast);  // F7 rides on the launch that produces the final flow
                    if (sink) {
                        rg.rsx = rgba->rsx;
                        rg.rsy = rgba->rsy;
                        for (int z = 0; z < gn; z++) {
                            rg.p[z] = rgba->p[z0 + z];
                            rg.step[z] = rgba->step[z0 + z];
                            rg.mu[z] = rgba->mu[z0 + z];
                            rg.mv[z] = rgba->mv[z0 + z];
                            rgba_fused[z0 + z] = true;
                        }
                    }
                    const bool prof = profile && k == 0 && k1 == kHaloIter && k2 == kHaloIter;  // the dominant kernel's launches
                    if (prof && (rc = ofxcv_prof_mark(ctx, s))) return rc;
                    rc = launch_col_steps(ctx, s, R0, R1, Mg[cur], Mg[cur ^ 1], ftab, out_tab, prc, w, h, k1, k2, hs, cur, G, sink ? &rg : nullptr);
                    if (rc) return rc;
                    if (prof && (rc = ofxcv_prof_mark(ctx, s))) return rc;
                    cur ^= 1;
                }
                continue;
            }
            if (!have_prev && (flags & OFXCV_OPTFLOW_USE_INITIAL_FLOW)) {
                // the caller's flow, area-resized to the top level and scaled; at k == 0 it is the flow buffer itself
                FlowTab init = sub_tab(out, z0, gn);
                if (k > 0) {
                    double scale = 1;
                    for (int i = 0; i < k; i++) scale *= pyr_scale;
                    init = sub_tab(coarse_tab(cflow[(k & 1) ^ 1], (size_t)w * 8), z0, gn);
                    for (int z = 0; z < gn; z++)
                        if ((rc = launch_initial_flow(ctx, s, (const float *)out.p[z0 + z], out.step[z0 + z], width, height, init.p[z], w, h, scale))) return rc;
                }
                if (halo_first) rc = launch_halo_iteration(ctx, s, R0, R1, nullptr, M0, init, no_pr, w, h, kHaloGiven, hs, 0, G);
                else rc = launch_update_matrices(ctx, s, 2, R0, R1, init, no_pr, w, h, M0, L.planes, gn, true);
            } else if (!have_prev) {
                if (halo_first) rc = launch_halo_iteration(ctx, s, R0, R1, nullptr, M0, no_flow, no_pr, w, h, kHaloZero, hs, 0, G);
                else rc = launch_update_matrices(ctx, s, 0, R0, R1, no_flow, no_pr, w, h, M0, L.planes, gn, true);
            } else {
                const Prolong pr = {pw, ph, 1. / pyr_scale, (double)pw / w, (double)ph / h, ctx->fb_filter_contraction};
                if (halo_first) rc = launch_halo_iteration(ctx, s, R0, R1, nullptr, M0, sub_tab(prev_all, z0, gn), pr, w, h, kHaloCoarse, hs, 0, G);
                else rc = launch_update_matrices(ctx, s, 1, R0, R1, sub_tab(prev_all, z0, gn), pr, w, h, M0, L.planes, gn, true);
            }
            if (rc) return rc;
            int cur = 0;
            for (int i = 0; i < iterations;) {
                const bool prof = profile && k == 0 && i < iterations - 1;  // the dominant kernel's launches
                const bool inner = prof && halo;  // marks set around the kernel inside launch_halo_iteration
                ctx->prof_now = inner;
                if (prof && !inner && (rc = ofxcv_prof_mark(ctx, s))) return rc;
                {
                    bool update = i < iterations - 1;
                    const FlowTab &ft = update ? no_flow : out_tab;
                    if (gaussian)
                        rc = launch_gauss_iteration(ctx, s, R0, R1, Mg[cur], Mg[cur ^ 1], ft, w, h, winsize, update, G);
                    else if (halo) {
                        // F7 rides on the level-0 launch that produces the final flow
                        RgbaTab rg = {};
                        const bool sink = rgba && k == 0 && !update;
                        if (sink) {
                            rg.rsx = rgba->rsx;
                            rg.rsy = rgba->rsy;
                            for (int z = 0; z < gn; z++) {
                                rg.p[z] = rgba->p[z0 + z];
                                rg.step[z] = rgba->step[z0 + z];
                                rg.mu[z] = rgba->mu[z0 + z];
                                rg.mv[z] = rgba->mv[z0 + z];
                                rgba_fused[z0 + z] = true;
                            }
                        }
                        rc = launch_halo_iteration(ctx, s, R0, R1, Mg[cur], Mg[cur ^ 1], ft, no_pr, w, h, update ? kHaloIter : kHaloLast, hs, cur, G, sink ? &rg : nullptr);
                    } else
                        rc = launch_iteration(ctx, s, R0, R1, Mg[cur], Mg[cur ^ 1], ft, w, h, winsize, update, G, true);
                    i += 1;
                }
                ctx->prof_now = false;
                if (rc) return rc;
                if (prof && !inner && (rc = ofxcv_prof_mark(ctx, s))) return rc;
                cur ^= 1;
            }
        }
        prev_all = out_all;
        have_prev = true;
        pw = w;
        ph = h;
    }
    if (rgba) {
        for (int z = 0; z < n; z++) {
            if (!rgba->p[z] || rgba_fused[z]) continue;
            rc = ofxcv_launch_flow_to_rgba(ctx, s_main, out.p[z], out.step[z], width, height, rgba->p[z], rgba->step[z], rgba->mu[z], rgba->mv[z], rgba->rsx, rgba->rsy);
            if (rc) return rc;
        }
    }
    return OFXCV_OK;
}

int ofxcv_calc_optical_flow_farneback_batch(ofxcv_ctx *ctx, int n, const uint8_t *const *d_prev, const size_t *prev_step,
                                            const uint8_t *const *d_next, const size_t *next_step, float *const *d_flow,
                                            const size_t *flow_step, int width, int height, double pyr_scale, int levels, int winsize,
                                            int iterations, int poly_n, double poly_sigma, int flags, void *stream) {
    return ofxcv_calc_optical_flow_farneback_batch_rgba(ctx, n, d_prev, prev_step, d_next, next_step, d_flow, flow_step, width, height, pyr_scale, levels,
                                                        winsize, iterations, poly_n, poly_sigma, flags, nullptr, nullptr, nullptr, nullptr, 1.0, 1.0, stream);
}

int ofxcv_calc_optical_flow_farneback_batch_rgba(ofxcv_ctx *ctx, int n, const uint8_t *const *d_prev, const size_t *prev_step,
                                                 const uint8_t *const *d_next, const size_t *next_step, float *const *d_flow,
                                                 const size_t *flow_step, int width, int height, double pyr_scale, int levels, int winsize,
                                                 int iterations, int poly_n, double poly_sigma, int flags, float *const *d_rgba,
                                                 const ptrdiff_t *rgba_row_bytes, const unsigned *chan_u_mask, const unsigned *chan_v_mask,
                                                 double render_scale_x, double render_scale_y, void *stream) {
    if (!ctx) return OFXCV_ERR_INVALID;
    OFXCV_HIP_CHECK(ctx, hipSetDevice(ctx->hip_device));  // a thread may hold contexts on several devices
    if (n < 1 || n > kMaxBatch) return ofxcv_fail(ctx, OFXCV_ERR_INVALID, "calc_optical_flow_farneback: batch of %d pairs outside 1..%d", n, kMaxBatch);
    if (!d_prev || !prev_step || !d_next || !next_step || !d_flow || !flow_step || width <= 0 || height <= 0)
        return ofxcv_fail(ctx, OFXCV_ERR_INVALID, "calc_optical_flow_farneback: bad argument");
    for (int z = 0; z < n; z++)
        if (!d_prev[z] || !d_next[z] || !d_flow[z] || prev_step[z] < (size_t)width || next_step[z] < (size_t)width ||
            flow_step[z] < (size_t)width * 8 || (flow_step[z] & 7) || (((uintptr_t)d_flow[z]) & 7))
            return ofxcv_fail(ctx, OFXCV_ERR_INVALID, "calc_optical_flow_farneback: bad argument (pair %d)", z);
    if (flags & ~(OFXCV_OPTFLOW_USE_INITIAL_FLOW | OFXCV_OPTFLOW_FARNEBACK_GAUSSIAN))
        return ofxcv_fail(ctx, OFXCV_ERR_UNSUPPORTED, "calc_optical_flow_farneback: flags 0x%x not supported (USE_INITIAL_FLOW, FARNEBACK_GAUSSIAN)", flags);
    if (!(pyr_scale > 0 && pyr_scale < 1) || levels < 0 || iterations < 1 || winsize < 1 || !(winsize & 1))
        return ofxcv_fail(ctx, OFXCV_ERR_INVALID, "calc_optical_flow_farneback: bad parameter");
    // a 5-plane field is addressed through one buffer resource with 32-bit byte offsets: 5 * pitch * height * 4 < 2^31
    if ((size_t)plane_pitch(width) * height * 20 >= (size_t)1 << 31)
        return ofxcv_fail(ctx, OFXCV_ERR_UNSUPPORTED, "calc_optical_flow_farneback: frames above %d pixels exceed the 32-bit buffer offsets",
                          (int)(((size_t)1 << 31) / 20));
    if (poly_n < 1 || poly_n > kMaxPolyN) return ofxcv_fail(ctx, OFXCV_ERR_UNSUPPORTED, "poly_n %d outside 1..%d", poly_n, kMaxPolyN);
    RgbaTab rgba = {};
    bool have_rgba = false;
    if (d_rgba) {
        if (!rgba_row_bytes || !chan_u_mask || !chan_v_mask || render_scale_x == 0 || render_scale_y == 0)
            return ofxcv_fail(ctx, OFXCV_ERR_INVALID, "calc_optical_flow_farneback: bad RGBA argument");
        rgba.rsx = render_scale_x;
        rgba.rsy = render_scale_y;
        for (int z = 0; z < n; z++) {
            if (!d_rgba[z]) continue;
            if ((((uintptr_t)d_rgba[z]) & 3) || (rgba_row_bytes[z] & 3))
                return ofxcv_fail(ctx, OFXCV_ERR_INVALID, "calc_optical_flow_farneback: RGBA image of pair %d is not float-aligned", z);
            rgba.p[z] = d_rgba[z];
            rgba.step[z] = rgba_row_bytes[z];
            rgba.mu[z] = chan_u_mask[z] & 15u;
            rgba.mv[z] = chan_v_mask[z] & 15u;
            have_rgba = true;
        }
    }
    const RgbaTab *rgba_p = have_rgba ? &rgba : nullptr;
    hipStream_t s = ofxcv_stream(ctx, stream);
    // measurement probe (tools/reuse_prep_probe.py), deliberately NOT an option of the C ABI: with OFXCV_DEBUG_REUSE_PREP=1 in the
    // environment a call skips its pyramid images and polynomial expansions and reads what the previous call left in the scratch -- only
    // meaningful when the frames, geometry and batch are those of that call
    {
        const char *e = std::getenv("OFXCV_DEBUG_REUSE_PREP");
        ctx->fb_reuse_prep = e && e[0] == '1';
    }
    levels = num_levels(width, height, pyr_scale, levels);
    if (levels > kMaxLevels) levels = kMaxLevels;

    // scratch per pair: M ping + M pong (level-0 size) + R0/R1 of every level, two coarse flows, the f64 column-sum scratch;
    // per frame: one pyramid image; shared: the row buffer of the two-pass pyramid fall-back
    Layout L;
    int rc = make_layout(ctx, n, width, height, pyr_scale, levels, L);
    if (rc) return rc;
    // (a context that serves calls of varying batch size -- the per-device batch context of the host path, vectorgen.hip -- sizes its scratch for
    // fb_reserve_pairs once: a scratch that grows is freed and re-allocated, and the captured launch sequences hold its addresses)
    Layout Lr = L;
    Lr.n = std::max(n, std::min(ctx->fb_reserve_pairs, kMaxBatch));
    rc = ofxcv_reserve(ctx, ctx->fb_planes, Lr.planes_bytes());
    if (rc) return rc;
    rc = ofxcv_reserve(ctx, ctx->fb_tmp, Lr.tmp_bytes());
    if (rc) return rc;
    if (levels > 0) {
        rc = ofxcv_reserve(ctx, ctx->fb_flow, Lr.flow_bytes());
        if (rc) return rc;
    }
    const bool need_vsum = ctx->fb_opencv_rounding || (flags & OFXCV_OPTFLOW_FARNEBACK_GAUSSIAN);
    if (need_vsum) {
        rc = ofxcv_reserve(ctx, ctx->fb_vsum, Lr.vsum_bytes());
        if (rc) return rc;
    }
    if (ctx->fb_col && !ctx->fb_col_flag.ptr) {  // the trace area of iterate_col_kernel
        rc = ofxcv_reserve(ctx, ctx->fb_col_flag, kColFlagBytes);
        if (rc) return rc;
        OFXCV_HIP_CHECK(ctx, hipMemsetAsync(ctx->fb_col_flag.ptr, 0, kColFlagBytes, s));
    }
    if (ctx->fb_col && !ctx->fb_col_abort) {  // its abort word: pinned, host-coherent (the host reads it at its synchronisation points without a copy)
        std::unique_lock<std::shared_mutex> lock(ofxcv_capture_mutex(ofxcv_lock_index(ctx)));
        void *p = nullptr;
        OFXCV_HIP_CHECK(ctx, hipHostMalloc(&p, 64, hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(p, 0, 64);
        ctx->fb_col_abort = (unsigned *)p;
    }
    rc = ofxcv_farneback_streams(ctx);
    if (rc) return rc;
    hipStream_t sp = ctx->fb_one_stream ? s : ctx->prep;

    ImgTab imgs = {};
    FlowTab out = {};
    for (int z = 0; z < n; z++) {
        imgs.p[2 * z] = d_prev[z];
        imgs.step[2 * z] = prev_step[z];
        imgs.p[2 * z + 1] = d_next[z];
        imgs.step[2 * z + 1] = next_step[z];
        out.p[z] = d_flow[z];
        out.step[z] = flow_step[z];
    }

    // hipGraph replay: the launches of a call are captured once per (pointers, geometry, parameters) and replayed
    // with one hipGraphLaunch.  The measurement hook needs its event pairs between launches and therefore uses the eager path.
    const bool use_graph = !ctx->prof_on && !ctx->fb_no_graph;
    if (!use_graph)
        return enqueue_farneback(ctx, s, sp, L, imgs, out, width, height, pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags,
                                 ctx->prof_on != 0, rgba_p);
    FbGraphKey key;
    std::memset(&key, 0, sizeof(key));
    key.n = n;
    key.width = width; key.height = height; key.levels = levels; key.winsize = winsize; key.iterations = iterations; key.poly_n = poly_n; key.flags = flags;
    key.pyr_scale = pyr_scale; key.poly_sigma = poly_sigma;
    key.planes = ctx->fb_planes.ptr; key.tmp = ctx->fb_tmp.ptr; key.cflow = ctx->fb_flow.ptr; key.vsum = need_vsum ? ctx->fb_vsum.ptr : nullptr;
    for (int z = 0; z < n; z++) {
        key.prev[z] = d_prev[z]; key.next[z] = d_next[z]; key.flow[z] = d_flow[z];
        key.prev_step[z] = prev_step[z]; key.next_step[z] = next_step[z]; key.flow_step[z] = flow_step[z];
        key.rgba[z] = rgba.p[z]; key.rgba_step[z] = rgba.step[z]; key.rgba_mu[z] = rgba.mu[z]; key.rgba_mv[z] = rgba.mv[z];
    }
    key.rsx = rgba.rsx; key.rsy = rgba.rsy;
    FbGraph *g = nullptr;
    for (FbGraph &c : ctx->fb_graphs)
        if (c.exec && !std::memcmp(&c.key, &key, sizeof(key))) g = &c;
    if (!g) {
        FbGraph *slot = &ctx->fb_graphs[0];  // an empty slot, else the least recently replayed one
        for (FbGraph &c : ctx->fb_graphs)
            if (!c.exec) { slot = &c; break; }
            else if (c.used < slot->used) slot = &c;
        // Relaxed mode: the captured region itself makes no capture-unsafe call, and other host threads (each with
        // its own context: allocations, synchronising copies) must not be able to invalidate this capture.  If the
        // capture cannot be completed anyway, the call falls back to plain launches and stops using graphs.
        hipGraph_t graph = nullptr;
        // one capture at a time per device, and no device allocation / free / context teardown of another host thread
        // during it (ofxcv_capture_mutex): either was seen to invalidate a capture on ROCm 7.2.  Launches, copies
        // and graph replays of other threads stay concurrent.
        std::unique_lock<std::shared_mutex> capture_lock(ofxcv_capture_mutex(ofxcv_lock_index(ctx)));
        const auto hold0 = std::chrono::steady_clock::now();
        if (slot->exec) {  // evicted entry: destroyed under the exclusive lock (see common.h)
            (void)hipGraphExecDestroy(slot->exec);
            slot->exec = nullptr;
        }
        bool ok = hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed) == hipSuccess;
        if (ok) {
            rc = enqueue_farneback(ctx, s, sp, L, imgs, out, width, height, pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags, false, rgba_p);
            ok = hipStreamEndCapture(s, &graph) == hipSuccess && rc == OFXCV_OK && graph != nullptr;
            if (ok) ok = hipGraphInstantiate(&slot->exec, graph, nullptr, nullptr, 0) == hipSuccess;
            if (graph) (void)hipGraphDestroy(graph);
        }
        ctx->lock_hold_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - hold0).count();
        ctx->lock_holds++;
        capture_lock.unlock();
        if (!ok) {
            // plain launches from here on, all on the caller's stream (the preparation stream may have been left in
            // the abandoned capture)
            slot->exec = nullptr;
            (void)hipGetLastError();  // clear the sticky capture error
            ctx->fb_no_graph = true;
            ctx->fb_one_stream = true;
            ctx->err[0] = 0;
            return enqueue_farneback(ctx, s, s, L, imgs, out, width, height, pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags, false, rgba_p);
        }
        slot->key = key;
        g = slot;
    }
    {
        // exclusive as well: two threads inside hipGraphLaunch at once (different graph execs, different streams) crashed in
        // hip::Graph::UpdateStreams on ROCm 7.2 -- about 1 in 30 runs of four concurrent render threads, backtrace under
        // rocgdb with every other locked operation parked on this lock
        std::unique_lock<std::shared_mutex> launch_lock(ofxcv_capture_mutex(ofxcv_lock_index(ctx)));
        const auto hold0 = std::chrono::steady_clock::now();
        g->used = ++ctx->fb_graph_clock;
        const hipError_t le = hipGraphLaunch(g->exec, s);
        ctx->lock_hold_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - hold0).count();
        ctx->lock_holds++;
        OFXCV_HIP_CHECK(ctx, le);
    }
    return OFXCV_OK;
}

int ofxcv_farneback_col_pairs(const ofxcv_ctx *ctx, int width, int height, int n) {
    if (!ctx || width <= 0 || height <= 0 || n <= 0) return 0;
    return col_pairs(ctx, width, height, n, ctx->fb_opencv_rounding == 1);
}

int ofxcv_calc_optical_flow_farneback(ofxcv_ctx *ctx, const uint8_t *d_prev, size_t prev_step, const uint8_t *d_next,
                                      size_t next_step, float *d_flow, size_t flow_step, int width, int height, double pyr_scale,
                                      int levels, int winsize, int iterations, int poly_n, double poly_sigma, int flags, void *stream) {
    // a batch of one: the same launch sequence with a grid z of 1
    return ofxcv_calc_optical_flow_farneback_batch(ctx, 1, &d_prev, &prev_step, &d_next, &next_step, &d_flow, &flow_step, width, height, pyr_scale,
                                                   levels, winsize, iterations, poly_n, poly_sigma, flags, stream);
}

}  // extern "C"
