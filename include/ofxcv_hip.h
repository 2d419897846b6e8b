/*
 * ofxcv_hip.h -- C ABI of libofxcv_hip.so: the MI355X (gfx950) implementation of the per-frame
 * OpenCV calls behind the openfx-opencv render() actions.
 *
 * Every entry point replaces one call the reference makes into OpenCV / openfx-supportext;
 * the reference call site is cited next to each declaration (paths relative to the reference
 * tree).  Plain pointers and sizes only; `stream` is a hipStream_t passed as void* (NULL = the
 * context's own compute stream).  All functions return OFXCV_OK (0) or a negative error code
 * and never throw; ofxcv_last_error() gives the text of the last failure on a context.
 *
 * Pointer conventions: parameters named d_* are DEVICE pointers (HBM), h_* are HOST pointers.
 * Strides (`*_step`, `*_row_bytes`) are in bytes, like cv::Mat::step / kOfxImagePropRowBytes.
 */
#ifndef OFXCV_HIP_H
#define OFXCV_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OFXCV_OK 0
#define OFXCV_ERR_INVALID (-1)     /* bad argument (NULL, non-positive size, ...)            */
#define OFXCV_ERR_HIP (-2)         /* a HIP runtime call failed (text in ofxcv_last_error)   */
#define OFXCV_ERR_MEMORY (-3)      /* device / pinned-host allocation failed                 */
#define OFXCV_ERR_UNSUPPORTED (-4) /* parameter combination outside the reference's use      */
#define OFXCV_ERR_NO_DEVICE (-5)   /* no gfx950 device visible                               */

typedef struct ofxcv_ctx ofxcv_ctx;

/* ---- context ------------------------------------------------------------------------------
 * One context = one device + its scratch (pyramid / polynomial-expansion planes, LUT, pinned
 * staging, two streams).  A context serves one call at a time; concurrent render() threads
 * (VectorGenerator is eRenderFullySafe, VectorGenerator.cpp:108) each take their own. */
/* Devices a host process can spread its render threads over (GenericOpenCVPlugin.cpp:350-357 / VectorGenerator.cpp:108: the reference is
 * eRenderFullySafe, i.e. the host calls render() from many threads).  Environment OFXCV_VIRTUAL_DEVICES=N presents N logical devices over
 * the physical ones (logical d -> physical d % count): the per-device state of a multi-GPU host process -- render threads' device choice,
 * per-device runtime lock, per-device caches of named frames -- then runs on a one-GPU box (tests, soak tools). */
int ofxcv_device_count(void);
int ofxcv_ctx_create(int device, ofxcv_ctx **out);
void ofxcv_ctx_destroy(ofxcv_ctx *ctx);
const char *ofxcv_last_error(const ofxcv_ctx *ctx);
const char *ofxcv_status_string(int status);
int ofxcv_ctx_device(const ofxcv_ctx *ctx);
/* measurement: nanoseconds this context's Farneback calls have held the process-wide (or per-device) runtime lock exclusively -- stream capture
 * + graph instantiation, hipGraphLaunch -- and the number of such holds.  hold x devices / call time = what the lock allows an in-process multi-GPU
 * host */
int ofxcv_lock_hold(const ofxcv_ctx *ctx, long *ns, long *holds);
/* the context's compute stream (hipStream_t), used whenever a `stream` argument is NULL */
void *ofxcv_ctx_stream(const ofxcv_ctx *ctx);
/* hipStreamSynchronize on the context's compute stream (or on `stream` if non-NULL); OFXCV_ERR_HIP if a bounded wait of the
 * column-owning Farneback kernel ever ran out on this context (its flows are then not valid; never observed) */
int ofxcv_ctx_synchronize(ofxcv_ctx *ctx, void *stream);

/* context options (25; unknown names and values outside an option's range are rejected with OFXCV_ERR_INVALID):
 * -- what the results are --
 *   "farneback.opencv_rounding" 1|0|2 how the 3x3 box window of FarnebackUpdateFlow_Blur is evaluated.
 *                                    1 (default): OpenCV's own order -- a running column sum in f64 to which every vertical row difference is
 *                                      added after being rounded to f32 -- evaluated strip-parallel: per level the pairs of the call that fill
 *                                      whole rounds of the chip (one workgroup per tile column and pair; ofxcv_farneback_col_pairs) by
 *                                      column-owning workgroups that run TWO iterations per launch (iterate_col_kernel), the others by
 *                                      overlapped strips, one launch per iteration (iterate3h_kernel).  Every sample within 1e-4 (relative)
 *                                      of the CPU result.
 *                                    0: direct sums (each window summed on its own in f64): at ill-conditioned pixels (6e-5 of the samples at
 *                                      1920x1080, 6e-4 at 3840x2160) the result leaves the 1e-4 band around the reference's.
 *                                    2: OpenCV's order as a serial one-thread-per-column scan (cross-check only, slow).
 *                                    Window sizes other than the reference's 3 always use direct sums.
 *   "farneback.gaussian_kernel_generation" 3|4   which cv::getGaussianKernel the pyramid blur follows: 3 (default) OpenCV 2.4 / 3.x (taps cast
 *                                    to float before they are normalised), 4 = 4.x (normalised in double, one cast);
 *   "farneback.filter_contraction" 0|1   0 (default) product and sum of a filter tap rounded separately (2.4 / 3.x, scalar builds), 1 = fused
 *                                    multiply-adds (the AVX2 / NEON paths of OpenCV 4.x): pyramid images and the flow prolongation;
 *   "farneback.resize_generation" 0|1|2   association of the 2x2 mean cv::resize(INTER_LINEAR) takes at an exact 2x reduction: 0 (default)
 *                                    ((a+b)+(c+d))/4 = bilinear = 4.x SIMD, 1 (((a+b)+c)+d)/4 scalar loop (2.4.x), 2 ((a+c)+(b+d))/4 3.x SSE2;
 *   "lut.luma"                 709|601  luma weights of the gray conversion (supportext's source is not in the reference tree);
 * -- how a Farneback call is planned and launched (never changes a result) --
 *   "farneback.graph"           0|1  0 (default): the launches of a call are enqueued one by one (0.3 ms of host time for a call of 8 pairs at
 *                                    1920x1080; no runtime lock held); 1: captured once into a hipGraph and replayed with one hipGraphLaunch under
 *                                    the process-wide runtime lock (the default of rounds 1-5; same speed);
 *   "farneback.col"             0|1  the column-owning form on the levels that qualify (default 1);
 *   "farneback.col_min"         n    workgroups (tile columns of 60 pixels x pairs) below which no launch takes it (128; from there on a cost
 *                                    model in rounds of the chip decides how many pairs do; smaller values force the form: tests);
 *   "farneback.col_ring"        0|1  that form gathers the second frame's expansion from a ring of rows in LDS (1, default) or from memory (0);
 *   "farneback.col_spin"        n    polls of one of its LDS waits before the kernel raises the abort word (2^22);
 *   "farneback.col_trace"       0|1  its (iterate, iterate) launches stamp the shader clock per phase (tools/col_trace.py);
 *   "farneback.batch_mb"        MiB  the strip-form levels are walked with as many pairs per launch as keep the level's working set under this
 *                                    (160: the Infinity Cache holds 256 MiB);
 *   "farneback.halo_geom"       n    test hook: geometry of the overlapped-strip form instead of the choice by level size (encoding: fb_strips.hip);
 *   "farneback.fused_pyramid"   0..3 1 (default) LDS-fused / direct pyramid kernels, 0 the two-pass kernels; test hooks: 2 the coarse levels by the
 *                                    byte-wise tile kernel, 3 the 3-tap levels without the wavefront-row form;
 * -- host images --
 *   "host.register"           0|1|2  how ofxcv_vectorgen_flow(s)_host moves host images: 1 (default) asynchronous copies straight from / into
 *                                    the host's pageable images; 2 the host's buffers registered (hipHostRegister) for the duration of the
 *                                    call when all four channels are mapped; 0 staged through a pinned ring (what bottom-up images always get);
 *   "host.split"              0|1|2  two directions: 0 one batched Farneback call after the third upload; 1 two single-pair calls, the first
 *                                    while the third frame is still on the wire; 2 (default) 1 while this is the only host-image call in
 *                                    flight on the device, else 0;
 *   "host.cache_mb"           n      budget of the device's cache of named frames (ofxcv_vectorgen_flows_host_keyed), default 512, 0 = off;
 *   "host.coalesce"           0|1|2  host-image calls of several render threads on one device (the reference is eRenderFullySafe,
 *                                    VectorGenerator.cpp:108): 1 (default) a call that finds "host.coalesce_min" (4) host-image calls in flight on
 *                                    its device, itself included, hands its frame pairs to the device's submission queue; the first caller that
 *                                    finds no coalesced call running runs everything queued as ONE batched Farneback call, the callers download
 *                                    their own images.  0 every call runs its own; 2 every call goes through the queue (tests).  Same results.
 *   "host.coalesce_max"       n      frame pairs per coalesced call (2 .. OFXCV_FARNEBACK_MAX_BATCH); 0 (default) = one round of the chip in the
 *                                    column-owning form of level 0: 8 pairs at 1920x1080, 4 at 3840x2160;
 *   "host.coalesce_min"       n      see host.coalesce (default 4: measured at 1920x1080, three threads are faster with their own calls);
 * -- inpaint --
 *   "inpaint.tiles" 0|1, "inpaint.max_tiles" n   tile schedule of the pipelined fill (1), workgroups per fill launch (0 = this call's share of the
 *                                    chip: 192 over the fills in flight, at least 48);
 *   "inpaint.portion" n              fill-order pixels per portion of the pipelined fill (0 = 8192);
 *   "inpaint.spin_limit" n           polls per awaited colour in the dataflow fill before the barrier-scheduled fall-back (-1 = 2^21);
 *   "inpaint.parallel_march" n       0 (default) serial front march; n > 0 the hole's 4-connected components marched side by side on host threads
 *                                    and merged into the exact fill order (1: from 8192 hole pixels; n > 1: from n).  Same maps; slower per call. */
int ofxcv_ctx_set_option(ofxcv_ctx *ctx, const char *name, int value);
/* current value of "farneback.opencv_rounding", "farneback.gaussian_kernel_generation", "farneback.resize_generation", "farneback.filter_contraction",
 * "lut.luma", "farneback.graph", "farneback.batch_mb", "farneback.col", "farneback.col_min", "host.register", "host.split", "host.cache_mb",
 * "host.coalesce", and two counters: "host.split_calls" (host-image calls that took the split form), "farneback.col_aborts" (1 if a bounded wait
 * inside iterate_col_kernel ever ran out; waits for the context's streams: a test hook) */
int ofxcv_ctx_get_option(const ofxcv_ctx *ctx, const char *name, int *value);

/* ---- measurement hook (bench.py's roofline leg) ---------------------------------------------
 * While enabled (enable = 1), the Farneback calls bracket every launch of the dominant kernel at pyramid level 0 with a hipEvent
 * pair on the stream it is launched on: the (iterate, iterate) launches of iterate_col_kernel (two iterations of every pair of the call) where
 * level 0 takes the column-owning form, otherwise the iterating launches of iterate3h_kernel (one iteration); in the other window modes the
 * iterating launches of the generic kernels.  ofxcv_profile_read synchronises, adds up the pairs and returns the total kernel time and the
 * number of launches since the last reset. */
int ofxcv_profile_enable(ofxcv_ctx *ctx, int enable);
int ofxcv_profile_read(ofxcv_ctx *ctx, double *total_ms, long *launches, int reset);

/* ---- F0: f32 linear RGB(A) -> 8-bit sRGB luma ---------------------------------------------
 * replaces OFX::Color::Lut::to_byte_grayscale_nodither as called by
 * GenericOpenCVPlugin::fetchCVImage8UGrayscale (OpenCV/GenericOpenCVPlugin.cpp:223-265, :261). */
int ofxcv_to_byte_grayscale(ofxcv_ctx *ctx, const float *d_src, ptrdiff_t src_row_bytes, int ncomp,
                            int width, int height, uint8_t *d_dst, ptrdiff_t dst_row_bytes, void *stream);

/* the same for n frames of one size in one launch (the 2n frames of a batched Farneback call; host arrays of device pointers).
 * Same bytes as n single calls; frames the four-pixel kernel cannot take (RGB, widths not a multiple of 4, unaligned rows)
 * are converted one by one. */
int ofxcv_to_byte_grayscale_batch(ofxcv_ctx *ctx, int n, const float *const *d_src, const ptrdiff_t *src_row_bytes,
                                  int ncomp, int width, int height, uint8_t *const *d_dst,
                                  const ptrdiff_t *dst_row_bytes, void *stream);

/* ---- F1-F6: dense Farneback optical flow --------------------------------------------------
 * replaces cv::calcOpticalFlowFarneback(prev, next, flow, pyr_scale, levels, winsize,
 * iterations, poly_n, poly_sigma, flags) at VectorGenerator/VectorGenerator.cpp:403
 * (the copyMakeBorder calls at :387-388 pad by zero pixels and are the identity).
 * prev/next: 8-bit single channel, flow: 2-channel interleaved f32 (CV_32FC2), all in HBM.
 * flags: 0 (what the reference passes), OFXCV_OPTFLOW_USE_INITIAL_FLOW (d_flow is read as the
 * initial flow: INTER_AREA-resized to the top pyramid level), OFXCV_OPTFLOW_FARNEBACK_GAUSSIAN
 * (separable Gaussian window instead of the box), or both; anything else is UNSUPPORTED. */
#define OFXCV_OPTFLOW_USE_INITIAL_FLOW 4     /* cv::OPTFLOW_USE_INITIAL_FLOW */
#define OFXCV_OPTFLOW_FARNEBACK_GAUSSIAN 256 /* cv::OPTFLOW_FARNEBACK_GAUSSIAN */
int ofxcv_calc_optical_flow_farneback(ofxcv_ctx *ctx, const uint8_t *d_prev, size_t prev_step,
                                      const uint8_t *d_next, size_t next_step, float *d_flow,
                                      size_t flow_step, int width, int height, double pyr_scale,
                                      int levels, int winsize, int iterations, int poly_n,
                                      double poly_sigma, int flags, void *stream);

/* The same for `n` independent frame pairs of one size in ONE call (1 <= n <= OFXCV_FARNEBACK_MAX_BATCH): every kernel
 * launch of the level walk carries all n pairs, so the pyramid levels that cannot fill the device with one pair (their
 * launches are latency-bound) are amortised over the batch.  Pair i is (d_prev[i], d_next[i]) -> d_flow[i]; pairs may
 * share images (a default VectorGenerator output frame is two pairs with the same first frame, forward t -> t+1 and
 * backward t -> t-1, VectorGenerator/VectorGenerator.cpp:597-638; BASELINE configs[4] batches 8 pairs per GPU).
 * The arrays are host arrays of device pointers / byte strides.  Results are bit-identical to n single calls;
 * ofxcv_calc_optical_flow_farneback IS this call with n = 1.  Scratch grows with n (about 0.2 GB per 1920x1080 pair). */
#define OFXCV_FARNEBACK_MAX_BATCH 16
int ofxcv_calc_optical_flow_farneback_batch(ofxcv_ctx *ctx, int n, const uint8_t *const *d_prev,
                                            const size_t *prev_step, const uint8_t *const *d_next,
                                            const size_t *next_step, float *const *d_flow,
                                            const size_t *flow_step, int width, int height, double pyr_scale,
                                            int levels, int winsize, int iterations, int poly_n,
                                            double poly_sigma, int flags, void *stream);

/* The batched call with F7 (below) fused in: pair i additionally writes flow[i] / render scale into the mapped channels of the
 * RGBA f32 image d_rgba[i] (NULL entry or NULL table: no image for that pair), exactly as ofxcv_flow_to_rgba would from
 * d_flow[i] afterwards -- in the default mode the last level-0 launch of the flow stores the pixels from its registers (no
 * second pass over the flow field, no extra launch); in the other window modes the library appends the F7 launches itself.
 * Replaces VectorGenerator/VectorGenerator.cpp:403 + :494-519 in one call.  Pairs that share a destination image must map
 * disjoint channels (forward flow -> R,G and backward flow -> B,A of one output frame, VectorGenerator.cpp:597-638). */
int ofxcv_calc_optical_flow_farneback_batch_rgba(ofxcv_ctx *ctx, int n, const uint8_t *const *d_prev,
                                                 const size_t *prev_step, const uint8_t *const *d_next,
                                                 const size_t *next_step, float *const *d_flow,
                                                 const size_t *flow_step, int width, int height, double pyr_scale,
                                                 int levels, int winsize, int iterations, int poly_n,
                                                 double poly_sigma, int flags, float *const *d_rgba,
                                                 const ptrdiff_t *rgba_row_bytes, const unsigned *chan_u_mask,
                                                 const unsigned *chan_v_mask, double render_scale_x,
                                                 double render_scale_y, void *stream);

/* how many of the n pairs of a batched call walk level 0 of a width x height frame in the column-owning form (two iterations per
 * launch) with the context's current options -- the first that many; the others keep the overlapped strips (bench.py: which
 * kernel dominates the timed workload) */
int ofxcv_farneback_col_pairs(const ofxcv_ctx *ctx, int width, int height, int n);

/* ---- F7: flow -> RGBA write-back ----------------------------------------------------------
 * replaces the loop at VectorGenerator/VectorGenerator.cpp:494-519.  chan_u_mask/chan_v_mask:
 * bit c set = RGBA channel c receives flow.x / flow.y (divided by the render scale); channels
 * in neither mask are left untouched. */
int ofxcv_flow_to_rgba(ofxcv_ctx *ctx, const float *d_flow, size_t flow_step, int width, int height,
                       float *d_dst, ptrdiff_t dst_row_bytes, unsigned chan_u_mask, unsigned chan_v_mask,
                       double render_scale_x, double render_scale_y, void *stream);

/* ---- whole VectorGenerator::calcOpticalFlow for host-resident OFX images --------------------
 * replaces VectorGeneratorPlugin::calcOpticalFlow (VectorGenerator/VectorGenerator.cpp:353-520,
 * Farneback branch) including the CVImageWrapper marshalling of GenericOpenCVPlugin.cpp:58-165:
 * both f32 frames are staged through pinned memory on a copy stream while the compute stream
 * runs the LUT / flow / scatter kernels, and only the mapped RGBA channels of h_dst are written. */
int ofxcv_vectorgen_flow_host(ofxcv_ctx *ctx, const float *h_ref, ptrdiff_t ref_row_bytes,
                              const float *h_other, ptrdiff_t other_row_bytes, int ncomp,
                              int width, int height, float *h_dst, ptrdiff_t dst_row_bytes,
                              unsigned chan_u_mask, unsigned chan_v_mask, double render_scale_x,
                              double render_scale_y, int levels, int iterations, int poly_n,
                              double poly_sigma);

/* Both directions of a default VectorGenerator output frame in one call (the render() body of
 * VectorGenerator/VectorGenerator.cpp:601-637: forward = frame t against t+1, backward = t against t-1): the reference
 * frame is staged, uploaded and converted once, the second frame pair overlaps the first flow, and the destination is
 * written in one pass.  Equivalent to ofxcv_vectorgen_flow_host(ref, fwd, ...) followed by (ref, bwd, ...); h_fwd or
 * h_bwd may be NULL. */
int ofxcv_vectorgen_flows_host(ofxcv_ctx *ctx, const float *h_ref, ptrdiff_t ref_row_bytes,
                               const float *h_fwd, ptrdiff_t fwd_row_bytes, const float *h_bwd,
                               ptrdiff_t bwd_row_bytes, int ncomp, int width, int height, float *h_dst,
                               ptrdiff_t dst_row_bytes, unsigned fwd_u_mask, unsigned fwd_v_mask,
                               unsigned bwd_u_mask, unsigned bwd_v_mask, double render_scale_x,
                               double render_scale_y, int levels, int iterations, int poly_n,
                               double poly_sigma);

/* The same with the frames NAMED.  Consecutive output frames of a sequence share two of their three source frames, and all
 * the flow needs of a source frame is its 8-bit gray image (F0: 1/16 of the f32 RGBA pixels): a frame whose name -- together
 * with the geometry -- was seen before on this device is neither uploaded nor converted again; its gray image is kept in a
 * per-device cache shared by all contexts (option "host.cache_mb", default 512 MB = 240 frames at 1920x1080; least-recently
 * used entries go first).  The name must identify the PIXELS: it has to change whenever they do -- OFX hosts provide exactly
 * that as kOfxImagePropUniqueIdentifier, which is what the plugin passes; the library trusts it.  NULL or "" = unnamed (the
 * frame takes the plain path; with all three unnamed the call is ofxcv_vectorgen_flows_host).  The reference has no
 * counterpart: it re-marshals every frame of every render() (VectorGenerator.cpp:601-637, GenericOpenCVPlugin.cpp:58-165). */
int ofxcv_vectorgen_flows_host_keyed(ofxcv_ctx *ctx, const float *h_ref, ptrdiff_t ref_row_bytes,
                                     const float *h_fwd, ptrdiff_t fwd_row_bytes, const float *h_bwd,
                                     ptrdiff_t bwd_row_bytes, int ncomp, int width, int height, float *h_dst,
                                     ptrdiff_t dst_row_bytes, unsigned fwd_u_mask, unsigned fwd_v_mask,
                                     unsigned bwd_u_mask, unsigned bwd_v_mask, double render_scale_x,
                                     double render_scale_y, int levels, int iterations, int poly_n,
                                     double poly_sigma, const char *ref_key, const char *fwd_key,
                                     const char *bwd_key);
/* named frames of this context's calls that were found on the device / that were uploaded, converted and kept */
long ofxcv_host_cache_hits(const ofxcv_ctx *ctx);
long ofxcv_host_cache_misses(const ofxcv_ctx *ctx);
/* the cache of the context's device: bytes and frames held; ofxcv_host_cache_clear drops every entry no call is using (and the idle batch
 * contexts of the device's submission queue: what a host calls before it unloads the plugin) */
int ofxcv_host_cache_stats(ofxcv_ctx *ctx, size_t *bytes, int *frames);
int ofxcv_host_cache_clear(ofxcv_ctx *ctx);

/* host-image calls of this context that were served by the device's submission queue (option "host.coalesce"), the frame pairs they brought,
 * and the sum over those calls of the pairs of the batched call each rode in (batch_pairs / calls = mean size of the call a render thread's
 * pairs ended up in) */
int ofxcv_host_coalesce_stats(const ofxcv_ctx *ctx, long *calls, long *pairs, long *batch_pairs);

/* How the host-image calls on this context moved their frames (context option "host.register"): copied straight from the
 * host's pageable images (1, default: ofxcv_host_direct_calls), with the host's buffers registered for the call and addressed
 * by the copy engine / the write-back kernel in place (2: ofxcv_host_zero_copy_calls), or staged through the pinned ring (0,
 * and whatever the other two cannot address: bottom-up images). */
long ofxcv_host_zero_copy_calls(const ofxcv_ctx *ctx);
long ofxcv_host_direct_calls(const ofxcv_ctx *ctx);

/* ---- I0-I2: inpaint hole mask -------------------------------------------------------------
 * replaces cvCvtColor(imgSrc, mask, CV_RGBA2GRAY) + cvThreshold(mask, mask, 0, 255, CV_THRESH_BINARY_INV)
 * + cvDilate(mask, mask, NULL, (int)t2) at opencv2fx/inpaint/inpaint.cpp:305-309.
 * d_rgba: 8-bit RGBA (4-byte aligned rows), d_mask: 8-bit, 255 = hole. */
int ofxcv_inpaint_mask(ofxcv_ctx *ctx, const uint8_t *d_rgba, ptrdiff_t row_bytes, int width, int height,
                       int dilate_iters, uint8_t *d_mask, ptrdiff_t mask_step, void *stream);

/* ---- I3-I4: Telea inpainting ---------------------------------------------------------------
 * replaces cvInpaint(image0, mask, image1, t1, CV_INPAINT_TELEA) at opencv2fx/inpaint/inpaint.cpp:311-318.
 * d_src/d_dst: 8-bit images with `channels` (3 or 4) bytes per pixel, the first three are inpainted (the
 * cvCvtColor RGBA2RGB copies of :303-304 become a pixel stride); d_src != d_dst.  Optional outputs for
 * parity checks (may be NULL): d_t_map (height+2)*(width+2) f32 final distance map of the padded image,
 * d_order_map width*height int32 fill order (1-based, 0 = not filled).  Synchronises `stream`: the
 * fast-marching front is advanced on the calling host thread. */
int ofxcv_inpaint_telea(ofxcv_ctx *ctx, const uint8_t *d_src, ptrdiff_t src_step, int channels,
                        const uint8_t *d_mask, ptrdiff_t mask_step, int width, int height, double radius,
                        uint8_t *d_dst, ptrdiff_t dst_step, float *d_t_map, int *d_order_map, void *stream);

/* cvInpaint(src, mask, dst, radius, method) with either method of photo/src/inpaint.cpp.  The reference plugin
 * hard-codes CV_INPAINT_TELEA (opencv2fx/inpaint/inpaint.cpp:311); CV_INPAINT_NS (Navier-Stokes, icvNSInpaintFMM) is the
 * other value of that argument: same front march and fill order, colour rule from the isophote direction. */
#define OFXCV_INPAINT_NS 0
#define OFXCV_INPAINT_TELEA 1
int ofxcv_inpaint(ofxcv_ctx *ctx, const uint8_t *d_src, ptrdiff_t src_step, int channels,
                  const uint8_t *d_mask, ptrdiff_t mask_step, int width, int height, double radius, int method,
                  uint8_t *d_dst, ptrdiff_t dst_step, float *d_t_map, int *d_order_map, void *stream);
/* The colour fill is a dataflow kernel whose wavefronts poll each other's results with a bounded number of polls
 * (context option "inpaint.spin_limit", default 2^21); a fill whose poll gave up is repeated with a barrier-scheduled
 * kernel (identical colours).  Number of such repeats on this context so far: */
long ofxcv_inpaint_fallback_count(const ofxcv_ctx *ctx);

/* ---- whole inpaint render() body for host-resident OFX images (noise == 0 path) --------------
 * replaces opencv2fx/inpaint/inpaint.cpp:286-358: RGBA in -> RGBA out, alpha forced to 255.  h_mask_out
 * (optional, width*height) receives the dilated hole mask so the caller can apply the libc-rand() noise of
 * :336-347 itself when the noise parameter is non-zero. */
int ofxcv_inpaint_render_host(ofxcv_ctx *ctx, const uint8_t *h_src, ptrdiff_t src_row_bytes, int width, int height,
                              double radius, double dilation, uint8_t *h_dst, ptrdiff_t dst_row_bytes,
                              uint8_t *h_mask_out);

/* ---- S: segmentation by pyramid mean-shift filtering ---------------------------------------
 * stands in for cvPyrSegmentation(image0, image1, storage, &comp, level, thr1, thr2) at
 * opencv2fx/segment/segment.cpp:296-302.  That routine (OpenCV <= 2.4 legacy module) is not in the reference
 * tree; BASELINE.json defines the workload as mean-shift, so the semantics are those of
 * cv::pyrMeanShiftFiltering(src, dst, sp, sr, max_level, TermCriteria(ITER+EPS, max_iter, eps)).
 * d_src/d_dst: 8-bit images with `channels` (3 or 4) bytes per pixel; a 4th channel is copied through. */
int ofxcv_pyr_mean_shift_filtering(ofxcv_ctx *ctx, const uint8_t *d_src, ptrdiff_t src_step, int channels,
                                   int width, int height, double sp, double sr, int max_level, int max_iter,
                                   double eps, uint8_t *d_dst, ptrdiff_t dst_step, void *stream);

/* whole segment render() body for host-resident OFX images (segment.cpp:266-323): RGBA in -> RGBA out,
 * alpha forced to 255 (:315-319); max_iter 5, eps 1 */
int ofxcv_segment_render_host(ofxcv_ctx *ctx, const uint8_t *h_src, ptrdiff_t src_row_bytes, int width, int height,
                              double sp, double sr, int max_level, uint8_t *h_dst, ptrdiff_t dst_row_bytes);

/* ---- stage-level entry points (the internal stages of calcOpticalFlowFarneback) ------------
 * Exposed so each stage can be parity-checked on its own against the oracle's restatement of
 * modules/video/src/optflowgf.cpp.  Planes: a 5-channel field is stored as 5 consecutive planes
 * of `plane_pitch` floats per row (ofxcv_farneback_plane_pitch(width)), plane stride pitch*height. */
int ofxcv_farneback_plane_pitch(int width);
int ofxcv_farneback_num_levels(int width, int height, double pyr_scale, int levels);
int ofxcv_farneback_level_geom(int width, int height, double pyr_scale, int k, int *lw, int *lh,
                               double *sigma, int *ksize);
/* convertTo(CV_32F) + GaussianBlur(ksize, sigma) + resize(lw x lh, INTER_LINEAR); d_I is lw*lh f32 packed */
int ofxcv_farneback_pyr_image(ofxcv_ctx *ctx, const uint8_t *d_img, size_t step, int width, int height,
                              int lw, int lh, double sigma, int ksize, float *d_I, void *stream);
/* FarnebackPolyExp: d_I w*h packed -> d_R 5 planes */
int ofxcv_farneback_polyexp(ofxcv_ctx *ctx, const float *d_I, int width, int height, float *d_R, int poly_n,
                            double poly_sigma, void *stream);
/* FarnebackUpdateMatrices over all rows; d_flow is 2-ch interleaved with flow_step bytes per row */
int ofxcv_farneback_update_matrices(ofxcv_ctx *ctx, const float *d_R0, const float *d_R1, const float *d_flow,
                                    size_t flow_step, int width, int height, float *d_M, void *stream);
/* FarnebackUpdateFlow_Blur (box window `winsize`): flow = solve(blur(M_in)); if update!=0 also
 * M_out = UpdateMatrices(R0,R1,flow).  d_flow may be NULL when update!=0 (flow stays on chip). */
int ofxcv_farneback_update_flow_blur(ofxcv_ctx *ctx, const float *d_R0, const float *d_R1, const float *d_M_in,
                                     float *d_M_out, float *d_flow, size_t flow_step, int width, int height,
                                     int winsize, int update, void *stream);

#ifdef __cplusplus
}
#endif
#endif
