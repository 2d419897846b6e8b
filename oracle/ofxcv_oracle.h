/*
 * ofxcv_oracle.h -- CPU oracle for the openfx-opencv render() hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and there only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED.  The reference (NatronGitHub/openfx-opencv) delegates every
 * arithmetic step of this path to un-vendored, un-pinned third-party code:
 *   - OpenCV (pkg-config opencv; 2.4.x for opencv2fx, 2.4-4.x for VectorGenerator)
 *       modules/video/src/optflowgf.cpp      cv::calcOpticalFlowFarneback
 *       modules/imgproc/src/smooth.cpp       getGaussianKernel / GaussianBlur
 *       modules/imgproc/src/filter.cpp       row/column filter evaluation order
 *       modules/imgproc/src/imgwarp.cpp      resize(INTER_LINEAR)
 *       modules/imgproc/src/color.cpp        RGBA2GRAY / RGBA2RGB
 *       modules/imgproc/src/morph.cpp        cvDilate
 *       modules/photo/src/inpaint.cpp        cvInpaint(CV_INPAINT_TELEA)
 *       modules/imgproc/src/segmentation.cpp pyrMeanShiftFiltering
 *   - openfx-supportext ofxsLut.h           Lut::to_byte_grayscale_nodither
 * None of it is present in /root/reference, no OpenCV build or cv2 wheel exists
 * in the build image, and the reference holds no tests, fixtures or golden
 * vectors.  This oracle restates the published algorithms of those files and is
 * anchored on the reference's own call sites:
 *   VectorGenerator/VectorGenerator.cpp:353-520  (Farneback branch 374-406)
 *   OpenCV/GenericOpenCVPlugin.cpp:223-265       (8-bit sRGB gray fetch)
 *   opencv2fx/inpaint/inpaint.cpp:286-358
 *   opencv2fx/segment/segment.cpp:266-323
 * It is pinned only by the analytic known-answer tests in tests/test_oracle_*.py.
 */
#ifndef OFXCV_ORACLE_H
#define OFXCV_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- helpers shared by all restatements ---------------------------------- */
int orc_cv_round(double v);            /* cvRound: round-half-to-even          */
int orc_cv_floor(double v);            /* cvFloor                              */
int orc_border_reflect101(int p, int len);

/* ---- Farneback (optflowgf.cpp) ------------------------------------------- */

/* getGaussianKernel(n, sigma, CV_32F)  [smooth.cpp] */
void orc_gaussian_kernel_f32(int n, double sigma, float *k);
/* which OpenCV generation's getGaussianKernel is restated: 3 (default: 2.4 / 3.x) or 4 (4.x, taps normalised in double);
 * only the cv2 cross-check flips it */
void orc_set_gaussian_kernel_generation(int generation);
int orc_get_gaussian_kernel_generation(void);

/* GaussianBlur(src,dst,Size(ksize,ksize),sigma,sigma), CV_32F, BORDER_REFLECT_101 */
void orc_gaussian_blur_f32(const float *src, int w, int h, float *dst, int ksize, double sigma);

/* resize(src,dst,Size(dw,dh),INTER_LINEAR) for CV_32FC(cn), interleaved.  cv::resize serves an exact 2x reduction with
 * INTER_AREA's 2x2 block mean; orc_set_resize_generation picks the association of its three additions: 0 (default)
 * ((a+b)+(c+d))/4 = the bilinear form = 4.x SIMD, 1 (((a+b)+c)+d)/4 scalar loop, 2 ((a+c)+(b+d))/4 3.x SSE2 */
void orc_set_resize_generation(int generation);
int orc_get_resize_generation(void);
/* 1: the separable filters and resize's vertical lerp with fused multiply-adds, as OpenCV 4.x's AVX2 / NEON paths (farneback.c) */
/* luma weights of orc_to_byte_grayscale: 709 (default) or 601 */
void orc_set_luma(int standard);
int orc_get_luma(void);
void orc_set_filter_contraction(int on);
int orc_get_filter_contraction(void);
void orc_resize_linear_f32(const float *src, int sw, int sh, int cn, float *dst, int dw, int dh);

/* FarnebackPrepareGaussian: g/xg/xxg hold 2n+1 entries each (index k+n), ig = {ig11,ig03,ig33,ig55} */
void orc_polyexp_prepare(int n, double sigma, float *g, float *xg, float *xxg, double ig[4]);

/* FarnebackPolyExp: I is w*h f32, R is w*h*5 f32 interleaved */
void orc_polyexp(const float *I, int w, int h, float *R, int n, double sigma);

/* FarnebackUpdateMatrices over rows [y0,y1); R0/R1/M 5-ch interleaved, flow 2-ch interleaved */
void orc_update_matrices(const float *R0, const float *R1, const float *flow, float *M,
                         int w, int h, int y0, int y1);

/* blur modes for FarnebackUpdateFlow_Blur */
#define ORC_BLUR_FAITHFUL 0 /* OpenCV's running sums (f32 row differences accumulated in f64) */
#define ORC_BLUR_DIRECT   1 /* same maths, direct 3x3 f64 sums (the order the HIP kernel uses)   */

/* FarnebackUpdateFlow_Blur (flags=0 box window) */
void orc_update_flow_blur(const float *R0, const float *R1, float *flow, float *M,
                          int w, int h, int block_size, int update_matrices, int mode);

/* number of pyramid resolutions actually used minus one (OpenCV's `levels` clip, min side 32) */
int orc_farneback_num_levels(int w, int h, double pyr_scale, int levels);
/* geometry of pyramid level k: size, blur sigma and blur kernel size */
void orc_farneback_level_geom(int w, int h, double pyr_scale, int k,
                              int *lw, int *lh, double *sigma, int *ksize);
/* one pyramid image: convertTo(32F) + GaussianBlur + resize */
void orc_farneback_pyr_image(const uint8_t *img, size_t step, int w, int h,
                             int lw, int lh, double sigma, int ksize, float *I);

/* FarnebackUpdateFlow_GaussianBlur (flags & OPTFLOW_FARNEBACK_GAUSSIAN): separable Gaussian window, f32 sums */
void orc_update_flow_gaussian(const float *R0, const float *R1, float *flow, float *M,
                              int w, int h, int block_size, int update_matrices);
/* resize(INTER_AREA) for f32 (shrinking), used on the initial flow */
void orc_resize_area_f32(const float *src, int sw, int sh, int cn, float *dst, int dw, int dh);

#define ORC_OPTFLOW_USE_INITIAL_FLOW   4   /* cv::OPTFLOW_USE_INITIAL_FLOW   */
#define ORC_OPTFLOW_FARNEBACK_GAUSSIAN 256 /* cv::OPTFLOW_FARNEBACK_GAUSSIAN */

/* cv::calcOpticalFlowFarneback(prev,next,flow,pyr_scale,levels,winsize,iterations,poly_n,poly_sigma,flags)
 * 8-bit single channel inputs with row stride `step`, flow is w*h*2 f32 interleaved (read as the initial flow with
 * ORC_OPTFLOW_USE_INITIAL_FLOW).  blur_mode applies to the box window only.  returns 0 on success. */
int orc_calc_optical_flow_farneback(const uint8_t *prev, const uint8_t *next, size_t step,
                                    int w, int h, float *flow,
                                    double pyr_scale, int levels, int winsize, int iterations,
                                    int poly_n, double poly_sigma, int flags, int blur_mode);

/* ---- ofxsLut: f32 linear RGB(A) -> 8-bit sRGB luma ------------------------ */
void orc_srgb_lut_build(uint16_t *lut /* 65536 entries, 8.8 fixed point */);
/* Lut::to_byte_grayscale_nodither; src row stride in bytes, ncomp 3 or 4 */
void orc_to_byte_grayscale(const float *src, ptrdiff_t src_row_bytes, int ncomp, int w, int h,
                           uint8_t *dst, ptrdiff_t dst_row_bytes);

/* VectorGenerator.cpp:494-519 write-back.  chan_u/chan_v: 4-entry 0/1 masks of the RGBA
 * channels receiving flow.x / flow.y; unmapped channels are left untouched. */
void orc_flow_to_rgba(const float *flow, int w, int h, float *dst, ptrdiff_t dst_row_bytes,
                      const int chan_u[4], const int chan_v[4], double rs_x, double rs_y);

/* ---- inpaint (opencv2fx/inpaint/inpaint.cpp:303-318) ---------------------- */
/* cvCvtColor(RGBA2GRAY) + cvThreshold(0,255,BINARY_INV) + cvDilate(3x3 rect, iters) */
void orc_inpaint_mask(const uint8_t *rgba, ptrdiff_t row_bytes, int w, int h, int dilate_iters,
                      uint8_t *mask /* w*h, 0 or 255 */);
/* cvInpaint(rgb, mask, out, radius, CV_INPAINT_TELEA) on packed 3-channel images.
 * Optional outputs (may be NULL): t_map (h+2)*(w+2) f32 final distance map,
 * f_map (h+2)*(w+2) u8 final flags, order w*h int32 fill order (1-based, 0 = not filled). */
#define ORC_INPAINT_NS    0 /* CV_INPAINT_NS    (never reachable in the reference plugin: inpaint.cpp:311) */
#define ORC_INPAINT_TELEA 1 /* CV_INPAINT_TELEA */
/* cvInpaint(src, mask, dst, radius, method) on 8-bit 3-channel images; maps as for orc_inpaint_telea */
int orc_inpaint(const uint8_t *rgb, const uint8_t *mask, int w, int h, double radius, int method,
                uint8_t *out, float *t_map, uint8_t *f_map, int32_t *order);
int orc_inpaint_telea(const uint8_t *rgb, const uint8_t *mask, int w, int h, double radius,
                      uint8_t *out, float *t_map, uint8_t *f_map, int32_t *order);
/* full render() body of the inpaint plugin for noise==0: RGBA in -> RGBA out (alpha 255) */
int orc_inpaint_render(const uint8_t *src_rgba, ptrdiff_t src_row_bytes, int w, int h,
                       double radius, double dilation, uint8_t *dst_rgba, ptrdiff_t dst_row_bytes);

/* ---- segment (BASELINE "mean-shift": cv::pyrMeanShiftFiltering) ----------- */
int orc_pyr_mean_shift(const uint8_t *rgb, int w, int h, double sp, double sr, int max_level,
                       int max_iter, double eps, uint8_t *out);

#ifdef __cplusplus
}
#endif
#endif
