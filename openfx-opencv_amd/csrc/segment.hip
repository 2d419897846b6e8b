// segment.hip -- pyramid mean-shift filtering for the segment plugin.
//
// The reference's segment render() calls cvPyrSegmentation(image0, image1, storage, &comp, 2, thr1, thr2)
// (opencv2fx/segment/segment.cpp:296-302), an OpenCV <= 2.4 "legacy" pyramid-linking routine whose source is
// not in the reference tree; BASELINE.json defines this workload as mean-shift, i.e. the semantics of
// cv::pyrMeanShiftFiltering(src, dst, sp, sr, maxLevel, TermCriteria(ITER+EPS, 5, 1))
// (OpenCV modules/imgproc/src/segmentation.cpp with pyrDown_/pyrUp_ of pyramids.cpp).  That is what is built
// here: all-integer, one pixel per thread, bit-exact against the oracle's restatement.
//
// Layout: every pyramid level holds packed pixels (R | G<<8 | B<<16 | X<<24), one aligned dword per pixel, so
// the (2sp+1)^2 window walk of the mean-shift iteration is one load per visited pixel, served by the L1
// (the windows of the 64 lanes of a wave overlap almost completely).
#include <algorithm>
#include <cmath>

#include "common.h"

namespace {

__device__ __forceinline__ int r101(int p, int len) {
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do {
        if (p < 0) p = -p;
        else p = 2 * (len - 1) - p;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}

template <int CN>
__global__ __launch_bounds__(256) void ms_pack_kernel(const uint8_t *__restrict__ src, ptrdiff_t step, int w, int h, uint32_t *__restrict__ dst) {
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const uint8_t *p = src + (ptrdiff_t)y * step + (size_t)x * CN;
    dst[(size_t)y * w + x] = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);  // top byte 0: the taps use 4-byte dot products
}

// result colours + (for 4-channel images) the alpha of the source pixel
template <int CN>
__global__ __launch_bounds__(256) void ms_unpack_kernel(const uint32_t *__restrict__ res, const uint8_t *__restrict__ src0, ptrdiff_t src0_step,
                                                        int w, int h, uint8_t *__restrict__ dst, ptrdiff_t step) {
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    uint32_t v = res[(size_t)y * w + x];
    uint8_t *p = dst + (ptrdiff_t)y * step + (size_t)x * CN;
    p[0] = (uint8_t)v;
    p[1] = (uint8_t)(v >> 8);
    p[2] = (uint8_t)(v >> 16);
    if (CN == 4) p[3] = src0[(ptrdiff_t)y * src0_step + (size_t)x * 4 + 3];
}

// pyrDown_ (8-bit): 5x5 [1 4 6 4 1]^2, BORDER_REFLECT_101, (sum + 128) >> 8
__global__ __launch_bounds__(256) void ms_pyr_down_kernel(const uint32_t *__restrict__ src, int sw, int sh, uint32_t *__restrict__ dst, int dw, int dh) {
    int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= dw || y >= dh) return;
    const int k[5] = {1, 4, 6, 4, 1};
    int s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const uint32_t *row = src + (size_t)r101(2 * y + i - 2, sh) * sw;
#pragma unroll
        for (int j = 0; j < 5; j++) {
            uint32_t v = row[r101(2 * x + j - 2, sw)];
            int wgt = k[i] * k[j];
            s0 += wgt * (int)(v & 255u);
            s1 += wgt * (int)((v >> 8) & 255u);
            s2 += wgt * (int)((v >> 16) & 255u);
        }
    }
    dst[(size_t)y * dw + x] = (uint32_t)((s0 + 128) >> 8) | ((uint32_t)((s1 + 128) >> 8) << 8) | ((uint32_t)((s2 + 128) >> 8) << 16);
}

// horizontal pass of pyrUp_ for destination column X of source row `s` (three channels at once)
__device__ __forceinline__ void up_hrow(const uint32_t *__restrict__ s, int sw, int X, int out[3]) {
    const int x = X >> 1;
    auto ch = [](uint32_t v, int c) { return (int)((v >> (8 * c)) & 255u); };
#pragma unroll
    for (int c = 0; c < 3; c++) {
        int v;
        if (sw == 1) v = ch(s[0], c) * 8;
        else if (!(X & 1)) {
            if (x == 0) v = ch(s[0], c) * 6 + ch(s[1], c) * 2;
            else if (x == sw - 1) v = ch(s[sw - 2], c) + ch(s[sw - 1], c) * 7;
            else v = ch(s[x - 1], c) + ch(s[x], c) * 6 + ch(s[x + 1], c);
        } else {
            if (x == sw - 1) v = ch(s[sw - 1], c) * 8;
            else v = (ch(s[x], c) + ch(s[x + 1], c)) * 4;
        }
        out[c] = v;
    }
}

// pyrUp_ (8-bit): dst is 2*src or 2*src - 1 in each dimension; (v + 32) >> 6
__global__ __launch_bounds__(256) void ms_pyr_up_kernel(const uint32_t *__restrict__ src, int sw, int sh, uint32_t *__restrict__ dst, int dw, int dh) {
    int X = blockIdx.x * 64 + threadIdx.x, Y = blockIdx.y * 4 + threadIdx.y;
    if (X >= dw || Y >= dh) return;
    const int y = Y >> 1;
    const int ym = r101((y - 1) * 2, sh * 2) / 2, yp = r101((y + 1) * 2, sh * 2) / 2;
    int a[3], b[3], c[3], o[3];
    up_hrow(src + (size_t)y * sw, sw, X, b);
    up_hrow(src + (size_t)yp * sw, sw, X, c);
    if (!(Y & 1)) {
        up_hrow(src + (size_t)ym * sw, sw, X, a);
#pragma unroll
        for (int q = 0; q < 3; q++) o[q] = (a[q] + b[q] * 6 + c[q] + 32) >> 6;
    } else {
#pragma unroll
        for (int q = 0; q < 3; q++) o[q] = ((b[q] + c[q]) * 4 + 32) >> 6;
    }
    dst[(size_t)Y * dw + X] = (uint32_t)(o[0] & 255) | ((uint32_t)(o[1] & 255) << 8) | ((uint32_t)(o[2] & 255) << 16);
}

__device__ __forceinline__ int cdist2(uint32_t a, uint32_t b) {
    int d0 = (int)(a & 255u) - (int)(b & 255u), d1 = (int)((a >> 8) & 255u) - (int)((b >> 8) & 255u),
        d2 = (int)((a >> 16) & 255u) - (int)((b >> 16) & 255u);
    return d0 * d0 + d1 * d1 + d2 * d2;
}

// segmentation.cpp: coarse result pixel (i,j), 1 <= i <= H1-2, 1 <= j <= W1-2, marks fine pixel (2i-1, 2j-1) when any
// of its 8 neighbours differs by >= isr22; the mark is then dilated 3x3.  Both steps in one pass per fine pixel.
__global__ __launch_bounds__(256) void ms_mask_kernel(const uint32_t *__restrict__ d1, int W1, int H1, int W, int H, int isr22,
                                                      uint8_t *__restrict__ mask) {
    int X = blockIdx.x * 64 + threadIdx.x, Y = blockIdx.y * 4 + threadIdx.y;
    if (X >= W || Y >= H) return;
    int m = 0;
    for (int dy = -1; dy <= 1 && !m; dy++)
        for (int dx = -1; dx <= 1 && !m; dx++) {
            int yy = Y + dy, xx = X + dx;
            if (yy < 0 || xx < 0 || yy >= H || xx >= W || !(yy & 1) || !(xx & 1)) continue;
            int i = (yy + 1) >> 1, j = (xx + 1) >> 1;
            if (i < 1 || j < 1 || i > H1 - 2 || j > W1 - 2) continue;
            uint32_t c = d1[(size_t)i * W1 + j];
            for (int a = -1; a <= 1 && !m; a++)
                for (int b = -1; b <= 1 && !m; b++)
                    if ((a | b) && cdist2(c, d1[(size_t)(i + a) * W1 + (j + b)]) >= isr22) m = 1;
        }
    mask[(size_t)Y * W + X] = (uint8_t)m;
}

// the mean-shift iteration of pyrMeanShiftFiltering for one pixel per thread
__global__ __launch_bounds__(256) void ms_iterate_kernel(const uint32_t *__restrict__ src, int W, int H, const uint8_t *__restrict__ mask,
                                                         float sp, int isr2, int max_iter, double eps, uint32_t *__restrict__ dst) {
    int j = blockIdx.x * 64 + threadIdx.x, i = blockIdx.y * 4 + threadIdx.y;
    if (j >= W || i >= H) return;
    if (mask && !mask[(size_t)i * W + j]) return;  // keeps the pyrUp value
    int x0 = j, y0 = i;
    uint32_t cpx = src[(size_t)i * W + j];
    int c0 = cpx & 255u, c1 = (cpx >> 8) & 255u, c2 = (cpx >> 16) & 255u;
    for (int iter = 0; iter < max_iter; iter++) {
        int count = 0, s0 = 0, s1 = 0, s2 = 0, sx = 0, sy = 0;
        int minx = max((int)rint((double)((float)x0 - sp)), 0), miny = max((int)rint((double)((float)y0 - sp)), 0);
        int maxx = min((int)rint((double)((float)x0 + sp)), W - 1), maxy = min((int)rint((double)((float)y0 + sp)), H - 1);
        // Window taps in integer dot products (the fourth byte of every packed pixel is 0):
        //   |t - c|^2 <= isr2  <=>  t.t - thr <= 2 t.c   with thr = isr2 - c.c,
        // and the three colour sums of the selected pixels are dot products with a one-hot byte vector, accumulated
        // by the instruction itself.  11 vector instructions per tap instead of 28; all-integer, so the same result.
        const uint32_t cpk = (uint32_t)c0 | ((uint32_t)c1 << 8) | ((uint32_t)c2 << 16);
        const uint32_t nthr = (uint32_t)(-(isr2 - (c0 * c0 + c1 * c1 + c2 * c2)));
        uint32_t u0 = 0, u1 = 0, u2 = 0;
        for (int y = miny; y <= maxy; y++) {
            const uint32_t *row = src + (size_t)y * W;
            int row_count = 0;
            auto tap = [&](uint32_t v, int x) {
                const int lhs = (int)__builtin_amdgcn_udot4(v, v, nthr, false);
                const int tc = (int)__builtin_amdgcn_udot4(v, cpk, 0u, false);
                const bool in = lhs <= 2 * tc;
                const uint32_t sel = in ? v : 0u;
                u0 = __builtin_amdgcn_udot4(sel, 0x00000001u, u0, false);
                u1 = __builtin_amdgcn_udot4(sel, 0x00000100u, u1, false);
                u2 = __builtin_amdgcn_udot4(sel, 0x00010000u, u2, false);
                sx += in ? x : 0;
                row_count += in ? 1 : 0;
            };
            int x = minx;
            for (; x + 7 <= maxx; x += 8) {  // eight loads in flight per lane
                uint32_t v[8];
#pragma unroll
                for (int q = 0; q < 8; q++) v[q] = row[x + q];
#pragma unroll
                for (int q = 0; q < 8; q++) tap(v[q], x + q);
            }
            for (; x + 3 <= maxx; x += 4) {
                const uint32_t v0 = row[x], v1 = row[x + 1], v2 = row[x + 2], v3 = row[x + 3];
                tap(v0, x);
                tap(v1, x + 1);
                tap(v2, x + 2);
                tap(v3, x + 3);
            }
            for (; x <= maxx; x++) tap(row[x], x);
            count += row_count;
            sy += y * row_count;
        }
        s0 = (int)u0;
        s1 = (int)u1;
        s2 = (int)u2;
        if (count == 0) break;
        double icount = 1. / count;
        int x1 = (int)rint(sx * icount), y1 = (int)rint(sy * icount);
        s0 = (int)rint(s0 * icount);
        s1 = (int)rint(s1 * icount);
        s2 = (int)rint(s2 * icount);
        int dd0 = s0 - c0, dd1 = s1 - c1, dd2 = s2 - c2;
        bool stop = (x0 == x1 && y0 == y1) || (double)(abs(x1 - x0) + abs(y1 - y0) + dd0 * dd0 + dd1 * dd1 + dd2 * dd2) <= eps;
        x0 = x1; y0 = y1;
        c0 = s0; c1 = s1; c2 = s2;
        if (stop) break;
    }
    dst[(size_t)i * W + j] = (uint32_t)(c0 & 255) | ((uint32_t)(c1 & 255) << 8) | ((uint32_t)(c2 & 255) << 16);
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

extern "C" {

int ofxcv_pyr_mean_shift_filtering(ofxcv_ctx *ctx, const uint8_t *d_src, ptrdiff_t src_step, int channels, int width, int height,
                                   double sp, double sr, int max_level, int max_iter, double eps, uint8_t *d_dst, ptrdiff_t dst_step,
                                   void *stream) {
    if (!ctx) return OFXCV_ERR_INVALID;
    OFXCV_HIP_CHECK(ctx, hipSetDevice(ctx->hip_device));  // a thread may hold contexts on several devices
    if (!d_src || !d_dst || width <= 0 || height <= 0 || (channels != 3 && channels != 4))
        return ofxcv_fail(ctx, OFXCV_ERR_INVALID, "pyr_mean_shift_filtering: bad argument");
    if (max_level < 0 || max_level > 8) return ofxcv_fail(ctx, OFXCV_ERR_INVALID, "pyr_mean_shift_filtering: max_level outside 0..8");
    max_iter = std::min(std::max(max_iter, 1), 100);
    if (eps < 0) eps = 0;
    hipStream_t s = ofxcv_stream(ctx, stream);
    const int isr2 = ofxcv_cv_round(sr * sr), isr22 = std::max(isr2, 16);

    int lw[9], lh[9];
    size_t off_src[9], off_dst[9], total = 0;
    lw[0] = width;
    lh[0] = height;
    for (int l = 0; l <= max_level; l++) {
        if (l > 0) {
            lw[l] = (lw[l - 1] + 1) / 2;
            lh[l] = (lh[l - 1] + 1) / 2;
        }
        off_src[l] = total;
        total += align_up((size_t)lw[l] * lh[l] * 4, 256);
        off_dst[l] = total;
        total += align_up((size_t)lw[l] * lh[l] * 4, 256);
    }
    const size_t off_mask = total;
    total += align_up((size_t)width * height, 256);
    int rc = ofxcv_reserve(ctx, ctx->seg_work, total);
    if (rc) return rc;
    char *base = (char *)ctx->seg_work.ptr;
    auto S = [&](int l) { return (uint32_t *)(base + off_src[l]); };
    auto D = [&](int l) { return (uint32_t *)(base + off_dst[l]); };
    uint8_t *mask = (uint8_t *)(base + off_mask);

    dim3 pb(256), pg(ofxcv_div_up(width, 256), height);
    if (channels == 4) hipLaunchKernelGGL(ms_pack_kernel<4>, pg, pb, 0, s, d_src, src_step, width, height, S(0));
    else hipLaunchKernelGGL(ms_pack_kernel<3>, pg, pb, 0, s, d_src, src_step, width, height, S(0));
    OFXCV_LAUNCH_CHECK(ctx, "ms_pack_kernel");
    dim3 tb(64, 4);
    for (int l = 1; l <= max_level; l++) {
        hipLaunchKernelGGL(ms_pyr_down_kernel, dim3(ofxcv_div_up(lw[l], 64), ofxcv_div_up(lh[l], 4)), tb, 0, s, (const uint32_t *)S(l - 1),
                           lw[l - 1], lh[l - 1], S(l), lw[l], lh[l]);
        OFXCV_LAUNCH_CHECK(ctx, "ms_pyr_down_kernel");
    }
    for (int l = max_level; l >= 0; l--) {
        float spl = (float)(sp / (1 << l));
        if (spl < 1) spl = 1;
        dim3 grid(ofxcv_div_up(lw[l], 64), ofxcv_div_up(lh[l], 4));
        const uint8_t *m = nullptr;
        if (l < max_level) {
            hipLaunchKernelGGL(ms_pyr_up_kernel, grid, tb, 0, s, (const uint32_t *)D(l + 1), lw[l + 1], lh[l + 1], D(l), lw[l], lh[l]);
            OFXCV_LAUNCH_CHECK(ctx, "ms_pyr_up_kernel");
            hipLaunchKernelGGL(ms_mask_kernel, grid, tb, 0, s, (const uint32_t *)D(l + 1), lw[l + 1], lh[l + 1], lw[l], lh[l], isr22, mask);
            OFXCV_LAUNCH_CHECK(ctx, "ms_mask_kernel");
            m = mask;
        }
        hipLaunchKernelGGL(ms_iterate_kernel, grid, tb, 0, s, (const uint32_t *)S(l), lw[l], lh[l], m, spl, isr2, max_iter, eps, D(l));
        OFXCV_LAUNCH_CHECK(ctx, "ms_iterate_kernel");
    }
    if (channels == 4)
        hipLaunchKernelGGL(ms_unpack_kernel<4>, pg, pb, 0, s, (const uint32_t *)D(0), d_src, src_step, width, height, d_dst, dst_step);
    else
        hipLaunchKernelGGL(ms_unpack_kernel<3>, pg, pb, 0, s, (const uint32_t *)D(0), d_src, src_step, width, height, d_dst, dst_step);
    OFXCV_LAUNCH_CHECK(ctx, "ms_unpack_kernel");
    return OFXCV_OK;
}

int ofxcv_segment_render_host(ofxcv_ctx *ctx, const uint8_t *h_src, ptrdiff_t src_row_bytes, int width, int height, double sp, double sr,
                              int max_level, uint8_t *h_dst, ptrdiff_t dst_row_bytes) {
    if (!ctx) return OFXCV_ERR_INVALID;
    if (!h_src || !h_dst || width <= 0 || height <= 0) return ofxcv_fail(ctx, OFXCV_ERR_INVALID, "segment_render_host: bad argument");
    OFXCV_HIP_CHECK(ctx, hipSetDevice(ctx->hip_device));
    hipStream_t s = ctx->compute;
    const size_t row = (size_t)width * 4, img = align_up(row * height, 256);
    int rc = ofxcv_reserve(ctx, ctx->ip_img, 2 * img);
    if (rc) return rc;
    uint8_t *d_src = (uint8_t *)ctx->ip_img.ptr, *d_dst = d_src + img;
    rc = ofxcv_upload_rows(ctx, d_src, row, h_src, src_row_bytes, height, s);
    if (rc) return rc;
    rc = ofxcv_pyr_mean_shift_filtering(ctx, d_src, (ptrdiff_t)row, 4, width, height, sp, sr, max_level, 5, 1.0, d_dst, (ptrdiff_t)row, s);
    if (rc) return rc;
    rc = ofxcv_download_rows(ctx, h_dst, dst_row_bytes, d_dst, row, height, s);
    if (rc) return rc;
    OFXCV_HIP_CHECK(ctx, hipStreamSynchronize(s));
    for (int y = 0; y < height; y++) {  // segment.cpp:315-319: alpha = 255
        uint8_t *d = h_dst + (ptrdiff_t)y * dst_row_bytes;
        for (int x = 0; x < width; x++) d[x * 4 + 3] = 255;
    }
    return OFXCV_OK;
}

}  // extern "C"
